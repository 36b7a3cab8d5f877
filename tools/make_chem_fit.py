#!/usr/bin/env python3
"""Polynomial form of the surface boxes' T-only equilibrium constants for the run kernels.

    python tools/make_chem_fit.py      ->  hector_amd/csrc/hx_chem_fit.inc  (+ profiles/chem_fit_report.json)

The reference evaluates, once per box and year, six functions of the box temperature alone
(src/ocean_csys.cpp:205-287: K0, Kw, Kh, K1, K2, Kb -- exponentials of a/T + b + c ln T + d T
polynomials at S = 34.5).  A box's temperature is SST anomaly + 18 + deltaT (oceanbox.cpp:97-99,
309-323), i.e. it stays within a few kelvin of a fixed centre for the whole run, and over
+-HALF_WIDTH kelvin each of the six is reproduced to better than 1e-16 relative by ONE polynomial
of degree DEGREE in t = (Tc - centre) / HALF_WIDTH: 13 multiply-adds instead of an argument
polynomial, a logarithm and a 23-instruction exponential.  Members whose SST leaves the interval
take the formulas themselves (hx_kernels.hip, phase A).

Method: Chebyshev interpolation at DEGREE + 1 Chebyshev nodes in 60-digit arithmetic (mpmath),
converted exactly to monomial coefficients in t and rounded to double; the report holds, for every
function, the truncation error of the 60-digit polynomial and the error of the rounded polynomial
evaluated by Horner's rule in IEEE double (what the kernel does), both against the 60-digit
formulas on 4 001 points of the interval."""
import json
import os
import struct

import mpmath as mp
import numpy as np

mp.mp.dps = 60
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEGREE = 13
HALF_WIDTH = 8        # kelvin: SST anomalies from -3 K to +13 K
BOXES = (("HL", mp.mpf("18") + mp.mpf("-16.4") + 5), ("LL", mp.mpf("18") + mp.mpf("2.9") + 5))
NAMES = ("K0", "Kw", "rKh", "K1", "K2", "Kb")   # the order of chem_exponents' six exponentials
M = mp.mpf


def constants(Tc):
    """src/ocean_csys.cpp:205-287 at S = 34.5, 60 digits."""
    S = M("34.5"); sqrtS = mp.sqrt(S); S15 = S ** M("1.5")
    Tk = Tc + M("273.15"); lnTk = mp.log(Tk); lnTk100 = mp.log(Tk / 100); T100 = Tk / 100
    K0 = mp.exp(M("-58.0931") + M("90.5069") * (100 / Tk) + M("22.2940") * lnTk100 +
                S * (M("0.027766") - M("0.025888") * T100 + M("0.0050578") * T100 * T100))
    Kw = mp.exp(M("-13847.26") / Tk + M("148.96502") - M("23.6521") * lnTk +
                (M("118.67") / Tk - M("5.977") + M("1.0495") * lnTk) * sqrtS - M("0.01615") * S)
    Kh = mp.exp(M("9345.17") / Tk - M("60.2409") + M("23.3585") * lnTk100 +
                S * (M("0.023517") - M("0.00023656") * Tk + M("0.0047036e-4") * Tk * Tk))
    pK1 = M("3633.86") / Tk - M("61.2172") + M("9.6777") * lnTk - M("0.011555") * S + M("0.0001152") * S * S
    pK2 = M("471.78") / Tk + M("25.9290") - M("3.16967") * lnTk - M("0.01781") * S + M("0.0001122") * S * S
    Kb = mp.exp((M("-8966.90") - M("2890.53") * sqrtS - M("77.942") * S + M("1.728") * S15 - M("0.0996") * S * S) / Tk +
                M("148.0248") + M("137.1942") * sqrtS + M("1.62142") * S +
                (M("-24.4344") - M("25.085") * sqrtS - M("0.2474") * S) * lnTk + M("0.053105") * sqrtS * Tk)
    return {"K0": K0, "Kw": Kw, "rKh": 1 / Kh, "K1": M(10) ** (-pK1), "K2": M(10) ** (-pK2), "Kb": Kb}


def cheb_to_monomial(c):
    """Chebyshev coefficients -> monomial coefficients (ascending), exact in mp arithmetic."""
    n = len(c)
    T = [[M(1)], [M(0), M(1)]]
    for k in range(2, n):
        a = [M(0)] + [2 * x for x in T[k - 1]]
        b = T[k - 2] + [M(0)] * (len(a) - len(T[k - 2]))
        T.append([x - y for x, y in zip(a, b)])
    out = [M(0)] * n
    for k in range(n):
        for i, x in enumerate(T[k]):
            out[i] += c[k] * x
    return out


def fit(f, centre):
    N = DEGREE + 1
    xs = [mp.cos(mp.pi * (k + M(1) / 2) / N) for k in range(N)]
    fs = [f(centre + HALF_WIDTH * x) for x in xs]
    c = [2 / M(N) * sum(fs[k] * mp.cos(mp.pi * j * (k + M(1) / 2) / N) for k in range(N)) for j in range(N)]
    c[0] /= 2
    return cheb_to_monomial(c)


def main():
    table = np.zeros((DEGREE + 1, 12))     # [power, descending][box * 6 + function]
    report = {"degree": DEGREE, "half_width_K": HALF_WIDTH, "functions": {}}
    ts = [M(k) / 2000 - 1 for k in range(4001)]
    for b, (box, centre) in enumerate(BOXES):
        ref = [constants(centre + HALF_WIDTH * t) for t in ts]
        for f, name in enumerate(NAMES):
            mono = fit(lambda T: constants(T)[name], centre)
            coef = np.array([float(x) for x in mono])
            trunc = max(abs(sum(m * t ** i for i, m in enumerate(mono)) / r[name] - 1) for t, r in zip(ts, ref))
            tt = np.array([float(t) for t in ts])
            p = np.full_like(tt, coef[DEGREE])
            for i in range(DEGREE - 1, -1, -1):
                p = p * tt + coef[i]           # (no FMA here: an upper bound for the kernel's Horner)
            rnd = max(abs(M(float(v)) / r[name] - 1) for v, r in zip(p, ref))
            report["functions"]["%s.%s" % (box, name)] = {
                "truncation_error_rel": float(trunc), "double_horner_error_rel": float(rnd),
                "centre_degC": float(centre)}
            table[:, b * 6 + f] = coef[::-1]
    report["max_double_horner_error_rel"] = max(v["double_horner_error_rel"] for v in report["functions"].values())
    lines = ["// GENERATED by tools/make_chem_fit.py -- do not edit.  The surface boxes' T-only equilibrium",
             "// constants (src/ocean_csys.cpp:205-287, S = 34.5) as degree-%d polynomials in" % DEGREE,
             "// t = (Tc - centre) / %d K: [power, descending][box * 6 + {K0, Kw, 1/Kh, K1, K2, Kb}]," % HALF_WIDTH,
             "// boxes HL (centre %.1f degC) and LL (centre %.1f degC); largest error of the double-precision"
             % (float(BOXES[0][1]), float(BOXES[1][1])),
             "// Horner evaluation against the 60-digit formulas: %.1e relative (profiles/chem_fit_report.json)."
             % report["max_double_horner_error_rel"],
             "#define HX_CHEM_FIT_DEGREE %d" % DEGREE,
             "#define HX_CHEM_FIT_HALF_WIDTH %d.0" % HALF_WIDTH,
             "#define HX_CHEM_FIT_CENTRE_HL %r" % float(BOXES[0][1]),
             "#define HX_CHEM_FIT_CENTRE_LL %r" % float(BOXES[1][1]),
             "static const double hx_chem_fit_table[(HX_CHEM_FIT_DEGREE + 1) * 12] = {"]
    for j in range(DEGREE + 1):
        lines.append("    " + ", ".join("%s" % float(x).hex() for x in table[j]) + ",")
    lines.append("};")
    with open(os.path.join(ROOT, "hector_amd", "csrc", "hx_chem_fit.inc"), "w") as fo:
        fo.write("\n".join(lines) + "\n")
    with open(os.path.join(ROOT, "profiles", "chem_fit_report.json"), "w") as fo:
        json.dump(report, fo, indent=1)
    print(json.dumps({k: v for k, v in report.items() if k != "functions"}))
    for k, v in report["functions"].items():
        print(k, "%.1e %.1e" % (v["truncation_error_rel"], v["double_horner_error_rel"]))


if __name__ == "__main__":
    main()
