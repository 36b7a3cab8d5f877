"""The surface boxes' T-only equilibrium constants as fitted polynomials (hx_chem_fit.inc,
tools/make_chem_fit.py, chem_constants_fit in hx_dev_chem.h).

* the committed table is what the generator produces and reproduces the reference's formulas
  (src/ocean_csys.cpp:205-287, restated in double here) to a few ulp over the whole interval;
* members whose SST leaves the interval (SSP5-8.5 with S = 8 and 10 K: +13 K in 2245 / 2171) take
  the formulas themselves: against the oracle, through the switch-over;
* a member's result does not depend on whether its wavefront holds such a member (bit for bit)."""
import json
import os
import re

import numpy as np
import pytest

import hector_amd
from conftest import ROOT

SSP585 = os.path.join(ROOT, "hector_amd", "data", "ssp585.hxs")
REL_CO2 = 2e-8
ABS_T = 2e-8


def _table():
    txt = open(os.path.join(ROOT, "hector_amd", "csrc", "hx_chem_fit.inc")).read()
    deg = int(re.search(r"HX_CHEM_FIT_DEGREE (\d+)", txt).group(1))
    hw = float(re.search(r"HX_CHEM_FIT_HALF_WIDTH ([\d.]+)", txt).group(1))
    cH = float(re.search(r"HX_CHEM_FIT_CENTRE_HL ([\d.eE+-]+)", txt).group(1))
    cL = float(re.search(r"HX_CHEM_FIT_CENTRE_LL ([\d.eE+-]+)", txt).group(1))
    body = txt[txt.index("{", txt.index("hx_chem_fit_table")) + 1:txt.rindex("}")]
    vals = np.array([float.fromhex(x.strip()) for x in body.replace("\n", " ").split(",") if x.strip()])
    return deg, hw, (cH, cL), vals.reshape(deg + 1, 12)


def _formulas(Tc):
    """K0, Kw, 1/Kh, K1, K2, Kb of src/ocean_csys.cpp:205-287 at S = 34.5 (numpy double)."""
    S = 34.5; sqrtS = np.sqrt(S); S15 = S ** 1.5
    Tk = Tc + 273.15; lnTk = np.log(Tk); lnTk100 = np.log(Tk / 100); T100 = Tk / 100
    K0 = np.exp(-58.0931 + 90.5069 * (100 / Tk) + 22.2940 * lnTk100 +
                S * (0.027766 - 0.025888 * T100 + 0.0050578 * T100 * T100))
    Kw = np.exp(-13847.26 / Tk + 148.96502 - 23.6521 * lnTk +
                (118.67 / Tk - 5.977 + 1.0495 * lnTk) * sqrtS - 0.01615 * S)
    Kh = np.exp(9345.17 / Tk - 60.2409 + 23.3585 * lnTk100 +
                S * (0.023517 - 0.00023656 * Tk + 0.0047036e-4 * Tk * Tk))
    pK1 = 3633.86 / Tk - 61.2172 + 9.6777 * lnTk - 0.011555 * S + 0.0001152 * S * S
    pK2 = 471.78 / Tk + 25.9290 - 3.16967 * lnTk - 0.01781 * S + 0.0001122 * S * S
    Kb = np.exp((-8966.90 - 2890.53 * sqrtS - 77.942 * S + 1.728 * S15 - 0.0996 * S * S) / Tk +
                148.0248 + 137.1942 * sqrtS + 1.62142 * S + (-24.4344 - 25.085 * sqrtS - 0.2474 * S) * lnTk +
                0.053105 * sqrtS * Tk)
    return [K0, Kw, 1 / Kh, 10.0 ** (-pK1), 10.0 ** (-pK2), Kb]


def test_fit_table_reproduces_the_formulas_over_its_interval():
    deg, hw, centres, tab = _table()
    assert (deg, hw) == (13, 8.0) and centres == (6.6, 25.9)   # SST anomaly + 18 + deltaT, +5 K
    rep = json.load(open(os.path.join(ROOT, "profiles", "chem_fit_report.json")))
    assert rep["max_double_horner_error_rel"] < 5e-16
    t = np.linspace(-1.0, 1.0, 20001)
    for b, c in enumerate(centres):
        ref = _formulas(c + hw * t)
        for f in range(6):
            p = np.full_like(t, tab[0, b * 6 + f])
            for j in range(1, deg + 1):
                p = p * t + tab[j, b * 6 + f]
            # (the double-precision formulas themselves -- the reference's arithmetic -- carry
            # 1e-14 ... 4e-13: their exponents are sums of terms of magnitude 60 ... 1000, Kb's
            # the largest; against the 60-digit formulas the table is good to 3.4e-16,
            # profiles/chem_fit_report.json)
            assert np.abs(p / ref[f] - 1).max() < 2e-12, (b, f)


def _leaving_the_interval(lib, oracle_mod, **kw):
    S = np.tile([3.0, 6.0, 8.0, 10.0], 16)          # every wavefront holds members that leave it
    n = S.size
    c = hector_amd.Core(SSP585, n, lib_path=lib, **kw).setvar("S", S, "degC")
    c.set_member_sorting(False)
    c.set_pair_kernel_limit(0)
    c.set_outputs(["CO2_concentration", "global_tas", "sst", "timesteps"])
    c.run(2300)
    assert (c.status() == 0).all()
    sst = c.fetchvars("sst", (2300, 2300))[0]
    assert sst[0] < 13 and sst[2] > 13 and sst[3] > 13
    o = oracle_mod.Oracle(SSP585)
    for i in range(4):
        p = o.default_params(); p.S = S[i]
        r, err, _ = o.run(p)
        assert err == 0
        ref = r["CO2_concentration"]
        assert (np.abs(c.fetchvars("CO2_concentration", (1745, 2300))[:, i] - ref) / ref).max() < REL_CO2, S[i]
        assert np.abs(c.fetchvars("global_tas", (1745, 2300))[:, i] - r["global_tas"]).max() < ABS_T, S[i]
    # the S = 3 and S = 6 members next to neighbours that left the interval, and among themselves
    d = hector_amd.Core(SSP585, n, lib_path=lib, **kw).setvar("S", np.tile([3.0, 6.0], 32), "degC")
    d.set_member_sorting(False)
    d.set_pair_kernel_limit(0)
    d.set_outputs(["CO2_concentration", "global_tas"])
    d.run(2300)
    for v in ("CO2_concentration", "global_tas"):
        x, y = c.fetchvars(v, (1745, 2300)), d.fetchvars(v, (1745, 2300))
        assert np.array_equal(x[:, 0], y[:, 0]) and np.array_equal(x[:, 1], y[:, 1]), v
    c.shutdown(); d.shutdown()


def test_members_leaving_the_fit_interval_in_the_host_build(emul_lib, oracle):
    import oracle_binding
    _leaving_the_interval(emul_lib, oracle_binding, allow_emulation=True)


@pytest.mark.gpu
@pytest.mark.parametrize("two_wave", [0, 1])
def test_members_leaving_the_fit_interval_on_gpu(hip_lib, oracle, two_wave, monkeypatch):
    import oracle_binding
    monkeypatch.setenv("HECTOR_AMD_TWO_WAVE_FROM", str(two_wave))
    _leaving_the_interval(hip_lib, oracle_binding, device=0)
