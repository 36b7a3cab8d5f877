// hx_dev_chem.h -- ocean carbonate chemistry (oceancsys), air-sea flux, alkalinity tuning (Brent); fast division
// Part of the device code of hx_kernels.hip (one translation unit; see its header for the
// reference file:line map).
#pragma once

namespace {

// Division where the last bit does not matter.  clang expands an IEEE fp64 division into 11
// dependent VALU instructions (div_scale x2, rcp, 4 fma, mul, fma, div_fmas, div_fixup); with
// one wavefront per SIMD that chain is fully exposed.  v_rcp_f64 is good to 4.6e-8 (measured
// on gfx950); one Newton step brings it to 2e-15, two to 1.1e-16 (<= 1 ulp), for normal-range
// operands, which is what the model has.  ~65 divisions per member-year.
__device__ __forceinline__ double hx_recip(double b) {
  double r = HX_RCP(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double hx_div(double a, double b) { return a * hx_recip(b); }
// a / b given r = hx_recip(b), with the final residual correction of a division routine: the
// quotient is correctly rounded (exact where a / b is representable) for two more FMAs
__device__ __forceinline__ double hx_div_cr(double a, double b, double r) {
  const double q = a * r;
  return fma(fma(-q, b, a), r, q);
}
// 2e-15: enough for a Newton correction, whose own error is squared away by the next iteration
__device__ __forceinline__ double hx_div1(double a, double b) {
  double r = HX_RCP(b);
  r = fma(fma(-b, r, 1.0), r, r);
  return a * r;
}

struct ChemK {  // T-dependent equilibrium constants of one surface box
  double K1, K2, Kb, Kw, Kh, Tr;
  double rKh;  // 1 / Kh (run kernels: exp(-a) from the year's batch; pCO2 = CO2* 1e6 rKh)
  double g;  // Tr * As * 12 / 1e15: annual flux per uatm of air-sea pCO2 difference
  // Run kernels: the quintic's coefficients as functions of DIC, p_i = A_i dic + C_i -- the box's
  // equilibrium constants and the member's alkalinity are fixed for the year, so a solve sets
  // its polynomial up with three multiply-adds instead of ~40 operations (chem_poly_constants):
  //   p4 = C4,  p3 = K1 dic + C3,  p2 = A2 dic + C2,  p1 = A1 dic + C1,  p0 = C0
  double A2, A1, C4, C3, C2, C1, C0, K1K2;
};

// oceancsys::ocean_csys_run, the part that depends only on T (S = 34.5, U = 6.7)
// src/ocean_csys.cpp:205-287, 349
__device__ __forceinline__ void chem_constants(double Tc, ChemK &k) {
  const double S = O_S;
  const double sqrtS = 5.873670062235365;      // sqrt(34.5)
  const double S15 = 202.64161714712009;       // 34.5^1.5
  const double Tk = Tc + 273.15;
  const double lnTk = log(Tk);
  const double lnTk100 = log(Tk / 100);
  double tmp1 = -58.0931 + 90.5069 * (100 / Tk) + 22.2940 * lnTk100;
  double tmp2 = S * (0.027766 - 0.025888 * (Tk / 100) +
                     0.0050578 * ((Tk / 100) * (Tk / 100)));
  const double K0 = exp(tmp1 + tmp2);
  const double Sc =
      2073.1 - (125.62 * Tc) + (3.6276 * Tc * Tc) - (0.043219 * Tc * Tc * Tc);
  tmp1 = -13847.26 / Tk + 148.96502 - 23.6521 * lnTk;
  tmp2 = +(118.67 / Tk - 5.977 + 1.0495 * lnTk) * sqrtS - 0.01615 * S;
  k.Kw = exp(tmp1 + tmp2);
  double tmp = 9345.17 / Tk - 60.2409 + 23.3585 * lnTk100;
  k.Kh = exp(tmp + S * (0.023517 - 0.00023656 * Tk + 0.0047036e-4 * Tk * Tk));
  const double pK1 = 3633.86 / Tk - 61.2172 + 9.6777 * lnTk - 0.011555 * S +
                     0.0001152 * S * S;
  k.K1 = exp10(-pK1);
  const double pK2 = 471.78 / Tk + 25.9290 - 3.16967 * lnTk - 0.01781 * S +
                     0.0001122 * S * S;
  k.K2 = exp10(-pK2);
  tmp1 = (-8966.90 - 2890.53 * sqrtS - 77.942 * S + 1.728 * S15 -
          0.0996 * S * S) / Tk;
  tmp2 = +148.0248 + 137.1942 * sqrtS + 1.62142 * S;
  double tmp3 = +(-24.4344 - 25.085 * sqrtS - 0.2474 * S) * lnTk +
                0.053105 * sqrtS * Tk;
  k.Kb = exp(tmp1 + tmp2 + tmp3);
  k.Tr = (0.585 * K0 * rsqrt(Sc) * O_U * O_U);
}

// Carbonate solve for one box: DIC + alk -> [H+] (largest real root of the
// quintic, src/ocean_csys.cpp:289-325) and pCO2 (:328-343).  The quintic has
// exactly one positive root (one sign change: p5,p4 < 0 < p2,p1,p0), so f > 0
// left of it and f < 0 right of it for h > 0; Newton from the previous [H+]
// with a sign-maintained bracket reaches the same root the reference's
// Fujiwara-bound Newton does (both to ~1e-16: measured <= 4e-15 against the oracle over random
// T, DIC, alk).  Stop rule = Boost's (|delta| <= |h| 2^-30, the step is applied first).
__device__ __forceinline__ double chem_solve(const ChemK &k, double carbon,
                                             double inv_vol, double alk,
                                             double &h_io, unsigned &status) {
  const double bor = 1 * (416.0 * (O_S / 35.0)) * 1.e-6;
  const double dic = ((carbon * 1e15) * (1.0 / 12.01) * (1.0 / 1027.0) * inv_vol);
  const double K1 = k.K1, K2 = k.K2, Kb = k.Kb, Kw = k.Kw;
  const double p4 = -alk - Kb - K1;
  const double p3 = dic * K1 - alk * (Kb + K1) + Kb * bor + Kw - Kb * K1 - K1 * K2;
  double tmp = dic * (Kb * K1 + 2.0 * K1 * K2) - alk * (Kb * K1 + K1 * K2) +
               Kb * bor * K1;
  const double p2 = tmp + (Kw * Kb + Kw * K1 - Kb * K1 * K2);
  tmp = 2.0 * dic * Kb * K1 * K2 - alk * Kb * K1 * K2 + Kb * bor * K1 * K2;
  const double p1 = tmp + (Kw * Kb * K1 + Kw * K1 * K2);
  const double p0 = Kw * Kb * K1 * K2;
  double h = h_io;
  double lo = 0.0, hi = 1.0;  // f(lo) > 0 > f(hi)
  const double factor = 0x1p-30;
  bool done = false;
  for (int it = 0; it < 200 && !done; ++it) {
    // Horner, top coefficient first (boost polynomial::evaluate)
    double f = -1.0;
    f = f * h + p4; f = f * h + p3; f = f * h + p2; f = f * h + p1; f = f * h + p0;
    double fp = -5.0;
    fp = fp * h + 4.0 * p4; fp = fp * h + 3.0 * p3; fp = fp * h + 2.0 * p2;
    fp = fp * h + p1;
    if (f == 0.0) { done = true; break; }
    // bracket of the largest root by the sign of f (sound from any start), bisect when Newton
    // leaves it -- except at convergence: there the step is a few ulps (or none: hn == h, which
    // is a bound by now) and its direction is the rounding noise of f; newton_raphson_iterate
    // accepts such a step and stops, so does this.
    if (f > 0) lo = h; else hi = h;
    double delta = f / fp;
    double hn = h - delta;
    if (!(hn > lo && hn < hi)) {
      if (fabs(delta) <= fabs(h) * 0x1p-48) {
        hn = h;
      } else {  // left the bracket (or fp == 0): bisect
        hn = 0.5 * (lo + hi);
        delta = h - hn;
      }
    }
    done = !(fabs(hn * factor) < fabs(delta));
    h = hn;
  }
  if (!done) status |= HX_ERR_ROOT;
  h_io = h;
  const double co2st = dic / (1.0 + K1 / h + K1 * K2 / h / h);
  return co2st * 1e6 / k.Kh;  // PCO2o, uatm
}

// The carbonate solve exactly as the reference iterates it: Fujiwara bound as the start
// (find_largest_root, src/ocean_csys.cpp:134-156) and boost::math::tools::
// newton_raphson_iterate (roots.hpp, Boost >= 1.71) with 31 bits, operation by operation and
// without FMA contraction.  Both this and chem_solve find the root to ~1e-16, but the alkalinity
// tuner (Brent on |flux - target|, keeping "the last point evaluated") branches on comparisons
// of objective values that differ in their last bits, so it gets the reference's own iteration
// and therefore the reference's objective values.  Used ~120 times per member, once per run.
__device__ __forceinline__ double chem_solve_ref(const ChemK &k, double carbon,
                                                           double inv_vol, double alk,
                                                           double &h_out, unsigned &status) {
#pragma clang fp contract(off)
  const double bor = 1 * (416.0 * (O_S / 35.0)) * 1.e-6;
  // convertToDIC returns umol/kg, ocean_csys_run divides by 1e6 again
  const double dic = ((((carbon * 1e15) * (1.0 / 12.01) * (1.0 / 1027.0) * inv_vol)) * 1e6) / 1e6;
  const double K1 = k.K1, K2 = k.K2, Kb = k.Kb, Kw = k.Kw;
  const double a5 = -1.0;
  const double a4 = -alk - Kb - K1;
  const double a3 = dic * K1 - alk * (Kb + K1) + Kb * bor + Kw - Kb * K1 - K1 * K2;
  double tmp = dic * (Kb * K1 + 2.0 * K1 * K2) - alk * (Kb * K1 + K1 * K2) + Kb * bor * K1;
  const double a2 = tmp + (Kw * Kb + Kw * K1 - Kb * K1 * K2);
  tmp = 2.0 * dic * Kb * K1 * K2 - alk * Kb * K1 * K2 + Kb * bor * K1 * K2;
  const double a1 = tmp + (Kw * Kb * K1 + Kw * K1 * K2);
  const double a0 = Kw * Kb * K1 * K2;
  const double d0 = a1 * 1.0, d1 = a2 * 2.0, d2 = a3 * 3.0, d3 = a4 * 4.0, d4 = a5 * 5.0;
  auto f_ = [&](double z) {
    double s = a5;
    s *= z; s += a4; s *= z; s += a3; s *= z; s += a2; s *= z; s += a1; s *= z; s += a0;
    return s;
  };
  auto fp_ = [&](double z) {
    double s = d4;
    s *= z; s += d3; s *= z; s += d2; s *= z; s += d1; s *= z; s += d0;
    return s;
  };
  auto sgn = [](double x) { return (double)((x > 0) - (x < 0)); };
  double mx = pow(fabs(a0 / (2.0 * a5)), 1.0 / 5);
  {
    double m_;
    m_ = pow(fabs(a1 / a5), 1.0 / 4.0); mx = (mx < m_) ? m_ : mx;
    m_ = pow(fabs(a2 / a5), 1.0 / 3.0); mx = (mx < m_) ? m_ : mx;
    m_ = pow(fabs(a3 / a5), 1.0 / 2.0); mx = (mx < m_) ? m_ : mx;
    m_ = pow(fabs(a4 / a5), 1.0 / 1.0); mx = (mx < m_) ? m_ : mx;
  }
  mx *= 2.0;
  double mn = 0.0, guess = mx - 0.001;
  double f0 = 0, f1, last_f0 = 0, result = guess;
  const double factor = 0x1p-30;  // ldexp(1, 1 - 31)
  const double BIG = 1.7976931348623157e308;
  double delta = BIG, delta1 = BIG, delta2 = BIG;
  double max_range_f = 0, min_range_f = 0;
  int count = 100000;
  bool go = true;
  while (go) {
    last_f0 = f0;
    delta2 = delta1;
    delta1 = delta;
    f0 = f_(result);
    f1 = fp_(result);
    --count;
    if (0 == f0) break;
    if (f1 == 0) {
      if (last_f0 == 0) {
        guess = (result == mn) ? mx : mn;
        last_f0 = f_(guess);
        delta = guess - result;
      }
      if (sgn(last_f0) * sgn(f0) < 0) delta = (delta < 0) ? (result - mn) / 2 : (result - mx) / 2;
      else delta = (delta < 0) ? (result - mx) / 2 : (result - mn) / 2;
    } else {
      delta = f0 / f1;
    }
    if (fabs(delta * 2) > fabs(delta2)) {
      const double shift = (delta > 0) ? (result - mn) / 2 : (result - mx) / 2;
      if ((result != 0) && (fabs(shift) > fabs(result))) delta = sgn(delta) * fabs(result) * (double)1.1f;
      else delta = shift;
      delta1 = 3 * delta;
      delta2 = 3 * delta;
    }
    guess = result;
    result -= delta;
    if (result <= mn) {
      delta = 0.5 * (guess - mn);
      result = guess - delta;
      if ((result == mn) || (result == mx)) break;
    } else if (result >= mx) {
      delta = 0.5 * (guess - mx);
      result = guess - delta;
      if ((result == mn) || (result == mx)) break;
    }
    if (delta > 0) { mx = guess; max_range_f = f0; }
    else { mn = guess; min_range_f = f0; }
    if (max_range_f * min_range_f > 0) { status |= HX_ERR_ROOT; result = guess; break; }
    go = count && (fabs(result * factor) < fabs(delta));
  }
  const double h = result;
  h_out = h;
  const double co2st = dic / (1.0 + K1 / h + K1 * K2 / h / h);
  return co2st * 1e6 / k.Kh;
}

// Both surface boxes at once.  Same formulas as chem_constants / chem_solve; the two
// boxes are independent, so writing them side by side gives the single resident
// wavefront two dependency chains to interleave, and the seven divisions by Tk
// share one reciprocal.
__device__ __forceinline__ void chem_constants2(double TcH, double TcL, ChemK &kH, ChemK &kL) {
  const double S = O_S;
  const double sqrtS = 5.873670062235365;      // sqrt(34.5)
  const double S15 = 202.64161714712009;       // 34.5^1.5
  const double Tc[2] = {TcH, TcL};
  const double As[2] = {O_AsHL, O_AsLL};
  ChemK *k[2] = {&kH, &kL};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const double Tk = Tc[b] + 273.15;
    const double rTk = hx_recip(Tk);
    const double T100 = Tk * 0.01;
    const double lnTk = log(Tk);
    const double lnTk100 = lnTk - 4.605170185988092;  // ln(Tk/100)
    double tmp1 = -58.0931 + 9050.69 * rTk + 22.2940 * lnTk100;
    double tmp2 = S * (0.027766 - 0.025888 * T100 + 0.0050578 * (T100 * T100));
    const double K0 = exp(tmp1 + tmp2);
    const double Sc = 2073.1 - (125.62 * Tc[b]) + (3.6276 * Tc[b] * Tc[b]) -
                      (0.043219 * Tc[b] * Tc[b] * Tc[b]);
    tmp1 = -13847.26 * rTk + 148.96502 - 23.6521 * lnTk;
    tmp2 = +(118.67 * rTk - 5.977 + 1.0495 * lnTk) * sqrtS - 0.01615 * S;
    k[b]->Kw = exp(tmp1 + tmp2);
    double tmp = 9345.17 * rTk - 60.2409 + 23.3585 * lnTk100;
    k[b]->Kh = exp(tmp + S * (0.023517 - 0.00023656 * Tk + 0.0047036e-4 * Tk * Tk));
    const double pK1 = 3633.86 * rTk - 61.2172 + 9.6777 * lnTk - 0.011555 * S +
                       0.0001152 * S * S;
    k[b]->K1 = exp10(-pK1);
    const double pK2 = 471.78 * rTk + 25.9290 - 3.16967 * lnTk - 0.01781 * S +
                       0.0001122 * S * S;
    k[b]->K2 = exp10(-pK2);
    tmp1 = (-8966.90 - 2890.53 * sqrtS - 77.942 * S + 1.728 * S15 - 0.0996 * S * S) * rTk;
    tmp2 = +148.0248 + 137.1942 * sqrtS + 1.62142 * S;
    double tmp3 = +(-24.4344 - 25.085 * sqrtS - 0.2474 * S) * lnTk + 0.053105 * sqrtS * Tk;
    k[b]->Kb = exp(tmp1 + tmp2 + tmp3);
    k[b]->Tr = (0.585 * K0 * rsqrt(Sc) * O_U * O_U);
    k[b]->g = k[b]->Tr * As[b] * (12.0 / 1e15);
  }
}

// The run kernel's version of chem_constants2, split around the year's batch of exponentials
// (hx_dev_math.h): chem_exponents() gives the arguments of the six exponentials of one box --
// K0, Kw, Kh, K1, K2, Kb, same formulas as above, 10^-pK as exp(-pK ln 10) -- and
// chem_from_exponentials() builds the box's constants from their values.  (The alkalinity tuner
// keeps chem_constants2: it runs once, and Brent's branch decisions see the last bits.)
// C: the formulas' constants as DATA (HxConst::ctab, hx_fill_chem_table's layout: wide scalar
// loads instead of two s_mov per constant and use), or null for the literals.
#define HXC(i, lit) (C ? C[(i)] : (lit))
__device__ __forceinline__ void chem_exponents(double Tc, double lnTk, double *a,
                                               const double *C = nullptr) {
  const double S = O_S;
  const double sqrtS = 5.873670062235365;      // sqrt(34.5)
  const double S15 = 202.64161714712009;       // 34.5^1.5
  const double LN10 = 2.302585092994045684;
  const double Tk = Tc + HXC(0, 273.15);
  const double rTk = hx_recip(Tk);
  const double T100 = Tk * HXC(1, 0.01);
  const double lnTk100 = lnTk - HXC(2, 4.605170185988092);  // ln(Tk/100)
  a[0] = (HXC(3, -58.0931) + HXC(4, 9050.69) * rTk + HXC(5, 22.2940) * lnTk100) +
         HXC(6, S) * (HXC(7, 0.027766) - HXC(8, 0.025888) * T100 + HXC(9, 0.0050578) * (T100 * T100));
  a[1] = (HXC(10, -13847.26) * rTk + HXC(11, 148.96502) - HXC(12, 23.6521) * lnTk) +
         ((HXC(13, 118.67) * rTk - HXC(14, 5.977) + HXC(15, 1.0495) * lnTk) * HXC(16, sqrtS) - HXC(17, 0.01615 * S));
  // (negated: the carbonate solve divides by Kh, so the batch delivers 1 / Kh = exp(-a))
  a[2] = -((HXC(18, 9345.17) * rTk - HXC(19, 60.2409) + HXC(20, 23.3585) * lnTk100) +
           HXC(6, S) * (HXC(21, 0.023517) - HXC(22, 0.00023656) * Tk + HXC(23, 0.0047036e-4) * Tk * Tk));
  const double pK1 = HXC(24, 3633.86) * rTk - HXC(25, 61.2172) + HXC(26, 9.6777) * lnTk - HXC(27, 0.011555 * S) + HXC(28, 0.0001152 * S * S);
  a[3] = -pK1 * HXC(29, LN10);
  const double pK2 = HXC(30, 471.78) * rTk + HXC(31, 25.9290) - HXC(32, 3.16967) * lnTk - HXC(33, 0.01781 * S) + HXC(34, 0.0001122 * S * S);
  a[4] = -pK2 * HXC(29, LN10);
  const double tmp1 = HXC(35, (-8966.90 - 2890.53 * sqrtS - 77.942 * S + 1.728 * S15 - 0.0996 * S * S)) * rTk;
  const double tmp2 = HXC(36, +148.0248 + 137.1942 * sqrtS + 1.62142 * S);
  const double tmp3 = HXC(37, +(-24.4344 - 25.085 * sqrtS - 0.2474 * S)) * lnTk + HXC(38, 0.053105 * sqrtS) * Tk;
  a[5] = tmp1 + tmp2 + tmp3;
}
#undef HXC
// the table behind chem_exponents' C (same expressions, evaluated by the host compiler in double)
inline void hx_fill_chem_table(double *t) {
  const double S = O_S;
  const double sqrtS = 5.873670062235365, S15 = 202.64161714712009, LN10 = 2.302585092994045684;
  const double v[39] = {273.15, 0.01, 4.605170185988092, -58.0931, 9050.69, 22.2940, S, 0.027766, 0.025888,
                        0.0050578, -13847.26, 148.96502, 23.6521, 118.67, 5.977, 1.0495, sqrtS, 0.01615 * S,
                        9345.17, 60.2409, 23.3585, 0.023517, 0.00023656, 0.0047036e-4, 3633.86, 61.2172, 9.6777,
                        0.011555 * S, 0.0001152 * S * S, LN10, 471.78, 25.9290, 3.16967, 0.01781 * S,
                        0.0001122 * S * S,
                        (-8966.90 - 2890.53 * sqrtS - 77.942 * S + 1.728 * S15 - 0.0996 * S * S),
                        +148.0248 + 137.1942 * sqrtS + 1.62142 * S,
                        +(-24.4344 - 25.085 * sqrtS - 0.2474 * S), 0.053105 * sqrtS};
  for (int i = 0; i < 39; ++i) t[i] = v[i];
  t[39] = 0.0;
}
// The six exponentials of BOTH boxes from the fitted polynomials (HxConst::kfit; hx_chem_fit.inc,
// tools/make_chem_fit.py): a box's temperature stays within a few kelvin of a fixed centre for a
// whole run (SST anomaly + 18 + deltaT), and over centre +- 8 K each of K0, Kw, 1/Kh, K1, K2, Kb
// is ONE degree-13 polynomial in t = (Tc - centre) / 8 K to 3.4e-16 relative (truncation < 2e-17;
// the rest is Horner's rounding, like the ~1 ulp of the exponential it replaces) -- 13
// multiply-adds instead of an argument polynomial, a logarithm and a 23-instruction exponential:
// twelve independent chains, coefficient row after coefficient row through scalar loads.
// e[0..5] = HL's, e[6..11] = LL's, in chem_exponents' order.  The caller checks the interval
// (chem_fit_applies) for the WHOLE wavefront and takes the formulas themselves otherwise.
__device__ __forceinline__ bool chem_fit_applies(double TcH, double TcL) {
  // (written so that a NaN temperature is outside)
  return fabs(TcH - HX_CHEM_FIT_CENTRE_HL) <= HX_CHEM_FIT_HALF_WIDTH &&
         fabs(TcL - HX_CHEM_FIT_CENTRE_LL) <= HX_CHEM_FIT_HALF_WIDTH;
}
__device__ __forceinline__ void chem_constants_fit(double TcH, double TcL, const double *F, double *e) {
  const double t[2] = {(TcH - HX_CHEM_FIT_CENTRE_HL) * (1.0 / HX_CHEM_FIT_HALF_WIDTH),
                       (TcL - HX_CHEM_FIT_CENTRE_LL) * (1.0 / HX_CHEM_FIT_HALF_WIDTH)};
  double p[12];
#ifndef HX_HOST_EMULATION
  // Row after row, twelve chains side by side: the next row's coefficients are requested
  // (scalar loads) while this row's multiply-adds issue, and a scheduling barrier keeps the
  // optimiser from asking for all 168 at once (336 scalar registers: it spills 90 of them).
  // v_fma_f64 takes the coefficient from its SGPR pair as the addend (left to the compiler a
  // coefficient that is used once becomes two v_mov into the destination of a v_fmac_f64: three
  // vector instructions per multiply-add).  The wait for a row sits BEHIND the multiply-adds of the
  // row before it (HX_ROW_ARRIVED: scalar loads return out of order, so a wait is for all of them,
  // and placed lazily -- in front of the row's first use -- it would also wait out the row
  // requested just before it).  (HX_CHEM_FIT_GROUP 6: one box at a time, half the scalar registers;
  // measured slower, profiles/r04_variant_log.md.)
#ifndef HX_CHEM_FIT_GROUP
#define HX_CHEM_FIT_GROUP 12
#endif
  constexpr int G = HX_CHEM_FIT_GROUP;
#if HX_CHEM_FIT_GROUP == 6
#define HX_ROW_ARRIVED(r) asm volatile("" :: "s"(r[0]), "s"(r[1]), "s"(r[2]), "s"(r[3]), "s"(r[4]), "s"(r[5]))
#else
#define HX_ROW_ARRIVED(r) asm volatile("" :: "s"(r[0]), "s"(r[1]), "s"(r[2]), "s"(r[3]), "s"(r[4]), "s"(r[5]), \
                                       "s"(r[6]), "s"(r[7]), "s"(r[8]), "s"(r[9]), "s"(r[10]), "s"(r[11]))
#endif
#pragma unroll
  for (int g0 = 0; g0 < 12; g0 += G) {
    double cur[G], nxt[G];
#pragma unroll
    for (int f = 0; f < G; ++f) { cur[f] = F[g0 + f]; nxt[f] = F[12 + g0 + f]; }
    HX_ROW_ARRIVED(cur);
#pragma unroll
    for (int f = 0; f < G; ++f) asm("v_mov_b64 %0, %1" : "=v"(p[g0 + f]) : "s"(cur[f]));   // (row 0: the leading coefficients)
    HX_ROW_ARRIVED(nxt);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < G; ++f) cur[f] = nxt[f];
#pragma unroll
    for (int j = 1; j <= HX_CHEM_FIT_DEGREE; ++j) {
      if (j < HX_CHEM_FIT_DEGREE) {
#pragma unroll
        for (int f = 0; f < G; ++f) nxt[f] = F[(j + 1) * 12 + g0 + f];
      }
#pragma unroll
      for (int f = 0; f < G; ++f)
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p[g0 + f]) : "v"(p[g0 + f]), "v"(t[(g0 + f) / 6]), "s"(cur[f]));
      if (j < HX_CHEM_FIT_DEGREE) HX_ROW_ARRIVED(nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < G; ++f) cur[f] = nxt[f];
    }
  }
#undef HX_ROW_ARRIVED
#else
#pragma unroll
  for (int f = 0; f < 12; ++f) p[f] = F[f];
#pragma unroll
  for (int j = 1; j <= HX_CHEM_FIT_DEGREE; ++j) {
#pragma unroll
    for (int f = 0; f < 12; ++f) p[f] = fma(p[f], t[f / 6], F[j * 12 + f]);
  }
#endif
#pragma unroll
  for (int f = 0; f < 12; ++f) e[f] = p[f];
}
__device__ __forceinline__ void chem_from_exponentials(double Tc, const double *e, double As,
                                                       ChemK &k) {
  const double Sc = 2073.1 - (125.62 * Tc) + (3.6276 * Tc * Tc) - (0.043219 * Tc * Tc * Tc);
  k.Kw = e[1]; k.rKh = e[2]; k.K1 = e[3]; k.K2 = e[4]; k.Kb = e[5];
  k.Tr = (0.585 * e[0] * rsqrt(Sc) * O_U * O_U);
  k.g = k.Tr * As * (12.0 / 1e15);
}
// ocean_csys.cpp:289-330 (the coefficients of the polynomial in [H+]) sorted by powers of DIC
__device__ __forceinline__ void chem_poly_constants(double alk, ChemK &k) {
  const double bor = 1 * (416.0 * (O_S / 35.0)) * 1.e-6;
  const double K1 = k.K1, K2 = k.K2, Kb = k.Kb, Kw = k.Kw;
  const double K1K2 = K1 * K2, KbK1 = Kb * K1, Kbbor = Kb * bor;
  k.K1K2 = K1K2;
  k.C4 = -alk - Kb - K1;
  k.C3 = (((-alk * (Kb + K1) + Kbbor) + Kw) - KbK1) - K1K2;
  k.A2 = KbK1 + 2.0 * K1K2;
  k.C2 = (-alk * (KbK1 + K1K2) + Kbbor * K1) + ((Kw * Kb + Kw * K1) - KbK1 * K2);
  k.A1 = 2.0 * KbK1 * K2;
  k.C1 = (-alk * KbK1 * K2 + Kbbor * K1K2) + (Kw * KbK1 + Kw * K1K2);
  k.C0 = Kw * KbK1 * K2;
}

// The yearly / per-stash solve of both boxes.  From a warm start (the previous [H+], a fraction
// of a percent away) the safeguards of the bracketed iteration never engage: the iterates approach the
// root from its right, where the quintic is concave, and stay inside the bracket -- so this is
// the plain Newton iteration with the same stop rule (Boost's: |delta| <= |h| 2^-30, the step
// applied first), i.e. the same iterates, without the bracket bookkeeping that made up half of
// the loop body (the two solves are ~17 % of a model year).  A lane whose iteration does not
// settle in 8 steps on a positive finite root (a cold or pathological start) redoes the solve
// from its original start with the safeguarded iteration (the cold branch below).
__device__ __forceinline__ void chem_solve2(const ChemK &kH, const ChemK &kL, double cH,
                                            double cL, double alkH, double alkL, double &hH,
                                            double &hL, double &pco2H, double &pco2L,
                                            unsigned &status) {
  const ChemK *k[2] = {&kH, &kL};
  const double carbon[2] = {cH, cL};
  (void)alkH; (void)alkL;   // (inside the year's polynomial constants)
  const double inv_vol[2] = {1.0 / O_vHL, 1.0 / O_vLL};
  double dic[2], p4[2], p3[2], p2[2], p1[2], p0[2], h[2] = {hH, hL};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    dic[b] = ((carbon[b] * 1e15) * (1.0 / 12.01) * (1.0 / 1027.0) * inv_vol[b]);
    p4[b] = k[b]->C4;
    p3[b] = fma(dic[b], k[b]->K1, k[b]->C3);
    p2[b] = fma(dic[b], k[b]->A2, k[b]->C2);
    p1[b] = fma(dic[b], k[b]->A1, k[b]->C1);
    p0[b] = k[b]->C0;
  }
  const double factor = 0x1p-30;
  const double q4[2] = {4.0 * p4[0], 4.0 * p4[1]}, q3[2] = {3.0 * p3[0], 3.0 * p3[1]},
               q2[2] = {2.0 * p2[0], 2.0 * p2[1]};
  // A lane stops moving the moment one of its steps meets the stop rule -- its result must not
  // depend on how long its 63 neighbours take (results are independent of the lane assignment,
  // bit for bit).  The freeze is arithmetic, not a branch or a pair of selects per value: every
  // step is scaled by act = 1.0 while the box iterates, 0.0 afterwards (h - 0 * delta = h
  // exactly), so both boxes' chains stay interleaved and an iteration is ~40 instructions.
  double act[2] = {1.0, 1.0};
#ifndef HX_NO_CHEM_BLIND
  // Two plain Newton steps first, without the stop test, the freeze and the wavefront's vote: from
  // a warm start a fraction of a percent away the errors go 1e-3 -> 1e-6 -> 1e-12, so no step
  // before the third can meet the stop rule (|step| <= |h| 2^-30) -- unless the start already
  // was the root, and then the steps are zero.  Every lane takes exactly these two steps whatever
  // its neighbours hold, so results stay independent of the lane assignment.  The steps divide by
  // the raw v_rcp_f64 (4.6e-8): a Newton correction's relative error is multiplied by the
  // correction's size, which the next iteration squares away -- here and in the tested iterations
  // below, whose last step is ~1e-11 of [H+].  24 instructions a step instead of 49.
  // (-DHX_NO_CHEM_BLIND: every iteration tested, divisions to 2e-15: the round-4 form.)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    HX_COUNT(0, 18);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const double x = h[b];
      double f = -1.0;
      f = f * x + p4[b]; f = f * x + p3[b]; f = f * x + p2[b]; f = f * x + p1[b];
      f = f * x + p0[b];
      double fp = -5.0;
      fp = fp * x + q4[b]; fp = fp * x + q3[b]; fp = fp * x + q2[b];
      fp = fp * x + p1[b];
      h[b] = x - f * HX_RCP(fp);
    }
  }
#define HX_CHEM_NEWTON_DIV(f, fp) ((f) * HX_RCP(fp))
#else
#define HX_CHEM_NEWTON_DIV(f, fp) hx_div1((f), (fp))
#endif
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    HX_COUNT(0, 18);  // (profiling build) Newton iterations
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const double x = h[b];
      double f = -1.0;
      f = f * x + p4[b]; f = f * x + p3[b]; f = f * x + p2[b]; f = f * x + p1[b];
      f = f * x + p0[b];
      double fp = -5.0;
      fp = fp * x + q4[b]; fp = fp * x + q3[b]; fp = fp * x + q2[b];
      fp = fp * x + p1[b];
      const double delta = HX_CHEM_NEWTON_DIV(f, fp) * act[b];
      const double hn = x - delta;
      h[b] = hn;
      // (a frozen box: delta = 0, the test holds again)
      act[b] = (fabs(hn * factor) < fabs(delta)) ? act[b] : 0.0;
    }
    if (!__any((act[0] + act[1]) != 0.0)) break;
  }
  const bool conv = (act[0] + act[1]) == 0.0;
  const bool ok = conv && h[0] > 0.0 && h[1] > 0.0 && h[0] < 1.0 && h[1] < 1.0;
  if (__any(!ok)) {
    HX_COUNT(0, 19);  // (profiling build) safeguarded restarts
    if (!ok) {
      // safeguarded restart from the original [H+]: Newton inside a sign-maintained bracket of
      // the largest root, bisection when a step leaves it (see chem_solve); sound from any start
      h[0] = hH; h[1] = hL;
      double lo[2] = {0.0, 0.0}, hi[2] = {1.0, 1.0};
      bool done[2] = {false, false};
      for (int it = 0; it < 200 && !(done[0] && done[1]); ++it) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const double x = h[b];
          double f = -1.0;
          f = f * x + p4[b]; f = f * x + p3[b]; f = f * x + p2[b]; f = f * x + p1[b];
          f = f * x + p0[b];
          double fp = -5.0;
          fp = fp * x + q4[b]; fp = fp * x + q3[b]; fp = fp * x + q2[b];
          fp = fp * x + p1[b];
          if (!done[b]) {
            if (f == 0.0) {
              done[b] = true;
            } else {
              if (f > 0) lo[b] = x; else hi[b] = x;
              double delta = hx_div1(f, fp);
              double hn = x - delta;
              if (!(hn > lo[b] && hn < hi[b])) {
                if (fabs(delta) <= fabs(x) * 0x1p-48) {  // converged, see chem_solve
                  hn = x;
                } else {  // left the bracket (or fp == 0): bisect
                  hn = 0.5 * (lo[b] + hi[b]);
                  delta = x - hn;
                }
              }
              done[b] = !(fabs(hn * factor) < fabs(delta));
              h[b] = hn;
            }
          }
        }
      }
      if (!(done[0] && done[1])) status |= HX_ERR_ROOT;
    }
  }
  hH = h[0]; hL = h[1];
  double pc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    // co2* = dic / (1 + K1/h + K1 K2/h^2), one division
    const double K1 = k[b]->K1, x = h[b];
    const double co2st = hx_div(dic[b] * (x * x), (x * x + K1 * x) + k[b]->K1K2);
    pc[b] = (co2st * 1e6) * k[b]->rKh;
  }
  pco2H = pc[0]; pco2L = pc[1];
}

// The same iteration for a list of boxes (the small-ensemble kernel gives each of its two
// wavefronts one box).
template <int N>
__device__ __forceinline__ void chem_solve_boxes(const ChemK *const (&k)[N], const double (&carbon)[N],
                                                 const double (&alk)[N], const double (&inv_vol)[N],
                                                 double (&h)[N], double (&pc)[N], unsigned &status) {
  double dic[N], p4[N], p3[N], p2[N], p1[N], p0[N], h0[N];
  (void)alk;   // (inside the year's polynomial constants)
#pragma unroll
  for (int b = 0; b < N; ++b) {
    h0[b] = h[b];
    dic[b] = ((carbon[b] * 1e15) * (1.0 / 12.01) * (1.0 / 1027.0) * inv_vol[b]);
    p4[b] = k[b]->C4;
    p3[b] = fma(dic[b], k[b]->K1, k[b]->C3);
    p2[b] = fma(dic[b], k[b]->A2, k[b]->C2);
    p1[b] = fma(dic[b], k[b]->A1, k[b]->C1);
    p0[b] = k[b]->C0;
  }
  const double factor = 0x1p-30;
  double q4[N], q3[N], q2[N];
  bool conv[N];
#pragma unroll
  for (int b = 0; b < N; ++b) { q4[b] = 4.0 * p4[b]; q3[b] = 3.0 * p3[b]; q2[b] = 2.0 * p2[b]; conv[b] = false; }
  auto all = [&](const bool (&f)[N]) { bool a = true;
#pragma unroll
    for (int b = 0; b < N; ++b) a = a && f[b];
    return a; };
  // (a box stops moving once a step met the stop rule, through act = 1.0 / 0.0: see chem_solve2)
  double act[N];
#pragma unroll
  for (int b = 0; b < N; ++b) act[b] = 1.0;
#ifndef HX_NO_CHEM_BLIND
  // (two untested steps on the raw reciprocal first: see chem_solve2)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    HX_COUNT(0, 18);
#pragma unroll
    for (int b = 0; b < N; ++b) {
      const double x = h[b];
      double f = -1.0;
      f = f * x + p4[b]; f = f * x + p3[b]; f = f * x + p2[b]; f = f * x + p1[b];
      f = f * x + p0[b];
      double fp = -5.0;
      fp = fp * x + q4[b]; fp = fp * x + q3[b]; fp = fp * x + q2[b];
      fp = fp * x + p1[b];
      h[b] = x - f * HX_RCP(fp);
    }
  }
#endif
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    HX_COUNT(0, 18);  // (profiling build) Newton iterations
    double left = 0.0;
#pragma unroll
    for (int b = 0; b < N; ++b) {
      const double x = h[b];
      double f = -1.0;
      f = f * x + p4[b]; f = f * x + p3[b]; f = f * x + p2[b]; f = f * x + p1[b];
      f = f * x + p0[b];
      double fp = -5.0;
      fp = fp * x + q4[b]; fp = fp * x + q3[b]; fp = fp * x + q2[b];
      fp = fp * x + p1[b];
      const double delta = HX_CHEM_NEWTON_DIV(f, fp) * act[b];
      const double hn = x - delta;
      h[b] = hn;
      act[b] = (fabs(hn * factor) < fabs(delta)) ? act[b] : 0.0;
      left += act[b];
    }
    if (!__any(left != 0.0)) break;
  }
#pragma unroll
  for (int b = 0; b < N; ++b) conv[b] = act[b] == 0.0;
  bool ok = all(conv);
#pragma unroll
  for (int b = 0; b < N; ++b) ok = ok && h[b] > 0.0 && h[b] < 1.0;
  if (__any(!ok)) {
    HX_COUNT(0, 19);  // (profiling build) safeguarded restarts
    // Safeguarded restart from the original [H+]: Newton inside a sign-maintained bracket of the
    // largest root, bisection when a step leaves it (see chem_solve); sound from any start.
    // EVERY lane runs it, under a wavefront-uniform loop, and the lanes that had converged keep
    // their result through selects: no divergent region.  (With `if (!ok) { ... }` ROCm 7.2's
    // register allocator parked live values in AGPRs at the top of the block where the branches
    // rejoin, BEFORE the execution mask was restored -- with no lane in the branch nothing was
    // saved and every lane read stale registers afterwards; tools/check_isa.py looks for that.)
    double hc[N], lo[N], hi[N];
    bool done[N];
#pragma unroll
    for (int b = 0; b < N; ++b) { hc[b] = h0[b]; lo[b] = 0.0; hi[b] = 1.0; done[b] = ok; }
    for (int it = 0; it < 200 && __any(!all(done)); ++it) {
#pragma unroll
      for (int b = 0; b < N; ++b) {
        const double x = hc[b];
        double f = -1.0;
        f = f * x + p4[b]; f = f * x + p3[b]; f = f * x + p2[b]; f = f * x + p1[b];
        f = f * x + p0[b];
        double fp = -5.0;
        fp = fp * x + q4[b]; fp = fp * x + q3[b]; fp = fp * x + q2[b];
        fp = fp * x + p1[b];
        const bool zero = f == 0.0;
        lo[b] = (f > 0) ? x : lo[b];
        hi[b] = (f > 0) ? hi[b] : x;
        const double delta = hx_div1(f, fp);
        const double hn = x - delta;
        const bool inside = hn > lo[b] && hn < hi[b];
        const bool tiny = fabs(delta) <= fabs(x) * 0x1p-48;  // converged, see chem_solve
        const double mid = 0.5 * (lo[b] + hi[b]);            // left the bracket (or fp == 0): bisect
        const double hn2 = inside ? hn : (tiny ? x : mid);
        const double d2 = (inside || tiny) ? delta : x - mid;
        const bool fin = zero || !(fabs(hn2 * factor) < fabs(d2));
        hc[b] = (done[b] || zero) ? hc[b] : hn2;
        done[b] = done[b] || fin;
      }
    }
    status |= all(done) ? 0u : HX_ERR_ROOT;
#pragma unroll
    for (int b = 0; b < N; ++b) h[b] = ok ? h[b] : hc[b];
  }
#pragma unroll
  for (int b = 0; b < N; ++b) {
    // co2* = dic / (1 + K1/h + K1 K2/h^2), one division
    const double K1 = k[b]->K1, x = h[b];
    const double co2st = hx_div(dic[b] * (x * x), (x * x + K1 * x) + k[b]->K1K2);
    pc[b] = (co2st * 1e6) * k[b]->rKh;
  }
}
// chem_solve2's interface on the select-based restart of chem_solve_boxes (no divergent region:
// -DHX_CHEM_SELECT routes the run kernels' solves here, experiment builds -- round 3 measured
// +16 % on the one-wavefront kernel (70 spilled SGPRs), round 4 re-measured it on the
// two-wavefront flavour: profiles/r04_variant_log.md)
__device__ __forceinline__ void chem_solve2_select(const ChemK &kH, const ChemK &kL, double cH, double cL,
                                                   double alkH, double alkL, double &hH, double &hL,
                                                   double &pco2H, double &pco2L, unsigned &status) {
  const ChemK *const k[2] = {&kH, &kL};
  const double carbon[2] = {cH, cL}, alk[2] = {alkH, alkL}, inv_vol[2] = {1.0 / O_vHL, 1.0 / O_vLL};
  double h[2] = {hH, hL}, pc[2];
  chem_solve_boxes<2>(k, carbon, alk, inv_vol, h, pc, status);
  hH = h[0]; hL = h[1]; pco2H = pc[0]; pco2L = pc[1];
}
#ifdef HX_CHEM_SELECT
#define HX_CHEM_SOLVE2 chem_solve2_select
#else
#define HX_CHEM_SOLVE2 chem_solve2
#endif
// one box (the small-ensemble kernel gives each of its two wavefronts one)
__device__ __forceinline__ void chem_solve1(const ChemK &kb, double carbon_, double alk_,
                                            double inv_vol_, double &h_, double &pco2,
                                            unsigned &status) {
  const ChemK *const k[1] = {&kb};
  const double carbon[1] = {carbon_}, alk[1] = {alk_}, inv_vol[1] = {inv_vol_};
  double h[1] = {h_}, pc[1];
  chem_solve_boxes<1>(k, carbon, alk, inv_vol, h, pc, status);
  h_ = h[0];
  pco2 = pc[0];
}

// calc_annual_surface_flux  src/ocean_csys.cpp:375-396
__device__ __forceinline__ double surf_flux(double co2, double pco2, double scale,
                                            double Tr, double As) {
  return (((co2 - pco2 * scale) * Tr) * As * 12.0) / 1e15;
}

// oceanbox::chem_equilibrate: tune alkalinity so that the chemistry reproduces
// the spinup flux at CO2 = co2 (src/oceanbox.cpp:382-445).  Boost's
// brent_find_minima restated; the alkalinity kept is the LAST point evaluated.
__device__ __forceinline__ double equilibrate_alk(const ChemK &k, double carbon,
                                                  double inv_vol, double As,
                                                  double co2, double f_target,
                                                  double &h, unsigned &status) {
  // Brent's branch decisions hinge on differences of nearly equal numbers (it is
  // minimising a V-shaped |flux - target|); its resolution here is only
  // tol/4 = 7.5e-9 absolute = 3e-6 of the alkalinity, so a different path ends
  // 1e-6..1e-5 away and moves CO2 by up to ~3e-6 relative.  Keep the arithmetic of
  // the decision logic exactly the reference's: no FMA contraction in here.
#pragma clang fp contract(off)
  auto fmin_ = [&](double alk) {
    const double p = chem_solve_ref(k, carbon, inv_vol, alk, h, status);
    return fabs(surf_flux(co2, p, 1.0, k.Tr, As) - f_target);
  };
  const double tolerance = 0x1p-25;  // bits = min(53/2, 31) = 26
  double mn = 2100e-6, mx = 2750e-6;
  double x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
  const double golden = 0.3819660f;
  x = w = v = mx;
  fw = fv = fx = fmin_(x);
  delta2 = delta = 0;
  u = x;
  for (int count = 0; count < 1000; ++count) {
    mid = (mn + mx) / 2;
    fract1 = tolerance * fabs(x) + tolerance / 4;
    fract2 = 2 * fract1;
    if (fabs(x - mid) <= (fract2 - (mx - mn) / 2)) break;
    if (fabs(delta2) > fract1) {
      double r = (x - w) * (fx - fv);
      double q = (x - v) * (fx - fw);
      double p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = fabs(q);
      const double td = delta2;
      delta2 = delta;
      if ((fabs(p) >= fabs(q * td / 2)) || (p <= q * (mn - x)) || (p >= q * (mx - x))) {
        delta2 = (x >= mid) ? mn - x : mx - x;
        delta = golden * delta2;
      } else {
        delta = p / q;
        u = x + delta;
        if (((u - mn) < fract2) || ((mx - u) < fract2))
          delta = (mid - x) < 0 ? -fabs(fract1) : fabs(fract1);
      }
    } else {
      delta2 = (x >= mid) ? mn - x : mx - x;
      delta = golden * delta2;
    }
    u = (fabs(delta) >= fract1) ? (x + delta)
                                : (delta > 0 ? (x + fabs(fract1)) : (x - fabs(fract1)));
    fu = fmin_(u);
    if (fu <= fx) {
      if (u >= x) mn = x; else mx = x;
      v = w; w = x; x = u; fv = fw; fw = fx; fx = fu;
    } else {
      if (u < x) mn = u; else mx = u;
      if ((fu <= fw) || (w == x)) { v = w; w = u; fv = fw; fw = fu; }
      else if ((fu <= fv) || (v == x) || (v == w)) { v = u; fv = fu; }
    }
  }
  return u;
}

}  // namespace
