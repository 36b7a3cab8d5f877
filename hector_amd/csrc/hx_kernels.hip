// hx_kernels.hip -- CDNA4 (gfx950) kernels for the Hector ensemble year loop.
//
// One ensemble member per lane, 64-lane workgroups (one wavefront each), all
// per-member state register-resident for the whole multi-year launch; HBM is
// touched only for (a) the coalesced SoA parameter/state rows at entry/exit,
// (b) one coalesced row per output variable per year and (c) DOECLIM's SST
// history, which is re-read ONCE PER BLOCK of HX_DBLK years instead of once per
// year (block-causal evaluation of the same ascending sum, partials parked in
// LDS).  Shared scenario series are wave-uniform and arrive through scalar
// loads.  No MFMA: this is elementwise fp64 ODE stepping.
//
// What each device function restates (reference file:line, /root/reference):
//   year loop / component order        src/core.cpp:483-504
//   OH, CH4, O3                         src/oh_component.cpp:137-178,
//                                       src/ch4_component.cpp:152-199,
//                                       src/o3_component.cpp:126-146
//   ocean year start / stash / RHS      src/ocean_component.cpp:356-407,653-763,603-626
//   box exchange                        src/oceanbox.cpp:203-323
//   carbonate chemistry                 src/ocean_csys.cpp:166-396
//   alkalinity tuning (Brent)           src/oceanbox.cpp:382-445 + Boost minima.hpp
//   land RHS / slow params / stash      src/simpleNbox-runtime.cpp:781-934,945-1072,270-609
//   dopri5 + controller + retry logic   src/carbon-cycle-solver.cpp:222-303 + odeint
//   forcing                             src/forcing_component.cpp:300-532
//   DOECLIM                             src/temperature_component.cpp:196-557
//
// Deliberate, tolerance-neutral departures from the reference's arithmetic
// (all <= a few ulp, see DESIGN.md "numerics"): FMA contraction on; quintic
// root by warm-started safeguarded Newton (same root, different path);
// T-only equilibrium constants computed once per year per box; LUC ratio via
// one division; 200-year Q10 window as a running sum; forcing summed in groups.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <hx_addrspace.h>

#include "hx_layout.h"

#define HX_DBLK 32  // DOECLIM block length = years per run-kernel launch
#define HX_DJT 8    // block years per thread in the history-pass kernel
#define HX_KPAD 32  // zero entries in front of / behind the Ker table

namespace {

constexpr double PGC2PPM = 1.0 / 2.13;  // carbon-cycle-model.hpp:29
constexpr double PG_C_TO_TG_CH4 = 1000.0 * 16.04 / 12.01;

// ---- DOECLIM constants  inst/include/temperature_component.hpp:77-98 -------
constexpr double D_ak = 0.31, D_bk = 1.59, D_csw = 0.13, D_earth_area = 5100656E8,
                 D_secs = 60.0 * 60.0 * 24.0 * 365.2422, D_rlam = 1.43,
                 D_zbot = 4000.0, D_bsi = 1.3, D_cal = 0.52, D_cas = 7.80,
                 D_flnd = 0.29, D_fso = 0.95;

// ---- ocean geometry  src/ocean_component.cpp:202-303 -----------------------
constexpr double O_part_high = 0.15, O_part_low = 1 - 0.15;
constexpr double O_spy = 60.0 * 60 * 24 * 365.25;
constexpr double O_area = 3.6e14;
constexpr double O_vLL = O_area * O_part_low * 100.0;
constexpr double O_vHL = O_area * O_part_high * 100.0;
constexpr double O_vI = O_area * 900.0;
constexpr double O_vD = O_area * (3777.0 - 900.0 - 100.0);
constexpr double O_AsHL = O_area * O_part_high, O_AsLL = O_area * O_part_low;
constexpr double O_S = 34.5, O_U = 6.7;

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
// 2e-15: enough for a Newton correction, whose own error is squared away by the next iteration
__device__ __forceinline__ double hx_div1(double a, double b) {
  double r = HX_RCP(b);
  r = fma(fma(-b, r, 1.0), r, r);
  return a * r;
}

struct ChemK {  // T-dependent equilibrium constants of one surface box
  double K1, K2, Kb, Kw, Kh, Tr;
  double g;  // Tr * As * 12 / 1e15: annual flux per uatm of air-sea pCO2 difference
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
// Fujiwara-bound Newton does.  Stop rule = Boost's (|delta| <= |h| 2^-30).
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
    if (f > 0) lo = h; else hi = h;
    double delta = f / fp;
    double hn = h - delta;
    if (!(hn > lo && hn < hi)) {  // left the bracket (or fp == 0): bisect
      hn = 0.5 * (lo + hi);
      delta = h - hn;
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
// newton_raphson_iterate (roots.hpp, Boost >= 1.71) with 31 bits, whose last step may be a
// bracket-halving one -- its root is only good to ~1e-9 relative, and WHICH 1e-9 depends on the
// iteration path.  The year-by-year solves do not care (chem_solve converges to the exact root),
// but the alkalinity tuner compares objective values that differ by less than that, so it gets
// the reference's own iteration.  Used ~120 times per member, once per run.
__device__ __attribute__((noinline)) double chem_solve_ref(const ChemK &k, double carbon,
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

__device__ __forceinline__ void chem_solve2(const ChemK &kH, const ChemK &kL, double cH,
                                            double cL, double alkH, double alkL, double &hH,
                                            double &hL, double &pco2H, double &pco2L,
                                            unsigned &status) {
  const double bor = 1 * (416.0 * (O_S / 35.0)) * 1.e-6;
  const ChemK *k[2] = {&kH, &kL};
  const double carbon[2] = {cH, cL}, alk[2] = {alkH, alkL};
  const double inv_vol[2] = {1.0 / O_vHL, 1.0 / O_vLL};
  double dic[2], p4[2], p3[2], p2[2], p1[2], p0[2], h[2] = {hH, hL};
  double lo[2] = {0.0, 0.0}, hi[2] = {1.0, 1.0};
  bool done[2] = {false, false};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const double K1 = k[b]->K1, K2 = k[b]->K2, Kb = k[b]->Kb, Kw = k[b]->Kw;
    dic[b] = ((carbon[b] * 1e15) * (1.0 / 12.01) * (1.0 / 1027.0) * inv_vol[b]);
    p4[b] = -alk[b] - Kb - K1;
    p3[b] = dic[b] * K1 - alk[b] * (Kb + K1) + Kb * bor + Kw - Kb * K1 - K1 * K2;
    double tmp = dic[b] * (Kb * K1 + 2.0 * K1 * K2) - alk[b] * (Kb * K1 + K1 * K2) +
                 Kb * bor * K1;
    p2[b] = tmp + (Kw * Kb + Kw * K1 - Kb * K1 * K2);
    tmp = 2.0 * dic[b] * Kb * K1 * K2 - alk[b] * Kb * K1 * K2 + Kb * bor * K1 * K2;
    p1[b] = tmp + (Kw * Kb * K1 + Kw * K1 * K2);
    p0[b] = Kw * Kb * K1 * K2;
  }
  const double factor = 0x1p-30;
  const double q4[2] = {4.0 * p4[0], 4.0 * p4[1]}, q3[2] = {3.0 * p3[0], 3.0 * p3[1]},
               q2[2] = {2.0 * p2[0], 2.0 * p2[1]};
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
          if (!(hn > lo[b] && hn < hi[b])) {  // left the bracket (or fp == 0): bisect
            hn = 0.5 * (lo[b] + hi[b]);
            delta = x - hn;
          }
          done[b] = !(fabs(hn * factor) < fabs(delta));
          h[b] = hn;
        }
      }
    }
  }
  if (!(done[0] && done[1])) status |= HX_ERR_ROOT;
  hH = h[0]; hL = h[1];
  double pc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    // co2* = dic / (1 + K1/h + K1 K2/h^2), one division
    const double K1 = k[b]->K1, K2 = k[b]->K2, x = h[b];
    const double co2st = hx_div(dic[b] * (x * x), (x * x + K1 * x) + K1 * K2);
    pc[b] = hx_div(co2st * 1e6, k[b]->Kh);
  }
  pco2H = pc[0]; pco2L = pc[1];
}

// calc_annual_surface_flux  src/ocean_csys.cpp:375-396
__device__ __forceinline__ double surf_flux(double co2, double pco2, double scale,
                                            double Tr, double As) {
  return (((co2 - pco2 * scale) * Tr) * As * 12.0) / 1e15;
}

// ---------------------------------------------------------------------------
// A compiler-only fence: values cached from memory may not be carried across it.
// The year loop is split into phases by these so that per-member constants that a
// phase needs are (re)loaded from HBM / L2 inside the phase instead of being kept
// in registers across the whole solver -- register pressure, not bandwidth, is what
// limits this kernel (DESIGN.md "registers").
#define HX_FENCE() asm volatile("" ::: "memory")

// Per-lane LDS scratchpad ("park"): year-level state and the constants that the
// phases and the stash block need a few times per year.  One wavefront per SIMD
// means every HBM/L2 access is an exposed ~1-2k-cycle stall; LDS answers in ~64.
// Filled from the HBM tables at kernel entry, state slots written back at exit.
enum HxPark {
  PK_CH4 = 0, PK_SST, PK_EOS, PK_TLAND, PK_TWIN, PK_TL_M1, PK_TL_M2, PK_F_PREV,
  PK_BASE_TOT, PK_BASE_CO2,            // <- year-level state (HBM state rows)
  PK_LN_CH4, PK_LN_CO2R,               // ln CH4 and ln(CO2/C0) of the year just finished: the
                                       // next year needs exactly these logarithms again
  PK_AERO, PK_VOL,
  PK_D0,                               // 14 DOECLIM constants HXD_A0..HXD_HFSCALE
  PK_K0 = PK_D0 + 14,                  // 7 ocean exchange coefficients HXD_KLH..HXD_KDI
  PK_FFROZEN0 = PK_K0 + 7,             // f_frozen per biome
};
// single-biome kernels also park the biome constants (11 more slots)
enum HxParkB1 { PKB_NPP0 = 0, PKB_F_NPPV, PKB_F_NPPD, PKB_F_LITTERD, PKB_RH_CH4_FRAC,
                PKB_FPF_STATIC, PKB_BETA, PKB_WF, PKB_LNQ10, PKB_MU, PKB_SIGMA, PKB_N };
// multi-biome kernels park the nine per-biome arrays of Member instead (see BiomeArr)
constexpr int HX_NBIOME_ARR = 9;
template <int B> constexpr int hx_npark() {
  return PK_FFROZEN0 + B + (B == 1 ? (int)PKB_N : HX_NBIOME_ARR * B);
}
template <int B> constexpr int hx_pkb1() { return PK_FFROZEN0 + B; }

// Per-biome arrays of a member.  One biome: plain registers.  More: the LDS park -- kept in
// registers, 36 doubles for B = 4 on top of the solver's working set overflow 256 VGPRs +
// 256 AGPRs and ~100 VGPRs spill to scratch, each reload an exposed memory stall; the solver
// steps themselves only touch the interval constants, not these arrays.
struct ParkArr {
  double (*base)[64];
  int lane;
  __device__ __forceinline__ double &operator[](int b) const { return base[b][lane]; }
};
struct RegArr1 {
  double v[1];
  __device__ __forceinline__ double &operator[](int b) { return v[b]; }
  __device__ __forceinline__ const double &operator[](int b) const { return v[b]; }
};
template <int B> struct BiomeArr { using type = ParkArr; };
template <> struct BiomeArr<1> { using type = RegArr1; };

// What stays in registers through the carbon-cycle solver of one year.
template <int B>
struct Member {
  double C0;
  // state
  double cHL, cLL, cIO, cDO, atmos, earth;
  typename BiomeArr<B>::type veg, det, soil, pf, thawed, tempferts;
  double cum_luc_va, cum_pf_ch4, masstot;
  double max_ts, lastflux_ann, sdt;
  int ts_timeout;
  double alkH, alkL, hH, hL;
  unsigned status;
  // per-year
  typename BiomeArr<B>::type co2fert, tempfertd, f_new_thaw;
  double luc_e, luc_u, ffi, daccs, npp_luc_adjust;
  ChemK kH, kL;
  double pco2H, pco2L;
  double annualflux_sum, nbp;
  int nstash, nsteps;
  double ode_start;
  bool chem_fresh;  // pco2H/L already computed for the current box carbon
  // where this lane's constants live
  hx_gcd par;  // params + mem   (row stride npad)
  hx_gcd der;  // derived + mem
  int npad;
  double (*pk)[64];  // LDS park
  int lane;
  hx_ccd upar;            // multi-biome kernels: the uniform-row table, or null if LandK rows vary
  const HxBuffers *bufp;  // run kernel only: for the diagnostics written inside the stash
  int iy;                 // year index being integrated
};
#define PKM(m, slot) ((m).pk[(slot)][(m).lane])

// biome constants of the land model, fetched where they are used
template <int B>
struct LandK {
  double npp0[B], f_nppv[B], f_nppd[B], f_litterd[B], rh_ch4_frac[B], fpf_static[B];
};
template <int B>
__device__ __forceinline__ void load_landk(const Member<B> &m, LandK<B> &k) {
  HX_FENCE();
  if constexpr (B == 1) {
    constexpr int o = hx_pkb1<B>();
    k.npp0[0] = PKM(m, o + PKB_NPP0); k.f_nppv[0] = PKM(m, o + PKB_F_NPPV);
    k.f_nppd[0] = PKM(m, o + PKB_F_NPPD); k.f_litterd[0] = PKM(m, o + PKB_F_LITTERD);
    k.rh_ch4_frac[0] = PKM(m, o + PKB_RH_CH4_FRAC); k.fpf_static[0] = PKM(m, o + PKB_FPF_STATIC);
    return;
  }
  if (m.upar) {
    // every member has the same biome constants (the usual case: ensembles perturb Q10, beta,
    // warming factors): wave-uniform scalar loads instead of 6 B vector loads from HBM
#pragma unroll
    for (int b = 0; b < B; ++b) {
      hx_ccd r = m.upar + (HXP_NGLOBAL + b * HXPB_N);
      k.npp0[b] = r[HXPB_NPP0]; k.f_nppv[b] = r[HXPB_F_NPPV]; k.f_nppd[b] = r[HXPB_F_NPPD];
      k.f_litterd[b] = r[HXPB_F_LITTERD]; k.rh_ch4_frac[b] = r[HXPB_RH_CH4_FRAC];
      k.fpf_static[b] = r[HXPB_FPF_STATIC];
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < B; ++b) {
    hx_gcd r = m.par + (size_t)(HXP_NGLOBAL + b * HXPB_N) * m.npad;
    k.npp0[b] = r[(size_t)HXPB_NPP0 * m.npad];
    k.f_nppv[b] = r[(size_t)HXPB_F_NPPV * m.npad];
    k.f_nppd[b] = r[(size_t)HXPB_F_NPPD * m.npad];
    k.f_litterd[b] = r[(size_t)HXPB_F_LITTERD * m.npad];
    k.rh_ch4_frac[b] = r[(size_t)HXPB_RH_CH4_FRAC * m.npad];
    k.fpf_static[b] = r[(size_t)HXPB_FPF_STATIC * m.npad];
  }
}

// rhs constants that only change at a stash (pools frozen in between,
// src/simpleNbox-runtime.cpp:809-840)
struct Interval {
  double P, npp, rh, v1, d2, s3, k4, k5, k7;
  double totC, surf, inv_surf;
};

template <int B> __device__ __forceinline__ double m_npp(const Member<B> &m, const LandK<B> &k, int b) {
  return (k.npp0[b] * m.co2fert[b]) * m.npp_luc_adjust;  // :622-635
}
template <int B> __device__ __forceinline__ double m_rh_fda(const Member<B> &m, int b) {
  return (m.det[b] * 0.25) * m.tempfertd[b];  // :653-665
}
template <int B> __device__ __forceinline__ double m_rh_fsa(const Member<B> &m, int b) {
  return (m.soil[b] * 0.02) * m.tempferts[b];  // :671-683
}
template <int B> __device__ __forceinline__ double m_rh_tp_co2(const Member<B> &m, const LandK<B> &k, int b) {
  return ((m.thawed[b] * (1 - k.fpf_static[b])) * 0.02) * m.tempferts[b] *
         (1.0 - k.rh_ch4_frac[b]);  // :689-701
}
template <int B> __device__ __forceinline__ double m_rh_tp_ch4(const Member<B> &m, const LandK<B> &k, int b) {
  return hx_div(m_rh_tp_co2(m, k, b), 1.0 - k.rh_ch4_frac[b]) * k.rh_ch4_frac[b];  // :707-711
}

// constraints of one model year, as the solver and the stash see them (CON kernels only)
struct YearCon {
  int mask;          // HXC_* bits
  double co2;        // CO2 constraint of the year that ends at tnew (NaN = none)
  double nbp_lo;     // NBP constraint of date tnew - 1  (round(t) for t < tnew - 0.5)
  double nbp_hi;     // NBP constraint of date tnew
  double t_half;     // tnew - 0.5: round(t) switches from tnew - 1 to tnew here
};

// the land flows of an interval with frozen pools
struct Flows {
  double npp, rh, fav, fad, fas, fda, fsa, tpc, tpm, litter, lfvd, lfvs, detsoil, thaw, refr;
};

template <int B>
__device__ __forceinline__ void make_interval(const Member<B> &m, const Flows &F, Interval &K) {
  K.npp = F.npp;
  K.rh = F.rh;
  K.P = ((m.ffi - m.daccs) + m.luc_e) - m.luc_u;
  K.v1 = F.fav - F.litter;
  K.d2 = ((F.fad + F.lfvd) - F.detsoil) - F.fda;
  K.s3 = ((F.fas + F.lfvs) + F.detsoil) - F.fsa;
  K.k4 = -F.thaw + F.refr;
  K.k5 = ((F.thaw - F.refr) - F.tpm) - F.tpc;
  K.k7 = -m.ffi + m.daccs;
  K.totC = m.cDO + m.cIO + m.cLL + m.cHL;  // ocean_component.cpp:325-328
  K.surf = m.cLL + m.cHL;
  K.inv_surf = hx_recip(K.surf);
}

// NBP constraint inside calcderivs: NPP and RH moved by +-diff/2, their parts scaled
// (simpleNbox-runtime.cpp:871-898)
template <int B>
__device__ __forceinline__ void make_interval_nbp(const Member<B> &m, Flows F, double target,
                                                  Interval &K) {
  if (!isnan(target)) {
    const double nbp = ((F.npp - F.rh) - m.luc_e) + m.luc_u;
    const double diff = target - nbp;
    const double npp_old = F.npp;
    F.npp = F.npp + diff / 2.0;
    const double npp_ratio = F.npp / npp_old;
    F.fav = F.fav * npp_ratio; F.fad = F.fad * npp_ratio; F.fas = F.fas * npp_ratio;
    const double rh_old = F.rh;
    F.rh = F.rh - diff / 2.0;
    const double rh_ratio = F.rh / rh_old;
    F.fda = F.fda * rh_ratio; F.fsa = F.fsa * rh_ratio; F.tpc = F.tpc * rh_ratio;
  }
  make_interval<B>(m, F, K);
}

template <int B, bool SPIN>
__device__ __forceinline__ void compute_flows(const Member<B> &m, const LandK<B> &lk,
                                              Flows &F) {
  double npp_c = 0, fav = 0, fad = 0, fas = 0, fda = 0, fsa = 0, tpc = 0, tpm = 0;
  double litter = 0, lfvd = 0, lfvs = 0, detsoil = 0, thaw = 0, refr = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const double n = m_npp(m, lk, b);
    npp_c += n;
    fav += n * lk.f_nppv[b];
    fad += n * lk.f_nppd[b];
    fas += n * (1 - lk.f_nppv[b] - lk.f_nppd[b]);
    fda += m_rh_fda(m, b);
    fsa += m_rh_fsa(m, b);
    const double co2 = m_rh_tp_co2(m, lk, b), ch4 = m_rh_tp_ch4(m, lk, b);
    tpc += co2;
    tpm += ch4;
    const double v = m.veg[b] * 0.035;
    litter += v;
    lfvd += v * lk.f_litterd[b];
    lfvs += v * (1 - lk.f_litterd[b]);
    detsoil += m.det[b] * 0.6;
    if (!SPIN) {  // compute_pf_thaw_refreeze :744-772
      double c_thaw = m.pf[b] * m.f_new_thaw[b];
      double r_tp = 0.0;
      if (c_thaw < 0) {
        const double want = -c_thaw;
        c_thaw = 0.0;
        r_tp = fmin(want, m.thawed[b] - co2 - ch4);
      }
      thaw += c_thaw;
      refr += r_tp;
    }
  }
  F.npp = npp_c; F.rh = fda + fsa + tpc;
  F.fav = fav; F.fad = fad; F.fas = fas; F.fda = fda; F.fsa = fsa; F.tpc = tpc; F.tpm = tpm;
  F.litter = litter; F.lfvd = lfvd; F.lfvs = lfvs; F.detsoil = detsoil;
  F.thaw = thaw; F.refr = refr;
}

// K: the interval's constants; K2 (CON kernels): the same for the second half of the year,
// where round(t) picks the next date's NBP constraint
template <int B, bool SPIN, bool CON = false>
__device__ __forceinline__ void prep_interval(const Member<B> &m, const LandK<B> &lk,
                                              Interval &K, Interval &K2, const YearCon &yc) {
  Flows F;
  compute_flows<B, SPIN>(m, lk, F);
  if constexpr (CON && !SPIN) {
    if (yc.mask & HXC_NBP) {
      make_interval_nbp<B>(m, F, yc.nbp_lo, K);
      make_interval_nbp<B>(m, F, yc.nbp_hi, K2);
      return;
    }
    make_interval<B>(m, F, K);
    K2 = K;
    return;
  }
  make_interval<B>(m, F, K);
}

// SimpleNbox::calcderivs + OceanComponent::calcderivs restricted to the five
// pools whose derivative depends on c[] (atmos, veg, det, soil, ocean)
template <int B, bool SPIN, bool CON = false>
__device__ __forceinline__ void rhs(const Member<B> &m, const Interval &K1, const Interval &K2,
                                    const YearCon &yc, double t, const double *y, double *d) {
  // CON kernels carry the thawed-permafrost pool as a sixth solver variable: with an NBP
  // constraint its derivative changes where round(t) does, so it is no longer constant
  const Interval &K = (CON && !SPIN && t >= yc.t_half) ? K2 : K1;
  if constexpr (CON) d[5] = K.k5;
  const double total = y[1] + y[2] + y[3];
  const double r = hx_div1(m.luc_e, total);  // 2e-15 on a term that is itself ~1e-3 of the flux
  double ao;
  if (SPIN) {
    ao = 0.0;  // preindustrial fluxes +1 / -1 PgC/yr  ocean_component.cpp:343-345
  } else {
    const double scale = (K.surf + (y[4] - K.totC)) * K.inv_surf;
    const double co2 = y[0] * PGC2PPM;
    ao = (co2 - m.pco2H * scale) * m.kH.g + (co2 - m.pco2L * scale) * m.kL.g;
  }
  d[0] = ((K.P - ao) - K.npp) + K.rh;
  d[1] = (K.v1 - r * y[1]) + m.luc_u;
  d[2] = K.d2 - r * y[2];
  d[3] = K.s3 - r * y[3];
  d[4] = ao;
}

// OceanComponent::stashCValues + SimpleNbox::stashCValues for one lane
template <int B, bool SPIN, bool CON = false>
__device__ __forceinline__ void stash(Member<B> &m, double t, const double *y,
                                      double c4, double c5, double c7, Interval &K,
                                      Interval &K2, const YearCon &yc, bool more) {
  LandK<B> lk;
  load_landk<B>(m, lk);
  const double kHD = PKM(m, PK_K0 + (HXD_KHD - HXD_KLH)), kLH = PKM(m, PK_K0 + 0),
               kLI = PKM(m, PK_K0 + (HXD_KLI - HXD_KLH)), kIL = PKM(m, PK_K0 + (HXD_KIL - HXD_KLH)),
               kIH = PKM(m, PK_K0 + (HXD_KIH - HXD_KLH)), kID = PKM(m, PK_K0 + (HXD_KID - HXD_KLH)),
               kDI = PKM(m, PK_K0 + (HXD_KDI - HXD_KLH));
  const double yf = t - m.ode_start;
  m.nstash++;
  const bool in_partial_year = (t != floor(t));
  const double co2 = y[0] * PGC2PPM;
  double aH, aL;
  if (SPIN) {
    aH = 1.000 * yf;
    aL = -1.000 * yf;
  } else {
    // compute_fluxes re-runs the chemistry with the PRE-update carbon; at the
    // first stash of a year that is the carbon the year-start solve already used
    // (same T, DIC, alk -> same result), so only later stashes need a new solve
    if (!m.chem_fresh)
      chem_solve2(m.kH, m.kL, m.cHL, m.cLL, m.alkH, m.alkL, m.hH, m.hL, m.pco2H, m.pco2L,
                  m.status);
    m.chem_fresh = false;
    aH = ((co2 - m.pco2H) * m.kH.g) * yf;
    aL = ((co2 - m.pco2L) * m.kL.g) * yf;
  }
  // box-to-box transports, oceanbox.cpp:244-257 (order: HL, LL, IO, DO)
  const double lHD = m.cHL * kHD * yf;
  const double lLH = m.cLL * kLH * yf, lLI = m.cLL * kLI * yf;
  const double lIL = m.cIO * kIL * yf, lIH = m.cIO * kIH * yf,
               lID = m.cIO * kID * yf;
  const double lDI = m.cDO * kDI * yf;
  const double currentflux = aH + aL;
  const double totC = m.cDO + m.cIO + m.cLL + m.cHL;
  const double solver_flux = y[4] - totC;
  double adj = 0.0;
  if (currentflux != 0.0) adj = (solver_flux - currentflux) / 2.0;
  aH += adj;
  aL += adj;
  const double inv_yf = hx_recip(yf);
  const double cdiff = solver_flux * inv_yf - m.lastflux_ann;
  if (cdiff > 0.1) {  // ocean_component.cpp:703-733
    m.max_ts = fmax(0.3, m.max_ts * 0.5);
    m.ts_timeout = 20;
  } else if (!in_partial_year && m.ts_timeout) {
    m.ts_timeout = max(0, m.ts_timeout - 1);
    if (!m.ts_timeout) {
      m.max_ts = fmin(1.0, m.max_ts / 0.5);
      if (m.max_ts < 1.0) m.ts_timeout = 20;
    }
  }
  bool diag = false;
  size_t dgo = 0;
  if constexpr (CON && !SPIN) {
    diag = m.bufp->stash_diag != 0;
    if (diag) {  // annualflux_sumHL/LL, annual_box_fluxes[HL->DO]: sums over the year's stashes
      const HxBuffers &buf = *m.bufp;
      dgo = (size_t)m.iy * buf.npad + (blockIdx.x * 64 + m.lane);
      if (buf.out[HXO_HL_UPTAKE]) HX_GD(buf.out[HXO_HL_UPTAKE])[dgo] += aH;
      if (buf.out[HXO_LL_UPTAKE]) HX_GD(buf.out[HXO_LL_UPTAKE])[dgo] += aL;
      if (buf.out[HXO_HL_DO]) HX_GD(buf.out[HXO_HL_DO])[dgo] += lHD;
    }
  }
  const double lastflux = aL + aH;
  m.annualflux_sum += lastflux;
  m.lastflux_ann = lastflux * inv_yf;
  // update_state: carbon + additions + ao - oa - subtractions (oceanbox.cpp:297-303)
  m.cHL = ((m.cHL + (lLH + lIH)) + aH) - lHD;
  m.cLL = ((m.cLL + lIL) + aL) - (lLH + lLI);
  m.cIO = (m.cIO + (lLI + lDI)) - ((lIL + lIH) + lID);
  m.cDO = (m.cDO + (lHD + lID)) - lDI;

  // ---- land: simpleNbox-runtime.cpp:270-609 --------------------------------
  double npp_t = 0, rh_t = 0, pf_t = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) npp_t += m_npp(m, lk, b);
#pragma unroll
  for (int b = 0; b < B; ++b)
    rh_t += (m_rh_fda(m, b) + m_rh_fsa(m, b)) + m_rh_tp_co2(m, lk, b);
#pragma unroll
  for (int b = 0; b < B; ++b) pf_t += m.pf[b];
  double alf = ((npp_t - rh_t) - m.luc_e) + m.luc_u;
  const double npp_rh = npp_t + rh_t;
  double tpf = c5;
  if (fabs(tpf) < 1e-10) tpf = 0.0;  // :337-341
  if (y[0] < 0 || y[1] < 0 || y[2] < 0 || y[3] < 0 || c4 < 0 || tpf < 0)
    m.status |= HX_ERR_NEGPOOL;
  double nveg = y[1], ndet = y[2], nsoil = y[3];
  double rh_adj = 1.0;
  double npp_fin_total = npp_t;  // npp_total after any NBP constraint (final_npp weights it)
  if constexpr (CON && !SPIN) {
    // NBP constraint in stashCValues :343-383: fluxes moved by +-diff/2, the pools by
    // diff * yf shared by size, the same amount taken out of the deep ocean
    const double target = (t >= yc.t_half) ? yc.nbp_hi : yc.nbp_lo;
    if ((yc.mask & HXC_NBP) && !isnan(target)) {
      const double diff = target - alf;
      const double npp2 = npp_t + diff / 2.0;
      npp_fin_total = npp2;
      rh_adj = (rh_t - diff / 2.0) / rh_t;
      const double rh2 = rh_t - diff / 2.0;
      const double pool_diff = diff * yf;
      const double total_land = ((y[2] + y[1]) + y[3]) + c5;
      ndet = ndet + pool_diff * y[2] / total_land;
      nveg = nveg + pool_diff * y[1] / total_land;
      nsoil = nsoil + pool_diff * y[3] / total_land;
      tpf = tpf + pool_diff * c5 / total_land;
      m.cDO = (-pool_diff) + m.cDO;
      alf = ((npp2 - rh2) - m.luc_e) + m.luc_u;
    }
  }
  m.nbp = alf;
  double fin_npp = 0, fin_rh = 0, fin_det = 0, fin_soil = 0;

  const double total = y[1] + y[2] + y[3];
  m.cum_luc_va += hx_div((m.luc_e - m.luc_u) * y[1], total);  // no yf: :388-393
  const double inv_nr = hx_recip(npp_rh);
  const double inv_pf = (pf_t > 0) ? hx_recip(pf_t) : 0.0;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const double wt = (B == 1) ? 1.0
        : (m_npp(m, lk, b) + ((m_rh_fda(m, b) + m_rh_fsa(m, b)) + m_rh_tp_co2(m, lk, b))) * inv_nr;
    const double wt_pf = (B == 1) ? ((pf_t > 0) ? 1.0 : 0.0) : m.pf[b] * inv_pf;
    if (diag) {  // final_npp / final_rh / final_rh_detritus / final_rh_soil :420-440
      const double a = m_rh_fda(m, b) * rh_adj, bb = m_rh_fsa(m, b) * rh_adj;
      const double cc = m_rh_tp_co2(m, lk, b) * rh_adj, dd = m_rh_tp_ch4(m, lk, b) * rh_adj;
      fin_npp += npp_fin_total * wt;
      fin_rh += ((a + bb) + cc) + dd;
      fin_det += a;
      fin_soil += bb;
    }
    if constexpr (CON) m.cum_pf_ch4 += (m_rh_tp_ch4(m, lk, b) * rh_adj) * yf;
    else m.cum_pf_ch4 += m_rh_tp_ch4(m, lk, b) * yf;  // :481
    m.veg[b] = nveg * wt;
    m.det[b] = ndet * wt;
    m.soil[b] = nsoil * wt;
    m.pf[b] = c4 * wt_pf;
    m.thawed[b] = tpf * wt_pf;
  }
  m.earth = c7;
  m.atmos = y[0];
  const double sum = ((((((y[0] + y[1]) + y[2]) + y[3]) + c4) + c5) + y[4]) + c7 +
                     m.cum_pf_ch4;
  if (m.masstot > 0.0 && fabs(sum - m.masstot) > 0.001) m.status |= HX_ERR_MASS;
  m.masstot = sum;
  double ca_residual = 0.0;
  if (SPIN) {  // pin the atmosphere to C0, residual to the deep box :567-603
    const double match = m.C0 / PGC2PPM;
    const double residual = m.atmos - match;
    m.cDO = residual + m.cDO;
    m.atmos = m.atmos - residual;
  } else if constexpr (CON) {
    // user-supplied [CO2] at this date: same transfer (:567-603); only whole dates exist
    if ((yc.mask & HXC_CO2) && !in_partial_year && !isnan(yc.co2)) {
      const double match = yc.co2 / PGC2PPM;
      const double residual = m.atmos - match;
      ca_residual = residual;
      m.cDO = residual + m.cDO;
      m.atmos = m.atmos - residual;
    }
  }
  if constexpr (CON && !SPIN) {
    if (diag) {  // the last stash of the year is the one that stays
      const HxBuffers &buf = *m.bufp;
      if (buf.out[HXO_NPP]) HX_GD(buf.out[HXO_NPP])[dgo] = fin_npp;
      if (buf.out[HXO_RH]) HX_GD(buf.out[HXO_RH])[dgo] = fin_rh;
      if (buf.out[HXO_RH_DET]) HX_GD(buf.out[HXO_RH_DET])[dgo] = fin_det;
      if (buf.out[HXO_RH_SOIL]) HX_GD(buf.out[HXO_RH_SOIL])[dgo] = fin_soil;
      if (buf.out[HXO_CA_RESIDUAL]) HX_GD(buf.out[HXO_CA_RESIDUAL])[dgo] = ca_residual;
    }
  }
  m.ode_start = t;
  if (more) prep_interval<B, SPIN, CON>(m, lk, K, K2, yc);  // constants of the next segment
}

// exp(p*log(x)) for the step-size controller (x in [5^-5, ~1e3]); the value only
// scales the next trial step, so ~1e-15 relative error is immaterial.
__device__ __forceinline__ double powr(double x, double p) { return exp(p * log(x)); }

// x^(-1/5) for the step-growth rule: single-precision seed, two Newton steps on
// y^-5 = x in fp64 (relative error e -> 3e^2: 1e-6 -> 3e-12 -> ~1e-16).  The
// fp64 log/exp pair it replaces is a ~75-instruction dependent chain, the longest
// in the step block, and a single resident wavefront cannot hide it.
__device__ __forceinline__ double pow_m15(double x) {
  double y = (double)exp2f(-0.2f * log2f((float)x));
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double y2 = y * y;
    const double y5 = y2 * y2 * y;
    y = y + y * ((1.0 - x * y5) * 0.2);
  }
  return y;
}

// CarbonCycleSolver::run for one model year t0 -> tnew (carbon-cycle-solver.cpp:
// 222-303).  The 64 lanes walk the reference's control flow in lock-step over
// SEGMENTS (one stash interval each): inner loop = dopri5 attempts until every
// lane has reached its own t_target (retries only move t_target), then ONE stash
// block for all lanes.  Lanes in reduced-timestep mode take up to 4 segments a
// year, the others idle through the extra ones; the expensive step and stash
// blocks are never interleaved lane by lane.
template <int B, bool SPIN, bool CON = false>
__device__ __forceinline__ void solve_year(Member<B> &m, const HxConst &kc,
                                           double t0, double tnew, const YearCon &yc) {
  constexpr int NP = CON ? 6 : 5;  // solver variables (see rhs)
  // dopri5 tableau (odeint runge_kutta_dopri5)
  constexpr double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40, b41 = 44.0 / 45,
                   b42 = -56.0 / 15, b43 = 32.0 / 9, b51 = 19372.0 / 6561,
                   b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729,
                   b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247,
                   b64 = 49.0 / 176, b65 = -5103.0 / 18656;
  constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192,
                   c5 = -2187.0 / 6784, c6 = 11.0 / 84;
  constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695,
                   dc4 = c4 - 393.0 / 640, dc5 = c5 - (-92097.0 / 339200),
                   dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
  constexpr double EPS = 2.220446049250313e-16;

  Interval K, K2s;
  Interval &K2 = CON ? K2s : K;
  {
    LandK<B> lk;
    load_landk<B>(m, lk);
    prep_interval<B, SPIN, CON>(m, lk, K, K2, yc);
  }
  // getCValues  simpleNbox-runtime.cpp:247-258
  double y[NP], l4, l5, l7;
  auto load_pools = [&]() {
    double v = 0, d = 0, s = 0, p = 0, th = 0;
#pragma unroll
    for (int b = 0; b < B; ++b) { v += m.veg[b]; d += m.det[b]; s += m.soil[b];
                                   p += m.pf[b]; th += m.thawed[b]; }
    y[0] = m.atmos; y[1] = v; y[2] = d; y[3] = s; y[4] = m.cDO + m.cIO + m.cLL + m.cHL;
    l4 = p; l5 = th; l7 = m.earth;
    if constexpr (CON) y[5] = th;
  };
  load_pools();
  m.ode_start = t0;
  double t = t0;   // time reached by accepted steps
  int retry = 0;
  bool alive = true;
  while (__any(alive && t < tnew)) {
    const bool seg = alive && t < tnew;
    // fresh integrate_adaptive call: by-value dt, fresh controlled stepper
    const double t_start = t;
    double t_target = tnew, dtl = m.sdt;
    double dxdt[NP];
    bool first_call = true;
    int fails = 0;
    bool stepping = seg;
    while (__any(stepping)) {
      if (stepping) {
        if (first_call) { rhs<B, SPIN, CON>(m, K, K2, yc, t, y, dxdt); first_call = false; }
        if (((t + dtl) - t_target) > EPS) dtl = t_target - t;
        // Every dopri5 stage time is <= t+dtl, and the model refuses any RHS
        // evaluation beyond max_timestep (ocean_component.cpp:621-625), so the
        // attempt throws CARBON_CYCLE_RETRY iff its last stage does.
        if (((t + dtl) - m.ode_start) > m.max_ts) {
          ++retry;  // carbon-cycle-solver.cpp:266-276
          t_target = t_start + (t_target - t_start) / 2.0;
          t = t_start;
          m.sdt = t_target - t;
          dtl = m.sdt;
          load_pools();
          first_call = true;
          fails = 0;
          if (retry >= 8) { m.status |= HX_ERR_RETRIES; alive = false; stepping = false; }
        } else {
          double k2[NP], k3[NP], k4[NP], k5[NP], k6[NP], xt[NP], xn[NP], dn[NP];
#pragma unroll
          for (int i = 0; i < NP; ++i) xt[i] = y[i] + dtl * b21 * dxdt[i];
          rhs<B, SPIN, CON>(m, K, K2, yc, t + dtl * (1.0 / 5), xt, k2);
#pragma unroll
          for (int i = 0; i < NP; ++i)
            xt[i] = y[i] + dtl * b31 * dxdt[i] + dtl * b32 * k2[i];
          rhs<B, SPIN, CON>(m, K, K2, yc, t + dtl * (3.0 / 10), xt, k3);
#pragma unroll
          for (int i = 0; i < NP; ++i)
            xt[i] = y[i] + dtl * b41 * dxdt[i] + dtl * b42 * k2[i] + dtl * b43 * k3[i];
          rhs<B, SPIN, CON>(m, K, K2, yc, t + dtl * (4.0 / 5), xt, k4);
#pragma unroll
          for (int i = 0; i < NP; ++i)
            xt[i] = y[i] + dtl * b51 * dxdt[i] + dtl * b52 * k2[i] +
                    dtl * b53 * k3[i] + dtl * b54 * k4[i];
          rhs<B, SPIN, CON>(m, K, K2, yc, t + dtl * (8.0 / 9), xt, k5);
#pragma unroll
          for (int i = 0; i < NP; ++i)
            xt[i] = y[i] + dtl * b61 * dxdt[i] + dtl * b62 * k2[i] +
                    dtl * b63 * k3[i] + dtl * b64 * k4[i] + dtl * b65 * k5[i];
          rhs<B, SPIN, CON>(m, K, K2, yc, t + dtl, xt, k6);
#pragma unroll
          for (int i = 0; i < NP; ++i)
            xn[i] = y[i] + dtl * c1 * dxdt[i] + dtl * c3 * k3[i] + dtl * c4 * k4[i] +
                    dtl * c5 * k5[i] + dtl * c6 * k6[i];
          rhs<B, SPIN, CON>(m, K, K2, yc, t + dtl, xn, dn);
          // default_error_checker: err = max_i |xe_i| / (eps_abs + eps_rel (|y_i| + dt |dy_i|)).
          // The maximum of the five quotients is found by cross-multiplication
          // (all denominators > 0) and divided once.
          double en = 0.0, ed = 1.0;
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            const double xe = dtl * dc1 * dxdt[i] + dtl * dc3 * k3[i] +
                              dtl * dc4 * k4[i] + dtl * dc5 * k5[i] +
                              dtl * dc6 * k6[i] + dtl * dc7 * dn[i];
            const double n = fabs(xe);
            const double d = kc.eps_abs + kc.eps_rel * (fabs(y[i]) + dtl * fabs(dxdt[i]));
            if (n * ed > en * d) { en = n; ed = d; }
          }
          double err = hx_div(en, ed);
          if (err > 1.0) {  // reject: default_step_adjuster::decrease_step
            dtl *= fmax(0.9 * powr(err, -1.0 / 3.0), 0.2);
            if (++fails > 500) { m.status |= HX_ERR_STEPFAIL; alive = false; stepping = false; }
          } else {          // accept
            // pools with a constant derivative over the interval advance exactly
            l4 += dtl * K.k4; l7 += dtl * K.k7;
            if constexpr (!CON) l5 += dtl * K.k5;
            t += dtl;
            // increase_step: err < 0.5 -> dt *= 0.9 * max(err, 5^-5)^(-1/5)
            const double grow = 0.9 * pow_m15(fmax(0.00032, err));
            if (err < 0.5) dtl *= grow;
#pragma unroll
            for (int i = 0; i < NP; ++i) { y[i] = xn[i]; dxdt[i] = dn[i]; }
            fails = 0;
            m.nsteps++;
            if (!((t_target - t) > EPS)) stepping = false;  // integrate_adaptive done
          }
        }
      }
    }
    if (seg && alive) {
      // the solver keeps integrating its own c[] afterwards (no getCValues,
      // carbon-cycle-solver.cpp:282-287); only the frozen-pool constants move
      retry = 0;
      stash<B, SPIN, CON>(m, t, y, l4, CON ? y[NP - 1] : l5, l7, K, K2, yc, t < tnew);
    }
  }
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

// ---- SoA helpers ------------------------------------------------------------
__device__ __forceinline__ double ldp(const HxBuffers &b, int row, int mem) {
  return HX_GCD(b.params)[(size_t)row * b.npad + mem];
}
__device__ __forceinline__ double ldd(const HxBuffers &b, int row, int mem) {
  return HX_GCD(b.derived)[(size_t)row * b.npad + mem];
}
__device__ __forceinline__ double lds_(const HxBuffers &b, int row, int mem) {
  return HX_GCD(b.state)[(size_t)row * b.npad + mem];
}
__device__ __forceinline__ void sts_(const HxBuffers &b, int row, int mem, double v) {
  HX_GD(b.state)[(size_t)row * b.npad + mem] = v;
}
__device__ __forceinline__ void sto_(const HxBuffers &b, int var, size_t off, double v) {
  HX_GD(b.out[var])[off] = v;
}

template <int B>
__device__ __forceinline__ void bind_member(const HxBuffers &buf, int mem, Member<B> &m,
                                            double (*park)[64], int lane) {
  m.par = HX_GCD(buf.params) + mem;
  m.der = HX_GCD(buf.derived) + mem;
  m.npad = buf.npad;
  m.pk = park;
  m.lane = lane;
  m.upar = (B > 1 && buf.uni_landk) ? HX_CCD(buf.uparams) : nullptr;
  if constexpr (B > 1) {
    constexpr int o = hx_pkb1<B>();
    ParkArr *arr[HX_NBIOME_ARR] = {&m.veg, &m.det, &m.soil, &m.pf, &m.thawed, &m.tempferts,
                                   &m.co2fert, &m.tempfertd, &m.f_new_thaw};
#pragma unroll
    for (int k = 0; k < HX_NBIOME_ARR; ++k) { arr[k]->base = park + o + k * B; arr[k]->lane = lane; }
  }
  m.C0 = ldp(buf, HXP_C0, mem);
  // constants -> park
  PKM(m, PK_AERO) = ldp(buf, HXP_AERO, mem);
  PKM(m, PK_VOL) = ldp(buf, HXP_VOL, mem);
#pragma unroll
  for (int k = 0; k < 14; ++k) PKM(m, PK_D0 + k) = ldd(buf, HXD_A0 + k, mem);
#pragma unroll
  for (int k = 0; k < 7; ++k) PKM(m, PK_K0 + k) = ldd(buf, HXD_KLH + k, mem);
  if constexpr (B == 1) {
    constexpr int o = hx_pkb1<B>();
    const int r = HXP_NGLOBAL;
    PKM(m, o + PKB_NPP0) = ldp(buf, r + HXPB_NPP0, mem);
    PKM(m, o + PKB_F_NPPV) = ldp(buf, r + HXPB_F_NPPV, mem);
    PKM(m, o + PKB_F_NPPD) = ldp(buf, r + HXPB_F_NPPD, mem);
    PKM(m, o + PKB_F_LITTERD) = ldp(buf, r + HXPB_F_LITTERD, mem);
    PKM(m, o + PKB_RH_CH4_FRAC) = ldp(buf, r + HXPB_RH_CH4_FRAC, mem);
    PKM(m, o + PKB_FPF_STATIC) = ldp(buf, r + HXPB_FPF_STATIC, mem);
    PKM(m, o + PKB_BETA) = ldp(buf, r + HXPB_BETA, mem);
    PKM(m, o + PKB_WF) = ldp(buf, r + HXPB_WF, mem);
    PKM(m, o + PKB_LNQ10) = ldd(buf, HXD_NGLOBAL, mem);
    PKM(m, o + PKB_MU) = ldp(buf, r + HXPB_PF_MU, mem);
    PKM(m, o + PKB_SIGMA) = ldp(buf, r + HXPB_PF_SIGMA, mem);
  }
}

// solver-resident state <-> HBM state table
template <int B>
__device__ __forceinline__ void load_state(const HxBuffers &buf, int mem, Member<B> &m) {
  m.cHL = lds_(buf, HXS_C_HL, mem); m.cLL = lds_(buf, HXS_C_LL, mem);
  m.cIO = lds_(buf, HXS_C_IO, mem); m.cDO = lds_(buf, HXS_C_DO, mem);
  m.atmos = lds_(buf, HXS_ATMOS, mem); m.earth = lds_(buf, HXS_EARTH, mem);
  m.cum_luc_va = lds_(buf, HXS_CUM_LUC_VA, mem);
  m.cum_pf_ch4 = lds_(buf, HXS_CUM_PF_CH4, mem);
  m.masstot = lds_(buf, HXS_MASSTOT, mem);
  m.max_ts = lds_(buf, HXS_MAX_TS, mem);
  m.ts_timeout = (int)lds_(buf, HXS_TS_TIMEOUT, mem);
  m.lastflux_ann = lds_(buf, HXS_LASTFLUX_ANN, mem);
  m.sdt = lds_(buf, HXS_SOLVER_DT, mem);
  m.alkH = lds_(buf, HXS_ALK_HL, mem); m.alkL = lds_(buf, HXS_ALK_LL, mem);
  m.hH = lds_(buf, HXS_H_HL, mem); m.hL = lds_(buf, HXS_H_LL, mem);
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const int r = HXS_NGLOBAL + b * HXSB_N;
    m.veg[b] = lds_(buf, r + HXSB_VEG, mem); m.det[b] = lds_(buf, r + HXSB_DET, mem);
    m.soil[b] = lds_(buf, r + HXSB_SOIL, mem); m.pf[b] = lds_(buf, r + HXSB_PF, mem);
    m.thawed[b] = lds_(buf, r + HXSB_THAWED, mem);
    m.tempferts[b] = lds_(buf, r + HXSB_TEMPFERTS, mem);
  }
  m.status = HX_GU(buf.status)[mem];
}

template <int B>
__device__ __forceinline__ void store_state(const HxBuffers &buf_, int mem,
                                            const Member<B> &m, double *base = nullptr) {
  // base == nullptr: the live state table; otherwise a per-year history slab
  HxBuffers buf = buf_;
  if (base) buf.state = base;
  sts_(buf, HXS_C_HL, mem, m.cHL); sts_(buf, HXS_C_LL, mem, m.cLL);
  sts_(buf, HXS_C_IO, mem, m.cIO); sts_(buf, HXS_C_DO, mem, m.cDO);
  sts_(buf, HXS_ATMOS, mem, m.atmos); sts_(buf, HXS_EARTH, mem, m.earth);
  sts_(buf, HXS_CUM_LUC_VA, mem, m.cum_luc_va);
  sts_(buf, HXS_CUM_PF_CH4, mem, m.cum_pf_ch4);
  sts_(buf, HXS_MASSTOT, mem, m.masstot);
  sts_(buf, HXS_MAX_TS, mem, m.max_ts);
  sts_(buf, HXS_TS_TIMEOUT, mem, (double)m.ts_timeout);
  sts_(buf, HXS_LASTFLUX_ANN, mem, m.lastflux_ann);
  sts_(buf, HXS_SOLVER_DT, mem, m.sdt);
  sts_(buf, HXS_ALK_HL, mem, m.alkH); sts_(buf, HXS_ALK_LL, mem, m.alkL);
  sts_(buf, HXS_H_HL, mem, m.hH); sts_(buf, HXS_H_LL, mem, m.hL);
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const int r = HXS_NGLOBAL + b * HXSB_N;
    sts_(buf, r + HXSB_VEG, mem, m.veg[b]); sts_(buf, r + HXSB_DET, mem, m.det[b]);
    sts_(buf, r + HXSB_SOIL, mem, m.soil[b]); sts_(buf, r + HXSB_PF, mem, m.pf[b]);
    sts_(buf, r + HXSB_THAWED, mem, m.thawed[b]);
    sts_(buf, r + HXSB_TEMPFERTS, mem, m.tempferts[b]);
  }
  if (!base) HX_GU(buf.status)[mem] = m.status;
}

// year-level state: park -> state rows of `base` (live table or history slab)
template <int B>
__device__ __forceinline__ void store_park_state(const HxBuffers &buf_, int mem,
                                                 const Member<B> &m, double *base = nullptr) {
  HxBuffers buf = buf_;
  if (base) buf.state = base;
  sts_(buf, HXS_CH4, mem, PKM(m, PK_CH4)); sts_(buf, HXS_SST, mem, PKM(m, PK_SST));
  sts_(buf, HXS_TLAND, mem, PKM(m, PK_TLAND)); sts_(buf, HXS_TWIN, mem, PKM(m, PK_TWIN));
  sts_(buf, HXS_TL_M1, mem, PKM(m, PK_TL_M1)); sts_(buf, HXS_TL_M2, mem, PKM(m, PK_TL_M2));
  sts_(buf, HXS_F_PREV, mem, PKM(m, PK_F_PREV));
  sts_(buf, HXS_BASE_TOT, mem, PKM(m, PK_BASE_TOT));
  sts_(buf, HXS_BASE_CO2, mem, PKM(m, PK_BASE_CO2));
  if (base) sts_(buf, HXS_EOS_VEGC, mem, PKM(m, PK_EOS));
#pragma unroll
  for (int b = 0; b < B; ++b)
    sts_(buf, HXS_NGLOBAL + b * HXSB_N + HXSB_F_FROZEN, mem, PKM(m, PK_FFROZEN0 + b));
}

}  // namespace

// ===========================================================================
// Per-member derived constants, once per parameter upload: DOECLIM matrices and
// time scales (temperature_component.cpp:251-412), ocean exchange coefficients
// (ocean_component.cpp:265-284), ln(q10).  ker: DOECLIM kernel table.
// ===========================================================================
__global__ __launch_bounds__(256) void hx_derive_kernel(const double *params, double *derived,
                                                        const double *ker, int ker_per_member,
                                                        int ns, int nbiome, int npad) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  if (mem >= npad) return;
  auto P = [&](int row) { return params[(size_t)row * npad + mem]; };
  auto D = [&](int row, double v) { derived[(size_t)row * npad + mem] = v; };
  const double S = P(HXP_S), qco2 = P(HXP_QCO2), diff = P(HXP_DIFF);
  const double flnd = D_flnd, bsi = D_bsi, rlam = D_rlam, ak = D_ak, bk = D_bk,
               cal = D_cal, cas = D_cas, fso = D_fso;
  const double cnum = rlam * flnd + bsi * (1.0 - flnd);
  const double cden = rlam * flnd - ak * (rlam - bsi);
  const double cfl = flnd * cnum / cden * qco2 / S - bk * (rlam - bsi) / cden;
  const double cfs = (rlam * flnd - ak / (1.0 - flnd) * (rlam - bsi)) * cnum / cden *
                         qco2 / S +
                     rlam * flnd / (1.0 - flnd) * bk * (rlam - bsi) / cden;
  const double kls = bk * rlam * flnd / cden - ak * flnd * cnum / cden * qco2 / S;
  const double keff = (D_secs / 10000) * diff;
  const double taucfs = cas / cfs, taucfl = cal / cfl;
  const double taudif = (cas * cas) / (D_csw * D_csw) * M_PI / keff;
  const double tauksl = (1.0 - flnd) * cas / kls, taukls = flnd * cal / kls;
  double C0_ = 1.0 / (taucfl * taucfl) + 1.0 / (taukls * taukls) +
               2.0 / taucfl / taukls + bsi / taukls / tauksl;
  double C1_ = -1 * bsi / (taukls * taukls) - bsi / taucfl / taukls -
               bsi / taucfs / taukls - (bsi * bsi) / taukls / tauksl;
  double C2_ = -1 * bsi / (tauksl * tauksl) - 1.0 / taucfs / tauksl -
               1.0 / taucfl / tauksl - 1.0 / taukls / tauksl;
  double C3_ = 1.0 / (taucfs * taucfs) + (bsi * bsi) / (tauksl * tauksl) +
               2.0 * bsi / taucfs / tauksl + bsi / taukls / tauksl;
  C0_ *= 1.0 / 12.0; C1_ *= 1.0 / 12.0; C2_ *= 1.0 / 12.0; C3_ *= 1.0 / 12.0;
  const double sq = sqrt(1.0 / taudif);
  const double ker_last = ker_per_member ? ker[(size_t)(ns - 1 + HX_KPAD) * npad + mem]
                                         : ker[ns - 1 + HX_KPAD];
  const double B0 = 1.0 + 1.0 / (2.0 * taucfl) + 1.0 / (2.0 * taukls) + C0_;
  const double B1 = -1.0 / (2.0 * taukls) * bsi + C1_;
  const double B2 = -1.0 / (2.0 * tauksl) + C2_;
  const double B3 = 1.0 + 1.0 / (2.0 * taucfs) + 1.0 / (2.0 * tauksl) * bsi +
                    2.0 * fso * sq + C3_;
  D(HXD_A0, 1.0 - 1.0 / (2.0 * taucfl) - 1.0 / (2.0 * taukls) + C0_);
  D(HXD_A1, 1.0 / (2.0 * taukls) * bsi + C1_);
  D(HXD_A2, 1.0 / (2.0 * tauksl) + C2_);
  D(HXD_A3, 1.0 - 1.0 / (2.0 * taucfs) - 1.0 / (2.0 * tauksl) * bsi + ker_last * fso * sq + C3_);
  const double det = B0 * B3 - B1 * B2;
  const double idet = 1 / det;
  D(HXD_IB0, idet * B3); D(HXD_IB1, idet * -1 * B1); D(HXD_IB2, idet * -1 * B2);
  D(HXD_IB3, idet * B0);
  // QC1/QC2 with DelQL == DelQO (temperature_component.cpp:462-477)
  D(HXD_QC1, ((1.0 / cal) * (1.0 / taucfl + 1.0 / taukls) - bsi / cas / taukls) / 12.0);
  D(HXD_QC2, ((1.0 / cas) * (1.0 / taucfs + bsi / tauksl) - 1.0 / cal / tauksl) / 12.0);
  D(HXD_DQ1, 0.5 / cal); D(HXD_DQ2, 0.5 / cas);
  D(HXD_DPSCALE, fso * sq);
  D(HXD_HFSCALE, cas * fso / sqrt(taudif));
  D(HXD_FLAG, det == 0 ? (double)HX_ERR_SINGULAR : 0.0);
  // exchange coefficients  src/ocean_component.cpp:265-284
  const double tt = P(HXP_TT), tu = P(HXP_TU), twi = P(HXP_TWI), tid = P(HXP_TID);
  D(HXD_KLH, (tt * O_spy) / O_vLL);
  D(HXD_KHD, ((tt + tu) * O_spy) / O_vHL);
  const double DO_IO = ((tt + tu) * O_spy) / O_vD;
  D(HXD_KIH, (tu * O_spy) / O_vI);
  const double IO_LL = (tt * O_spy) / O_vI;
  const double IO_LLex = (twi * O_spy) / O_vI;
  D(HXD_KLI, (twi * O_spy) / O_vLL);
  const double DO_IOex = (tid * O_spy) / O_vD;
  D(HXD_KID, (tid * O_spy) / O_vI);
  D(HXD_KIL, IO_LL + IO_LLex);
  D(HXD_KDI, DO_IO + DO_IOex);
  for (int b = 0; b < nbiome; ++b)
    D(HXD_NGLOBAL + b, log(P(HXP_NGLOBAL + b * HXPB_N + HXPB_Q10)));
}


// In-kernel form of the DOECLIM history pass (see hx_doeclim_pass_kernel below for
// the algorithm): one lane = one member, all HX_DBLK block years, two sweeps of 16
// accumulators.  Deliberately NOT inlined: as a real call it gets its own register
// allocation (16 loads in flight need landing registers the year loop does not have),
// and the caller's live registers are saved around it once per HX_DBLK years.
template <bool KERPM, bool HF>
__device__ __attribute__((noinline)) void doeclim_pass_dev(const double *sst_hist,
                                                           const double *ker, double *part,
                                                           double *part2, int ns, int npad,
                                                           int blk0, int mem) {
  hx_gcd hist = HX_GCD(sst_hist) + mem;
  const size_t np = (size_t)npad;
  auto ldk = [&](int idx) -> double {
    if constexpr (KERPM) return HX_GCD(ker)[(size_t)idx * np + mem];
    else return HX_CCD(ker)[idx];
  };
  for (int j0 = 0; j0 < HX_DBLK; j0 += 16) {
    double acc[16], acc2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = 0; acc2[j] = 0; }
    // window entry w of chunk i0 = Ker[(ns - (blk0 + j0) - 1) + i0 - 15 + w]
    const int k0 = ns - (blk0 + j0) - 1 - 15 + HX_KPAD;
    if (blk0 + j0 < ns) {
      // software pipeline: the 16 loads of the next chunk are in flight while the
      // 256 FMAs of the current one execute
      auto load_chunk = [&](double *T, int i0) {
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
          const int i = i0 + ii;
          // rows >= blk0 may hold stale values of an earlier run: mask them
          const double v = hist[(size_t)(i < ns ? i : ns - 1) * np];
          T[ii] = (i < blk0) ? v : 0.0;
        }
      };
      auto compute_chunk = [&](const double *T, int i0) {
        double kw[32];
#pragma unroll
        for (int w = 0; w < 32; ++w) kw[w] = ldk(k0 + i0 + w);
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            acc[j] += T[ii] * kw[15 + ii - j];
            if (HF) acc2[j] += T[ii] * kw[16 + ii - j];
          }
        }
      };
      double Ta[16], Tb[16];
      load_chunk(Ta, 0);
      for (int i0 = 0; i0 < blk0; i0 += 32) {
        load_chunk(Tb, i0 + 16);
        compute_chunk(Ta, i0);
        if (i0 + 16 < blk0) {
          load_chunk(Ta, i0 + 32);
          compute_chunk(Tb, i0 + 16);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      HX_GD(part)[(size_t)(j0 + j) * np + mem] = acc[j];
      if (HF) HX_GD(part2)[(size_t)(j0 + j) * np + mem] = acc2[j];
    }
  }
}

// ===========================================================================
// DOECLIM history pass.  sum_{i<t} Tsst[i] * Ker[ns - t + i - 1]
// (temperature_component.cpp:488-491; :534-537 for the heat-flux diagnostic, Ker
// index + 1) is a causal convolution: evaluated per year it re-reads the whole SST
// history every year (83 % of the algorithmic HBM bytes).  The run
// kernel is launched per block of <= HX_DBLK years; before each launch this kernel
// reads the history BEFORE the block once and leaves, for every year of the block,
// the partial sum over that history in part[j][member] -- same ascending summation
// order per year as the reference; the run kernel appends the in-block terms.
// Thread = (member, HX_DJT block years); history in chunks of 16 years: 16 coalesced
// loads in flight + a 23-entry window of Ker (scalar loads when the diffusivity is
// shared) per 128 FMAs.  Ker is zero-padded by HX_KPAD entries on both sides, so
// block years beyond the end of the run and the ragged last chunk need no branches.
// ===========================================================================
template <bool KERPM, bool HF>
__global__ __launch_bounds__(64) void hx_doeclim_pass_kernel(const double *sst_hist,
                                                             const double *ker, double *part,
                                                             double *part2, int ns, int npad,
                                                             int blk0) {
  const int mem = blockIdx.x * 64 + threadIdx.x;
  const int j0 = blockIdx.y * HX_DJT;
  if (mem >= npad) return;
  double acc[HX_DJT], acc2[HX_DJT];
#pragma unroll
  for (int j = 0; j < HX_DJT; ++j) { acc[j] = 0; acc2[j] = 0; }
  hx_gcd hist = HX_GCD(sst_hist) + mem;
  const size_t np = (size_t)npad;
  // window entry w of chunk i0 = Ker[(ns - (blk0 + j0) - 1) + i0 - (HX_DJT - 1) + w]
  const int k0 = ns - (blk0 + j0) - 1 - (HX_DJT - 1) + HX_KPAD;
  auto ldk = [&](int idx) -> double {
    if constexpr (KERPM) return HX_GCD(ker)[(size_t)idx * np + mem];
    else return HX_CCD(ker)[idx];
  };
  for (int i0 = 0; i0 < blk0; i0 += 16) {
    double T[16], kw[HX_DJT + 16];
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
      const int i = i0 + ii;
      // rows >= blk0 may hold stale values of an earlier run: mask them
      const double v = hist[(size_t)(i < ns ? i : ns - 1) * np];
      T[ii] = (i < blk0) ? v : 0.0;
    }
#pragma unroll
    for (int w = 0; w < HX_DJT + 16; ++w) kw[w] = ldk(k0 + i0 + w);
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
#pragma unroll
      for (int j = 0; j < HX_DJT; ++j) {
        acc[j] += T[ii] * kw[HX_DJT - 1 + ii - j];
        if (HF) acc2[j] += T[ii] * kw[HX_DJT + ii - j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < HX_DJT; ++j) {
    HX_GD(part)[(size_t)(j0 + j) * np + mem] = acc[j];
    if (HF) HX_GD(part2)[(size_t)(j0 + j) * np + mem] = acc2[j];
  }
}

// ===========================================================================
// Spinup: Core::run_spinup (src/core.cpp:394-420) + CarbonCycleSolver::
// run_spinup (src/carbon-cycle-solver.cpp:313-370).  Pseudo-years 0,1,2...
// until max |c(step) - c(step-1)| < eps_spinup; emissions 0, all fertilisation
// factors 1, chemistry off (spinup_chem = 0), atmosphere pinned to C0.
// Initialises the whole state table from the parameter rows.
// ===========================================================================
template <int B>
__global__ __launch_bounds__(64) void hx_spinup_kernel(const HxArgs *__restrict__ args,
                                                       int *spinup_steps) {
  const HxBuffers &buf = args->buf;
  const HxConst &kc = args->kc;
  const int mem = blockIdx.x * 64 + threadIdx.x;
  if (mem >= buf.npad) return;
  __shared__ double s_park[hx_npark<B>()][64];
  Member<B> m;
  bind_member<B>(buf, mem, m, s_park, (int)threadIdx.x);
  // initial conditions: ocean_component.cpp:234-260, simpleNbox.cpp:45-79,
  // simpleNbox-runtime.cpp:146-172
  {
    const double LLf = O_vLL / (O_vLL + O_vHL), HLf = 1 - LLf;
    const double If = O_vI / (O_vI + O_vD), Df = 1 - If;
    const double ps = ldp(buf, HXP_PRE_SURF, mem), pid = ldp(buf, HXP_PRE_ID, mem);
    m.cLL = LLf * ps; m.cHL = HLf * ps; m.cIO = If * pid; m.cDO = Df * pid;
  }
  m.earth = 5500;
  m.atmos = m.C0 * (1.0 / PGC2PPM);
  m.cum_luc_va = 0; m.cum_pf_ch4 = 0; m.masstot = 0;
  m.max_ts = 1.0; m.ts_timeout = 0; m.lastflux_ann = 0; m.sdt = kc.dt0;
  m.alkH = 0; m.alkL = 0; m.hH = 1e-8; m.hL = 1e-8;
  m.status = (unsigned)ldd(buf, HXD_FLAG, mem);
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const int r = HXP_NGLOBAL + b * HXPB_N;
    m.veg[b] = ldp(buf, r + HXPB_VEG0, mem); m.det[b] = ldp(buf, r + HXPB_DET0, mem);
    m.soil[b] = ldp(buf, r + HXPB_SOIL0, mem); m.pf[b] = ldp(buf, r + HXPB_PF0, mem);
    m.thawed[b] = 0; m.tempferts[b] = 1;
    m.co2fert[b] = 1; m.tempfertd[b] = 1; m.f_new_thaw[b] = 0;
  }
  m.luc_e = m.luc_u = m.ffi = m.daccs = 0;
  m.npp_luc_adjust = 1.0;  // (eos - 0)/eos
  m.pco2H = m.pco2L = 0; m.annualflux_sum = 0; m.nbp = 0; m.nsteps = 0;
  m.kH.Tr = m.kL.Tr = 0; m.kH.g = m.kL.g = 0;
  m.chem_fresh = false;

  bool spun = (kc.max_spinup <= 1);
  int steps = 0;
  for (int step = 1; step < kc.max_spinup && __any(!spun); ++step) {
    if (!spun) {
      const double o0 = m.atmos, oO = m.cDO + m.cIO + m.cLL + m.cHL, oE = m.earth;
      double ov = 0, od = 0, os = 0, op = 0, ot = 0;
#pragma unroll
      for (int b = 0; b < B; ++b) { ov += m.veg[b]; od += m.det[b]; os += m.soil[b];
                                     op += m.pf[b]; ot += m.thawed[b]; }
      m.nstash = 0;
      solve_year<B, true>(m, kc, (double)(step - 1), (double)step, YearCon{});
      double nv = 0, nd = 0, nso = 0, np = 0, nt = 0;
#pragma unroll
      for (int b = 0; b < B; ++b) { nv += m.veg[b]; nd += m.det[b]; nso += m.soil[b];
                                     np += m.pf[b]; nt += m.thawed[b]; }
      double mx = fabs(m.atmos - o0);
      mx = fmax(mx, fabs(nv - ov)); mx = fmax(mx, fabs(nd - od));
      mx = fmax(mx, fabs(nso - os)); mx = fmax(mx, fabs(np - op));
      mx = fmax(mx, fabs(nt - ot));
      mx = fmax(mx, fabs((m.cDO + m.cIO + m.cLL + m.cHL) - oO));
      mx = fmax(mx, fabs(m.earth - oE));
      steps = step;
      spun = (mx < kc.eps_spinup) || (m.status != 0);
    }
  }
  if (!spun) m.status |= HX_ERR_SPINUP;
  store_state<B>(buf, mem, m);
  // SimpleNbox::run, first call: end_of_spinup_vegc  runtime.cpp:209-213
  double v1 = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    v1 += m.veg[b];
    sts_(buf, HXS_NGLOBAL + b * HXSB_N + HXSB_F_FROZEN, mem, 1.0);
  }
  sts_(buf, HXS_EOS_VEGC, mem, v1);
  sts_(buf, HXS_CH4, mem, kc.M0f);  // CH4(startDate): M0 or the constraint of that date
  sts_(buf, HXS_TLAND, mem, 0.0); sts_(buf, HXS_SST, mem, 0.0);
  sts_(buf, HXS_F_PREV, mem, 0.0); sts_(buf, HXS_BASE_TOT, mem, 0.0);
  sts_(buf, HXS_BASE_CO2, mem, 0.0);
  sts_(buf, HXS_TL_M1, mem, 0.0); sts_(buf, HXS_TL_M2, mem, 0.0);
  sts_(buf, HXS_TWIN, mem, 0.0);
  // year-0 (startDate) outputs = state recorded at the end of spinup
  const size_t o = (size_t)mem;
  auto put = [&](int var, double v) { if (buf.out[var]) sto_(buf, var, o, v); };
  put(HXO_SST, 0.0); put(HXO_TLAND, 0.0);
  put(HXO_CO2, m.atmos * PGC2PPM); put(HXO_TGAV, 0.0);
  put(HXO_RF_TOT, 0.0); put(HXO_RF_CO2, 0.0);
  put(HXO_OCEAN_C, m.cDO + m.cIO + m.cLL + m.cHL);
  put(HXO_HL_PH, 0.0); put(HXO_LL_PH, 0.0); put(HXO_ATMOS_C, m.atmos);
  {
    double v = 0, d = 0, s = 0, p = 0, th = 0;
#pragma unroll
    for (int b = 0; b < B; ++b) { v += m.veg[b]; d += m.det[b]; s += m.soil[b];
                                   p += m.pf[b]; th += m.thawed[b]; }
    put(HXO_PERMAFROST_C, p); put(HXO_VEG_C, v); put(HXO_DET_C, d);
    put(HXO_SOIL_C, s); put(HXO_THAWED_C, th);
  }
  put(HXO_HEATFLUX, 0.0); put(HXO_CH4, kc.M0f); put(HXO_O3, 0.0);
  put(HXO_EARTH_C, m.earth); put(HXO_NBP, 0.0); put(HXO_OCEAN_UPTAKE, 0.0);
  put(HXO_NSTASH, 0.0); put(HXO_NSTEPS, 0.0);
  put(HXO_C_HL, m.cHL); put(HXO_C_LL, m.cLL); put(HXO_C_IO, m.cIO); put(HXO_C_DO, m.cDO);
  put(HXO_F_FROZEN, 1.0); put(HXO_TAU_OH, kc.TOH0);
#pragma unroll
  for (int b = 0; b < B; ++b) {
    put(HXO_BIOME0 + 0 * HX_MAXB + b, m.veg[b]); put(HXO_BIOME0 + 1 * HX_MAXB + b, m.det[b]);
    put(HXO_BIOME0 + 2 * HX_MAXB + b, m.soil[b]); put(HXO_BIOME0 + 3 * HX_MAXB + b, m.pf[b]);
    put(HXO_BIOME0 + 4 * HX_MAXB + b, m.thawed[b]);
  }
  if (spinup_steps) spinup_steps[mem] = steps;
}

// ===========================================================================
// Main run: years (iy_from, iy_to] (indices relative to startDate).  Each year
// is three phases separated by compiler fences:
//   A  gases, ocean year start, slow parameters   (year-level constants from HBM)
//   B  carbon-cycle solver                        (registers only)
//   C  forcing, DOECLIM, outputs                  (year-level constants from HBM)
// Year-level state (Tland, SST, forcing[t-1], CH4, Q10 window...) lives in the HBM
// state table between phases; the solver's pools stay in registers for the launch.
// ===========================================================================
// CON: the extended kernel -- the scenario holds constraints (HxConst::con_mask), a member has
// a land-ocean warming ratio, or diagnostics beyond HXO_SST_LO are recorded.  A separate
// instantiation, so that plain runs carry none of it.
template <int B, bool HF, bool KERPM, bool CON>
__global__ __launch_bounds__(64) void hx_run_kernel(const HxArgs *__restrict__ args,
                                                    int iy_from, int iy_to) {
  // LDS: the SSTs produced inside this launch's block of years (<= HX_DBLK), per lane
  // (multi-biome kernels need that LDS for the per-biome arrays and re-read the block's SSTs
  // from the output array instead)
  __shared__ double s_tblk[B == 1 ? HX_DBLK : 1][64];
  const int lane = threadIdx.x;
  const int mem = blockIdx.x * 64 + lane;
  if (mem >= args->buf.npad) return;
  __shared__ double s_park[hx_npark<B>()][64];
  Member<B> m;
  bind_member<B>(args->buf, mem, m, s_park, lane);
  load_state<B>(args->buf, mem, m);
  {  // year-level state -> park
    const HxBuffers &buf = args->buf;
    PKM(m, PK_CH4) = lds_(buf, HXS_CH4, mem); PKM(m, PK_SST) = lds_(buf, HXS_SST, mem);
    PKM(m, PK_EOS) = lds_(buf, HXS_EOS_VEGC, mem); PKM(m, PK_TLAND) = lds_(buf, HXS_TLAND, mem);
    PKM(m, PK_TWIN) = lds_(buf, HXS_TWIN, mem); PKM(m, PK_TL_M1) = lds_(buf, HXS_TL_M1, mem);
    PKM(m, PK_TL_M2) = lds_(buf, HXS_TL_M2, mem); PKM(m, PK_F_PREV) = lds_(buf, HXS_F_PREV, mem);
    PKM(m, PK_BASE_TOT) = lds_(buf, HXS_BASE_TOT, mem);
    PKM(m, PK_BASE_CO2) = lds_(buf, HXS_BASE_CO2, mem);
    PKM(m, PK_LN_CH4) = log(lds_(buf, HXS_CH4, mem));
    PKM(m, PK_LN_CO2R) = log(hx_div(m.atmos * PGC2PPM, m.C0));
#pragma unroll
    for (int b = 0; b < B; ++b)
      PKM(m, PK_FFROZEN0 + b) = lds_(buf, HXS_NGLOBAL + b * HXSB_N + HXSB_F_FROZEN, mem);
  }
  constexpr bool want_hf = HF;  // heat-flux diagnostic needs a second history sum
  int blk0 = -1;  // first year index of the current DOECLIM block
  if constexpr (CON) m.bufp = &args->buf;

  for (int iy = iy_from + 1; iy <= iy_to; ++iy) {
    HX_FENCE();
    if constexpr (CON) m.iy = iy;
    double ch4, o3;
    // ======================= phase A ========================================
    {
      const HxBuffers &buf = args->buf;
      const HxConst &kc = args->kc;
      hx_ccd sh = HX_CCD(buf.shared) + (size_t)iy * HXSH_STRIDE;
      // every HBM value this phase needs, issued back to back (one exposed latency:
      // with one wavefront per SIMD nothing else hides it)
      const double prev_ch4 = PKM(m, PK_CH4);
      double sst = PKM(m, PK_SST);
      const double eos = PKM(m, PK_EOS);
      double tland = PKM(m, PK_TLAND);
      if constexpr (CON) {
        // land-ocean warming ratio: the carbon cycle and the ocean see temperatures derived
        // from global tas, DOECLIM keeps its own (temperature_component.cpp:586-625,722-739)
        const double lo = ldp(buf, HXP_LO_RATIO, mem);
        if (lo != 0 && iy > 1) {
          const double tg = D_flnd * tland + (1.0 - D_flnd) * D_bsi * sst;
          const double toa = tg / ((lo * D_flnd) + (1 - D_flnd));
          tland = toa * lo;
          sst = toa / D_bsi;
        }
      }
      double twin = PKM(m, PK_TWIN);
      const double tl_m2 = PKM(m, PK_TL_M2);
      const int iold = iy - 203;
      const double tl_old =
          HX_GCD(buf.out[HXO_TLAND])[(size_t)(iold >= 1 ? iold : 0) * buf.npad + mem];
      double p_beta[B], p_wf[B], p_lnq10[B], p_mu[B], p_sigma[B], s_ffrozen[B];
      LandK<B> lk;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        s_ffrozen[b] = PKM(m, PK_FFROZEN0 + b);
        if constexpr (B == 1) {
          constexpr int o = hx_pkb1<B>();
          p_beta[b] = PKM(m, o + PKB_BETA); p_wf[b] = PKM(m, o + PKB_WF);
          p_mu[b] = PKM(m, o + PKB_MU); p_sigma[b] = PKM(m, o + PKB_SIGMA);
          p_lnq10[b] = PKM(m, o + PKB_LNQ10);
          lk.fpf_static[b] = PKM(m, o + PKB_FPF_STATIC);
          lk.rh_ch4_frac[b] = PKM(m, o + PKB_RH_CH4_FRAC);
        } else {
          const int pr = HXP_NGLOBAL + b * HXPB_N;
          p_wf[b] = ldp(buf, pr + HXPB_WF, mem);
          if (buf.uni_bio) {  // beta, permafrost mu/sigma uniform over members: scalar loads
            hx_ccd u = HX_CCD(buf.uparams) + pr;
            p_beta[b] = u[HXPB_BETA]; p_mu[b] = u[HXPB_PF_MU]; p_sigma[b] = u[HXPB_PF_SIGMA];
          } else {
            p_beta[b] = ldp(buf, pr + HXPB_BETA, mem);
            p_mu[b] = ldp(buf, pr + HXPB_PF_MU, mem);
            p_sigma[b] = ldp(buf, pr + HXPB_PF_SIGMA, mem);
          }
          p_lnq10[b] = ldd(buf, HXD_NGLOBAL + b, mem);
          if (m.upar) {
            lk.fpf_static[b] = m.upar[pr + HXPB_FPF_STATIC];
            lk.rh_ch4_frac[b] = m.upar[pr + HXPB_RH_CH4_FRAC];
          } else {
            lk.fpf_static[b] = ldp(buf, pr + HXPB_FPF_STATIC, mem);
            lk.rh_ch4_frac[b] = ldp(buf, pr + HXPB_RH_CH4_FRAC, mem);
          }
        }
      }
      // ---- OH, CH4, O3 ----
      double rh_ch4 = 0;  // D_RH_CH4 as recorded at the end of last year
      if (iy > 1) {
#pragma unroll
        for (int b = 0; b < B; ++b) rh_ch4 += m_rh_tp_ch4(m, lk, b);
      }
      double toh = 0.0;
      if (prev_ch4 != kc.M0)
        toh = ((kc.CCH4 * (PKM(m, PK_LN_CH4) - kc.lnM0) + sh[HXSH_OH_B]) + sh[HXSH_OH_C]) +
              sh[HXSH_OH_D];
      const double tau_oh = kc.TOH0 * exp(-toh);
      if constexpr (CON) {
      if (buf.out[HXO_TAU_OH]) sto_(buf, HXO_TAU_OH, (size_t)iy * buf.npad + mem, tau_oh);
      if (buf.stash_diag) {  // sums over the year's stashes start at zero (oceanbox::new_year)
        const size_t o = (size_t)iy * buf.npad + mem;
        if (buf.out[HXO_HL_UPTAKE]) sto_(buf, HXO_HL_UPTAKE, o, 0.0);
        if (buf.out[HXO_LL_UPTAKE]) sto_(buf, HXO_LL_UPTAKE, o, 0.0);
        if (buf.out[HXO_HL_DO]) sto_(buf, HXO_HL_DO, o, 0.0);
      }
      }
      {
        double ch4_em = sh[HXSH_CH4_EM];
        if constexpr (CON) {
          if (buf.mseries[HXM_CH4_EM])
            ch4_em = HX_GCD(buf.mseries[HXM_CH4_EM])[(size_t)iy * buf.npad + mem];
        }
        const double emisTocon =
            ((ch4_em + rh_ch4 * PG_C_TO_TG_CH4) + sh[HXSH_CH4N]) * kc.inv_UC_CH4;
        const double dCH4 = ((emisTocon - prev_ch4 * kc.inv_Tsoil) - prev_ch4 * kc.inv_Tstrat) -
                            hx_div(prev_ch4, tau_oh);
        ch4 = prev_ch4 + dCH4;
      }
      if constexpr (CON) {  // ch4_component.cpp:156-157
        if (kc.con_mask & HXC_CH4) {
          const double c = sh[HXSH_CH4_CON];
          if (!isnan(c)) ch4 = c;
        }
      }
      PKM(m, PK_CH4) = ch4;
      const double ln_ch4 = log(ch4);
      PKM(m, PK_LN_CH4) = ln_ch4;
      o3 = ((5 * ln_ch4 + sh[HXSH_O3_NOX]) + sh[HXSH_O3_CO]) + sh[HXSH_O3_NMVOC];
      // ---- ocean: new year ----
      chem_constants2(sst + 18 + (-16.4), sst + 18 + 2.9, m.kH, m.kL);
      m.annualflux_sum = 0; m.nstash = 0; m.nsteps = 0;
      // (the alkalinities were tuned once, right after the spinup: hx_alk_kernel)
      chem_solve2(m.kH, m.kL, m.cHL, m.cLL, m.alkH, m.alkL, m.hH, m.hL, m.pco2H, m.pco2L,
                  m.status);
      m.chem_fresh = true;
      // ---- slowparameval (t = year-1) ----
      m.ffi = sh[HXSH_FFI]; m.daccs = sh[HXSH_DACCS];
      m.luc_e = sh[HXSH_LUC_E]; m.luc_u = sh[HXSH_LUC_U];
      if constexpr (CON) {  // emissions that differ between members
        const size_t o = (size_t)iy * buf.npad + mem;
        if (buf.mseries[HXM_FFI]) m.ffi = HX_GCD(buf.mseries[HXM_FFI])[o];
        if (buf.mseries[HXM_DACCS]) m.daccs = HX_GCD(buf.mseries[HXM_DACCS])[o];
        if (buf.mseries[HXM_LUC_E]) m.luc_e = HX_GCD(buf.mseries[HXM_LUC_E])[o];
        if (buf.mseries[HXM_LUC_U]) m.luc_u = HX_GCD(buf.mseries[HXM_LUC_U])[o];
      }
      m.npp_luc_adjust = hx_div(eos - m.cum_luc_va, eos);
      const double lnc = PKM(m, PK_LN_CO2R);  // = log((atmos C * PGC2PPM) / C0), from last year's phase C
      // Q10 window: mean over i in [t-200, t-1] of Tland_record(i) =
      // Tland(i-1), 0 before the first record (runtime.cpp:1041-1052)
      if (iy >= 3) {
        twin += tl_m2;  // Tland of year iy-3 enters
        if (iold >= 1) twin -= tl_old;
        PKM(m, PK_TWIN) = twin;
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        m.co2fert[b] = 1 + p_beta[b] * lnc;
        const double Tb = tland * p_wf[b];
        m.tempfertd[b] = exp(p_lnq10[b] * (Tb * 0.1));
        m.f_new_thaw[b] = 0.0;
        if (m.pf[b] != 0.0) {
          double ff = 1.0;
          if (Tb > 0) {
            const double d = hx_div(log(Tb) - p_mu[b], p_sigma[b] * 1.4142135623730951);
            ff = 1 - erfc(-d) / 2;
          }
          m.f_new_thaw[b] = s_ffrozen[b] - ff;
          PKM(m, PK_FFROZEN0 + b) = ff;
        }
        const double Trm = (iy > 1) ? (twin * p_wf[b]) * 0.005 : 0.0;
        const double tfs = exp(p_lnq10[b] * (Trm * 0.1));
        const double last = (iy > 1) ? m.tempferts[b] : 0.0;
        m.tempferts[b] = fmax(tfs, last);  // sticky :1054-1059
      }
    }
    HX_FENCE();
    // ======================= phase B: carbon-cycle solver ====================
    {
      const double year = (double)(args->kc.start_year + iy);
      YearCon yc{};
      if constexpr (CON) {
        hx_ccd sh = HX_CCD(args->buf.shared) + (size_t)iy * HXSH_STRIDE;
        yc.mask = args->kc.con_mask;
        yc.co2 = sh[HXSH_CO2_CON];
        yc.nbp_hi = sh[HXSH_NBP_CON];
        yc.nbp_lo = (sh - HXSH_STRIDE)[HXSH_NBP_CON];
        yc.t_half = year - 0.5;
      }
      solve_year<B, false, CON>(m, args->kc, year - 1.0, year, yc);
    }
    HX_FENCE();
    // ======================= phase C ========================================
    {
      const HxBuffers &buf = args->buf;
      const HxConst &kc = args->kc;
      const int ns = kc.ns;
      hx_ccd sh = HX_CCD(buf.shared) + (size_t)iy * HXSH_STRIDE;
      if (blk0 < 0 || iy >= blk0 + HX_DBLK) {
        // new DOECLIM block: this lane's partial sums over its SST history
        blk0 = iy;
        doeclim_pass_dev<KERPM, HF>(buf.out[HXO_SST], buf.ker, const_cast<double *>(buf.dpart),
                                    const_cast<double *>(buf.dpart2), ns, buf.npad, blk0, mem);
        HX_FENCE();
      }
      // every HBM value this phase needs, issued back to back
      const double tland = PKM(m, PK_TLAND), sst = PKM(m, PK_SST);
      const double f_prev = PKM(m, PK_F_PREV);
      const double base_tot = PKM(m, PK_BASE_TOT), base_co2 = PKM(m, PK_BASE_CO2);
      const double tl_m1 = PKM(m, PK_TL_M1);
      const double p_aero = PKM(m, PK_AERO), p_vol = PKM(m, PK_VOL);
#define HXDK(row) PKM(m, PK_D0 + ((row) - HXD_A0))
      const double dA0 = HXDK(HXD_A0), dA1 = HXDK(HXD_A1), dA2 = HXDK(HXD_A2), dA3 = HXDK(HXD_A3),
                   dIB0 = HXDK(HXD_IB0), dIB1 = HXDK(HXD_IB1), dIB2 = HXDK(HXD_IB2),
                   dIB3 = HXDK(HXD_IB3), dQC1 = HXDK(HXD_QC1), dQC2 = HXDK(HXD_QC2),
                   dDQ1 = HXDK(HXD_DQ1), dDQ2 = HXDK(HXD_DQ2), dDPS = HXDK(HXD_DPSCALE),
                   dHFS = want_hf ? HXDK(HXD_HFSCALE) : 0.0;
#undef HXDK
      const int jb = iy - blk0;
      double dpast = HX_GCD(buf.dpart)[(size_t)jb * buf.npad + mem];
      double hint = want_hf ? HX_GCD(buf.dpart2)[(size_t)jb * buf.npad + mem] : 0.0;
      // ---- forcing ----
      const double co2c = m.atmos * PGC2PPM;
      const double ln_co2r = log(hx_div(co2c, m.C0));
      PKM(m, PK_LN_CO2R) = ln_co2r;
      double rf_tot = 0, rf_co2 = 0;
      if (iy >= kc.baseyear_idx) {
        const double a1 = -2.4785e-7, b1 = 7.5906e-4, c1 = -2.1492e-3, d1 = 5.2488;
        const double a2 = -3.4197e-4, b2 = 2.5455e-4, c2 = -2.4357e-4, d2 = 0.12173;
        const double a3 = -8.9603e-5, b3 = -1.2462e-4, d3 = 0.045194;
        const double sqN = sh[HXSH_SQRT_N2O], sqM = sqrt(ch4), sqC = sqrt(co2c);
        const double C_alpha_max = m.C0 - (b1 / (2 * a1));
        double alpha_prime;
        if (co2c > C_alpha_max) alpha_prime = d1 - ((b1 * b1) / (4 * a1));
        else if (m.C0 < co2c && co2c < C_alpha_max)
          alpha_prime = d1 + a1 * ((co2c - m.C0) * (co2c - m.C0)) + b1 * (co2c - m.C0);
        else alpha_prime = d1;
        const double sarf_co2 = (alpha_prime + c1 * sqN) * ln_co2r;
        const double fco2 = (sarf_co2 * kc.delta_co2) + sarf_co2;
        const double sarf_n2o = (a2 * sqC + b2 * sqN + c2 * sqM + d2) * (sqN - kc.sqrtN0);
        const double fn2o = (kc.delta_n2o * sarf_n2o) + sarf_n2o;
        const double sarf_ch4 = (a3 * sqM + b3 * sqN + d3) * (sqM - kc.sqrtM0);
        const double fch4 = (kc.delta_ch4 * sarf_ch4) + sarf_ch4;
        const double fh2o = 0.0485 * ((ch4 - kc.M0f) * kc.inv_h2o_span);
        const double fo3 = 0.042 * o3;
        double ftot = ((((((fco2 + fn2o) + fch4) + fh2o) + fo3) + sh[HXSH_RF_OTHER]) +
                       p_aero * sh[HXSH_RF_AERO]) +
                      p_vol * sh[HXSH_RF_VOL];
        if constexpr (CON) {  // forcing_component.cpp:498-505
          if (kc.con_mask & HXC_FTOT) {
            const double c = sh[HXSH_FTOT_CON];
            if (!isnan(c)) ftot = c;
          }
        }
        if (iy == kc.baseyear_idx) {
          PKM(m, PK_BASE_TOT) = ftot;
          PKM(m, PK_BASE_CO2) = fco2;
          rf_tot = 0; rf_co2 = 0;  // x - x
        } else {
          rf_tot = ftot - base_tot;
          rf_co2 = fco2 - base_co2;
        }
      }
      // ---- DOECLIM: history before the block (hx_doeclim_pass_kernel) + in-block terms ----
      double tl_new, sst_new, heatflux = 0, tgav, flux_mixed = 0, flux_interior = 0;
      {
        const int j = jb;
        // Ker is stored with HX_KPAD zeros in front: entry k lives at k + HX_KPAD
        const int kq = ns - iy - 1 + HX_KPAD;
        auto ldk = [&](int idx) -> double {
          if constexpr (KERPM) return HX_GCD(buf.ker)[(size_t)idx * buf.npad + mem];
          else return HX_CCD(buf.ker)[idx];
        };
        for (int i = blk0; i < iy; ++i) {
          double T;
          if constexpr (B == 1) T = s_tblk[i - blk0][lane];
          else T = HX_GCD(buf.out[HXO_SST])[(size_t)i * buf.npad + mem];
          dpast += T * ldk(kq + i);
          if (want_hf) hint += T * ldk(kq + i + 1);
        }
        dpast *= dDPS;
        const double DelQ = rf_tot - f_prev;
        const double DQ1 = dDQ1 * (rf_tot + f_prev) + DelQ * dQC1;
        const double DQ2 = dDQ2 * (rf_tot + f_prev) + DelQ * dQC2;
        const double X1 = DQ1 + (dA0 * tland + dA1 * sst);
        const double X2 = (DQ2 + dpast) + (dA2 * tland + dA3 * sst);
        tl_new = dIB0 * X1 + dIB1 * X2;
        sst_new = dIB2 * X1 + dIB3 * X2;
        tgav = D_flnd * tl_new + (1.0 - D_flnd) * D_bsi * sst_new;
        if constexpr (CON) {  // user-supplied temperature :510-525
          if (kc.con_mask & HXC_TAS) {
            const double c = sh[HXSH_TAS_CON];
            if (!isnan(c)) {
              tgav = c;
              tl_new = (tgav - (1.0 - D_flnd) * D_bsi * sst_new) / D_flnd;
              sst_new = (tgav - D_flnd * tl_new) / ((1.0 - D_flnd) * D_bsi);
            }
          }
        }
        if (want_hf) {
          const double hmix = D_cas * (sst_new - sst);
          const double hi = dHFS * (2.0 * sst_new - hint);
          heatflux = hmix + D_fso * hi;
          flux_mixed = hmix;
          flux_interior = hi;
        }
        if constexpr (B == 1) s_tblk[j][lane] = sst_new;
      }
      double tl_seen = tland, tl_rep = tl_new, sst_rep = sst_new;  // what D_LAND_TAS / D_SST return
      if constexpr (CON) {
        const double lo = ldp(buf, HXP_LO_RATIO, mem);
        if (lo != 0) {
          if (iy > 1) {
            const double tg0 = D_flnd * tland + (1.0 - D_flnd) * D_bsi * sst;
            tl_seen = (tg0 / ((lo * D_flnd) + (1 - D_flnd))) * lo;
          }
          const double toa = tgav / ((lo * D_flnd) + (1 - D_flnd));
          tl_rep = toa * lo;
          sst_rep = toa / D_bsi;
        }
      }
      PKM(m, PK_F_PREV) = rf_tot;
      PKM(m, PK_TL_M2) = tl_m1;  // Tland of years iy-2, iy-1
      PKM(m, PK_TL_M1) = tl_seen;  // for the next year
      PKM(m, PK_TLAND) = tl_new;
      PKM(m, PK_SST) = sst_new;
      // ---- outputs ----
      const size_t o = (size_t)iy * buf.npad + mem;
      sto_(buf, HXO_SST, o, sst_new);
      sto_(buf, HXO_TLAND, o, tl_rep);
      if constexpr (CON) { if (buf.out[HXO_SST_LO]) sto_(buf, HXO_SST_LO, o, sst_rep); }
      if (buf.out[HXO_CO2]) sto_(buf, HXO_CO2, o, co2c);
      if (buf.out[HXO_TGAV]) sto_(buf, HXO_TGAV, o, tgav);
      if (buf.out[HXO_RF_TOT]) sto_(buf, HXO_RF_TOT, o, rf_tot);
      if (buf.out[HXO_RF_CO2]) sto_(buf, HXO_RF_CO2, o, rf_co2);
      if (buf.out[HXO_OCEAN_C]) sto_(buf, HXO_OCEAN_C, o, m.cDO + m.cIO + m.cLL + m.cHL);
      if (buf.out[HXO_HL_PH]) sto_(buf, HXO_HL_PH, o, -log10(m.hH));
      if (buf.out[HXO_LL_PH]) sto_(buf, HXO_LL_PH, o, -log10(m.hL));
      if (buf.out[HXO_ATMOS_C]) sto_(buf, HXO_ATMOS_C, o, m.atmos);
      if (buf.out[HXO_HEATFLUX]) sto_(buf, HXO_HEATFLUX, o, heatflux);
      if (buf.out[HXO_CH4]) sto_(buf, HXO_CH4, o, ch4);
      if (buf.out[HXO_O3]) sto_(buf, HXO_O3, o, o3);
      if (buf.out[HXO_EARTH_C]) sto_(buf, HXO_EARTH_C, o, m.earth);
      if (buf.out[HXO_NBP]) sto_(buf, HXO_NBP, o, m.nbp);
      if (buf.out[HXO_OCEAN_UPTAKE]) sto_(buf, HXO_OCEAN_UPTAKE, o, m.annualflux_sum);
      if (buf.out[HXO_NSTASH]) sto_(buf, HXO_NSTASH, o, (double)m.nstash);
      if (buf.out[HXO_NSTEPS]) sto_(buf, HXO_NSTEPS, o, (double)m.nsteps);
      if (buf.out[HXO_PERMAFROST_C] || buf.out[HXO_VEG_C] || buf.out[HXO_DET_C] ||
          buf.out[HXO_SOIL_C] || buf.out[HXO_THAWED_C]) {
        double v = 0, d = 0, s = 0, p = 0, th = 0;
#pragma unroll
        for (int b = 0; b < B; ++b) { v += m.veg[b]; d += m.det[b]; s += m.soil[b];
                                       p += m.pf[b]; th += m.thawed[b]; }
        if (buf.out[HXO_PERMAFROST_C]) sto_(buf, HXO_PERMAFROST_C, o, p);
        if (buf.out[HXO_VEG_C]) sto_(buf, HXO_VEG_C, o, v);
        if (buf.out[HXO_DET_C]) sto_(buf, HXO_DET_C, o, d);
        if (buf.out[HXO_SOIL_C]) sto_(buf, HXO_SOIL_C, o, s);
        if (buf.out[HXO_THAWED_C]) sto_(buf, HXO_THAWED_C, o, th);
      }
      if constexpr (CON) {  // diagnostics of the extended kernel
      if (buf.out[HXO_GMST]) sto_(buf, HXO_GMST, o, D_flnd * tl_new + (1.0 - D_flnd) * sst_new);
      if (buf.out[HXO_FLUX_MIXED]) sto_(buf, HXO_FLUX_MIXED, o, flux_mixed);
      if (buf.out[HXO_FLUX_INTERIOR]) sto_(buf, HXO_FLUX_INTERIOR, o, flux_interior);
      if (buf.out[HXO_C_HL]) sto_(buf, HXO_C_HL, o, m.cHL);
      if (buf.out[HXO_C_LL]) sto_(buf, HXO_C_LL, o, m.cLL);
      if (buf.out[HXO_C_IO]) sto_(buf, HXO_C_IO, o, m.cIO);
      if (buf.out[HXO_C_DO]) sto_(buf, HXO_C_DO, o, m.cDO);
      if (buf.out[HXO_PCO2_HL]) sto_(buf, HXO_PCO2_HL, o, m.pco2H);
      if (buf.out[HXO_PCO2_LL]) sto_(buf, HXO_PCO2_LL, o, m.pco2L);
      if constexpr (B > 1) {  // "<biome>.veg_c" ...: the pools of each biome
#pragma unroll
        for (int b = 0; b < B; ++b) {
          if (buf.out[HXO_BIOME0 + 0 * HX_MAXB + b]) sto_(buf, HXO_BIOME0 + 0 * HX_MAXB + b, o, m.veg[b]);
          if (buf.out[HXO_BIOME0 + 1 * HX_MAXB + b]) sto_(buf, HXO_BIOME0 + 1 * HX_MAXB + b, o, m.det[b]);
          if (buf.out[HXO_BIOME0 + 2 * HX_MAXB + b]) sto_(buf, HXO_BIOME0 + 2 * HX_MAXB + b, o, m.soil[b]);
          if (buf.out[HXO_BIOME0 + 3 * HX_MAXB + b]) sto_(buf, HXO_BIOME0 + 3 * HX_MAXB + b, o, m.pf[b]);
          if (buf.out[HXO_BIOME0 + 4 * HX_MAXB + b]) sto_(buf, HXO_BIOME0 + 4 * HX_MAXB + b, o, m.thawed[b]);
        }
      }
      if (buf.out[HXO_RH_CH4] || buf.out[HXO_F_FROZEN]) {
        // record_state: RH_ch4 = rh_ftpa_ch4 of the year-end pools (simpleNbox.cpp:800-812);
        // f_frozen: permafrost-weighted mean over biomes, 1 without permafrost (:492-514)
        LandK<B> lk;
        load_landk<B>(m, lk);
        double rch4 = 0, ptot = 0, ff = 0;
#pragma unroll
        for (int b = 0; b < B; ++b) { rch4 += m_rh_tp_ch4(m, lk, b); ptot += m.pf[b]; }
        if (ptot > 0.0) {
#pragma unroll
          for (int b = 0; b < B; ++b) ff += (m.pf[b] / ptot) * PKM(m, PK_FFROZEN0 + b);
        } else ff = 1.0;
        if (buf.out[HXO_RH_CH4]) sto_(buf, HXO_RH_CH4, o, rch4);
        if (buf.out[HXO_F_FROZEN]) sto_(buf, HXO_F_FROZEN, o, ff);
      }
      }
      if (buf.hist) {  // Core::reset(date) needs every component's state of every year
        double *slab = buf.hist + (size_t)iy * (size_t)HX_NSTATE(B) * buf.npad;
        store_state<B>(buf, mem, m, slab);
        store_park_state<B>(buf, mem, m, slab);
      }
    }
  }
  HX_FENCE();
  store_state<B>(args->buf, mem, m);
  store_park_state<B>(args->buf, mem, m);
}

// ===========================================================================
// Broadcast member 0's state/outputs row to every member (shared spinup).
// ===========================================================================
__global__ void hx_broadcast_rows_kernel(double *table, int nrows, int npad) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  if (mem >= npad || mem == 0) return;
  for (int r = 0; r < nrows; ++r) table[(size_t)r * npad + mem] = table[(size_t)r * npad];
}
__global__ void hx_broadcast_u32_kernel(unsigned *v, int npad) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  if (mem >= npad || mem == 0) return;
  v[mem] = v[0];
}

// ===========================================================================
// Output gather: members are assigned to lanes in a behaviour-sorted order (see
// EnsembleCore::upload_params); results go back to the caller in member order.
// dst[y][member] = src[y][lane_of_member[member]]
// ===========================================================================
__global__ __launch_bounds__(256) void hx_gather_kernel(const double *src, const int *lane_of_member,
                                                        double *dst, int n, int npad, int nyears) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (mem >= n || y >= nyears) return;
  dst[(size_t)y * n + mem] = src[(size_t)y * npad + lane_of_member[mem]];
}

// ===========================================================================
// Per-year ensemble statistics of one output variable over members [0, n):
// count, sum, sum of squares, min, max -> stats[year][5].  One workgroup per
// year; wave-level DPP/shuffle reduction, then one LDS hop across the waves.
// ===========================================================================
__global__ __launch_bounds__(256) void hx_stats_kernel(const double *var, int n, int npad,
                                                       int iy0, double *stats) {
  const int iy = iy0 + blockIdx.x;
  const double *row = var + (size_t)iy * npad;
  // pure streaming: four 16-byte loads in flight per lane (rows are 512-byte aligned: npad is a
  // multiple of 64), four independent accumulator sets
  double sa[4] = {0, 0, 0, 0}, qa[4] = {0, 0, 0, 0};
  double mn = INFINITY, mx = -INFINITY;
  const int n2 = n >> 1;                       // number of double2 elements
  const double2 *row2 = reinterpret_cast<const double2 *>(row);
  int i = threadIdx.x;
  for (; i + 3 * (int)blockDim.x < n2; i += 4 * blockDim.x) {
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = row2[i + k * (int)blockDim.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sa[k] += v[k].x + v[k].y;
      qa[k] += v[k].x * v[k].x + v[k].y * v[k].y;
      mn = fmin(mn, fmin(v[k].x, v[k].y));
      mx = fmax(mx, fmax(v[k].x, v[k].y));
    }
  }
  for (; i < n2; i += blockDim.x) {
    const double2 v = row2[i];
    sa[0] += v.x + v.y; qa[0] += v.x * v.x + v.y * v.y;
    mn = fmin(mn, fmin(v.x, v.y)); mx = fmax(mx, fmax(v.x, v.y));
  }
  if ((n & 1) && threadIdx.x == 0) {
    const double v = row[n - 1];
    sa[0] += v; qa[0] += v * v; mn = fmin(mn, v); mx = fmax(mx, v);
  }
  double s = (sa[0] + sa[1]) + (sa[2] + sa[3]), s2 = (qa[0] + qa[1]) + (qa[2] + qa[3]);
  double cnt = (threadIdx.x == 0) ? (double)n : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64); s2 += __shfl_down(s2, off, 64);
    cnt += __shfl_down(cnt, off, 64);
    mn = fmin(mn, __shfl_down(mn, off, 64)); mx = fmax(mx, __shfl_down(mx, off, 64));
  }
  __shared__ double red[4][5];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = cnt; red[w][1] = s; red[w][2] = s2;
                                 red[w][3] = mn; red[w][4] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) { red[0][0] += red[k][0]; red[0][1] += red[k][1];
      red[0][2] += red[k][2]; red[0][3] = fmin(red[0][3], red[k][3]);
      red[0][4] = fmax(red[0][4], red[k][4]); }
    double *o = stats + (size_t)blockIdx.x * 5;
    o[0] = red[0][0]; o[1] = red[0][1]; o[2] = red[0][2]; o[3] = red[0][3]; o[4] = red[0][4];
  }
}

// ===========================================================================
// DOECLIM convolution kernel table Ker[i] (temperature_component.cpp:303-371)
// for `count` diffusivities: ker[i * stride + mem].  count = 1, stride = 1 when
// every member shares the diffusivity.
// ===========================================================================
__global__ __launch_bounds__(256) void hx_doeclim_table_kernel(const double *diff_row,
                                                               double *ker, int ns, int count,
                                                               int stride) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (mem >= count || i >= ns) return;
  const double keff = (D_secs / 10000) * diff_row[mem];
  const double tb = (D_zbot * D_zbot) / keff;  // taubot / dt, dt = 1
  const double sq2 = sqrt(2.0), sqpt = sqrt(M_PI * tb);
  double KT0, KTA1, KTB1, KTA2, KTB2, KTA3, KTB3;
  if (i == ns - 1) {
    KT0 = 4.0 - 2.0 * sq2;
    KTA1 = -8.0 * exp(-tb) + 4.0 * sq2 * exp(-0.5 * tb);
    KTB1 = 4.0 * sqpt * (1.0 + erf(sqrt(0.5 * tb)) - 2.0 * erf(sqrt(tb)));
    KTA2 = 8.0 * exp(-4.0 * tb) - 4.0 * sq2 * exp(-2.0 * tb);
    KTB2 = -8.0 * sqpt * (1.0 + erf(sqrt(2.0 * tb)) - 2.0 * erf(2.0 * sqrt(tb)));
    KTA3 = -8.0 * exp(-9.0 * tb) + 4.0 * sq2 * exp(-4.5 * tb);
    KTB3 = 12.0 * sqpt * (1.0 + erf(sqrt(4.5 * tb)) - 2.0 * erf(3.0 * sqrt(tb)));
  } else {
    const double a = (double)(ns - i), b = (double)(ns + 1 - i), c = (double)(ns - 1 - i);
    const double ra = sqrt(a), rb = sqrt(b), rc = sqrt(c);
    const double ua = sqrt(tb / a), ub = sqrt(tb / b), uc = sqrt(tb / c);
    KT0 = 4.0 * ra - 2.0 * rb - 2.0 * rc;
    KTA1 = -8.0 * ra * exp(-tb / a) + 4.0 * rb * exp(-tb / b) + 4.0 * rc * exp(-tb / c);
    KTB1 = 4.0 * sqpt * (erf(uc) + erf(ub) - 2.0 * erf(ua));
    KTA2 = 8.0 * ra * exp(-4.0 * tb / a) - 4.0 * rb * exp(-4.0 * tb / b) -
           4.0 * rc * exp(-4.0 * tb / c);
    KTB2 = -8.0 * sqpt * (erf(2.0 * uc) + erf(2.0 * ub) - 2.0 * erf(2.0 * ua));
    KTA3 = -8.0 * ra * exp(-9.0 * tb / a) + 4.0 * rb * exp(-9.0 * tb / b) +
           4.0 * rc * exp(-9.0 * tb / c);
    KTB3 = 12.0 * sqpt * (erf(3.0 * uc) + erf(3.0 * ub) - 2.0 * erf(3.0 * ua));
  }
  ker[(size_t)(i + HX_KPAD) * stride + mem] = KT0 + KTA1 + KTB1 + KTA2 + KTB2 + KTA3 + KTB3;
}

// ===========================================================================
// oceanbox::chem_equilibrate for both surface boxes (src/oceanbox.cpp:382-445,
// ocean_component.cpp:392-400): the first post-spinup ocean.run() turns the chemistry on and
// tunes each box's alkalinity so that it reproduces the spinup flux at the spun-up CO2.  Its
// inputs are all spinup results (box carbon, atmosphere, SST = 0), so it runs once, directly
// after the spinup kernel, and the run kernel finds the alkalinities in the state table.
// ===========================================================================
__global__ __launch_bounds__(64) void hx_alk_kernel(const HxArgs *__restrict__ args, int nmem) {
  const int mem = blockIdx.x * 64 + threadIdx.x;
  if (mem >= nmem) return;
  const HxBuffers &buf = args->buf;
  const double cHL = lds_(buf, HXS_C_HL, mem), cLL = lds_(buf, HXS_C_LL, mem);
  const double co2 = lds_(buf, HXS_ATMOS, mem) * PGC2PPM;
  const double sst = lds_(buf, HXS_SST, mem);
  unsigned status = HX_GU(buf.status)[mem];
  ChemK kH, kL;
  chem_constants2(sst + 18 + (-16.4), sst + 18 + 2.9, kH, kL);
  double hH = 0, hL = 0;
  const double alkH = equilibrate_alk(kH, cHL, 1.0 / O_vHL, O_AsHL, co2, 1.000, hH, status);
  const double alkL = equilibrate_alk(kL, cLL, 1.0 / O_vLL, O_AsLL, co2, -1.000, hL, status);
  sts_(buf, HXS_ALK_HL, mem, alkH); sts_(buf, HXS_ALK_LL, mem, alkL);
  sts_(buf, HXS_H_HL, mem, hH); sts_(buf, HXS_H_LL, mem, hL);
  HX_GU(buf.status)[mem] = status;
}

// ===========================================================================
// Diagnostics derived from recorded outputs, one (year, member) element per thread.
// The reference keeps them as members of the carbonate-chemistry object of the last solve
// (ocean_csys.cpp:328-366) or recomputes them on request (ocean_component.cpp:440-512);
// here they follow from what the run recorded:
//   [H+] = 10^-pH and pCO2 of the last solve, the box temperature Tbox = SST(year-1) + 18 +
//   deltaT (oceanbox.cpp:97-99, 309-323), and the year-end box carbon:
//   CO2* = pCO2 Kh, CO3 = CO2* K1 K2 / [H+]^2  (identical to DIC / (1 + h/K2 + h^2/(K1 K2)))
// ===========================================================================
__device__ __forceinline__ double diag_rf(int kind, const HxDiagArgs &a, int iy, int mem) {
  hx_ccd sh = HX_CCD(a.shared) + (size_t)iy * HXSH_STRIDE;
  const size_t o = (size_t)iy * a.npad + mem;
  const double a2 = -3.4197e-4, b2 = 2.5455e-4, c2 = -2.4357e-4, d2 = 0.12173;
  const double a3 = -8.9603e-5, b3 = -1.2462e-4, d3 = 0.045194;
  const double sqN = sh[HXSH_SQRT_N2O];
  if (kind == HXG_RF_O3) return 0.042 * a.o3[o];
  const double ch4 = a.ch4[o], sqM = sqrt(ch4);
  if (kind == HXG_RF_H2O) return 0.0485 * ((ch4 - a.M0f) / (1831 - a.M0f));
  if (kind == HXG_RF_CH4) {
    const double sarf = (a3 * sqM + b3 * sqN + d3) * (sqM - a.sqrtM0);
    return (a.delta_ch4 * sarf) + sarf;
  }
  const double sqC = sqrt(a.co2[o]);
  const double sarf = (a2 * sqC + b2 * sqN + c2 * sqM + d2) * (sqN - a.sqrtN0);
  return (a.delta_n2o * sarf) + sarf;
}

__global__ __launch_bounds__(256) void hx_diag_kernel(int kind, HxDiagArgs a, double *out) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  const int iy = a.iy0 + blockIdx.y;
  if (mem >= a.npad) return;
  const size_t o = (size_t)iy * a.npad + mem;
  double r = 0.0;
  if (kind >= HXG_RF_N2O) {
    // forcings are reported relative to the base year (forcing_component.cpp:507-527)
    if (iy >= a.base_idx) r = diag_rf(kind, a, iy, mem) - diag_rf(kind, a, a.base_idx, mem);
  } else if (kind == HXG_OCEAN_TAS) {
    const double lo = a.lo_ratio[mem];
    r = (lo != 0) ? a.tgav[o] / ((lo * D_flnd) + (1 - D_flnd)) : D_bsi * a.sst[o];
  } else {
    const double sst_prev = (iy >= 1) ? a.sst[o - a.npad] : 0.0;
    const double Tc = sst_prev + 18 + a.deltaT;
    // convertToDIC  ocean_csys.cpp:403-408 (umol/kg)
    const double dic = ((((a.carbon ? a.carbon[o] : 0.0) * 1e15) * (1.0 / 12.01)) * (1.0 / 1027.0) *
                        a.inv_vol) * 1e6;
    if (kind == HXG_TEMP) r = Tc;
    else if (kind == HXG_DIC) r = dic;
    else {
      ChemK k;
      chem_constants(Tc, k);
      const double h = exp10(-a.ph[o]);
      const double co2st = a.pco2[o] * k.Kh;            // umol/kg
      const double co3 = co2st * ((k.K1 * k.K2) / (h * h));  // umol/kg
      if (kind == HXG_CO3) r = co3;
      else if (kind == HXG_REVELLE) r = dic / co3;        // oceanbox.cpp:278-292
      else {
        const double S = O_S, Tk = Tc + 273.15, sqrtS = 5.873670062235365,
                     S15 = 202.64161714712009;
        double t1, t2, t3;
        if (kind == HXG_OMEGA_CA) {  // ocean_csys.cpp:266-274
          t1 = -171.9065 - 0.077993 * Tk + 2839.319 / Tk + 71.595 * log10(Tk);
          t2 = +(-0.77712 + 0.0028426 * Tk + 178.34 / Tk) * sqrtS;
          t3 = -0.07711 * S + 0.0041249 * S15;
        } else {
          t1 = -171.945 - 0.077993 * Tk + 2903.293 / Tk + 71.595 * log10(Tk);
          t2 = +(-0.068393 + 0.0017276 * Tk + 88.135 / Tk) * sqrtS;
          t3 = -0.10018 * S + 0.0059415 * S15;
        }
        const double Ksp = exp10(t1 + t2 + t3);
        const double calcium = 0.02128 / 40.087 * (S / 1.80655);
        r = (((co3 * 1e-6) * calcium) / Ksp);
      }
    }
  }
  out[(size_t)blockIdx.y * a.npad + mem] = r;
}

// slrComponent (slr_component.cpp:116-232, Vermeer & Rahmstorf 2009): sea-level rise from the
// global tas series of each member; one thread walks one member's years.  Nothing exists
// before the reference period 1951-1980 has been run; dT/dt is the derivative of the series as
// known when the date was computed (h_interpolator.cpp:132-167).  out: 4 arrays [ns][npad]
// (slr, sl_rc, slr_no_ice, sl_rc_no_ice), zero where the reference has no value.
__global__ __launch_bounds__(64) void hx_slr_kernel(const double *tgav, int npad, int start_year,
                                                    int iy_to, double *out, size_t var_stride) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  if (mem >= npad) return;
  const int lo = 1951 - start_year, hi = 1980 - start_year, first = 1;
  if (lo < first || iy_to < hi) return;
  auto TG = [&](int iy) { return tgav[(size_t)iy * npad + mem]; };
  double sum = 0.0;
  for (int i = lo; i <= hi; ++i) sum += TG(i);
  const double ref = sum / (hi - lo + 1);
  double slr = 0.0, slr_ni = 0.0;
  for (int iy = first; iy <= iy_to; ++iy) {
    const int last = (iy <= hi) ? hi : iy;  // the series' last date when this date was computed
    double dTdt = 0.0;
    if (last - first + 1 > 2) {
      if (iy == first) dTdt = (TG(first + 1) - TG(first)) / 1.0;
      else if (iy == last) dTdt = (TG(iy) - TG(iy - 1)) / 1.0;
      else dTdt = (((TG(iy) - TG(iy - 1)) / 1.0) + ((TG(iy + 1) - TG(iy)) / 1.0)) / 2.0;
    }
    const double T = TG(iy) - ref;
    const double dHdt = 0.56 * (T - (-0.41)) + (-4.9) * dTdt;
    const double dHdt_ni = 0.08 * (T - (-0.375)) + 2.5 * dTdt;
    slr = slr + dHdt;
    slr_ni = slr_ni + dHdt_ni;
    const size_t o = (size_t)iy * npad + mem;
    out[o] = slr;
    out[var_stride + o] = dHdt;
    out[2 * var_stride + o] = slr_ni;
    out[3 * var_stride + o] = dHdt_ni;
  }
}

// Unit vector of the carbonate chemistry (a7): n independent (T, carbon, alkalinity) triples
// through chem_constants + chem_solve from a cold start (the Fujiwara bound is not needed: the
// bracketed Newton accepts any positive start), one per thread.  Test hook for parity at the
// function level; the run kernel uses the same two functions.
__global__ void hx_unit_csys_kernel(int n, const double *Tc, const double *carbon,
                                    const double *alk, double inv_vol, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ChemK k;
  chem_constants(Tc[i], k);
  double h = 1e-8;
  unsigned status = 0;
  const double pco2 = chem_solve(k, carbon[i], inv_vol, alk[i], h, status);
  out[4 * i + 0] = pco2;
  out[4 * i + 1] = -log10(h);
  out[4 * i + 2] = k.Tr;
  out[4 * i + 3] = (double)status;
}

// ---------------------------------------------------------------------------
// host-callable launchers (the only symbols the host runtime uses)
// ---------------------------------------------------------------------------
extern "C++" {
hipError_t hx_launch_spinup(int B, const HxArgs *d_args, int nmem_launch, int *d_steps,
                            hipStream_t st) {
  const int blocks = (nmem_launch + 63) / 64;
  switch (B) {
    case 1: hipLaunchKernelGGL(hx_spinup_kernel<1>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
    case 2: hipLaunchKernelGGL(hx_spinup_kernel<2>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
    case 3: hipLaunchKernelGGL(hx_spinup_kernel<3>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
    case 4: hipLaunchKernelGGL(hx_spinup_kernel<4>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t hx_launch_alk(const HxArgs *d_args, int nmem_launch, hipStream_t st) {
  hipLaunchKernelGGL(hx_alk_kernel, dim3((nmem_launch + 63) / 64), dim3(64), 0, st, d_args,
                     nmem_launch);
  return hipGetLastError();
}

template <int B>
static void launch_run_b(const HxArgs *d_args, int npad, bool hf, bool kpm, bool con,
                         int iy_from, int iy_to, hipStream_t st) {
  const int blocks = npad / 64;
  const size_t lds = 0;
  if (con && kpm)
    hipLaunchKernelGGL((hx_run_kernel<B, true, true, true>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (con)
    hipLaunchKernelGGL((hx_run_kernel<B, true, false, true>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (hf && kpm)
    hipLaunchKernelGGL((hx_run_kernel<B, true, true, false>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (hf)
    hipLaunchKernelGGL((hx_run_kernel<B, true, false, false>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (kpm)
    hipLaunchKernelGGL((hx_run_kernel<B, false, true, false>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else
    hipLaunchKernelGGL((hx_run_kernel<B, false, false, false>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
}
hipError_t hx_launch_run(int B, const HxArgs *d_args, int npad, bool heatflux, bool kpm, bool con,
                         int iy_from, int iy_to, hipStream_t st) {
  switch (B) {
    case 1: launch_run_b<1>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break;
    case 2: launch_run_b<2>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break;
    case 3: launch_run_b<3>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break;
    case 4: launch_run_b<4>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t hx_launch_doeclim_pass(const double *sst_hist, const double *ker, double *part,
                                  double *part2, int ns, int npad, int blk0, int nyears,
                                  bool heatflux, bool kpm, hipStream_t st) {
  const dim3 grid(npad / 64, (nyears + HX_DJT - 1) / HX_DJT), block(64);
  if (heatflux && kpm)
    hipLaunchKernelGGL((hx_doeclim_pass_kernel<true, true>), grid, block, 0, st, sst_hist, ker, part, part2, ns, npad, blk0);
  else if (heatflux)
    hipLaunchKernelGGL((hx_doeclim_pass_kernel<false, true>), grid, block, 0, st, sst_hist, ker, part, part2, ns, npad, blk0);
  else if (kpm)
    hipLaunchKernelGGL((hx_doeclim_pass_kernel<true, false>), grid, block, 0, st, sst_hist, ker, part, part2, ns, npad, blk0);
  else
    hipLaunchKernelGGL((hx_doeclim_pass_kernel<false, false>), grid, block, 0, st, sst_hist, ker, part, part2, ns, npad, blk0);
  return hipGetLastError();
}
int hx_doeclim_block_years() { return HX_DBLK; }
int hx_doeclim_kernel_pad() { return HX_KPAD; }
hipError_t hx_launch_unit_csys(int n, const double *Tc, const double *carbon, const double *alk,
                               double inv_vol, double *out, hipStream_t st) {
  hipLaunchKernelGGL(hx_unit_csys_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, Tc, carbon, alk,
                     inv_vol, out);
  return hipGetLastError();
}
hipError_t hx_launch_diag(int kind, const HxDiagArgs &a, double *out, hipStream_t st) {
  hipLaunchKernelGGL(hx_diag_kernel, dim3((a.npad + 255) / 256, a.ny), dim3(256), 0, st, kind, a, out);
  return hipGetLastError();
}
hipError_t hx_launch_slr(const double *tgav, int npad, int start_year, int iy_to, double *out,
                         size_t var_stride, hipStream_t st) {
  hipLaunchKernelGGL(hx_slr_kernel, dim3((npad + 63) / 64), dim3(64), 0, st, tgav, npad,
                     start_year, iy_to, out, var_stride);
  return hipGetLastError();
}
hipError_t hx_launch_broadcast(double *table, int nrows, int npad, hipStream_t st) {
  hipLaunchKernelGGL(hx_broadcast_rows_kernel, dim3((npad + 255) / 256), dim3(256), 0, st,
                     table, nrows, npad);
  return hipGetLastError();
}
hipError_t hx_launch_broadcast_u32(unsigned *v, int npad, hipStream_t st) {
  hipLaunchKernelGGL(hx_broadcast_u32_kernel, dim3((npad + 255) / 256), dim3(256), 0, st, v, npad);
  return hipGetLastError();
}
hipError_t hx_launch_doeclim_kernel(const double *diff_row, double *ker, int ns, int count,
                                    int stride, hipStream_t st) {
  hipLaunchKernelGGL(hx_doeclim_table_kernel, dim3((count + 255) / 256, ns), dim3(256), 0, st,
                     diff_row, ker, ns, count, stride);
  return hipGetLastError();
}
hipError_t hx_launch_derive(const double *params, double *derived, const double *ker,
                            int ker_per_member, int ns, int nbiome, int npad, hipStream_t st) {
  hipLaunchKernelGGL(hx_derive_kernel, dim3((npad + 255) / 256), dim3(256), 0, st, params,
                     derived, ker, ker_per_member, ns, nbiome, npad);
  return hipGetLastError();
}
hipError_t hx_launch_gather(const double *src, const int *lane_of_member, double *dst, int n,
                            int npad, int nyears, hipStream_t st) {
  hipLaunchKernelGGL(hx_gather_kernel, dim3((n + 255) / 256, nyears), dim3(256), 0, st, src,
                     lane_of_member, dst, n, npad, nyears);
  return hipGetLastError();
}
hipError_t hx_launch_stats(const double *var, int n, int npad, int iy0, int nyears,
                           double *stats, hipStream_t st) {
  hipLaunchKernelGGL(hx_stats_kernel, dim3(nyears), dim3(256), 0, st, var, n, npad, iy0, stats);
  return hipGetLastError();
}
}
