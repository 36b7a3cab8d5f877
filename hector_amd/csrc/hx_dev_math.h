// hx_dev_math.h -- exp / log / sqrt for the run kernel, written for ONE resident wavefront per SIMD
// Part of the device code of hx_kernels.hip (one translation unit).
//
// With a single wavefront on a SIMD every instruction -- VALU, SALU, a v_mov of a literal --
// costs one ~4-clock issue slot, so instruction COUNT is what a model year costs.  The device
// library's fp64 exp is 17 arithmetic instructions wrapped in 25 that materialise its
// polynomial coefficients (v_mov pairs, because v_fmac wants the addend in the destination) and
// 17 that handle overflow / underflow; its log is 98 VALU instructions of double-double
// arithmetic; a model year calls them ~20 times = a quarter of all instructions of the year.
// Here:
//  * hx_exp_batch<N>: N independent exponentials evaluated coefficient by coefficient, so that
//    each coefficient is loaded into an SGPR pair ONCE and serves as the addend of N v_fma_f64
//    (17 VALU per exponential, no special cases: arguments are model quantities, |x| < 700);
//  * hx_log: the classic  log(m 2^k) = k ln2 + f - f^2/2 + s (f^2/2 + R(s^2)),  s = f / (2 + f)
//    reduction with a degree-7 minimax R (fdlibm's scheme; < 1 ulp), the division through
//    v_rcp_f64 + two Newton steps: ~30 VALU; normal arguments (<= 0 gives -inf / NaN like libm);
//  * hx_sqrt: v_rsq_f64 seed + two coupled Newton steps (Goldschmidt), ~9 VALU, <= 1 ulp for
//    normal-range arguments.
// The reference calls std::exp / std::log / std::sqrt / std::pow (libm, <= 1 ulp); these agree
// with libm to ~1 ulp, i.e. ~1e-16 relative -- inside the kernels' other tolerance-neutral
// departures (DESIGN.md "numerics").
#pragma once

namespace {

// exp(x[i]) in place, i < N.  Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2; degree-13 Taylor
// polynomial in r (remainder r^14/14! < 5e-18); scaled by 2^k with v_ldexp_f64.
// T: the constants as DATA (HxConst::mtab, hx_fill_math_table's layout) behind wide scalar loads,
// or null for literals.  A 64-bit literal is two s_mov_b32 where it is used; the year loop's exp /
// log batches materialise ~110 of them a year, the table costs a dozen s_load.
template <int N>
__device__ __forceinline__ void hx_exp_batch(double (&x)[N], const double *T = nullptr) {
  constexpr double LOG2E = 1.4426950408889634074, LN2_HI = 6.93147180369123816490e-01,
                   LN2_LO = 1.90821492927058770002e-10;
  double r[N], p[N];
  int k[N];
  const double log2e = T ? T[12] : LOG2E, mln2hi = T ? T[13] : -LN2_HI, mln2lo = T ? T[14] : -LN2_LO;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double kd = rint(x[i] * log2e);
    k[i] = (int)kd;
    r[i] = fma(kd, mln2hi, x[i]);
    r[i] = fma(kd, mln2lo, r[i]);
  }
  // 1/13! ... 1/2!
  constexpr double c[12] = {1.6059043836821613e-10, 2.0876756987868099e-09, 2.5052108385441719e-08,
                            2.7557319223985893e-07, 2.7557319223985888e-06, 2.4801587301587302e-05,
                            1.9841269841269841e-04, 1.3888888888888889e-03, 8.3333333333333332e-03,
                            4.1666666666666664e-02, 1.6666666666666666e-01, 0.5};
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = T ? T[0] : c[0];
#pragma unroll
  for (int j = 1; j < 12; ++j) {
    const double cj = T ? T[j] : c[j];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fma(p[i], r[i], cj);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    p[i] = fma(p[i], r[i], 1.0);
    p[i] = fma(p[i], r[i], 1.0);
    x[i] = ldexp(p[i], k[i]);
  }
}
// the same for a longer list, eight at a time (the working set of a batch is 5 VGPRs per entry)
template <int N>
__device__ __forceinline__ void hx_exp_chunks(double (&x)[N], const double *T = nullptr) {
  constexpr int C = 8;
#pragma unroll
  for (int i0 = 0; i0 + C <= N; i0 += C) {
    double t[C];
#pragma unroll
    for (int i = 0; i < C; ++i) t[i] = x[i0 + i];
    hx_exp_batch<C>(t, T);
#pragma unroll
    for (int i = 0; i < C; ++i) x[i0 + i] = t[i];
  }
  constexpr int R = N % C;
  if constexpr (R > 0) {
    double t[R];
#pragma unroll
    for (int i = 0; i < R; ++i) t[i] = x[N - R + i];
    hx_exp_batch<R>(t, T);
#pragma unroll
    for (int i = 0; i < R; ++i) x[N - R + i] = t[i];
  }
}
// (single exponentials clamp their argument from below so that exp(-inf) = 0 like libm's; the
// batches of the year start take finite model quantities -- pK polynomials in T, Q10 exponents,
// the OH lifetime's log terms, whose -inf the reference does not survive either: a CH4
// concentration of 0 gives tau_OH = 0 there and a division by it in the same year)
__device__ __forceinline__ double hx_exp(double x) {
  double a[1] = {fmax(x, -746.0)};
  hx_exp_batch<1>(a);
  return a[0];
}

// log(x[i]) in place, i < N; x positive and normal.  Evaluated coefficient by coefficient like
// hx_exp_batch.
template <int N>
__device__ __forceinline__ void hx_log_batch(double (&x)[N], const double *T = nullptr) {
  const double LN2_HI = T ? T[23] : 6.93147180369123816490e-01, LN2_LO = T ? T[24] : 1.90821492927058770002e-10;
  const double Lg1 = T ? T[16] : 6.666666666666735130e-01, Lg2 = T ? T[17] : 3.999999999940941908e-01,
               Lg3 = T ? T[18] : 2.857142874366239149e-01, Lg4 = T ? T[19] : 2.222219843214978396e-01,
               Lg5 = T ? T[20] : 1.818357216161805012e-01, Lg6 = T ? T[21] : 1.531383769920937332e-01,
               Lg7 = T ? T[22] : 1.479819860511658591e-01;
  double dk[N], f[N], s[N], w[N], z[N], t1[N], t2[N], x0[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    x0[i] = x[i];
    int e;
    double m = frexp(x[i], &e);  // [0.5, 1)
    const bool lo = m < (T ? T[25] : 0.70710678118654752440);
    m = lo ? m + m : m;
    dk[i] = (double)(lo ? e - 1 : e);
    f[i] = m - 1.0;
    const double d = 2.0 + f[i];
    double rc = HX_RCP(d);
    rc = fma(fma(-d, rc, 1.0), rc, rc);
    rc = fma(fma(-d, rc, 1.0), rc, rc);
    s[i] = f[i] * rc;
    z[i] = s[i] * s[i];
    w[i] = z[i] * z[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { t1[i] = fma(w[i], Lg6, Lg4); t2[i] = fma(w[i], Lg7, Lg5); }
#pragma unroll
  for (int i = 0; i < N; ++i) { t1[i] = fma(w[i], t1[i], Lg2); t2[i] = fma(w[i], t2[i], Lg3); }
#pragma unroll
  for (int i = 0; i < N; ++i) t2[i] = fma(w[i], t2[i], Lg1);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double R = fma(z[i], t2[i], w[i] * t1[i]);
    const double hfsq = 0.5 * f[i] * f[i];
    const double r = dk[i] * LN2_HI - ((hfsq - fma(s[i], hfsq + R, dk[i] * LN2_LO)) - f[i]);
    // (a concentration or pool that has gone to zero or below: libm's answers, so that the member
    // turns NaN and raises its flag like it would in the reference instead of carrying on)
    // log(+inf) = +inf: frexp's mantissa of inf is inf and the reduction would give NaN
    x[i] = (x0[i] > 0.0) ? ((x0[i] == __builtin_inf()) ? x0[i] : r)
                         : ((x0[i] == 0.0) ? -__builtin_inf() : __builtin_nan(""));
  }
}
__device__ __forceinline__ double hx_log(double x, const double *T = nullptr) {
  double a[1] = {x};
  hx_log_batch<1>(a, T);
  return a[0];
}

// The frozen-permafrost fraction of N biomes from their lognormal arguments
// d = (ln Tb - mu) / (sigma sqrt 2):  1 - erfc(-d) / 2  (simpleNbox-runtime.cpp:1006-1034; boost's
// lognormal cdf is erfc(-d) / 2).  One argument range for every d (tools/make_erfc_fit.py):
//   erfc(z) = t exp(-z^2 + P(u)),  t = 2 / (2 + z),  u = 2 t - 1,  z = |d|;  erfc(-z) = 2 - erfc(z)
// P of degree 27 with coefficients below 0.7, evaluated coefficient by coefficient over the N
// biomes like hx_exp_batch: ~65 vector instructions per biome where the device library's erfc (both
// of its ranges evaluated, then selected) takes ~160, 60 of them moves of coefficients into
// vector registers.  Absolute error <= 4.4e-16 (profiles/erfc_fit_report.json).
#include "hx_erfc_fit.inc"
template <int N>
__device__ __forceinline__ void hx_frozen_fraction_batch(const double (&d)[N], double (&ff)[N]) {
#ifdef HX_OCML_ERFC   // experiment builds: the device library's erfc
#pragma unroll
  for (int i = 0; i < N; ++i) ff[i] = 1 - erfc(-d[i]) / 2;
  return;
#endif
  constexpr double c[HX_ERFC_FIT_DEGREE + 1] = HX_ERFC_FIT_COEFFS;
  double a[N], t[N], u[N], p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    a[i] = fabs(d[i]);
    const double den = 2.0 + a[i];
    double rc = HX_RCP(den);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    t[i] = rc + rc;
    u[i] = (t[i] + t[i]) - 1.0;
    p[i] = c[0];
  }
  // (the coefficient as the SCALAR addend of a v_fma_f64, spelled out: left to the compiler a
  // coefficient that is used once becomes two v_mov into the destination of a v_fmac_f64 -- three
  // vector instructions per term instead of one)
#pragma unroll
  for (int j = 1; j <= HX_ERFC_FIT_DEGREE; ++j) {
    [[maybe_unused]] const double cj = c[j];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#ifndef HX_HOST_EMULATION
      double r;
      asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p[i]), "v"(u[i]), "s"(cj));
      p[i] = r;
#else
      p[i] = fma(p[i], u[i], c[j]);
#endif
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = fmax(fma(-a[i], a[i], p[i]), -746.0);   // (exp(-inf) = 0)
  hx_exp_batch<N>(p);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double half = 0.5 * (t[i] * p[i]);   // erfc(|d|) / 2
    ff[i] = (d[i] > 0.0) ? half : 1.0 - half;
  }
}

// sqrt(x), x >= 0 and normal (sqrt(0) = 0: rsq(0) = inf and 0 * inf would be NaN -- a zero
// concentration in the forcing formulas, e.g. a constraint of 0, must stay finite like libm's)
__device__ __forceinline__ double hx_sqrt(double x) {
  double y = HX_RSQ(x);            // ~1e-8 relative
  double g = x * y, h = 0.5 * y;   // g -> sqrt(x), h -> 1 / (2 sqrt(x))
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-g, g, x);               // residual
  g = fma(r, h, g);
  r = fma(-g, g, x);
  return (x == 0.0) ? x : fma(r, h, g);
}

}  // namespace
