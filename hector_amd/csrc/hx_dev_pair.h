// hx_dev_pair.h -- the small-ensemble run kernel: TWO wavefronts per 64 members
// Part of the device code of hx_kernels.hip (one translation unit).
//
// With one resident wavefront per SIMD a model year costs its instructions and its dependency
// chains (nothing else hides a latency), and an ensemble
// that does not fill the machine (BASELINE configs[1]: 1 024 members = 16 wavefronts on 1 024
// SIMDs) runs as long as one wavefront's sequential chain.  The chain is shortened by giving the
// 64 members of a workgroup two wavefronts that work on DIFFERENT PARTS of each member's year at
// the same time, on two SIMDs of the CU, and hand values over through LDS (lane i of one
// wavefront to lane i of the other) at workgroup barriers:
//
//   wavefront 0 "ocean":  equilibrium constants, the HL carbonate solves, the atmosphere and ocean
//                         variables of the carbon-cycle solver, the ocean half of a stash (box
//                         exchange, timestep controller), the in-block terms of the DOECLIM history
//                         sum, forcing, the DOECLIM year step, outputs
//   wavefront 1 "land":   Q10 factors, permafrost curve, the LL carbonate solves, the three land
//                         variables of the solver and the constant-derivative pools, the land half
//                         of a stash, OH / CH4 / O3, the matrix-pipe pass over the SST history
//
// Within a stash interval the land and the atmosphere-ocean subsystems are decoupled (land
// fluxes come from frozen pools, simpleNbox-runtime.cpp:809-840); they meet only in the error
// norm of a dopri5 step, so a step attempt needs ONE hand-off: each side's largest error
// quotient (and the ocean side's candidate atmosphere / ocean values, which the land side's mass
// check needs if the step ends a segment).  Both wavefronts carry the solver's control state
// (t, dt, target, retry count ...) and take every decision themselves from identical inputs, so
// they walk the reference's control flow in step and meet the same barriers.
//
// Same formulas, same decisions as hx_run_kernel<1,false,false,0> (parity tests compare both with
// the oracle); for one biome, no constraints, the outputs listed at its launch site
// (ensemble_core.cpp: CO2, tas, forcings, pools, NBP / NPP / RH, pH, heat flux ...), with shared or
// per-member diffusivity and with or without the per-year state history: the configuration
// small perturbed-parameter ensembles use.  The host picks it up to hx_set_pair_kernel_limit members
// (default 32 768: one workgroup per two SIMDs).
#pragma once

namespace {

enum PairStep {  // per step attempt, double-buffered
  PS_N0 = 0, PS_X3, PS_N4, PS_D4, PS_X0, PS_X4,  // ocean -> land: largest error quotient of its variables; candidates
                                                 // (atmosphere, ocean total, and the soil pool, which this side integrates)
  PS_NL, PS_DL,                                  // land -> ocean: largest quotient of vegetation and detritus
  PS_N
};
enum PairYear {  // per year / per stash, each slot written and read on opposite sides of a barrier
  PY_PN = 0, PY_CH4, PY_O3, PY_STATUS1, PY_PCO2L,               // land -> ocean
  PY_S3C, PY_DTOT, PY_SOIL, PY_TOT0,                             // land -> ocean, per interval: the soil pool's constant
                                                                 // inflow, d(veg + det + soil)/dt, the pool, the three pools' sum
  PY_MAXTS, PY_STATUS0, PY_TLAND, PY_LNC, PY_CLL,                // ocean -> land
  PY_KL_K1, PY_KL_K2, PY_KL_KB, PY_KL_KW, PY_KL_KH,              // ocean -> land, once a year
  PY_HSTAT,                                                      // ocean -> land at year end (state history)
  PY_HL, PY_HLO,                                                 // [H+] of the LL box: land -> ocean at year end (the
                                                                 // year-start solve of BOTH boxes is the ocean side's), and back
  PY_N
};

struct PairCtl {  // the solver's control state: identical in both wavefronts
  double t, dtl, t_target, t_start, sdt, ode_start, max_ts;
  int retry, fails, nsteps;
  bool first_call, stepping, alive;
};

// dopri5 tableau (odeint runge_kutta_dopri5)
struct Dp5 {
  static constexpr double b21 = 1.0 / 5, b31 = 3.0 / 40, b32 = 9.0 / 40, b41 = 44.0 / 45,
                          b42 = -56.0 / 15, b43 = 32.0 / 9, b51 = 19372.0 / 6561,
                          b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729,
                          b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247,
                          b64 = 49.0 / 176, b65 = -5103.0 / 18656;
  static constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192,
                          c5 = -2187.0 / 6784, c6 = 11.0 / 84;
  static constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695,
                          dc4 = c4 - 393.0 / 640, dc5 = c5 - (-92097.0 / 339200),
                          dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
};

// one dopri5 attempt on NP variables: candidate xn, its derivative dn, and the error quotients
// en_i = |xe_i| / (eps_abs + eps_rel (|y_i| + dt |dy_i|)) of every variable (the divisions side
// by side here, BEFORE the hand-off: after the barrier an attempt's error is two maxima away)
// (T: the tableau as data, HxConst::tab in hx_fill_tableau's order, or null for the literals)
template <int NP, class Rhs>
__device__ __forceinline__ void pair_attempt(const Rhs &rhs, double dtl, double eps_abs,
                                             double eps_rel, const double *y, const double *dxdt,
                                             double *xn, double *dn, double *en, double *ed,
                                             const double *T = nullptr) {
  double k2[NP], k3[NP], k4[NP], k5[NP], k6[NP], xt[NP];
  struct Tab {
    const double *T;
    // stage 2: b21, 1/5 | 3: b31 b32 3/10 | 4: b41 b42 b43 4/5 | 5: b51..b54 8/9 | 6: b61..b65 |
    // candidate: c1 c3 c4 c5 c6 | error: dc1 dc3 dc4 dc5 dc6 dc7
    __device__ __forceinline__ double operator()(int i, double lit) const { return T ? T[i] : lit; }
  } const tb{T};
  struct {
    double b21, b31, b32, b41, b42, b43, b51, b52, b53, b54, b61, b62, b63, b64, b65, c1, c3, c4, c5,
        c6, dc1, dc3, dc4, dc5, dc6, dc7;
  } const Dp5 = {tb(0, ::Dp5::b21), tb(2, ::Dp5::b31), tb(3, ::Dp5::b32), tb(5, ::Dp5::b41),
                 tb(6, ::Dp5::b42), tb(7, ::Dp5::b43), tb(9, ::Dp5::b51), tb(10, ::Dp5::b52),
                 tb(11, ::Dp5::b53), tb(12, ::Dp5::b54), tb(14, ::Dp5::b61), tb(15, ::Dp5::b62),
                 tb(16, ::Dp5::b63), tb(17, ::Dp5::b64), tb(18, ::Dp5::b65), tb(19, ::Dp5::c1),
                 tb(20, ::Dp5::c3), tb(21, ::Dp5::c4), tb(22, ::Dp5::c5), tb(23, ::Dp5::c6),
                 tb(24, ::Dp5::dc1), tb(25, ::Dp5::dc3), tb(26, ::Dp5::dc4), tb(27, ::Dp5::dc5),
                 tb(28, ::Dp5::dc6), tb(29, ::Dp5::dc7)};
#pragma unroll
  for (int i = 0; i < NP; ++i) xt[i] = y[i] + dtl * Dp5.b21 * dxdt[i];
  rhs(xt, k2, 1);
#pragma unroll
  for (int i = 0; i < NP; ++i) xt[i] = y[i] + dtl * Dp5.b31 * dxdt[i] + dtl * Dp5.b32 * k2[i];
  rhs(xt, k3, 2);
#pragma unroll
  for (int i = 0; i < NP; ++i)
    xt[i] = y[i] + dtl * Dp5.b41 * dxdt[i] + dtl * Dp5.b42 * k2[i] + dtl * Dp5.b43 * k3[i];
  rhs(xt, k4, 3);
#pragma unroll
  for (int i = 0; i < NP; ++i)
    xt[i] = y[i] + dtl * Dp5.b51 * dxdt[i] + dtl * Dp5.b52 * k2[i] + dtl * Dp5.b53 * k3[i] +
            dtl * Dp5.b54 * k4[i];
  rhs(xt, k5, 4);
#pragma unroll
  for (int i = 0; i < NP; ++i)
    xt[i] = y[i] + dtl * Dp5.b61 * dxdt[i] + dtl * Dp5.b62 * k2[i] + dtl * Dp5.b63 * k3[i] +
            dtl * Dp5.b64 * k4[i] + dtl * Dp5.b65 * k5[i];
  rhs(xt, k6, 5);
#pragma unroll
  for (int i = 0; i < NP; ++i)
    xn[i] = y[i] + dtl * Dp5.c1 * dxdt[i] + dtl * Dp5.c3 * k3[i] + dtl * Dp5.c4 * k4[i] +
            dtl * Dp5.c5 * k5[i] + dtl * Dp5.c6 * k6[i];
  rhs(xn, dn, 5);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double xe = dtl * Dp5.dc1 * dxdt[i] + dtl * Dp5.dc3 * k3[i] + dtl * Dp5.dc4 * k4[i] +
                      dtl * Dp5.dc5 * k5[i] + dtl * Dp5.dc6 * k6[i] + dtl * Dp5.dc7 * dn[i];
    en[i] = hx_div(fabs(xe), eps_abs + eps_rel * (fabs(y[i]) + dtl * fabs(dxdt[i])));
  }
  (void)ed;
}

// The ocean side's attempt (round 5): the atmosphere-ocean pair as ONE flux chain z (hx_dev_solver.h,
// hx_zchain: dz/dt = alp - lam z within an interval, the pair follows from the stage fluxes) AND
// the soil pool, whose equation d soil/dt = s3c - r(t) soil needs nothing of the land side within
// an interval but the loss rates r -- and those follow from the three pools' sum, which moves at
// the constant rate dtot (both sides carry the sum by the same recurrence).  Two dependency chains
// here, two (vegetation, detritus) on the land side, where it used to be two and three: the
// attempt is as long as its longer side.  -> candidates x0n / x4n / x3n, z at the candidate, the
// soil pool's derivative there, and the largest error quotient of the three variables.
struct PairZs { double Pn, aoA, aoB, pG, totC, s3c; };
// (Both attempts below work on SCALED stage derivatives H = h k like solve_year's: the affine
//  right-hand sides take h into their constants once per attempt, the stage combinations read the
//  tableau straight from scalar registers.  The loss rates of the stage times are part of the
//  attempt: luc_e / (tot0 + h dtot f_j), scaled by h, and unscaled at the candidate for the
//  derivative that the next attempt starts from.)
struct PairRates { double hr[5], r5; };
__device__ __forceinline__ void pair_rates(double h, double luc_e, double tot0, double dtot, PairRates &R) {
  const double hC = h * dtot, hle = h * luc_e;
  R.hr[0] = hx_div1(hle, fma(hC, 0.2, tot0));
  R.hr[1] = hx_div1(hle, fma(hC, 0.3, tot0));
  R.hr[2] = hx_div1(hle, fma(hC, 0.8, tot0));
  R.hr[3] = hx_div1(hle, fma(hC, 8.0 / 9.0, tot0));
  double inv = HX_RCP(tot0 + hC);
  inv = fma(fma(-(tot0 + hC), inv, 1.0), inv, inv);
  R.hr[4] = hle * inv;
  R.r5 = luc_e * inv;
}
__device__ __forceinline__ void pair_attempt_zs(const PairZs &k, double h, double eps_abs, double eps_rel,
                                                double luc_e, double tot0, double dtot,
                                                double y0, double y4, double z1, double y3, double dx3,
                                                double &x0n, double &x4n, double &z7, double &x3n,
                                                double &dn3, double &q, const double *T) {
  const double b21 = T[0], b31 = T[2], b32 = T[3], b41 = T[5], b42 = T[6], b43 = T[7], b51 = T[9],
               b52 = T[10], b53 = T[11], b54 = T[12], b61 = T[14], b62 = T[15], b63 = T[16], b64 = T[17],
               b65 = T[18], c1 = T[19], c3 = T[20], c4 = T[21], c5 = T[22], c6 = T[23], dc1 = T[24],
               dc3 = T[25], dc4 = T[26], dc5 = T[27], dc6 = T[28], dc7 = T[29];
  PairRates R;
  pair_rates(h, luc_e, tot0, dtot, R);
  const double hl = h * (k.aoA + k.aoB), ha = h * (k.aoA * k.Pn), hs = h * k.s3c;
  const double H1 = h * dx3;
  const double Hz1 = fma(-hl, z1, ha);
  const double z2 = z1 + b21 * Hz1;
  double xt = y3 + b21 * H1;
  const double H2 = fma(-R.hr[0], xt, hs);
  const double Hz2 = fma(-hl, z2, ha);
  const double z3 = z1 + b31 * Hz1 + b32 * Hz2;
  xt = y3 + b31 * H1 + b32 * H2;
  const double H3 = fma(-R.hr[1], xt, hs);
  const double Hz3 = fma(-hl, z3, ha);
  const double z4 = z1 + b41 * Hz1 + b42 * Hz2 + b43 * Hz3;
  xt = y3 + b41 * H1 + b42 * H2 + b43 * H3;
  const double H4 = fma(-R.hr[2], xt, hs);
  const double Hz4 = fma(-hl, z4, ha);
  const double z5 = z1 + b51 * Hz1 + b52 * Hz2 + b53 * Hz3 + b54 * Hz4;
  xt = y3 + b51 * H1 + b52 * H2 + b53 * H3 + b54 * H4;
  const double H5 = fma(-R.hr[3], xt, hs);
  const double Hz5 = fma(-hl, z5, ha);
  const double z6 = z1 + b61 * Hz1 + b62 * Hz2 + b63 * Hz3 + b64 * Hz4 + b65 * Hz5;
  xt = y3 + b61 * H1 + b62 * H2 + b63 * H3 + b64 * H4 + b65 * H5;
  const double H6 = fma(-R.hr[4], xt, hs);
  x3n = y3 + c1 * H1 + c3 * H3 + c4 * H4 + c5 * H5 + c6 * H6;
  dn3 = k.s3c - R.r5 * x3n;
  const double H7 = fma(-R.hr[4], x3n, hs);
  const double Z = h * (c1 * z1 + c3 * z3 + c4 * z4 + c5 * z5 + c6 * z6);
  x0n = fma(h, k.Pn, y0) - Z;
  x4n = y4 + Z;
  z7 = fma(x0n, k.aoA, -fma(x4n - k.totC, k.aoB, k.pG));
  const double E = h * (dc1 * z1 + dc3 * z3 + dc4 * z4 + dc5 * z5 + dc6 * z6 + dc7 * z7);
  const double d0 = eps_abs + eps_rel * (fabs(y0) + h * fabs(k.Pn - z1));
  const double d4 = eps_abs + eps_rel * (fabs(y4) + h * fabs(z1));
  const double xe3 = dc1 * H1 + dc3 * H3 + dc4 * H4 + dc5 * H5 + dc6 * H6 + dc7 * H7;
  const double d3 = eps_abs + eps_rel * (fabs(y3) + fabs(H1));
  q = fmax(hx_div(fabs(E), fmin(d0, d4)), hx_div(fabs(xe3), d3));
}
// The land side's attempt: vegetation (c[0] = v1, + luc_u) and detritus (c[1] = d2c), scaled stages.
__device__ __forceinline__ void pair_attempt_ld(double v1, double luc_u, double d2c, double h, double eps_abs,
                                                double eps_rel, double luc_e, double tot0, double dtot,
                                                const double *y, const double *dxdt, double *xn, double *dn,
                                                double &q, const double *T) {
  const double b21 = T[0], b31 = T[2], b32 = T[3], b41 = T[5], b42 = T[6], b43 = T[7], b51 = T[9],
               b52 = T[10], b53 = T[11], b54 = T[12], b61 = T[14], b62 = T[15], b63 = T[16], b64 = T[17],
               b65 = T[18], c1 = T[19], c3 = T[20], c4 = T[21], c5 = T[22], c6 = T[23], dc1 = T[24],
               dc3 = T[25], dc4 = T[26], dc5 = T[27], dc6 = T[28], dc7 = T[29];
  PairRates R;
  pair_rates(h, luc_e, tot0, dtot, R);
  const double cs[2] = {h * (v1 + luc_u), h * d2c};
  double H1[2], H2[2], H3[2], H4[2], H5[2], H6[2], H7[2], xt[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { H1[i] = h * dxdt[i]; xt[i] = y[i] + b21 * H1[i]; H2[i] = fma(-R.hr[0], xt[i], cs[i]); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { xt[i] = y[i] + b31 * H1[i] + b32 * H2[i]; H3[i] = fma(-R.hr[1], xt[i], cs[i]); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { xt[i] = y[i] + b41 * H1[i] + b42 * H2[i] + b43 * H3[i]; H4[i] = fma(-R.hr[2], xt[i], cs[i]); }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    xt[i] = y[i] + b51 * H1[i] + b52 * H2[i] + b53 * H3[i] + b54 * H4[i];
    H5[i] = fma(-R.hr[3], xt[i], cs[i]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    xt[i] = y[i] + b61 * H1[i] + b62 * H2[i] + b63 * H3[i] + b64 * H4[i] + b65 * H5[i];
    H6[i] = fma(-R.hr[4], xt[i], cs[i]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    xn[i] = y[i] + c1 * H1[i] + c3 * H3[i] + c4 * H4[i] + c5 * H5[i] + c6 * H6[i];
    H7[i] = fma(-R.hr[4], xn[i], cs[i]);
  }
  dn[0] = (v1 - R.r5 * xn[0]) + luc_u;   // (unscaled, the form of the step loop's first right-hand side)
  dn[1] = d2c - R.r5 * xn[1];
  double e[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double xe = dc1 * H1[i] + dc3 * H3[i] + dc4 * H4[i] + dc5 * H5[i] + dc6 * H6[i] + dc7 * H7[i];
    e[i] = hx_div(fabs(xe), eps_abs + eps_rel * (fabs(y[i]) + fabs(H1[i])));
  }
  q = fmax(e[0], e[1]);
}

// default_error_checker over all five variables: the maximum of the quotients (the land side
// hands over the largest of its three), like odeint and like solve_year
__device__ __forceinline__ double pair_err(double q0, double ql, double q4) {
  return fmax(fmax(fmax(0.0, q0), ql), q4);
}

// what both wavefronts do with the error of an attempt (controlled_runge_kutta::try_step +
// integrate_adaptive, as in solve_year); returns true if the step was accepted
// (fp contraction off in the three control functions: both wavefronts must get bit-identical
// t / dt / targets from them, whatever code surrounds the inlined copy)
__device__ __forceinline__ bool pair_control(PairCtl &c, double err, unsigned &status) {
#pragma clang fp contract(off)
  constexpr double EPS = 2.220446049250313e-16;
  if (__builtin_expect(err > 1.0, 0)) {  // reject (rare): default_step_adjuster::decrease_step
    c.dtl *= fmax(0.9 * pow_m13(err), 0.2);
    if (++c.fails > 500) { status |= HX_ERR_STEPFAIL; c.alive = false; c.stepping = false; }
    return false;
  }
  c.t += c.dtl;
  const double grow = 0.9 * pow_m15(fmax(0.00032, err));
  if (err < 0.5) c.dtl *= grow;
  c.fails = 0;
  c.nsteps++;
  if (!((c.t_target - c.t) > EPS)) c.stepping = false;  // integrate_adaptive done
  if (c.nsteps > HX_MAX_STEPS_PER_YEAR) { status |= HX_ERR_STEPFAIL; c.alive = false; c.stepping = false; }
  return true;
}

// before an attempt: the clip of dt to the target, and whether a retry is due (see solve_year)
__device__ __forceinline__ bool pair_clip_need(PairCtl &c) {
#pragma clang fp contract(off)
  constexpr double EPS = 2.220446049250313e-16;
  if (c.stepping && ((c.t + c.dtl) - c.t_target) > EPS) c.dtl = c.t_target - c.t;
  return c.stepping && ((c.t + c.dtl) - c.ode_start) > c.max_ts;
}

// the retries before an attempt (carbon-cycle-solver.cpp:266-276; see solve_year): bookkeeping
// only; returns true if the caller has to reload its pools
__device__ __forceinline__ bool pair_retry(PairCtl &c, unsigned &status) {
#pragma clang fp contract(off)
  constexpr double EPS = 2.220446049250313e-16;
  if (((c.t + c.dtl) - c.t_target) > EPS) c.dtl = c.t_target - c.t;
  bool reload = false;
  while (c.stepping && ((c.t + c.dtl) - c.ode_start) > c.max_ts) {
    ++c.retry;
    c.t_target = c.t_start + (c.t_target - c.t_start) / 2.0;
    c.t = c.t_start;
    c.sdt = c.t_target - c.t;
    c.dtl = c.sdt;
    reload = true;
    c.fails = 0;
    if (c.retry >= 8) { status |= HX_ERR_RETRIES; c.alive = false; c.stepping = false; }
  }
  if (reload) c.first_call = true;
  return reload;
}

#ifdef HX_PHASE_CLOCK
__device__ __forceinline__ void hx_pstamp(long long *clkp, int k) {
  volatile long long *clk = clkp;
  asm volatile("" ::: "memory");
  const long long now = (long long)__builtin_readcyclecounter();
  clk[k] += now - clk[23];
  clk[23] = now;
  asm volatile("" ::: "memory");
}
#define PSTAMP(k) hx_pstamp(s_pclk[role], (k))
#ifdef HX_PHASE_CLOCK_FINE
#define PSTAMPF(k) PSTAMP(k)
#else
#define PSTAMPF(k)
#endif
#else
#define PSTAMP(k)
#define PSTAMPF(k)
#endif

}  // namespace

// ===========================================================================
// hx_pair_kernel: years (iy_from, iy_to], 128-thread workgroups = 2 wavefronts per 64 members
//
// One model year, left to right (| = workgroup barrier):
//   ocean:  solve HL               | interval | steps ... stash | (solve HL) | ... forcing, DOECLIM, next year's constants of both boxes |
//   land:   Tland factors, solve LL | flows    | steps ... stash | (solve LL) | ... next year's CH4/OH/O3, Q10 window, history sums   |
// The carbonate solve a stash needs (pre-update carbon = the carbon left by the stash before) is
// done right after that earlier stash, one box per wavefront, so no stash waits for it.
// ===========================================================================
// KERPM: the members differ in ocean heat diffusivity, so each has its own DOECLIM kernel table
// ([ns + pad][npad] in HBM instead of one shared table) and the history pass runs on the vector ALU.
// HF: the ocean heat flux is recorded ("heatflux"; its two parts are extended diagnostics of the run
// kernel): a second history sum with the kernel table shifted by a year.
// CONS: the scenario holds a CO2, tas, RF_tot or CH4 constraint (the reference's concentration-driven
// runs: simpleNbox-runtime.cpp:567-603, temperature_component.cpp:510-525, forcing_component.cpp:498-505,
// ch4_component.cpp:156-157) -- each is a test of the year's shared-table entry on the side that owns the
// variable; an NBP constraint (inside the solver), per-member constraint series and a land-ocean
// warming ratio take the extended run kernel.
// NB: the biome count (1-4, unrolled): the land side owns the per-biome pools, factors and the
// biome loops of the flows and of the stash (simpleNbox-runtime.cpp:270-609); the solver and the ocean
// side see totals and are the same for every NB.
template <bool KERPM, bool HF, bool CONS = false, int NB = 1>
#ifdef HX_PAIR_TWO_PER_SIMD   // experiment builds: the register / LDS budget of two blocks' wavefronts per SIMD
#define HX_PAIR_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define HX_PAIR_OCC
#endif
__global__ __launch_bounds__(128) HX_PAIR_OCC void hx_pair_kernel(const HxArgs *__restrict__ args, int iy_from,
                                                      int iy_to) {
  __shared__ double s_tblk[HX_DBLK + 1][64];  // SSTs of years blk0-1 .. blk0+31 (the ocean side's)
  __shared__ double s_st[2][PS_N][64];        // step hand-offs, double-buffered
  __shared__ double s_yr[PY_N][64];           // year / stash hand-offs
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const int mem = blockIdx.x * 64 + lane;
  const HxBuffers &buf = args->buf;
  const HxConst &kc = args->kc;
  const size_t np = (size_t)buf.npad;
  hx_wave_stamp(buf, 2 * blockIdx.x + role, 0, lane);
  unsigned status = HX_GU(buf.status)[mem];
  PairCtl c;
  c.sdt = lds_(buf, HXS_SOLVER_DT, mem);
  c.max_ts = lds_(buf, HXS_MAX_TS, mem);
  c.alive = status == 0;
  c.nsteps = 0;
  int par = 0;      // which half of s_st the next step hand-off uses
  int blk0 = -1;    // first year index of the current DOECLIM block
  const double eps_abs = kc.eps_abs, eps_rel = kc.eps_rel;
#ifdef HX_PHASE_CLOCK
  __shared__ long long s_pclk[2][24];
  if (lane == 0) { for (int k = 0; k < 23; ++k) s_pclk[role][k] = 0; s_pclk[role][23] = (long long)__builtin_readcyclecounter(); }
#endif

  if (role == 0) {
    // ======================= wavefront 0: atmosphere, ocean, climate =======================
    double cHL = lds_(buf, HXS_C_HL, mem), cLL = lds_(buf, HXS_C_LL, mem),
           cIO = lds_(buf, HXS_C_IO, mem), cDO = lds_(buf, HXS_C_DO, mem);
    double atmos = lds_(buf, HXS_ATMOS, mem);
    const double alkH = lds_(buf, HXS_ALK_HL, mem), alkL = lds_(buf, HXS_ALK_LL, mem);
    double hH = lds_(buf, HXS_H_HL, mem);
    int ts_timeout = (int)lds_(buf, HXS_TS_TIMEOUT, mem);
    double lastflux_ann = lds_(buf, HXS_LASTFLUX_ANN, mem);
    double tland = lds_(buf, HXS_TLAND, mem), sst = lds_(buf, HXS_SST, mem),
           f_prev = lds_(buf, HXS_F_PREV, mem), base_tot = lds_(buf, HXS_BASE_TOT, mem),
           base_co2 = lds_(buf, HXS_BASE_CO2, mem);
    const double C0 = ldp(buf, HXP_C0, mem), p_aero = ldp(buf, HXP_AERO, mem),
                 p_vol = ldp(buf, HXP_VOL, mem);
    const double kLH = ldd(buf, HXD_KLH, mem), kLI = ldd(buf, HXD_KLI, mem),
                 kHD = ldd(buf, HXD_KHD, mem), kIL = ldd(buf, HXD_KIL, mem),
                 kIH = ldd(buf, HXD_KIH, mem), kID = ldd(buf, HXD_KID, mem),
                 kDI = ldd(buf, HXD_KDI, mem);
    const double dA0 = ldd(buf, HXD_A0, mem), dA1 = ldd(buf, HXD_A1, mem), dA2 = ldd(buf, HXD_A2, mem),
                 dA3 = ldd(buf, HXD_A3, mem), dIB0 = ldd(buf, HXD_IB0, mem), dIB1 = ldd(buf, HXD_IB1, mem),
                 dIB2 = ldd(buf, HXD_IB2, mem), dIB3 = ldd(buf, HXD_IB3, mem), dQC1 = ldd(buf, HXD_QC1, mem),
                 dQC2 = ldd(buf, HXD_QC2, mem), dDQ1 = ldd(buf, HXD_DQ1, mem), dDQ2 = ldd(buf, HXD_DQ2, mem),
                 dDPS = ldd(buf, HXD_DPSCALE, mem);
    auto ldk = [&](int idx) -> double {
      if constexpr (KERPM) return HX_GCD(buf.ker)[(size_t)idx * np + mem];
      else return HX_CCD(buf.ker)[idx];
    };
    const double ker_lag1 = ldk(kc.ns - 2 + HX_KPAD);  // Ker entry of last year's SST
    const double ker2_lag1 = HF ? ldk(kc.ns - 1 + HX_KPAD) : 0.0;  // ... in the heat-flux sum
    const double dHFS = HF ? ldd(buf, HXD_HFSCALE, mem) : 0.0;
    ChemK kH, kL;
    double pco2H = 0, pco2L = 0;
    // The history sum of a year (all but last year's SST, which is added from the register): the
    // pass over the years before the block was run by the land side, the block's own years are in
    // LDS (this side wrote them).  Done in the wait for the land side's first step of the year
    // (its attempt on three variables is the longer one), the HBM value requested at year start.
    s_tblk[0][lane] = sst;  // the year before the first block
    // this side's rows of the state table (base == nullptr) or of a year's history slab
    auto store_ocean = [&](double *base) {
      HxBuffers b2 = buf;
      if (base) b2.state = base;
      sts_(b2, HXS_C_HL, mem, cHL); sts_(b2, HXS_C_LL, mem, cLL);
      sts_(b2, HXS_C_IO, mem, cIO); sts_(b2, HXS_C_DO, mem, cDO);
      sts_(b2, HXS_ATMOS, mem, atmos);
      sts_(b2, HXS_MAX_TS, mem, c.max_ts); sts_(b2, HXS_TS_TIMEOUT, mem, (double)ts_timeout);
      sts_(b2, HXS_LASTFLUX_ANN, mem, lastflux_ann); sts_(b2, HXS_SOLVER_DT, mem, c.sdt);
      sts_(b2, HXS_H_HL, mem, hH);
      sts_(b2, HXS_TLAND, mem, tland); sts_(b2, HXS_SST, mem, sst);
      sts_(b2, HXS_F_PREV, mem, f_prev); sts_(b2, HXS_BASE_TOT, mem, base_tot);
      sts_(b2, HXS_BASE_CO2, mem, base_co2);
      if (base) sts_(b2, HXS_ALK_HL, mem, alkH);  // (constant rows: a slab is a whole table)
    };
    double dpart_pf = 0, dpast_in = 0;
    [[maybe_unused]] double dpart2_pf = 0, hint_in = 0;
    bool sums_done = true;
    auto history_sums = [&](int iy) {
      const int jb = iy - blk0;
      double acc = dpart_pf;
      [[maybe_unused]] double acc2 = dpart2_pf;
      const int kq = kc.ns - iy - 1 + HX_KPAD + (blk0 - 1);  // Ker index of slot 0 (year blk0 - 1)
      const int nchunk = (jb + 7) >> 3;
      for (int cc = 0; cc < nchunk; ++cc) {
        double T[8], K[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { T[q] = s_tblk[8 * cc + q][lane]; K[q] = ldk(kq + 8 * cc + q); }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += ((8 * cc + q < jb) ? T[q] : 0.0) * K[q];
        if constexpr (HF) {
#pragma unroll
          for (int q = 0; q < 8; ++q) K[q] = ldk(kq + 8 * cc + q + 1);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc2 += ((8 * cc + q < jb) ? T[q] : 0.0) * K[q];
        }
      }
      dpast_in = acc;
      if constexpr (HF) hint_in = acc2;
      sums_done = true;
    };

    // The equilibrium constants of both boxes for a year whose starting SST is sst_v; LL's go to the
    // land side through LDS.  Evaluated at the END of the year before (the SST is known there, and
    // this side would otherwise wait for the land side's year end), ahead of the loop for a launch's
    // first year: nothing of it is left on the year start's critical path, and the barrier that
    // used to publish LL's constants there is gone (they are read after barrier C / the entry barrier).
    auto year_consts = [&](double sst_v) {
      const double TcH = sst_v + 18 + (-16.4), TcL = sst_v + 18 + 2.9;
      double ex[12];
#ifndef HX_NO_CHEM_FIT
      // (see hx_run_kernel, phase A: the fitted polynomials; the formulas themselves for the lanes
      // of a wavefront that holds a box temperature outside the fit's interval)
      const bool fit_in = chem_fit_applies(TcH, TcL);
      chem_constants_fit(TcH, TcL, kc.kfit, ex);
      if (__builtin_expect(__any(!fit_in), 0))
#endif
      {
        double lg[2] = {TcH + 273.15, TcL + 273.15}, e12[12];
        hx_log_batch<2>(lg);
        chem_exponents(TcH, lg[0], &e12[0]);
        chem_exponents(TcL, lg[1], &e12[6]);
        hx_exp_chunks<12>(e12);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
#ifndef HX_NO_CHEM_FIT
          ex[i] = fit_in ? ex[i] : e12[i];
#else
          ex[i] = e12[i];
#endif
        }
      }
      chem_from_exponentials(TcH, &ex[0], O_AsHL, kH);
      chem_from_exponentials(TcL, &ex[6], O_AsLL, kL);
      chem_poly_constants(alkH, kH);
      chem_poly_constants(alkL, kL);
      s_yr[PY_KL_K1][lane] = kL.K1; s_yr[PY_KL_K2][lane] = kL.K2; s_yr[PY_KL_KB][lane] = kL.Kb;
      s_yr[PY_KL_KW][lane] = kL.Kw; s_yr[PY_KL_KH][lane] = kL.rKh;
    };
    year_consts(sst);
    __syncthreads();  // ---- entry barrier: LL constants of the launch's first year published

    for (int iy = iy_from + 1; iy <= iy_to; ++iy) {
      hx_ccd sh = HX_CCD(buf.shared) + (size_t)iy * HXSH_STRIDE;
      // ---- year start (ocean): the constants are in hand (year_consts) ----
      PSTAMP(0);
      PSTAMP(1);
      // The year-start solve of BOTH boxes, interleaved (1.4x the time of one), while the land side
      // evaluates its Tland-dependent factors and flows: it used to do those AND the LL solve ahead
      // of barrier A, with this side waiting.  [H+] of LL comes from the land side (its solves after
      // the stashes keep it) and goes back with the result.
      {
        double hL = s_yr[PY_HL][lane];
        chem_solve2_select(kH, kL, cHL, cLL, alkH, alkL, hH, hL, pco2H, pco2L, status);
        s_yr[PY_HLO][lane] = hL;
        s_yr[PY_PCO2L][lane] = pco2L;   // (what the year's first stash reads)
      }
      s_yr[PY_MAXTS][lane] = c.max_ts;
      s_yr[PY_STATUS0][lane] = (double)status;
      PSTAMP(2);
      __syncthreads();  // ---- barrier A: the land side's flows, CH4, O3, history sum, LL pCO2
      PSTAMP(3);
      double Pn = s_yr[PY_PN][lane];
      double annualflux_sum = 0;
      const double ch4 = s_yr[PY_CH4][lane], o3 = s_yr[PY_O3][lane];
      if (blk0 < 0 || iy >= blk0 + HX_DBLK) blk0 = iy;
      dpart_pf = HX_GCD(buf.dpart)[(size_t)(iy - blk0) * np + mem];  // (the land side ran the pass last year end)
      if constexpr (HF) dpart2_pf = HX_GCD(buf.dpart2)[(size_t)(iy - blk0) * np + mem];
      sums_done = false;
      status |= (unsigned)s_yr[PY_STATUS1][lane];
      // the soil pool of the interval (this side integrates it: pair_attempt_zs), the three land
      // pools' sum and its rate
      double s3c = s_yr[PY_S3C][lane], dtot = s_yr[PY_DTOT][lane], soil_seg = s_yr[PY_SOIL][lane],
             tot_seg = s_yr[PY_TOT0][lane];
      const double luc_e = sh[HXSH_LUC_E];
      // flux constants of the interval (make_interval)
      double totC = cDO + cIO + cLL + cHL;
      double pG = pco2H * kH.g + pco2L * kL.g;
      double aoA = PGC2PPM * (kH.g + kL.g);
      double aoB = pG * hx_recip(cLL + cHL);
      // ---- solver (ocean): y[0] = atmosphere, y[1] = ocean total (as the flux chain z), y3 = soil ----
      const double year = (double)(kc.start_year + iy);
      const double t0 = year - 1.0, tnew = year;
      double y[2], z1, y3, dx3, tot0;
      // (the fresh stepper's first right-hand side: z at (t, y), the soil pool's derivative)
      auto first_rhs = [&]() {
        z1 = fma(y[0], aoA, -fma(y[1] - totC, aoB, pG));
        dx3 = s3c - hx_div1(luc_e, tot0) * y3;
      };
      auto load_pools = [&]() { y[0] = atmos; y[1] = cDO + cIO + cLL + cHL; y3 = soil_seg; tot0 = tot_seg; };
      load_pools();
      c.ode_start = t0; c.t = t0; c.retry = 0; c.nsteps = 0;
      c.alive = status == 0;
      PSTAMP(4);
      // (bottom-tested, like solve_year's loops: the vote may not be duplicated into a loop header)
      for (bool go_seg = __any(c.alive && c.t < tnew); go_seg; go_seg = __any(c.alive && c.t < tnew)) {
        const bool seg = c.alive && c.t < tnew;
        c.t_start = c.t; c.t_target = tnew; c.dtl = c.sdt; c.first_call = false; c.fails = 0;
        c.stepping = seg;
        // (as in solve_year: the fresh stepper's first RHS ahead of the loop, the attempt as
        // straight-line code for every lane, retries behind a uniform rare branch)
        first_rhs();
        for (bool go_ = __any(c.stepping); go_; go_ = __any(c.stepping)) {
          if (__builtin_expect(__any(pair_clip_need(c)), 0)) {
            if (c.stepping && pair_retry(c, status)) { load_pools(); first_rhs(); }
          }
          const bool tried = c.stepping;
          const double *Tp;
#ifndef HX_PAIR_TAB_LITERALS
          { int toff = 0; asm volatile("" : "+s"(toff)); Tp = kc.tab + toff; }   // (read in this pass: see solve_year)
#else
          Tp = kc.tab;
#endif
          double x0n, x4n, z7, x3n, dn3, qo;
          const PairZs zk{Pn, aoA, aoB, pG, totC, s3c};
          pair_attempt_zs(zk, c.dtl, eps_abs, eps_rel, luc_e, tot0, dtot, y[0], y[1], z1, y3, dx3, x0n, x4n, z7,
                          x3n, dn3, qo, Tp);
          if (!sums_done) history_sums(iy);
          s_st[par][PS_N0][lane] = qo;
          s_st[par][PS_X0][lane] = x0n; s_st[par][PS_X4][lane] = x4n; s_st[par][PS_X3][lane] = x3n;
          PSTAMPF(12);
          __syncthreads();
          PSTAMPF(13);
          if (tried) {
            const double err = pair_err(qo, s_st[par][PS_NL][lane], 0.0);
            const double used = c.dtl;
            if (pair_control(c, err, status)) {
              y[0] = x0n; y[1] = x4n; z1 = z7; y3 = x3n; dx3 = dn3;
              tot0 = fma(used, dtot, tot0);   // (the land side's copy moves by the same operation)
            }
          }
          par ^= 1;
          PSTAMPF(14);
        }
        PSTAMP(5);
        // ---- stash, ocean half (OceanComponent::stashCValues) ----
        const bool more = seg && c.alive && c.t < tnew;
        if (seg && c.alive) {
          c.retry = 0;
          const double t = c.t, yf = t - c.ode_start;
          const bool in_partial_year = (t != floor(t));
          const double co2 = y[0] * PGC2PPM;
          // (pCO2 of the carbon this stash starts from: the year-start solve at the year's first
          // stash, the solve that followed the previous stash otherwise)
          pco2L = s_yr[PY_PCO2L][lane];
          double aH = ((co2 - pco2H) * kH.g) * yf, aL = ((co2 - pco2L) * kL.g) * yf;
          const double lHD = cHL * kHD * yf;
          const double lLH = cLL * kLH * yf, lLI = cLL * kLI * yf;
          const double lIL = cIO * kIL * yf, lIH = cIO * kIH * yf, lID = cIO * kID * yf;
          const double lDI = cDO * kDI * yf;
          const double currentflux = aH + aL;
          const double tot = cDO + cIO + cLL + cHL;
          const double solver_flux = y[1] - tot;
          double adj = 0.0;
          if (currentflux != 0.0) adj = (solver_flux - currentflux) / 2.0;
          aH += adj; aL += adj;
          const double inv_yf = hx_recip(yf);
          const double cdiff = solver_flux * inv_yf - lastflux_ann;
          if (cdiff > 0.1) {  // ocean_component.cpp:703-733
            c.max_ts = fmax(0.3, c.max_ts * 0.5);
            ts_timeout = 20;
          } else if (!in_partial_year && ts_timeout) {
            ts_timeout = max(0, ts_timeout - 1);
            if (!ts_timeout) {
              c.max_ts = fmin(1.0, c.max_ts / 0.5);
              if (c.max_ts < 1.0) ts_timeout = 20;
            }
          }
          const double lastflux = aL + aH;
          annualflux_sum += lastflux;
          lastflux_ann = lastflux * inv_yf;
          cHL = ((cHL + (lLH + lIH)) + aH) - lHD;
          cLL = ((cLL + lIL) + aL) - (lLH + lLI);
          cIO = (cIO + (lLI + lDI)) - ((lIL + lIH) + lID);
          cDO = (cDO + (lHD + lID)) - lDI;
          if (y[0] < 0) status |= HX_ERR_NEGPOOL;
          atmos = y[0];
          if constexpr (CONS) {
            // user-supplied [CO2] at this date: the atmosphere is set to it, the residual goes to the
            // deep ocean (simpleNbox-runtime.cpp:567-603); only whole dates exist
            const double cc = sh[HXSH_CO2_CON];
            if ((kc.con_mask & HXC_CO2) && !in_partial_year && !isnan(cc)) {
              const double residual = atmos - cc / PGC2PPM;
              cDO = residual + cDO;
              atmos = atmos - residual;
            }
          }
          c.ode_start = t;
        }
        s_yr[PY_MAXTS][lane] = c.max_ts;
        s_yr[PY_STATUS0][lane] = (double)status;
        s_yr[PY_CLL][lane] = cLL;
        PSTAMP(6);
        __syncthreads();  // ---- stash hand-off
        PSTAMP(7);
        status |= (unsigned)s_yr[PY_STATUS1][lane];
        if (seg && c.alive) {
          Pn = s_yr[PY_PN][lane];  // the next interval's land flows
          s3c = s_yr[PY_S3C][lane]; dtot = s_yr[PY_DTOT][lane];
          soil_seg = s_yr[PY_SOIL][lane]; tot_seg = s_yr[PY_TOT0][lane];
          y3 = soil_seg; tot0 = tot_seg;
          totC = cDO + cIO + cLL + cHL;
          pG = pco2H * kH.g + pco2L * kL.g;
          aoA = PGC2PPM * (kH.g + kL.g);
          aoB = pG * hx_recip(cLL + cHL);
          if (status != 0) c.alive = false;
        }
        // the solve the next stash of this year will need (error flags travel with the next hand-off)
        if (__any(more)) {
          double h2 = hH, p2 = pco2H;
          unsigned st2 = 0;
          chem_solve1(kH, cHL, alkH, 1.0 / O_vHL, h2, p2, st2);
          if (more) { hH = h2; pco2H = p2; status |= st2; }
        }
        PSTAMP(8);
        __syncthreads();  // (the hand-off slots are free again)
        PSTAMP(9);
      }
      // ---- year end (ocean): forcing, DOECLIM year step, outputs ----
      if (!sums_done) history_sums(iy);  // (a year without a step: every lane retired)
      const int slot = iy - (blk0 - 1);
      const double co2c = atmos * PGC2PPM;
      const double ln_co2r = hx_log(hx_div(co2c, C0));
      double rf_tot = 0, rf_co2 = 0;
      if (iy >= kc.baseyear_idx) {
        const double a1 = -2.4785e-7, b1 = 7.5906e-4, c1 = -2.1492e-3, d1 = 5.2488;
        const double a2 = -3.4197e-4, b2 = 2.5455e-4, c2 = -2.4357e-4, d2 = 0.12173;
        const double a3 = -8.9603e-5, b3 = -1.2462e-4, d3 = 0.045194;
        const double sqN = sh[HXSH_SQRT_N2O], sqN0 = kc.sqrtN0, rf_other = sh[HXSH_RF_OTHER];
        const double sqM = hx_sqrt(ch4), sqC = hx_sqrt(co2c);
        const double C_alpha_max = C0 - (b1 / (2 * a1));
        double alpha_prime;
        if (co2c > C_alpha_max) alpha_prime = d1 - ((b1 * b1) / (4 * a1));
        else if (C0 < co2c && co2c < C_alpha_max)
          alpha_prime = d1 + a1 * ((co2c - C0) * (co2c - C0)) + b1 * (co2c - C0);
        else alpha_prime = d1;
        const double sarf_co2 = (alpha_prime + c1 * sqN) * ln_co2r;
        const double fco2 = (sarf_co2 * kc.delta_co2) + sarf_co2;
        const double sarf_n2o = (a2 * sqC + b2 * sqN + c2 * sqM + d2) * (sqN - sqN0);
        const double fn2o = (kc.delta_n2o * sarf_n2o) + sarf_n2o;
        const double sarf_ch4 = (a3 * sqM + b3 * sqN + d3) * (sqM - kc.sqrtM0);
        const double fch4 = (kc.delta_ch4 * sarf_ch4) + sarf_ch4;
        const double fh2o = 0.0485 * ((ch4 - kc.M0f) * kc.inv_h2o_span);
        const double fo3 = kc.o3_rf * o3;
        double ftot = ((((((fco2 + fn2o) + fch4) + fh2o) + fo3) + rf_other) +
                       p_aero * sh[HXSH_RF_AERO]) + p_vol * sh[HXSH_RF_VOL];
        if constexpr (CONS) {  // forcing_component.cpp:498-505
          const double cf = sh[HXSH_FTOT_CON];
          if ((kc.con_mask & HXC_FTOT) && !isnan(cf)) ftot = cf;
        }
        if (iy == kc.baseyear_idx) { base_tot = ftot; base_co2 = fco2; }
        else { rf_tot = ftot - base_tot; rf_co2 = fco2 - base_co2; }
      }
      // history sum: the land side's part (years before last) + last year's SST
      const double dpast = (dpast_in + sst * ker_lag1) * dDPS;
      const double DelQ = rf_tot - f_prev;
      const double DQ1 = dDQ1 * (rf_tot + f_prev) + DelQ * dQC1;
      const double DQ2 = dDQ2 * (rf_tot + f_prev) + DelQ * dQC2;
      const double X1 = DQ1 + (dA0 * tland + dA1 * sst);
      const double X2 = (DQ2 + dpast) + (dA2 * tland + dA3 * sst);
      double tl_new = dIB0 * X1 + dIB1 * X2;
      double sst_new = dIB2 * X1 + dIB3 * X2;
      double tgav = D_flnd * tl_new + (1.0 - D_flnd) * D_bsi * sst_new;
      if constexpr (CONS) {  // user-supplied temperature: temperature_component.cpp:510-525
        const double ct = sh[HXSH_TAS_CON];
        if ((kc.con_mask & HXC_TAS) && !isnan(ct)) {
          tgav = ct;
          tl_new = (tgav - (1.0 - D_flnd) * D_bsi * sst_new) / D_flnd;
          sst_new = (tgav - D_flnd * tl_new) / ((1.0 - D_flnd) * D_bsi);
        }
      }
      s_tblk[slot][lane] = sst_new;
      if (slot == HX_DBLK) s_tblk[0][lane] = sst_new;  // the year before the next block
      s_yr[PY_TLAND][lane] = tl_new;
      s_yr[PY_LNC][lane] = ln_co2r;
      const size_t o = (size_t)iy * np + mem;
      if constexpr (HF) {  // heat fluxes into the mixed layer and the interior ocean (DOECLIM)
        const double hint = hint_in + sst * ker2_lag1;
        const double hmix = D_cas * (sst_new - sst);
        const double hi = dHFS * (2.0 * sst_new - hint);
        sto_(buf, HXO_HEATFLUX, o, hmix + D_fso * hi);
      }
      f_prev = rf_tot; tland = tl_new; sst = sst_new;
      sto_(buf, HXO_SST, o, sst_new);
      sto_(buf, HXO_TLAND, o, tl_new);
      if (buf.out[HXO_CO2]) sto_(buf, HXO_CO2, o, co2c);
      if (buf.out[HXO_TGAV]) sto_(buf, HXO_TGAV, o, tgav);
      if (buf.out[HXO_RF_TOT]) sto_(buf, HXO_RF_TOT, o, rf_tot);   // (with CO2 and tas: the R
      if (buf.out[HXO_RF_CO2]) sto_(buf, HXO_RF_CO2, o, rf_co2);   //  wrapper's default variables)
      // (one test for the rest of what this side can record; marked unlikely, as the other optional
      // blocks of this kernel: with the default outputs that keeps their code out of the hot path's
      // register allocation -- 2.7 % of the launch)
      if (__builtin_expect(!!(buf.out_rare), 0)) {
        if (buf.out[HXO_ATMOS_C]) sto_(buf, HXO_ATMOS_C, o, atmos);
        if (buf.out[HXO_OCEAN_C]) sto_(buf, HXO_OCEAN_C, o, cDO + cIO + cLL + cHL);
        if (buf.out[HXO_OCEAN_UPTAKE]) sto_(buf, HXO_OCEAN_UPTAKE, o, annualflux_sum);
        if (buf.out[HXO_HL_PH]) sto_(buf, HXO_HL_PH, o, -log10(hH));
        if (buf.out[HXO_CH4]) sto_(buf, HXO_CH4, o, ch4);
        if (buf.out[HXO_O3]) sto_(buf, HXO_O3, o, o3);
        if (buf.out[HXO_GMST]) sto_(buf, HXO_GMST, o, D_flnd * tl_new + (1.0 - D_flnd) * sst_new);
      }
      if (__builtin_expect(!!(buf.hist), 0)) {  // Core::reset(date) needs every component's state of every year
        store_ocean(buf.hist + (size_t)iy * (size_t)HX_NSTATE(NB) * np);
        s_yr[PY_HSTAT][lane] = (double)status;
      }
      if (iy < iy_to) year_consts(sst);   // next year's equilibrium constants (sst is this year's result now)
      PSTAMP(10);
      __syncthreads();  // ---- barrier C: year end (SST and land temperature published)
      PSTAMP(11);
    }
    store_ocean(nullptr);  // state back to the table
  } else {
    // ======================= wavefront 1: land, gases, history sums =======================
    double veg[NB], det[NB], soil[NB], pf[NB], thawed[NB], tempferts[NB], ffrozen[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int rs = HXS_NGLOBAL + b * HXSB_N;
      veg[b] = lds_(buf, rs + HXSB_VEG, mem); det[b] = lds_(buf, rs + HXSB_DET, mem);
      soil[b] = lds_(buf, rs + HXSB_SOIL, mem); pf[b] = lds_(buf, rs + HXSB_PF, mem);
      thawed[b] = lds_(buf, rs + HXSB_THAWED, mem); tempferts[b] = lds_(buf, rs + HXSB_TEMPFERTS, mem);
      ffrozen[b] = lds_(buf, rs + HXSB_F_FROZEN, mem);
    }
    double earth = lds_(buf, HXS_EARTH, mem), cum_luc_va = lds_(buf, HXS_CUM_LUC_VA, mem),
           cum_pf_ch4 = lds_(buf, HXS_CUM_PF_CH4, mem), masstot = lds_(buf, HXS_MASSTOT, mem);
    const double eos = lds_(buf, HXS_EOS_VEGC, mem);
    double ch4 = lds_(buf, HXS_CH4, mem), ln_ch4 = hx_log(ch4), o3 = 0;
    double twin = lds_(buf, HXS_TWIN, mem), tl_m1 = lds_(buf, HXS_TL_M1, mem),
           tl_m2 = lds_(buf, HXS_TL_M2, mem);
    double tland = lds_(buf, HXS_TLAND, mem);
    double lnc = hx_log(hx_div(lds_(buf, HXS_ATMOS, mem) * PGC2PPM, ldp(buf, HXP_C0, mem)));
    double cLL = lds_(buf, HXS_C_LL, mem), hL = lds_(buf, HXS_H_LL, mem);
    const double alkL = lds_(buf, HXS_ALK_LL, mem);
    double npp0[NB], f_nppv[NB], f_nppd[NB], f_litterd[NB], rh_ch4_frac[NB], fpf_static[NB], beta[NB], wf[NB],
        lnq10[NB], pmu[NB], psigma[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int r = HXP_NGLOBAL + b * HXPB_N;
      npp0[b] = ldp(buf, r + HXPB_NPP0, mem); f_nppv[b] = ldp(buf, r + HXPB_F_NPPV, mem);
      f_nppd[b] = ldp(buf, r + HXPB_F_NPPD, mem); f_litterd[b] = ldp(buf, r + HXPB_F_LITTERD, mem);
      rh_ch4_frac[b] = ldp(buf, r + HXPB_RH_CH4_FRAC, mem); fpf_static[b] = ldp(buf, r + HXPB_FPF_STATIC, mem);
      beta[b] = ldp(buf, r + HXPB_BETA, mem); wf[b] = ldp(buf, r + HXPB_WF, mem);
      lnq10[b] = ldd(buf, HXD_NGLOBAL + b, mem);
      pmu[b] = ldp(buf, r + HXPB_PF_MU, mem); psigma[b] = ldp(buf, r + HXPB_PF_SIGMA, mem);
    }
    auto rh_tp_co2 = [&](int b) { return ((thawed[b] * (1 - fpf_static[b])) * 0.02) * tempferts[b] * (1.0 - rh_ch4_frac[b]); };
    auto rh_tp_ch4 = [&](int b) { return ((thawed[b] * (1 - fpf_static[b])) * 0.02) * tempferts[b] * rh_ch4_frac[b]; };  // (see m_rh_tp_ch4)
    auto rh_ch4_total = [&]() { double x = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) x += rh_tp_ch4(b);
      return x; };
    ChemK kL;
    kL.Tr = 0; kL.g = 0;
    double pco2L = 0;
    // what a year needs that does not depend on the year before's climate -- OH / CH4 / O3
    // (ch4_component.cpp, oh_component.cpp, o3_component.cpp), the Q10 window, at the start of a
    // DOECLIM block the pass over the SST history (years before blk0 - 1; the block's own years are
    // summed by the ocean side from LDS) -- done while the ocean side finishes that year.  The
    // window's load from memory is issued a phase earlier (prefetch()): with one wavefront on the
    // SIMD a load's latency is otherwise waited out in full.
    double tfs_cand[NB], tl_old_pf = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) tfs_cand[b] = 0;
    // this side's rows of the state table (base == nullptr) or of a year's history slab
    auto store_land = [&](double *base) {
      HxBuffers b2 = buf;
      if (base) b2.state = base;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int rr = HXS_NGLOBAL + b * HXSB_N;
        sts_(b2, rr + HXSB_VEG, mem, veg[b]); sts_(b2, rr + HXSB_DET, mem, det[b]);
        sts_(b2, rr + HXSB_SOIL, mem, soil[b]); sts_(b2, rr + HXSB_PF, mem, pf[b]);
        sts_(b2, rr + HXSB_THAWED, mem, thawed[b]); sts_(b2, rr + HXSB_TEMPFERTS, mem, tempferts[b]);
        sts_(b2, rr + HXSB_F_FROZEN, mem, ffrozen[b]);
      }
      sts_(b2, HXS_EARTH, mem, earth); sts_(b2, HXS_CUM_LUC_VA, mem, cum_luc_va);
      sts_(b2, HXS_CUM_PF_CH4, mem, cum_pf_ch4); sts_(b2, HXS_MASSTOT, mem, masstot);
      sts_(b2, HXS_CH4, mem, ch4); sts_(b2, HXS_TWIN, mem, twin);
      sts_(b2, HXS_TL_M1, mem, tl_m1); sts_(b2, HXS_TL_M2, mem, tl_m2);
      sts_(b2, HXS_H_LL, mem, hL);
      if (base) { sts_(b2, HXS_ALK_LL, mem, alkL); sts_(b2, HXS_EOS_VEGC, mem, eos); }
    };
    auto prefetch = [&](int iyn) {
      const int iold = iyn - 203;
      tl_old_pf = HX_GCD(buf.out[HXO_TLAND])[(size_t)(iold >= 1 ? iold : 0) * np + mem];
    };
    auto prepare = [&](int iyn) {
      hx_ccd shn = HX_CCD(buf.shared) + (size_t)iyn * HXSH_STRIDE;
      const int iold = iyn - 203;
      const double prev_ch4 = ch4;
      const double rh_ch4 = (iyn > 1) ? rh_ch4_total() : 0.0;
      double toh = 0.0;
      if (prev_ch4 != kc.M0)
        toh = ((kc.CCH4 * (ln_ch4 - kc.lnM0) + shn[HXSH_OH_B]) + shn[HXSH_OH_C]) + shn[HXSH_OH_D];
      if (iyn >= 3) {  // Q10 window (runtime.cpp:1041-1052)
        twin += tl_m2;
        if (iold >= 1) twin -= tl_old_pf;
      }
      double ex[1 + NB];
      ex[0] = -toh;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const double Trm = (iyn > 1) ? (twin * wf[b]) * 0.005 : 0.0;
        ex[1 + b] = lnq10[b] * (Trm * 0.1);
      }
      hx_exp_batch<1 + NB>(ex);
#pragma unroll
      for (int b = 0; b < NB; ++b) tfs_cand[b] = ex[1 + b];
      const double tau_oh = kc.TOH0 * ex[0];
      const double emisTocon = ((shn[HXSH_CH4_EM] + rh_ch4 * PG_C_TO_TG_CH4) + shn[HXSH_CH4N]) * kc.inv_UC_CH4;
      const double dCH4 = ((emisTocon - prev_ch4 * kc.inv_Tsoil) - prev_ch4 * kc.inv_Tstrat) -
                          hx_div(prev_ch4, tau_oh);
      ch4 = prev_ch4 + dCH4;
      if constexpr (CONS) {  // ch4_component.cpp:156-157
        const double cm = shn[HXSH_CH4_CON];
        if ((kc.con_mask & HXC_CH4) && !isnan(cm)) ch4 = cm;
      }
      ln_ch4 = hx_log(ch4);
      o3 = ((5 * ln_ch4 + shn[HXSH_O3_NOX]) + shn[HXSH_O3_CO]) + shn[HXSH_O3_NMVOC];
      if (blk0 < 0 || iyn >= blk0 + HX_DBLK) {
        blk0 = iyn;
        if constexpr (KERPM)
          doeclim_pass_dev<true, HF>(buf.out[HXO_SST], buf.ker, const_cast<double *>(buf.dpart),
                                        const_cast<double *>(buf.dpart2), kc.ns, buf.npad, blk0, mem, blk0 - 1);
        else
          doeclim_pass_mfma<HF>(buf.out[HXO_SST], buf.ker, const_cast<double *>(buf.dpart),
                                   const_cast<double *>(buf.dpart2), kc.ns, buf.npad, blk0, mem, blk0 - 1);
        HX_FENCE();
      }
    };
    prefetch(iy_from + 1);
    prepare(iy_from + 1);
    s_yr[PY_HL][lane] = hL;
    __syncthreads();  // ---- entry barrier (the ocean side's year_consts of the first year; this side's [H+] of LL)

    for (int iy = iy_from + 1; iy <= iy_to; ++iy) {
      hx_ccd sh = HX_CCD(buf.shared) + (size_t)iy * HXSH_STRIDE;
      // ---- year start (land): what depends on last year's land temperature and CO2 ----
      const double ffi = sh[HXSH_FFI], daccs = sh[HXSH_DACCS], luc_e = sh[HXSH_LUC_E], luc_u = sh[HXSH_LUC_U];
      const double npp_luc_adjust = hx_div(eos - cum_luc_va, eos);
      double co2fert[NB], tempfertd[NB], f_new_thaw[NB];
      if constexpr (NB == 1) {
        co2fert[0] = 1 + beta[0] * lnc;
        const double Tb = tland * wf[0];
        tempfertd[0] = hx_exp(lnq10[0] * (Tb * 0.1));
        f_new_thaw[0] = 0.0;
        if (pf[0] != 0.0) {
          double ff = 1.0;
          if (Tb > 0) {
            const double d[1] = {hx_div(hx_log(Tb) - pmu[0], psigma[0] * 1.4142135623730951)};
            double f1[1];
            hx_frozen_fraction_batch<1>(d, f1);
            ff = f1[0];
          }
          f_new_thaw[0] = ffrozen[0] - ff;
          ffrozen[0] = ff;
        }
      } else {
        // (several biomes: the logarithms, exponentials and frozen fractions as batches, like
        // hx_run_kernel's phase A; a biome at or below 0 degC is frozen through, one without
        // permafrost keeps what it has)
        double Tb[NB], lg[NB], ex[NB], dfr[NB], ffb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          co2fert[b] = 1 + beta[b] * lnc;
          Tb[b] = tland * wf[b];
          lg[b] = (Tb[b] > 0) ? Tb[b] : 1.0;
          ex[b] = lnq10[b] * (Tb[b] * 0.1);
        }
        hx_log_batch<NB>(lg);
        hx_exp_batch<NB>(ex);
#pragma unroll
        for (int b = 0; b < NB; ++b) dfr[b] = hx_div(lg[b] - pmu[b], psigma[b] * 1.4142135623730951);
        hx_frozen_fraction_batch<NB>(dfr, ffb);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          tempfertd[b] = ex[b];
          const bool has_pf = pf[b] != 0.0;
          const double ff = (Tb[b] > 0) ? ffb[b] : 1.0;
          f_new_thaw[b] = has_pf ? ffrozen[b] - ff : 0.0;
          ffrozen[b] = has_pf ? ff : ffrozen[b];
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) tempferts[b] = fmax(tfs_cand[b], (iy > 1) ? tempferts[b] : 0.0);  // sticky :1054-1059
      tl_m2 = tl_m1; tl_m1 = tland;  // Tland of years iy-2, iy-1 for the next year
      // ---- interval constants of the land side (compute_flows + make_interval: sums over the biomes
      // in biome order, operation by operation hx_dev_solver.h's) ----
      double v1, d2c, s3c, k4, k5, k7, Pn;
      auto prep = [&]() {
        double npp_c = 0, fav = 0, fad = 0, fas = 0, fda = 0, fsa = 0, tpc = 0, tpm = 0;
        double litter = 0, lfvd = 0, lfvs = 0, detsoil = 0, thaw = 0, refr = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const double n = (npp0[b] * co2fert[b]) * npp_luc_adjust;
          npp_c += n;
          fav += n * f_nppv[b]; fad += n * f_nppd[b]; fas += n * (1 - f_nppv[b] - f_nppd[b]);
          fda += (det[b] * 0.25) * tempfertd[b]; fsa += (soil[b] * 0.02) * tempferts[b];
          const double co2 = rh_tp_co2(b), ch4b = rh_tp_ch4(b);
          tpc += co2; tpm += ch4b;
          const double v = veg[b] * 0.035;
          litter += v; lfvd += v * f_litterd[b]; lfvs += v * (1 - f_litterd[b]);
          detsoil += det[b] * 0.6;
          double c_thaw = pf[b] * f_new_thaw[b], r_tp = 0.0;
          if (c_thaw < 0) { const double want = -c_thaw; c_thaw = 0.0; r_tp = fmin(want, thawed[b] - co2 - ch4b); }
          thaw += c_thaw; refr += r_tp;
        }
        const double rh = fda + fsa + tpc;
        Pn = ((((ffi - daccs) + luc_e) - luc_u) - npp_c) + rh;
        v1 = fav - litter;
        d2c = ((fad + lfvd) - detsoil) - fda;
        s3c = ((fas + lfvs) + detsoil) - fsa;
        k4 = -thaw + refr;
        k5 = ((thaw - refr) - tpm) - tpc;
        k7 = -ffi + daccs;
      };
      // pool totals (what the solver starts a year from: getCValues sums the biomes)
      auto total = [&](const double (&x)[NB]) { double t = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) t += x[b];
        return t; };
      prep();
      PSTAMP(0);
      PSTAMP(1);
      // (LL's constants of this year: published before barrier C of the year before / the entry barrier)
      kL.K1 = s_yr[PY_KL_K1][lane]; kL.K2 = s_yr[PY_KL_K2][lane]; kL.Kb = s_yr[PY_KL_KB][lane];
      kL.Kw = s_yr[PY_KL_KW][lane]; kL.rKh = s_yr[PY_KL_KH][lane];
      chem_poly_constants(alkL, kL);   // (for this side's solves after the stashes; the year-start solve is the ocean side's)
      s_yr[PY_PN][lane] = Pn; s_yr[PY_CH4][lane] = ch4; s_yr[PY_O3][lane] = o3;
      s_yr[PY_STATUS1][lane] = (double)status;
      // the soil pool is integrated by the ocean side (pair_attempt_zs): its constants, the pool, and
      // the three pools' sum with its rate -- both sides carry the sum by the same recurrence
      const double veg_t = total(veg), det_t = total(det), soil_t = total(soil);
      double dtot = (((v1 + luc_u) + d2c) + s3c) - luc_e, tot_seg = (veg_t + det_t) + soil_t;
      s_yr[PY_S3C][lane] = s3c; s_yr[PY_DTOT][lane] = dtot; s_yr[PY_SOIL][lane] = soil_t; s_yr[PY_TOT0][lane] = tot_seg;
      PSTAMP(2);
      __syncthreads();  // ---- barrier A
      PSTAMP(3);
      c.max_ts = s_yr[PY_MAXTS][lane];
      status |= (unsigned)s_yr[PY_STATUS0][lane];
      hL = s_yr[PY_HLO][lane];   // the year-start solve's [H+] of LL
      // the pools' sum moves at a constant rate within an interval (every LUC loss rr * y_i adds up to
      // luc_e), and a Runge-Kutta stage preserves that: the loss rates of all stages of an attempt are
      // known before it starts -- six independent divisions instead of one at the head of each stage
      double rrs[6];
      auto rhs = [&](const double *y, double *d, int s) {
        const double rr = rrs[s];
        d[0] = (v1 - rr * y[0]) + luc_u;
        d[1] = d2c - rr * y[1];
      };
      // ---- solver (land): y = veg, detritus (soil: the ocean side); permafrost, thawed, earth advance exactly ----
      const double year = (double)(kc.start_year + iy);
      const double t0 = year - 1.0, tnew = year;
      double y[2], dxdt[2], l4, l5, l7, tot0;
      auto load_pools = [&]() { y[0] = total(veg); y[1] = total(det); l4 = total(pf); l5 = total(thawed); l7 = earth; tot0 = tot_seg; };
      load_pools();
      c.ode_start = t0; c.t = t0; c.retry = 0; c.nsteps = 0;
      c.alive = status == 0;
      int nstash = 0;
      double nbp = 0;
      const bool want_nbp = buf.out_rare && buf.out[HXO_NBP];
      // NPP / RH and its parts: the fluxes of the year's LAST interval (simpleNbox-runtime.cpp:420-440);
      // every stash writes them, the last one stays (nothing carried through the step loop)
      const bool want_flux = buf.stash_diag != 0;
      if (iy < iy_to) prefetch(iy + 1);
      PSTAMP(4);
      // (bottom-tested, like solve_year's loops: the vote may not be duplicated into a loop header)
      for (bool go_seg = __any(c.alive && c.t < tnew); go_seg; go_seg = __any(c.alive && c.t < tnew)) {
        const bool seg = c.alive && c.t < tnew;
        c.t_start = c.t; c.t_target = tnew; c.dtl = c.sdt; c.first_call = false; c.fails = 0;
        c.stepping = seg;
        double y0c = 0, y4c = 0, y3c = total(soil);  // the ocean side's atmosphere / ocean totals / soil pool after the last accepted step
        auto first_rhs = [&]() { rrs[0] = hx_div1(luc_e, tot0); rhs(y, dxdt, 0); };
        first_rhs();  // (the fresh stepper's first RHS, ahead of the loop: see the ocean side)
        for (bool go_ = __any(c.stepping); go_; go_ = __any(c.stepping)) {
          double xn[2], dn[2];
          if (__builtin_expect(__any(pair_clip_need(c)), 0)) {
            if (c.stepping && pair_retry(c, status)) { load_pools(); first_rhs(); }
          }
          const bool tried = c.stepping;
          const double *Tp;
#ifndef HX_PAIR_TAB_LITERALS
          { int toff = 0; asm volatile("" : "+s"(toff)); Tp = kc.tab + toff; }   // (read in this pass: see solve_year)
#else
          Tp = kc.tab;
#endif
          double bn;   // the larger error quotient of this side's two
          pair_attempt_ld(v1, luc_u, d2c, c.dtl, eps_abs, eps_rel, luc_e, tot0, dtot, y, dxdt, xn, dn, bn, Tp);
          s_st[par][PS_NL][lane] = bn;
          PSTAMPF(12);
          __syncthreads();
          PSTAMPF(13);
          if (tried) {
            const double err = pair_err(s_st[par][PS_N0][lane], bn, 0.0);
            const double used = c.dtl;
            if (pair_control(c, err, status)) {
              l4 += used * k4; l7 += used * k7; l5 += used * k5;
#pragma unroll
              for (int i = 0; i < 2; ++i) { y[i] = xn[i]; dxdt[i] = dn[i]; }
              y0c = s_st[par][PS_X0][lane]; y4c = s_st[par][PS_X4][lane]; y3c = s_st[par][PS_X3][lane];
              tot0 = fma(used, dtot, tot0);
            }
          }
          par ^= 1;
          PSTAMPF(14);
        }
        PSTAMP(5);
        // ---- stash, land half (SimpleNbox::stashCValues, one biome) ----
        const bool more = seg && c.alive && c.t < tnew;
        if (seg && c.alive) {
          c.retry = 0;
          ++nstash;
          const double t = c.t, yf = t - c.ode_start;
          // the interval's NPP and RH from the pools it started with (their sum weighs the biomes)
          double npp_b[NB], rhb[NB], npp_t = 0, rh_t = 0, pf_t = 0;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            npp_b[b] = (npp0[b] * co2fert[b]) * npp_luc_adjust;
            rhb[b] = ((det[b] * 0.25) * tempfertd[b] + (soil[b] * 0.02) * tempferts[b]) + rh_tp_co2(b);
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) npp_t += npp_b[b];
#pragma unroll
          for (int b = 0; b < NB; ++b) rh_t += rhb[b];
#pragma unroll
          for (int b = 0; b < NB; ++b) pf_t += pf[b];
          if (__builtin_expect(!!(want_nbp), 0)) nbp = ((npp_t - rh_t) - luc_e) + luc_u;  // NBP of the interval that ends here
          if constexpr (NB == 1) {
            if (__builtin_expect(!!(want_flux), 0)) {
              const size_t o = (size_t)iy * np + mem;
              const double rhd = (det[0] * 0.25) * tempfertd[0], rhs = (soil[0] * 0.02) * tempferts[0];
              if (buf.out[HXO_NPP]) sto_(buf, HXO_NPP, o, npp_b[0]);
              if (buf.out[HXO_RH]) sto_(buf, HXO_RH, o, ((rhd + rhs) + rh_tp_co2(0)) + rh_tp_ch4(0));
              if (buf.out[HXO_RH_DET]) sto_(buf, HXO_RH_DET, o, rhd);
              if (buf.out[HXO_RH_SOIL]) sto_(buf, HXO_RH_SOIL, o, rhs);
            }
          }
          double tpf = l5;
          if (fabs(tpf) < 1e-10) tpf = 0.0;  // :337-341
          if (y[0] < 0 || y[1] < 0 || y3c < 0 || l4 < 0 || tpf < 0) status |= HX_ERR_NEGPOOL;
          const double total3 = y[0] + y[1] + y3c;
          cum_luc_va += hx_div((luc_e - luc_u) * y[0], total3);  // no yf: :388-393
          if constexpr (NB == 1) {
            cum_pf_ch4 += rh_tp_ch4(0) * yf;  // :481
            const double wt_pf = (pf[0] > 0) ? 1.0 : 0.0;
            veg[0] = y[0]; det[0] = y[1]; soil[0] = y3c;
            pf[0] = l4 * wt_pf; thawed[0] = tpf * wt_pf;
          } else {
            // the new totals go to the biomes by their share of NPP + RH (permafrost: by their share
            // of it), correctly rounded quotients like hx_dev_solver.h's stash (hx_div_cr)
            const double npp_rh = npp_t + rh_t;
            const double inv_nr = hx_recip(npp_rh), inv_pf = (pf_t > 0) ? hx_recip(pf_t) : 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const double wt = hx_div_cr(npp_b[b] + rhb[b], npp_rh, inv_nr);
              const double wt_pf = hx_div_cr(pf[b], pf_t, inv_pf);
              cum_pf_ch4 += rh_tp_ch4(b) * yf;  // :481
              veg[b] = y[0] * wt; det[b] = y[1] * wt; soil[b] = y3c * wt;
              pf[b] = l4 * wt_pf; thawed[b] = tpf * wt_pf;
            }
          }
          earth = l7;
          const double sum = ((((((y0c + y[0]) + y[1]) + y3c) + l4) + l5) + y4c) + l7 + cum_pf_ch4;
          if (masstot > 0.0 && !(fabs(sum - masstot) <= 0.001)) status |= HX_ERR_MASS;
          masstot = sum;
          c.ode_start = t;
          if (t < tnew) {  // constants of the next segment
            prep();
            // (the solver goes on from its own values, carbon-cycle-solver.cpp:282-287: their sum,
            // not the sum of the biomes' new pools)
            dtot = (((v1 + luc_u) + d2c) + s3c) - luc_e; tot_seg = (y[0] + y[1]) + y3c;
            tot0 = tot_seg;
          }
        }
        s_yr[PY_PN][lane] = Pn;
        s_yr[PY_S3C][lane] = s3c; s_yr[PY_DTOT][lane] = dtot; s_yr[PY_SOIL][lane] = y3c; s_yr[PY_TOT0][lane] = tot_seg;
        s_yr[PY_STATUS1][lane] = (double)status;
        PSTAMP(6);
        __syncthreads();  // ---- stash hand-off
        PSTAMP(7);
        c.max_ts = s_yr[PY_MAXTS][lane];
        status |= (unsigned)s_yr[PY_STATUS0][lane];
        if (seg && c.alive) {
          cLL = s_yr[PY_CLL][lane];
          if (status != 0) c.alive = false;
        }
        if (__any(more)) {  // the LL solve the next stash of this year will need
          double h2 = hL, p2 = pco2L;
          unsigned st2 = 0;
          chem_solve1(kL, cLL, alkL, 1.0 / O_vLL, h2, p2, st2);
          if (more) { hL = h2; pco2L = p2; status |= st2; s_yr[PY_PCO2L][lane] = p2; }
        }
        PSTAMP(8);
        __syncthreads();
        PSTAMP(9);
      }
      // ---- year end (land): the stash count ("timesteps"), next year's climate-independent part ----
      if (buf.out[HXO_NSTASH]) sto_(buf, HXO_NSTASH, (size_t)iy * np + mem, (double)nstash);
      if (__builtin_expect(!!(buf.out_rare), 0)) {  // (one test for the rest of what this side can record)
        const size_t o = (size_t)iy * np + mem;
        if (buf.out[HXO_NBP]) sto_(buf, HXO_NBP, o, nbp);
        if (buf.out[HXO_VEG_C]) sto_(buf, HXO_VEG_C, o, total(veg));
        if (buf.out[HXO_DET_C]) sto_(buf, HXO_DET_C, o, total(det));
        if (buf.out[HXO_SOIL_C]) sto_(buf, HXO_SOIL_C, o, total(soil));
        if (buf.out[HXO_PERMAFROST_C]) sto_(buf, HXO_PERMAFROST_C, o, total(pf));
        if (buf.out[HXO_THAWED_C]) sto_(buf, HXO_THAWED_C, o, total(thawed));
        if (buf.out[HXO_EARTH_C]) sto_(buf, HXO_EARTH_C, o, earth);
        if (buf.out[HXO_LL_PH]) sto_(buf, HXO_LL_PH, o, -log10(hL));
        // record_state: RH_ch4 of the year-end pools (simpleNbox.cpp:800-812); f_frozen is 1 without
        // permafrost (:492-514)
        if (buf.out[HXO_RH_CH4]) sto_(buf, HXO_RH_CH4, o, rh_ch4_total());
        if (buf.out[HXO_F_FROZEN]) {   // permafrost-weighted mean over the biomes
          const double ptot = total(pf);
          double ff = 1.0;
          if (ptot > 0.0) {
            ff = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) ff += (pf[b] / ptot) * ffrozen[b];
          }
          sto_(buf, HXO_F_FROZEN, o, ff);
        }
      }
      if (__builtin_expect(!!(buf.hist), 0)) store_land(buf.hist + (size_t)iy * (size_t)HX_NSTATE(NB) * np);
      if (iy < iy_to) prepare(iy + 1);
      s_yr[PY_HL][lane] = hL;
      PSTAMP(10);
      __syncthreads();  // ---- barrier C
      PSTAMP(11);
      tland = s_yr[PY_TLAND][lane];
      lnc = s_yr[PY_LNC][lane];
      if (__builtin_expect(!!(buf.hist), 0))
        HX_GU(buf.hist_status)[(size_t)iy * np + mem] = status | (unsigned)s_yr[PY_HSTAT][lane];
    }
    store_land(nullptr);
  }
  // the two halves' error flags, merged
  __syncthreads();
  s_yr[role ? PY_STATUS1 : PY_STATUS0][lane] = (double)status;
  __syncthreads();
  if (role == 0)
    HX_GU(buf.status)[mem] = status | (unsigned)s_yr[PY_STATUS1][lane];
  hx_wave_stamp(buf, 2 * blockIdx.x + role, 1, lane);
#ifdef HX_PHASE_CLOCK
  __syncthreads();
  if (role == 0 && buf.out[HXO_TGAV])
    for (int k = 0; k < 46; ++k)
      sto_(buf, HXO_TGAV, (size_t)(1 + k) * np + mem, (double)s_pclk[k / 23][k % 23]);
#endif
}
