// hx_kernels.hip -- CDNA4 (gfx950) kernels for the Hector ensemble year loop.
//
// One ensemble member per lane, 64-lane workgroups (one wavefront each), all
// per-member state register-resident for the whole multi-year launch; HBM is
// touched only for (a) the coalesced SoA parameter/state rows at entry/exit,
// (b) one coalesced row per output variable per year and (c) DOECLIM's SST
// history, which is re-read ONCE PER BLOCK of HX_DBLK years instead of once per
// year (block-causal evaluation of the same ascending sum, partials parked in
// LDS).  Shared scenario series are wave-uniform and arrive through scalar
// loads.  Elementwise fp64 ODE stepping on the vector ALU; the one dense
// contraction, that history pass, on the fp64 matrix pipe (doeclim_pass_mfma).
// Ensembles that leave SIMDs idle are run by hx_pair_kernel (hx_dev_pair.h), two
// wavefronts per 64 members.
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
// (all <= a few ulp, see DESIGN.md "numerics"): FMA contraction on; exp / log /
// sqrt of hx_dev_math.h; quintic root by warm-started Newton with a safeguarded
// restart (same root, different path);
// T-only equilibrium constants once per year per box, from fitted polynomials inside their
// interval (hx_chem_fit.inc); the permafrost curve's erfc as one polynomial range
// (hx_erfc_fit.inc); LUC ratio via one division; 200-year Q10 window as a running sum; forcing
// summed in groups.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <math.h>
#include <stdint.h>

#include <hx_addrspace.h>

#include "hx_layout.h"

#define HX_DBLK 32  // DOECLIM block length = years per run-kernel launch
#define HX_KPAD 32  // zero entries in front of the Ker table (the host pads 64 behind it)

// hx_run_kernel<HX_B1W2, ...> is compiled for two resident wavefronts per SIMD (hx_dev_member.h);
// every other instantiation keeps the compiler's default bounds
#ifndef HX_HOST_EMULATION
#define HX_WAVES_PER_SIMD(B) __attribute__((amdgpu_waves_per_eu(hx_w2<B>() ? 2 : 1, hx_w2<B>() ? 2 : 8)))
#else
#define HX_WAVES_PER_SIMD(B)
#endif

// (the fitted equilibrium constants of next year evaluated at the end of phase C: see hx_run_kernel)
#if !defined(HX_NO_CHEM_FIT) && !defined(HX_FIT_IN_PHASE_A)
#define HX_FIT_CARRIED 1
#endif

#include "hx_dev_const.h"
#include "hx_dev_clock.h"
#include "hx_dev_math.h"
#include "hx_dev_chem.h"
#include "hx_dev_member.h"
#include "hx_dev_solver.h"

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


// The DOECLIM history pass (in-kernel device call; see DESIGN.md section 4 for
// the algorithm): one lane = one member, all HX_DBLK block years, two sweeps of 16
// accumulators.  (Round 1 kept it out of line for its own register allocation; with ROCm 7.2's
// compiler a real call inside this kernel turned out fragile -- two of the 32 instantiations
// faulted on the device after unrelated edits, see doeclim_pass_mfma -- so it is inlined.  It
// now serves only ensembles whose members differ in diffusivity.)
// NA: block years per sweep (16 accumulators in two sweeps; 8 in four for the two-wavefront
// flavour, whose 256 registers do not hold 16 + a 31-entry kernel window + two chunks of history)
template <bool KERPM, bool HF, int NA = 16>
__device__ __forceinline__ void doeclim_pass_dev(const double *sst_hist,
                                                           const double *ker, double *part,
                                                           double *part2, int ns, int npad,
                                                           int blk0, int mem, int hist_end) {
  // (hist_end: the years before it enter the sums -- blk0, or blk0 - 1 where the caller adds
  // last year's term itself, as doeclim_pass_mfma)
  hx_gcd hist = HX_GCD(sst_hist) + mem;
  const size_t np = (size_t)npad;
  auto ldk = [&](int idx) -> double {
    if constexpr (KERPM) return HX_GCD(ker)[(size_t)idx * np + mem];
    else return HX_CCD(ker)[idx];
  };
  for (int j0 = 0; j0 < HX_DBLK; j0 += NA) {
    double acc[NA], acc2[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { acc[j] = 0; acc2[j] = 0; }
    // window entry w of chunk i0 = Ker[(ns - (blk0 + j0) - 1) + i0 - 15 + w]
    const int k0 = ns - (blk0 + j0) - 1 - 15 + HX_KPAD;
#ifndef HX_PASS_NO_HALVES
    if constexpr (KERPM && NA == 16) {
      // Per-member kernel tables, one wavefront per SIMD (round 5).  The window of a chunk of 16
      // history years is 32 entries of THIS member's table -- vector loads -- of which the upper
      // 16 are the next chunk's lower 16: the window moves through the table in HALVES of 16
      // (half q = entries k0 + 16 q ... + 15; chunk c uses halves c / 16 and c / 16 + 1), each
      // loaded once, and like the history itself one chunk AHEAD of its use -- the old form asked
      // for all 32 right before the chunk's 256 multiply-adds, one exposed HBM latency per chunk.
      // Four names in turn (P Q / Q R / R S / S P), so that no half is ever copied.
      if (blk0 + j0 < ns) {
        auto load_T = [&](double (&T)[16], int i0) {
#pragma unroll
          for (int ii = 0; ii < 16; ++ii) {
            const int i = i0 + ii;
            const double v = hist[(size_t)(i < ns ? i : ns - 1) * np];   // (rows >= hist_end may be stale)
            T[ii] = (i < hist_end) ? v : 0.0;
          }
        };
        auto load_H = [&](double (&H)[16], int q) {
#pragma unroll
          for (int x = 0; x < 16; ++x) H[x] = HX_GCD(ker)[(size_t)(k0 + 16 * q + x) * np + mem];
        };
        auto compute = [&](const double (&T)[16], const double (&lo)[16], const double (&hi)[16]) {
#pragma unroll
          for (int ii = 0; ii < 16; ++ii) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int w = 15 + ii - j, w2 = 16 + ii - j;
              acc[j] += T[ii] * (w < 16 ? lo[w] : hi[w - 16]);
              if (HF) acc2[j] += T[ii] * (w2 < 16 ? lo[w2] : hi[w2 - 16]);
            }
          }
        };
        double Ta[16], Tb[16], P[16], Q[16], R[16], S[16];
        load_T(Ta, 0); load_H(P, 0); load_H(Q, 1);
        for (int i0 = 0; i0 < hist_end; i0 += 64) {
          const int q = i0 >> 4;
          load_T(Tb, i0 + 16); load_H(R, q + 2);
          compute(Ta, P, Q);
          if (i0 + 16 < hist_end) {
            load_T(Ta, i0 + 32); load_H(S, q + 3);
            compute(Tb, Q, R);
            if (i0 + 32 < hist_end) {
              load_T(Tb, i0 + 48); load_H(P, q + 4);
              compute(Ta, R, S);
              if (i0 + 48 < hist_end) {
                load_T(Ta, i0 + 64); load_H(Q, q + 5);
                compute(Tb, S, P);
              }
            }
          }
        }
      }
    } else
#endif
    if (blk0 + j0 < ns) {
      // software pipeline: the 16 loads of the next chunk are in flight while the
      // 256 FMAs of the current one execute
      auto load_chunk = [&](double *T, int i0) {
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
          const int i = i0 + ii;
          // rows >= blk0 may hold stale values of an earlier run: mask them
          const double v = hist[(size_t)(i < ns ? i : ns - 1) * np];
          T[ii] = (i < hist_end) ? v : 0.0;
        }
      };
      auto compute_chunk = [&](const double *T, int i0) {
        // (entries 16 - NA ... 31 of the window are the ones NA accumulators touch)
        double kw[32];
#pragma unroll
        for (int w = 16 - NA; w < 32; ++w) kw[w] = ldk(k0 + i0 + w);
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            acc[j] += T[ii] * kw[15 + ii - j];
            if (HF) acc2[j] += T[ii] * kw[16 + ii - j];
          }
        }
      };
      double Ta[16], Tb[16];
      load_chunk(Ta, 0);
      for (int i0 = 0; i0 < hist_end; i0 += 32) {
        load_chunk(Tb, i0 + 16);
        compute_chunk(Ta, i0);
        if (i0 + 16 < hist_end) {
          load_chunk(Ta, i0 + 32);
          compute_chunk(Tb, i0 + 16);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      HX_GD(part)[(size_t)(j0 + j) * np + mem] = acc[j];
      if (HF) HX_GD(part2)[(size_t)(j0 + j) * np + mem] = acc2[j];
    }
  }
}

#if HX_HAS_MFMA
// The same history pass on the fp64 matrix pipe, for ensembles that share the diffusivity (one
// Ker table).  part[j][m] = sum_{i < blk0} Ker[kb + i - j] T[i][m] is a (32 x H) Toeplitz matrix
// times the (H x 64) history of the wavefront's members: the one dense contraction of the model.
// v_mfma_f64_16x16x4_f64 takes a 16 x 4 slice of the Toeplitz matrix (A: lane l holds
// A[l & 15][l >> 4] = Ker[kb + (i0 + (l >> 4)) - (j0 + (l & 15))]) and the 4 x 16 slice of the
// history of 16 members (B: lane l holds T[i0 + (l >> 4)][16 g + (l & 15)]) and accumulates the
// 16 x 16 tile D[(l >> 4) + 4 r][l & 15], r < 4, in ascending history order like the scalar loop.
// Per four history years: 6 loads and 8 MFMAs (two halves of the block x four member groups)
// instead of 128 v_fma_f64 -- with one wavefront per SIMD the 128 issue slots were the cost.
// The history is read once per block instead of twice.
// (Inlined: as a real call like doeclim_pass_dev it is as fast, but hx_run_kernel<2,0,0,0> built
// with ROCm 7.2's compiler then faults on the device -- one instantiation of 32, found by the
// test suite; nothing in the source distinguishes it.)
// ACCV: the accumulators in VGPRs ("v" pin) -- the two-wavefront flavour: a kernel that names
// AGPRs gets its 256 registers split 128 + 128 by the compiler, one that does not gets 256 VGPRs.
template <bool HF, bool ACCV = false>
__device__ __forceinline__ void doeclim_pass_mfma(const double *sst_hist,
                                                            const double *ker, double *part,
                                                            double *part2, int ns, int npad,
                                                            int blk0, int mem, int hist_end) {
  // (hist_end: history years i < hist_end enter the sums; the run kernel passes blk0, the
  // small-ensemble kernel blk0 - 1 and adds the last year itself)
  typedef double d4 __attribute__((ext_vector_type(4)));
#ifndef HX_HOST_EMULATION
  // (two-wavefront flavour: the lane's part of the addresses below is computed HERE, not ahead of
  // the year loop where the optimiser would hoist it and keep six more registers live for good)
  if constexpr (ACCV) asm volatile("" : "+v"(mem));
#endif
  const int lane = mem & 63, q = lane >> 4, c = lane & 15;
  const size_t np = (size_t)npad;
  hx_gcd hist = HX_GCD(sst_hist) + (mem - lane) + c;  // member 16 g + c of this wavefront
  hx_gcd kg = HX_GCD(ker) + (ns - blk0 - 1 + HX_KPAD) - c + q;  // Ker[kb + (i0 + q) - c] at [i0]
  d4 acc[2][4], acc2[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[h][g] = d4{0, 0, 0, 0}; acc2[h][g] = d4{0, 0, 0, 0}; }
  struct Slice { double a0, a1, e0, e1, b[4]; };  // operands of four history years
  auto load = [&](int i0, Slice &S) {
    const int i = i0 + q;
    S.a0 = kg[i0];
    S.a1 = kg[i0 - 16];
    if (HF) { S.e0 = kg[i0 + 1]; S.e1 = kg[i0 - 15]; } else { S.e0 = 0; S.e1 = 0; }
    // rows >= hist_end must not contribute (they may hold values of an earlier run): read row 0
    // instead, the SST anomaly of startDate, which is 0 for every member -- the operands of the
    // MFMAs then come straight from loads, no VALU select in between
    const size_t row = (size_t)(i < hist_end ? i : 0) * np;
#pragma unroll
    for (int g = 0; g < 4; ++g) S.b[g] = hist[row + 16 * g];
  };
  auto mma = [&](const Slice &S) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc[0][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.a0, S.b[g], acc[0][g], 0, 0, 0);
      acc[1][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.a1, S.b[g], acc[1][g], 0, 0, 0);
      if (HF) {
        acc2[0][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.e0, S.b[g], acc2[0][g], 0, 0, 0);
        acc2[1][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.e1, S.b[g], acc2[1][g], 0, 0, 0);
      }
    }
  };
  // The accumulators are pinned to the AGPR half of the register file across the loop's back
  // edge by passing them through an asm statement (left alone the compiler carries them in VGPRs
  // and copies all 64 / 128 registers into and out of AGPRs around every group of MFMAs).  To the
  // compiler that statement becomes the accumulators' last writer, so its hazard recogniser no
  // longer sees that an MFMA result is read -- as SrcC of the next iteration's first MFMAs, by
  // the stores after the loop: the statement therefore carries the wait itself, 48 states, more
  // than the 8 passes of a v_mfma_f64_16x16x4_f64 issued just before it (once per 32 MFMAs =
  // 1024 cycles of matrix pipe).  Without it the code was only correct for instruction orders
  // that happened to keep an accumulator's reuse 7 MFMAs apart.
  auto pin = [&]() {
    if constexpr (ACCV && HF)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                     "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),
                     "+v"(acc2[0][0]), "+v"(acc2[0][1]), "+v"(acc2[0][2]), "+v"(acc2[0][3]),
                     "+v"(acc2[1][0]), "+v"(acc2[1][1]), "+v"(acc2[1][2]), "+v"(acc2[1][3]));
    else if constexpr (ACCV)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                     "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
    else if constexpr (HF)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]),
                     "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                     "+a"(acc2[0][0]), "+a"(acc2[0][1]), "+a"(acc2[0][2]), "+a"(acc2[0][3]),
                     "+a"(acc2[1][0]), "+a"(acc2[1][1]), "+a"(acc2[1][2]), "+a"(acc2[1][3]));
    else
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]),
                     "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]));
  };
  // Four register sets in turn, three of them in flight behind the one the matrix pipe is
  // working on: a load from HBM / L2 takes several times the 8 MFMAs of a slice, and nothing else
  // runs on this SIMD meanwhile.  (Sets past the end load row 0 = zeros and Ker entries of the
  // table's zero padding: they add 0.)
  Slice s0, s1, s2, s3;
  load(0, s0); load(4, s1); load(8, s2);
  pin();
  for (int i0 = 0; i0 < hist_end; i0 += 16) {
    load(i0 + 12, s3);
    mma(s0);
    load(i0 + 16, s0);
    mma(s1);
    load(i0 + 20, s1);
    mma(s2);
    load(i0 + 24, s2);
    mma(s3);
    pin();
  }
  hx_gd po = HX_GD(part) + (mem - lane) + c, po2 = HX_GD(part2) + (mem - lane) + c;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t o = (size_t)(16 * h + q + 4 * r) * np + 16 * g;
        po[o] = acc[h][g][r];
        if (HF) po2[o] = acc2[h][g][r];
      }
}
#endif

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
#ifndef HX_HOST_EMULATION
  // (a chain of ~500 dependent solver steps, often on ONE wavefront: ahead of the prewarm loop's
  // wavefronts -- hx_prewarm_kernel -- in the SIMD's instruction arbiter)
  __builtin_amdgcn_s_setprio(3);
#endif
  __shared__ double s_park_fixed[B == HX_DYN ? 1 : hx_npark<B>()][64];
  double (*s_park)[64] = (B == HX_DYN) ? hx_dyn_park : s_park_fixed;
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
  // (the bits found while deriving the member's constants -- a singular DOECLIM system -- depend
  // on S and diff, which the spinup never sees: they are OR'ed in per member after the spinup,
  // hx_or_flags_kernel, so that a shared spinup neither loses nor spreads them)
  m.status = 0;
#pragma unroll hx_ur<B>()
  for (int b = 0; b < nbio<B>(m); ++b) {
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
  m.spin_row = nullptr;

  bool spun = (kc.max_spinup <= 1);
  int steps = 0;
  for (int step = 1; step < kc.max_spinup && __any(!spun); ++step) {
    if (!spun) {
      const double o0 = m.atmos, oO = m.cDO + m.cIO + m.cLL + m.cHL, oE = m.earth;
      double ov = 0, od = 0, os = 0, op = 0, ot = 0;
#pragma unroll hx_ur<B>()
      for (int b = 0; b < nbio<B>(m); ++b) { ov += m.veg[b]; od += m.det[b]; os += m.soil[b];
                                     op += m.pf[b]; ot += m.thawed[b]; }
      m.nstash = 0; m.nsteps = 0;
      m.annualflux_sum = 0;
      if (buf.spin_rec) {
        m.spin_row = buf.spin_rec + ((size_t)(step - 1) * HXSR_N * buf.npad + mem);
        hx_gd r = HX_GD(m.spin_row);
        r[(size_t)HXSR_HL_UPTAKE * buf.npad] = 0.0; r[(size_t)HXSR_LL_UPTAKE * buf.npad] = 0.0;
        r[(size_t)HXSR_HL_DO * buf.npad] = 0.0;
      }
      solve_year<B, true>(m, kc, (double)(step - 1), (double)step, YearCon{});
      double nv = 0, nd = 0, nso = 0, np = 0, nt = 0;
#pragma unroll hx_ur<B>()
      for (int b = 0; b < nbio<B>(m); ++b) { nv += m.veg[b]; nd += m.det[b]; nso += m.soil[b];
                                     np += m.pf[b]; nt += m.thawed[b]; }
      double mx = fabs(m.atmos - o0);
      mx = fmax(mx, fabs(nv - ov)); mx = fmax(mx, fabs(nd - od));
      mx = fmax(mx, fabs(nso - os)); mx = fmax(mx, fabs(np - op));
      mx = fmax(mx, fabs(nt - ot));
      mx = fmax(mx, fabs((m.cDO + m.cIO + m.cLL + m.cHL) - oO));
      mx = fmax(mx, fabs(m.earth - oE));
      steps = step;
      spun = (mx < kc.eps_spinup) || (m.status != 0);
      if (m.spin_row) {  // the pools as recorded after the step (record_state)
        hx_gd r = HX_GD(m.spin_row);
        const size_t npd = (size_t)buf.npad;
        r[HXSR_ATMOS_C * npd] = m.atmos; r[HXSR_VEG_C * npd] = nv; r[HXSR_DET_C * npd] = nd;
        r[HXSR_SOIL_C * npd] = nso; r[HXSR_PERMAFROST_C * npd] = np; r[HXSR_THAWED_C * npd] = nt;
        r[HXSR_EARTH_C * npd] = m.earth;
        r[HXSR_C_DO * npd] = m.cDO; r[HXSR_C_HL * npd] = m.cHL; r[HXSR_C_IO * npd] = m.cIO;
        r[HXSR_C_LL * npd] = m.cLL;
        r[HXSR_OCEAN_UPTAKE * npd] = m.annualflux_sum;
      }
    }
  }
  if (!spun) m.status |= HX_ERR_SPINUP;
  store_state<B>(buf, mem, m);
  // SimpleNbox::run, first call: end_of_spinup_vegc  runtime.cpp:209-213
  double v1 = 0;
#pragma unroll hx_ur<B>()
  for (int b = 0; b < nbio<B>(m); ++b) {
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
#pragma unroll hx_ur<B>()
    for (int b = 0; b < nbio<B>(m); ++b) { v += m.veg[b]; d += m.det[b]; s += m.soil[b];
                                   p += m.pf[b]; th += m.thawed[b]; }
    put(HXO_PERMAFROST_C, p); put(HXO_VEG_C, v); put(HXO_DET_C, d);
    put(HXO_SOIL_C, s); put(HXO_THAWED_C, th);
  }
  put(HXO_HEATFLUX, 0.0); put(HXO_CH4, kc.M0f); put(HXO_O3, 0.0);
  put(HXO_EARTH_C, m.earth); put(HXO_NBP, 0.0); put(HXO_OCEAN_UPTAKE, 0.0);
  put(HXO_NSTASH, 0.0); put(HXO_NSTEPS, 0.0);
  put(HXO_C_HL, m.cHL); put(HXO_C_LL, m.cLL); put(HXO_C_IO, m.cIO); put(HXO_C_DO, m.cDO);
  put(HXO_F_FROZEN, 1.0); put(HXO_TAU_OH, kc.TOH0);
#pragma unroll hx_ur<B>()
  for (int b = 0; b < nbio<B>(m); ++b) {
    put(HXO_B(HXOB_VEG, b), m.veg[b]); put(HXO_B(HXOB_DET, b), m.det[b]);
    put(HXO_B(HXOB_SOIL, b), m.soil[b]); put(HXO_B(HXOB_PF, b), m.pf[b]);
    put(HXO_B(HXOB_THAWED, b), m.thawed[b]); put(HXO_B(HXOB_F_FROZEN, b), 1.0);
    put(HXO_B(HXOB_TEMPFERTD, b), 1.0); put(HXO_B(HXOB_TEMPFERTS, b), 1.0);
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
// CON = 2: ... with carbon tracking inside the stash (hx_dev_track.h); CON = 3 (one biome): ...
// with carbon tracking on two companion wavefronts -- 192 threads a block, waves 1 and 2 only track.
// the year loop's own rows (outputs, the values requested a phase ahead) as row address + lane
// offset: the lean-park kernels (hx_tbl); -DHX_ROWIO_ALL: every kernel (experiment builds)
#ifdef HX_ROWIO_ALL
template <int B> constexpr bool hx_rowio() { return true; }
#else
template <int B> constexpr bool hx_rowio() { return hx_tbl<B>(); }
#endif
template <int B, bool HF, bool KERPM, int CON>
__global__ __launch_bounds__(CON == 3 ? 64 * (1 + trk_waves<(B < 1 || B > 4 ? 1 : B)>()) : 64) HX_WAVES_PER_SIMD(B)
void hx_run_kernel(const HxArgs *__restrict__ args, int iy_from, int iy_to) {
  static_assert(CON != 3 || (B >= 1 && B <= 4), "tracking companions: the unrolled kernels");
  static_assert(!hx_w2<B>() || CON <= 1, "two-wavefront flavour: the plain and the extended kernel");
  // LDS: the SSTs produced inside this launch's block of years (<= HX_DBLK), per lane
  // (multi-biome kernels need that LDS for the per-biome arrays and re-read the block's SSTs
  // from the output array instead)
  __shared__ double s_tblk[B == 1 ? HX_DBLK : 1][64];
  const int lane = threadIdx.x & 63;
  const int mem = blockIdx.x * 64 + lane;
  if (mem >= args->buf.npad) return;
  [[maybe_unused]] double (*s_trk_rec)[64] = nullptr;
  [[maybe_unused]] int *s_trk_cmd = nullptr;
  if constexpr (CON == 3) {
    __shared__ double s_rec[2 * trkr_n(B)][64];  // two sets of hand-over slots
    __shared__ int s_cmd[5];                  // {what, year} of each set; wave 0's event counter
    s_trk_rec = s_rec; s_trk_cmd = s_cmd;
    if (threadIdx.x >= 64) {
      if (threadIdx.x < 128) {
        s_rec[TRKR_ACTIVE][lane] = 0.0; s_rec[trkr_n(B) + TRKR_ACTIVE][lane] = 0.0;
        s_cmd[4] = 0;
      }
      __syncthreads();
      track_companion<B>(args, iy_from, lane, ((int)(threadIdx.x >> 6) - 1) * trk_nc<B>(), s_trk_rec, s_trk_cmd);
      return;
    }
    __syncthreads();  // (the companion has cleared the hand-over flags)
  }
  __shared__ double s_park_fixed[B == HX_DYN ? 1 : hx_npark<B>()][64];
  double (*s_park)[64] = (B == HX_DYN) ? hx_dyn_park : s_park_fixed;
  Member<B> m;
  hx_wave_stamp(args->buf, blockIdx.x, 0, lane);
  bind_member<B>(args->buf, mem, m, s_park, lane);
#ifdef HX_PHASE_CLOCK
  for (int k = 0; k < HX_NCLK; ++k) hx_s_clk[k] = 0;
  hx_s_clk[HX_NCLK] = (long long)__builtin_readcyclecounter();
#endif
  load_state<B>(args->buf, mem, m);
  {  // year-level state -> park
    const HxBuffers &buf = args->buf;
    PKM(m, PK_CH4) = lds_(buf, HXS_CH4, mem); PKM(m, PK_SST) = lds_(buf, HXS_SST, mem);
    if constexpr (!hx_w2<B>()) PKM(m, PK_EOS) = lds_(buf, HXS_EOS_VEGC, mem);
    PKM(m, PK_TLAND) = lds_(buf, HXS_TLAND, mem);
    PKM(m, PK_TWIN) = lds_(buf, HXS_TWIN, mem); PKM(m, PK_TL_M1) = lds_(buf, HXS_TL_M1, mem);
    PKM(m, PK_TL_M2) = lds_(buf, HXS_TL_M2, mem); PKM(m, PK_F_PREV) = lds_(buf, HXS_F_PREV, mem);
    PKM(m, PK_BASE_TOT) = lds_(buf, HXS_BASE_TOT, mem);
    PKM(m, PK_BASE_CO2) = lds_(buf, HXS_BASE_CO2, mem);
    PKM(m, PK_LN_CH4) = hx_log(lds_(buf, HXS_CH4, mem));
    PKM(m, PK_LN_CO2R) = hx_log(hx_div(m.atmos * PGC2PPM, m.C0));
    if constexpr (!hx_slim_park<B>()) {
#pragma unroll hx_ur<B>()
    for (int b = 0; b < nbio<B>(m); ++b)
      PKM(m, pk_ff0<B>() + b) = lds_(buf, HXS_NGLOBAL + b * HXSB_N + HXSB_F_FROZEN, mem);
    }
  }
  constexpr bool want_hf = HF;  // heat-flux diagnostic needs a second history sum
  int blk0 = -1;  // first year index of the current DOECLIM block
  // (-DHX_BLOCK_STAGGER, experiment builds: the launch's FIRST block cut short by a
  // wavefront-dependent amount, so that the wavefronts do not all read their SST history in the
  // same model years.  The sums do not depend on where the blocks start, bit for bit -- the pass
  // and the in-block terms add the history years in ascending order either way -- and neither
  // does the time: 65 536 / 131 072 / 262 144 members 6.77-6.82 / 10.11-10.17 / 19.2-19.6 ms with
  // and without it.  The pass is bound by the matrix pipe, 64 cycles per v_mfma_f64_16x16x4_f64
  // on this part, not by an HBM burst.)
#ifdef HX_BLOCK_STAGGER
  const int blk_len0 = 1 + (int)((blockIdx.x * 13u) & (HX_DBLK - 1));
#else
  const int blk_len0 = HX_DBLK;
#endif
  int blk_end = 0;  // first year index after the current block
  // Two values a year come straight from HBM -- land temperature of 203 years ago (Q10 window)
  // and the block's history partial sum -- and with one wavefront on the SIMD a load's latency is
  // waited out in full: they are requested a phase ahead, before the solver (pf_*).
  double pf_tl_old = 0.0, pf_dpart = 0.0;
  // Per-member diffusivity: the in-block terms of the history sum multiply the block's SSTs with
  // THIS member's kernel entries of lags 1 ... 32 -- rows of the [ns][npad] kernel table, vector
  // loads where the shared table takes scalar ones, and a lone wavefront waited out one HBM
  // latency per chunk of eight (2.5 a model year).  They are requested before the solver like the
  // two values above and wait in registers (33 doubles; the KERPM kernels have them to spare):
  // 65 536 members with their own diffusivity 8.56 -> see profiles/r05_variant_log.md 11.
  constexpr bool KPF = KERPM && !hx_w2<B>() && hx_nbc<B>() >= 1 && hx_nbc<B>() <= 4;
  [[maybe_unused]] double kpf[KPF ? HX_DBLK + 1 : 1];
  if constexpr (KPF) {
#pragma unroll
    for (int r = 0; r <= HX_DBLK; ++r) kpf[r] = 0.0;   // (an entry not loaded yet multiplies a zero SST)
  }
  if constexpr (!hx_w2<B>()) {
    const int iold0 = iy_from + 1 - 203;
    pf_tl_old = HX_GCD(args->buf.out[HXO_TLAND])[(size_t)(iold0 >= 1 ? iold0 : 0) * args->buf.npad + mem];
  }
  // The same for the year's entries of the shared table (scalar loads, one exposed latency per
  // phase otherwise): phase A's of next year and phase C's of this year are requested before the
  // solver and wait in SGPRs.
  double ya[12], yc4[4];  // OH_B/C/D, CH4_EM, CH4N, O3_NOX/CO/NMVOC, FFI, DACCS, LUC_E, LUC_U; SQRT_N2O, RF_OTHER, RF_AERO, RF_VOL
  auto load_year_a = [&](int iyn) {
    hx_ccd shn = HX_CCD(args->buf.shared) + (size_t)(iyn < args->kc.ns ? iyn : args->kc.ns - 1) * HXSH_STRIDE;
    ya[0] = shn[HXSH_OH_B]; ya[1] = shn[HXSH_OH_C]; ya[2] = shn[HXSH_OH_D]; ya[3] = shn[HXSH_CH4_EM];
    ya[4] = shn[HXSH_CH4N]; ya[5] = shn[HXSH_O3_NOX]; ya[6] = shn[HXSH_O3_CO]; ya[7] = shn[HXSH_O3_NMVOC];
    ya[8] = shn[HXSH_FFI]; ya[9] = shn[HXSH_DACCS]; ya[10] = shn[HXSH_LUC_E]; ya[11] = shn[HXSH_LUC_U];
  };
  // Round 6: the year's 16 entries of the shared table are requested where they are used, in every
  // kernel.  Requested ahead of the solver (round 2: one exposed scalar-load latency a phase less)
  // their 32 scalar registers did not survive the step loop: the allocator moved them into the
  // lanes of a spill VGPR and back -- 64 v_writelane / v_readlane a model year to save two ~55-clock
  // waits.  Here: 42 -> 16 spilled scalars in the plain kernel, 65 536 members 5.78 -> 5.74 ms
  // (profiles/r06_variant_log.md 8; -DHX_YA_PREFETCH: the old form).
#ifndef HX_YA_PREFETCH
  constexpr bool YAL = true;
#else
  constexpr bool YAL = hx_w2<B>();
#endif
  if constexpr (!YAL) load_year_a(iy_from + 1);
  // Two-wavefront flavour (round 6): the build uses 227 of its 256 registers, so four values it
  // re-read from the tables at every year start -- constants of the launch: end-of-spinup
  // vegetation, both alkalinities, ln Q10 -- stay in eight of the spare ones, and the one value that
  // does change, the land temperature leaving the Q10 window, is requested at the END of the
  // previous year's phase C, ahead of the equilibrium-constant fit's 156 multiply-adds.  The year
  // start then waits for no table trip at all.  (-DHX_W2_NO_KEEP: the round-4 form.)
#ifndef HX_W2_NO_KEEP
  constexpr bool W2K = hx_w2<B>();
#else
  constexpr bool W2K = false;
#endif
  [[maybe_unused]] double w2k_eos = 0, w2k_alkH = 0, w2k_alkL = 0, w2k_lnq = 0;
  if constexpr (W2K) {
    const HxBuffers &buf = args->buf;
    HX_W2_LOCAL(m);
    w2k_eos = w2_ld<true>(buf.state, m.npad, HXS_EOS_VEGC, m.moff);
    w2k_alkH = w2_ld<true>(buf.state, m.npad, HXS_ALK_HL, m.moff);
    w2k_alkL = w2_ld<true>(buf.state, m.npad, HXS_ALK_LL, m.moff);
    w2k_lnq = w2_ld<true>(buf.derived, m.npad, HXD_NGLOBAL, m.moff);
    const int iold0 = iy_from + 1 - 203;
    pf_tl_old = hx_ldm<true>(HX_GCD(buf.out[HXO_TLAND]) + (size_t)(iold0 >= 1 ? iold0 : 0) * buf.npad, m.moff);
  }
  if constexpr (CON >= 2) m.trk_iy = args->kc.trk_iy;
  if constexpr (CON == 3) { m.trk_rec = s_trk_rec; m.trk_cmd = s_trk_cmd; }
  int cost_steps = 0, cost_stash = 0;  // this lane's solver work (the host's lane-ordering key)
#ifdef HX_FIT_CARRIED
  // Next year's fitted equilibrium constants (chem_constants_fit) are evaluated at the END of
  // phase C -- the SST they depend on is known there -- and carried to the year start in 24
  // vector registers: in phase A the two coefficient rows in flight (48 scalar registers) pushed
  // ~60 long-lived scalars into spill lanes (164 v_readlane / v_writelane a year against 63).
  // (Plain kernels; with constraints or a land-ocean warming ratio the year start may see another
  // SST than DOECLIM's, and the extended kernels evaluate the fit there.)
  // (Extended kernels, one wavefront per SIMD: carried too, and used unless some member has a
  // land-ocean warming ratio -- HXC_LO, a scenario-wide bit -- which is the one thing that lets the
  // year start see another SST than the one phase C left: a temperature constraint replaces
  // sst_new before it is parked.  The two-wavefront flavour's extended instantiations sit at
  // their 256-register budget and keep the evaluation in phase A.)
  constexpr bool FITC = !hx_cons<CON>() || (!hx_w2<B>() && (CON == 1 || CON == -1));
  [[maybe_unused]] double fitc[12];
  if constexpr (FITC) {
    const double s0 = PKM(m, PK_SST);
    chem_constants_fit(s0 + 18 + (-16.4), s0 + 18 + 2.9, args->kc.kfit, fitc);
  }
#endif

  // Two to four biomes (parked arrays, no table rows in scalar form): the biomes' warming factors
  // and ln Q10 -- constants of the launch, and the only HBM values the year start asked for, one
  // exposed round trip a model year -- are read once and stay in 2 x NB registers.
#ifndef HX_NO_CWF
  constexpr bool CWF = !hx_w2<B>() && !hx_tbl<B>() && hx_nbc<B>() >= 2 && hx_nbc<B>() <= 4;
#else
  constexpr bool CWF = false;
#endif
  // (The block's SSTs, which these kernels re-read from the output array every year, requested
  // before the solver as well -- 64 more registers through the step loop -- lose 1.3 %:
  // profiles/r05_variant_log.md 13.)
  [[maybe_unused]] double c_wf[CWF ? hx_nbc<B>() : 1], c_lnq[CWF ? hx_nbc<B>() : 1];
  if constexpr (CWF) {
#pragma unroll
    for (int b = 0; b < hx_nbc<B>(); ++b) {
      c_wf[b] = ldp(args->buf, HXP_NGLOBAL + b * HXPB_N + HXPB_WF, mem);
      c_lnq[b] = ldd(args->buf, HXD_NGLOBAL + b, mem);
    }
  }
  // Extended kernels: which constraints / per-member series / outputs exist -- HxConst::con_mask,
  // HxBuffers::ms_mask, ::out_mask0 -- read ONCE and kept in scalar registers (laundered through an
  // empty asm so that they stay values, not loads): a scalar load + test inside the year loop waits
  // for every scalar load and LDS read in flight (one counter), e.g. for next year's table entries
  // requested ahead of the solver.
  [[maybe_unused]] unsigned cmk = 0, msk = 0;
  [[maybe_unused]] unsigned long long omk = 0;
  if constexpr (CON) {
    cmk = (unsigned)args->kc.con_mask; msk = args->buf.ms_mask; omk = args->buf.out_mask0;
    m.omk = omk;
  }
  // (... and laundered again at the head of every region that tests them: a test of an opaque
  // but loop-invariant value is itself loop-invariant, and the optimiser would compute all ~50 of
  // them ahead of the year loop and keep them in -- spilled -- scalar registers)
#ifndef HX_HOST_EMULATION
#define HX_MASKS_LOCAL() do { if constexpr (CON) asm volatile("" : "+s"(cmk), "+s"(msk), "+s"(omk)); } while (0)
#else
#define HX_MASKS_LOCAL() do { } while (0)
#endif
  for (int iy = iy_from + 1; iy <= iy_to; ++iy) {
    HX_FENCE();
    HX_STAMP(m, 0);   // (kernel entry / loop overhead)
    HX_MASKS_LOCAL();
    if constexpr (CON) m.iy = iy;
    double ch4, o3;
    // ======================= phase A ========================================
    {
      const HxBuffers &buf = args->buf;
      const HxConst &kc = args->kc;
      hx_ccd sh = HX_CCD(buf.shared) + (size_t)iy * HXSH_STRIDE;
      // every HBM value this phase needs, issued back to back (one exposed latency:
      // with one wavefront per SIMD nothing else hides it)
      if constexpr (CON == 2) {
        if (buf.track_out_f && iy == kc.trk_iy) track_start<B>(m);
      }
      const double prev_ch4 = PKM(m, PK_CH4);
      double sst = PKM(m, PK_SST);
      double eos;
      if constexpr (hx_tbl<B>() && !hx_w2<B>()) hx_tbl_local<B>(m);
      if constexpr (hx_w2<B>()) {   // values that live in the tables (hx_dev_member.h)
        HX_W2_LOCAL(m);
        if constexpr (W2K) {
          eos = w2k_eos; m.alkH = w2k_alkH; m.alkL = w2k_alkL;
        } else {
        eos = w2_ld<true>(buf.state, m.npad, HXS_EOS_VEGC, m.moff);
        m.alkH = w2_ld<true>(buf.state, m.npad, HXS_ALK_HL, m.moff);
        m.alkL = w2_ld<true>(buf.state, m.npad, HXS_ALK_LL, m.moff);
        }
      } else {
        eos = PKM(m, PK_EOS);
      }
      double tland = PKM(m, PK_TLAND);
      if constexpr (hx_cons<CON>()) {
        // land-ocean warming ratio: the carbon cycle and the ocean see temperatures derived
        // from global tas, DOECLIM keeps its own (temperature_component.cpp:586-625,722-739)
        // (HXC_LO: some member has one -- no load of the parameter row otherwise)
        // (every lane of the wavefront evaluates it and a lane without a ratio keeps its values:
        // selects, no divergent region -- lo = 0 divides by 1 - flnd)
        if (HX_RARE(cmk & HXC_LO)) {
          const double lo = ldp(buf, HXP_LO_RATIO, mem);
          const bool on = lo != 0 && iy > 1;
          const double tg = D_flnd * tland + (1.0 - D_flnd) * D_bsi * sst;
          const double toa = tg / ((lo * D_flnd) + (1 - D_flnd));
          tland = on ? toa * lo : tland;
          sst = on ? toa / D_bsi : sst;
        }
      }
      double twin = PKM(m, PK_TWIN);
      const double tl_m2 = PKM(m, PK_TL_M2);
      const int iold = iy - 203;
      if constexpr (YAL && !hx_w2<B>()) load_year_a(iy);
      if constexpr (hx_w2<B>()) {
        // (two resident wavefronts hide a load's latency: nothing is requested a phase ahead, so
        // nothing waits in registers through the solver)
        load_year_a(iy);
        if constexpr (!W2K)
        pf_tl_old = hx_ldm<true>(HX_GCD(buf.out[HXO_TLAND]) + (size_t)(iold >= 1 ? iold : 0) * buf.npad, m.moff);
      }
      const double tl_old = pf_tl_old;
#ifndef HX_PF_BEFORE_SOLVER
      // Round 6: next year's window entry and this year's history partial sum are requested HERE, at
      // the head of the year, not ahead of the solver.  They wait in registers through the solver
      // either way -- but the allocator parks them in AGPRs when the step loop starts, a move that
      // needs the value to have arrived: requested 86 instructions before it (the round-6 register
      // allocation; 285 in round 5) the wavefront sat out most of an HBM round trip every year
      // (tools/isa_sim.py: 933 of phase A's 7 382 clocks on that one s_waitcnt vmcnt).  From here
      // they have all of phase A, ~4 000 clocks.
      if constexpr (!hx_w2<B>()) {
        const int iold1 = iy + 1 - 203;
        const bool newblk_a = blk0 < 0 || iy >= blk_end;
        if constexpr (hx_rowio<B>()) {
          pf_tl_old = hx_ldm(HX_GCD(buf.out[HXO_TLAND]) + (size_t)(iold1 >= 1 ? iold1 : 0) * buf.npad, m.moff);
          pf_dpart = hx_ldm(HX_GCD(buf.dpart) + (size_t)(newblk_a ? 0 : iy - blk0) * buf.npad, m.moff);
        } else {
          pf_tl_old = HX_GCD(buf.out[HXO_TLAND])[(size_t)(iold1 >= 1 ? iold1 : 0) * buf.npad + mem];
          pf_dpart = HX_GCD(buf.dpart)[(size_t)(newblk_a ? 0 : iy - blk0) * buf.npad + mem];
        }
      }
#endif
      constexpr int NB = hx_nbc<B>();
      constexpr int SB = (B == HX_DYN) ? 1 : NB;  // (the looped kernels read these where they use them)
      double p_beta[SB], p_wf[SB], p_lnq10[SB], p_mu[SB], p_sigma[SB], s_ffrozen[SB];
      LandK<B> lk;
      if constexpr (B == HX_DYN) load_landk<B>(m, lk);
      if constexpr (B != HX_DYN) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        s_ffrozen[b] = ffrozen_of<B>(m, b);
        if constexpr (hx_w2<B>()) {
          const int pr = HXP_NGLOBAL;
          if (buf.uni_wf) p_wf[b] = HX_CCD(buf.uparams)[pr + HXPB_WF];
          else p_wf[b] = w2_ld<true>(buf.params, m.npad, pr + HXPB_WF, m.moff);
          if (buf.uni_bio) {
            hx_ccd u = HX_CCD(buf.uparams) + pr;
            p_beta[b] = u[HXPB_BETA]; p_mu[b] = u[HXPB_PF_MU]; p_sigma[b] = u[HXPB_PF_SIGMA];
          } else {
            p_beta[b] = w2_ld<true>(buf.params, m.npad, pr + HXPB_BETA, m.moff);
            p_mu[b] = w2_ld<true>(buf.params, m.npad, pr + HXPB_PF_MU, m.moff);
            p_sigma[b] = w2_ld<true>(buf.params, m.npad, pr + HXPB_PF_SIGMA, m.moff);
          }
          if constexpr (W2K) p_lnq10[b] = w2k_lnq;
          else p_lnq10[b] = w2_ld<true>(buf.derived, m.npad, HXD_NGLOBAL, m.moff);
          load_landk<B>(m, lk);
        } else if constexpr (B == 1) {
          constexpr int o = hx_pkb1<B>();
          p_beta[b] = PKM(m, o + PKB_BETA); p_wf[b] = PKM(m, o + PKB_WF);
          p_mu[b] = PKM(m, o + PKB_MU); p_sigma[b] = PKM(m, o + PKB_SIGMA);
          p_lnq10[b] = PKM(m, o + PKB_LNQ10);
          lk.fpf_static[b] = PKM(m, o + PKB_FPF_STATIC);
          lk.rh_ch4_frac[b] = PKM(m, o + PKB_RH_CH4_FRAC);
        } else {
          const int pr = HXP_NGLOBAL + b * HXPB_N;
          // (lean park: row address + the lane's offset, hx_tbl)
          auto ldpm = [&](int row) -> double {
            if constexpr (hx_tbl<B>()) return w2_ld(buf.params, m.npad, row, m.moff);
            else return ldp(buf, row, mem);
          };
          if constexpr (CWF) p_wf[b] = c_wf[b];
          else p_wf[b] = ldpm(pr + HXPB_WF);
          if (buf.uni_bio) {  // beta, permafrost mu/sigma uniform over members: scalar loads
            hx_ccd u = HX_CCD(buf.uparams) + pr;
            p_beta[b] = u[HXPB_BETA]; p_mu[b] = u[HXPB_PF_MU]; p_sigma[b] = u[HXPB_PF_SIGMA];
          } else {
            p_beta[b] = ldpm(pr + HXPB_BETA);
            p_mu[b] = ldpm(pr + HXPB_PF_MU);
            p_sigma[b] = ldpm(pr + HXPB_PF_SIGMA);
          }
          if constexpr (CWF) p_lnq10[b] = c_lnq[b];
          else if constexpr (hx_tbl<B>()) p_lnq10[b] = w2_ld(buf.derived, m.npad, HXD_NGLOBAL + b, m.moff);
          else p_lnq10[b] = ldd(buf, HXD_NGLOBAL + b, mem);
          if (m.upar) {
            lk.fpf_static[b] = m.upar[pr + HXPB_FPF_STATIC];
            lk.rh_ch4_frac[b] = m.upar[pr + HXPB_RH_CH4_FRAC];
          } else {
            lk.fpf_static[b] = ldpm(pr + HXPB_FPF_STATIC);
            lk.rh_ch4_frac[b] = ldpm(pr + HXPB_RH_CH4_FRAC);
          }
        }
      }
      }
      // ---- OH, CH4, O3 ----
      double rh_ch4 = 0;  // D_RH_CH4 as recorded at the end of last year
      if (iy > 1) {
#pragma unroll hx_ur<B>()
        for (int b = 0; b < nbio<B>(m); ++b) rh_ch4 += m_rh_tp_ch4(m, lk, b);
      }
      // (a select, not a branch: four operations, and a divergent region here is one of the
      // places where ROCm 7.2 put register saves ahead of the exec restore -- tools/check_isa.py)
      const double toh_x = ((kc.CCH4 * (PKM(m, PK_LN_CH4) - kc.lnM0) + ya[0]) + ya[1]) + ya[2];
      const double toh = (prev_ch4 != kc.M0) ? toh_x : 0.0;
      // Q10 window: mean over i in [t-200, t-1] of Tland_record(i) =
      // Tland(i-1), 0 before the first record (runtime.cpp:1041-1052)
      if (iy >= 3) {
        twin += tl_m2;  // Tland of year iy-3 enters
        if (iold >= 1) twin -= tl_old;
        PKM(m, PK_TWIN) = twin;
      }
      // ---- the year's logarithms and exponentials, in two batches (hx_dev_math.h): every
      // argument is known here -- both boxes' temperatures (SST of last year), the OH lifetime
      // exponent, the biomes' temperatures
      const double TcH = sst + 18 + (-16.4), TcL = sst + 18 + 2.9;
      double Tb[SB];
      double lg[2 + NB];
      lg[0] = TcH + 273.15; lg[1] = TcL + 273.15;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        Tb[b] = tland * p_wf[b];
        lg[2 + b] = (Tb[b] > 0) ? Tb[b] : 1.0;  // ln(Tb) of the permafrost curve, if Tb > 0
      }
      double ex[13 + 2 * NB];
      ex[12] = -toh;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const double Trm = (iy > 1) ? (twin * p_wf[b]) * 0.005 : 0.0;
        ex[13 + 2 * b] = p_lnq10[b] * (Tb[b] * 0.1);
        ex[14 + 2 * b] = p_lnq10[b] * (Trm * 0.1);
      }
      // The year's other logarithms and exponentials (permafrost curve, OH lifetime, Q10 factors)
      // in their own batches; the twelve T-only constants of the two boxes from the fitted
      // polynomials (chem_constants_fit) -- except for a lane whose box temperature has left the
      // fit's interval (SST anomalies outside -3 ... +13 K): when the wavefront holds such a lane
      // every lane also evaluates the formulas themselves, and each lane keeps the fit's values if
      // it is inside and the formulas' if it is not, so that a member's result does not depend on
      // its neighbours.  (-DHX_NO_CHEM_FIT: the formulas for every lane, experiment builds.)
      if constexpr (NB > 0) {
        double lb[NB > 0 ? NB : 1];
#pragma unroll
        for (int b = 0; b < NB; ++b) lb[b] = lg[2 + b];
        hx_log_batch<(NB > 0 ? NB : 1)>(lb, kc.mtab);
#pragma unroll
        for (int b = 0; b < NB; ++b) lg[2 + b] = lb[b];
      }
      {
        double eb[1 + 2 * NB];
#pragma unroll
        for (int i = 0; i < 1 + 2 * NB; ++i) eb[i] = ex[12 + i];
        hx_exp_chunks<1 + 2 * NB>(eb, kc.mtab);
#pragma unroll
        for (int i = 0; i < 1 + 2 * NB; ++i) ex[12 + i] = eb[i];
      }
#ifndef HX_NO_CHEM_FIT
      const bool fit_in = chem_fit_applies(TcH, TcL);
#ifdef HX_FIT_CARRIED
      if constexpr (!hx_cons<CON>()) {
#pragma unroll
        for (int i = 0; i < 12; ++i) ex[i] = fitc[i];
      } else if constexpr (FITC) {
        if (HX_RARE(cmk & HXC_LO)) chem_constants_fit(TcH, TcL, kc.kfit, ex);
        else {
#pragma unroll
          for (int i = 0; i < 12; ++i) ex[i] = fitc[i];
        }
      } else
#endif
      chem_constants_fit(TcH, TcL, kc.kfit, ex);
      if (__builtin_expect(__any(!fit_in), 0))
#endif
      {
        double l2[2] = {lg[0], lg[1]}, e12[12];
        hx_log_batch<2>(l2, kc.mtab);
      // (literals here: as data -- HxConst::ctab, -DHX_CHEM_TABLE -- the 39 constants arrive in
      // one sweep of scalar loads, 78 SGPRs at once, and 73 of the kernel's SGPRs spill)
#ifdef HX_CHEM_TABLE
        chem_exponents(TcH, l2[0], &e12[0], kc.ctab);
        chem_exponents(TcL, l2[1], &e12[6], kc.ctab);
#else
        chem_exponents(TcH, l2[0], &e12[0]);
        chem_exponents(TcL, l2[1], &e12[6]);
#endif
        hx_exp_chunks<12>(e12, kc.mtab);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
#ifndef HX_NO_CHEM_FIT
          ex[i] = fit_in ? ex[i] : e12[i];
#else
          ex[i] = e12[i];
#endif
        }
      }
      const double tau_oh = kc.TOH0 * ex[12];
      if constexpr (CON) {
      if (HX_RARE(omk & (1ull << HXO_TAU_OH))) sto_(buf, HXO_TAU_OH, (size_t)iy * buf.npad + mem, tau_oh);
      if (HX_RARE(omk & ((1ull << HXO_HL_UPTAKE) | (1ull << HXO_LL_UPTAKE) | (1ull << HXO_HL_DO)))) {
        // sums over the year's stashes start at zero (oceanbox::new_year)
        const size_t o = (size_t)iy * buf.npad + mem;
        if (HX_RARE(omk & (1ull << HXO_HL_UPTAKE))) sto_(buf, HXO_HL_UPTAKE, o, 0.0);
        if (HX_RARE(omk & (1ull << HXO_LL_UPTAKE))) sto_(buf, HXO_LL_UPTAKE, o, 0.0);
        if (HX_RARE(omk & (1ull << HXO_HL_DO))) sto_(buf, HXO_HL_DO, o, 0.0);
      }
      }
      {
        double ch4_em = ya[3];
        if constexpr (hx_cons<CON>()) {
          if (HX_RARE(msk & (1u << HXM_CH4_EM)))
            ch4_em = HX_GCD(buf.mseries[HXM_CH4_EM])[(size_t)iy * buf.npad + mem];
        }
        const double emisTocon =
            ((ch4_em + rh_ch4 * PG_C_TO_TG_CH4) + ya[4]) * kc.inv_UC_CH4;
        const double dCH4 = ((emisTocon - prev_ch4 * kc.inv_Tsoil) - prev_ch4 * kc.inv_Tstrat) -
                            hx_div(prev_ch4, tau_oh);
        ch4 = prev_ch4 + dCH4;
      }
      if constexpr (hx_cons<CON>()) {  // ch4_component.cpp:156-157
        if (HX_RARE(cmk & HXC_CH4)) {
          double c = sh[HXSH_CH4_CON];
          if (HX_RARE(msk & (1u << HXM_CH4_CON))) c = HX_GCD(buf.mseries[HXM_CH4_CON])[(size_t)iy * buf.npad + mem];
          if (!isnan(c)) ch4 = c;
        }
      }
      PKM(m, PK_CH4) = ch4;
      const double ln_ch4 = hx_log(ch4, kc.mtab);
      PKM(m, PK_LN_CH4) = ln_ch4;
      o3 = ((5 * ln_ch4 + ya[5]) + ya[6]) + ya[7];
      // ---- ocean: new year ----
      HX_STAMP(m, 1);   // park reads, the year's log / exp batches, OH / CH4 / O3
      chem_from_exponentials(TcH, &ex[0], O_AsHL, m.kH);
      chem_from_exponentials(TcL, &ex[6], O_AsLL, m.kL);
      chem_poly_constants(m.alkH, m.kH);
      chem_poly_constants(m.alkL, m.kL);
      HX_STAMP(m, 2);   // T-only equilibrium constants of both boxes (from the exponentials)
      m.annualflux_sum = 0; m.nstash = 0; m.nsteps = 0;
      // (the alkalinities were tuned once, right after the spinup: hx_alk_kernel)
      HX_CHEM_SOLVE2(m.kH, m.kL, m.cHL, m.cLL, m.alkH, m.alkL, m.hH, m.hL, m.pco2H, m.pco2L,
                  m.status);
      m.chem_fresh = true;
      HX_STAMP(m, 3);   // year-start carbonate solve
      // ---- slowparameval (t = year-1) ----
      m.ffi = ya[8]; m.daccs = ya[9];
      m.luc_e = ya[10]; m.luc_u = ya[11];
      if constexpr (hx_cons<CON>()) {  // emissions that differ between members
        if (HX_RARE(msk & ((1u << HXM_FFI) | (1u << HXM_DACCS) | (1u << HXM_LUC_E) | (1u << HXM_LUC_U)))) {
        const size_t o = (size_t)iy * buf.npad + mem;
        if (HX_RARE(msk & (1u << HXM_FFI))) m.ffi = HX_GCD(buf.mseries[HXM_FFI])[o];
        if (HX_RARE(msk & (1u << HXM_DACCS))) m.daccs = HX_GCD(buf.mseries[HXM_DACCS])[o];
        if (HX_RARE(msk & (1u << HXM_LUC_E))) m.luc_e = HX_GCD(buf.mseries[HXM_LUC_E])[o];
        if (HX_RARE(msk & (1u << HXM_LUC_U))) m.luc_u = HX_GCD(buf.mseries[HXM_LUC_U])[o];
        }
      }
      m.npp_luc_adjust = hx_div(eos - m.cum_luc_va, eos);
      const double lnc = PKM(m, PK_LN_CO2R);  // = log((atmos C * PGC2PPM) / C0), from last year's phase C
      if constexpr (B == HX_DYN) {
        // looped kernels: the same in chunks of HX_DYN_CHUNK biomes -- a chunk's parameters, pools
        // and last year's values requested together, its exponentials, logarithms and frozen
        // fractions evaluated as batches (hx_dev_math.h), then stored (hx_dev_solver.h: load_bio)
        constexpr int CH = HX_DYN_CHUNK;
        const int nb = m.nb;
        // (the chunk's table values one chunk ahead, like the solver's chunk loops: hx_dev_solver.h,
        // chunk_loop_bio -- a chunk ends in stores to such rows)
        struct ATab { double wf, lnq, beta, pmu, psg, ffz, tfl; };
        auto load_atab = [&](int c0, ATab (&t)[CH]) {
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const int b = min(c0 + j, nb - 1);
            const int pr = HXP_NGLOBAL + b * HXPB_N;
            t[j].wf = w2_ld(buf.params, m.npad, pr + HXPB_WF, m.moff);
            t[j].lnq = w2_ld(buf.derived, m.npad, HXD_NGLOBAL + b, m.moff);
            t[j].beta = w2_ld(buf.params, m.npad, pr + HXPB_BETA, m.moff);
            t[j].pmu = w2_ld(buf.params, m.npad, pr + HXPB_PF_MU, m.moff);
            t[j].psg = w2_ld(buf.params, m.npad, pr + HXPB_PF_SIGMA, m.moff);
            t[j].ffz = ffrozen_of<B>(m, b);
            t[j].tfl = m.tempferts[b];
          }
        };
        ATab anxt[CH];
        load_atab(0, anxt);
        for (int b0 = 0; b0 < nb; b0 += CH) {
          double wf[CH], lnq[CH], beta[CH], pmu[CH], psg[CH], pfv[CH], ffz[CH], tfl[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            wf[j] = anxt[j].wf; lnq[j] = anxt[j].lnq; beta[j] = anxt[j].beta; pmu[j] = anxt[j].pmu;
            psg[j] = anxt[j].psg; ffz[j] = anxt[j].ffz; tfl[j] = anxt[j].tfl;
            pfv[j] = m.pf[min(b0 + j, nb - 1)];
          }
#ifndef HX_DYN_NO_AHEAD_A
          if (b0 + CH < nb) load_atab(b0 + CH, anxt);
#endif
          double Tbc[CH], exc[2 * CH], lgc[CH], dfc[CH], ffc[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            Tbc[j] = tland * wf[j];
            const double Trm = (iy > 1) ? (twin * wf[j]) * 0.005 : 0.0;
            exc[2 * j] = fmax(lnq[j] * (Tbc[j] * 0.1), -746.0);
            exc[2 * j + 1] = fmax(lnq[j] * (Trm * 0.1), -746.0);
            lgc[j] = (Tbc[j] > 0) ? Tbc[j] : 1.0;
          }
          hx_log_batch<CH>(lgc);
          hx_exp_chunks<2 * CH>(exc);
#pragma unroll
          for (int j = 0; j < CH; ++j) dfc[j] = hx_div(lgc[j] - pmu[j], psg[j] * 1.4142135623730951);
          hx_frozen_fraction_batch<CH>(dfc, ffc);
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            if (b0 + j < nb) {
              const int b = b0 + j;
              m.co2fert[b] = 1 + beta[j] * lnc;
              m.tempfertd[b] = exc[2 * j];
              const bool has_pf = pfv[j] != 0.0;
              const double ff = (Tbc[j] > 0) ? ffc[j] : 1.0;
              m.f_new_thaw[b] = has_pf ? ffz[j] - ff : 0.0;
              set_ffrozen<B>(m, b, has_pf ? ff : ffz[j]);
              const double last = (iy > 1) ? tfl[j] : 0.0;
              m.tempferts[b] = fmax(exc[2 * j + 1], last);
            }
          }
#ifdef HX_DYN_NO_AHEAD_A
          if (b0 + CH < nb) load_atab(b0 + CH, anxt);
#endif
        }
      }
      // the frozen fractions of all biomes as one batch (hx_dev_math.h); a biome at or below
      // 0 degC is frozen through (lg holds ln 1 there), one without permafrost keeps what it has
      [[maybe_unused]] double ffb[NB > 0 ? NB : 1];
      if constexpr (NB > 0) {
        double dfr[NB > 0 ? NB : 1];
#pragma unroll
        for (int b = 0; b < NB; ++b) dfr[b] = hx_div(lg[2 + b] - p_mu[b], p_sigma[b] * 1.4142135623730951);
        hx_frozen_fraction_batch<(NB > 0 ? NB : 1)>(dfr, ffb);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        m.co2fert[b] = 1 + p_beta[b] * lnc;
        m.tempfertd[b] = ex[13 + 2 * b];  // exp(ln q10 * Tb / 10)
        {
          const bool has_pf = m.pf[b] != 0.0;
          const double ff = (Tb[b] > 0) ? ffb[b] : 1.0;
          m.f_new_thaw[b] = has_pf ? s_ffrozen[b] - ff : 0.0;
          set_ffrozen<B>(m, b, has_pf ? ff : s_ffrozen[b]);
        }
        const double tfs = ex[14 + 2 * b];  // exp(ln q10 * Trm / 10), Trm = 200-year mean
        const double last = (iy > 1) ? m.tempferts[b] : 0.0;
        m.tempferts[b] = fmax(tfs, last);  // sticky :1054-1059
      }
    }
    HX_FENCE();
    HX_STAMP(m, 4);     // slow parameters
    // ======================= phase B: carbon-cycle solver ====================
    {
      if constexpr (!hx_w2<B>()) {  // next year's window entry, this year's history partial sum (see pf_* above)
        const HxBuffers &buf = args->buf;
        const bool newblk = blk0 < 0 || iy >= blk_end;  // (then the pass has not run yet)
#ifdef HX_PF_BEFORE_SOLVER   // (the round-2 ... round-5 place of the two requests: experiments)
        const int iold1 = iy + 1 - 203;
        if constexpr (hx_rowio<B>()) {
          pf_tl_old = hx_ldm(HX_GCD(buf.out[HXO_TLAND]) + (size_t)(iold1 >= 1 ? iold1 : 0) * buf.npad, m.moff);
          pf_dpart = hx_ldm(HX_GCD(buf.dpart) + (size_t)(newblk ? 0 : iy - blk0) * buf.npad, m.moff);
        } else {
        pf_tl_old = HX_GCD(buf.out[HXO_TLAND])[(size_t)(iold1 >= 1 ? iold1 : 0) * buf.npad + mem];
        pf_dpart = HX_GCD(buf.dpart)[(size_t)(newblk ? 0 : iy - blk0) * buf.npad + mem];
        }
#endif
        if constexpr (KPF) {  // this year's in-block kernel entries (chunks of eight, like their use)
          const int jbp = newblk ? 0 : iy - blk0;
          const int kqp = args->kc.ns - iy - 1 + HX_KPAD + (newblk ? 0 : blk0);
#pragma unroll
          for (int c = 0; c < HX_DBLK / 8; ++c) {
            if (8 * c < jbp + (HF ? 1 : 0)) {   // (the heat-flux sum reads one entry further)
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                if constexpr (hx_rowio<B>()) kpf[8 * c + r] = hx_ldm(HX_GCD(buf.ker) + (size_t)(kqp + 8 * c + r) * buf.npad, m.moff);
                else kpf[8 * c + r] = HX_GCD(buf.ker)[(size_t)(kqp + 8 * c + r) * buf.npad + mem];
              }
            }
          }
        }
        if constexpr (!YAL) {
        hx_ccd shc = HX_CCD(buf.shared) + (size_t)iy * HXSH_STRIDE;
        yc4[0] = shc[HXSH_SQRT_N2O]; yc4[1] = shc[HXSH_RF_OTHER]; yc4[2] = shc[HXSH_RF_AERO]; yc4[3] = shc[HXSH_RF_VOL];
        load_year_a(iy + 1);
        }
      }
      const double year = (double)(args->kc.start_year + iy);
      HX_MASKS_LOCAL();
      YearCon yc{};
      if constexpr (hx_cons<CON>()) {
        hx_ccd sh = HX_CCD(args->buf.shared) + (size_t)iy * HXSH_STRIDE;
        yc.mask = (int)cmk;
        // (the columns are read only under their mask bit -- the stash looks at the values only
        // then; per-member series by their bit of HxBuffers::ms_mask)
        if (yc.mask & HXC_CO2) yc.co2 = sh[HXSH_CO2_CON];
        if constexpr (hx_nbp<CON>()) {
          yc.nbp_hi = sh[HXSH_NBP_CON];
          yc.nbp_lo = (sh - HXSH_STRIDE)[HXSH_NBP_CON];
        }
        if (HX_RARE(msk & ((1u << HXM_CO2_CON) | (1u << HXM_NBP_CON)))) {  // constraints that differ between members
          const HxBuffers &buf = args->buf;
          const size_t o = (size_t)iy * buf.npad + mem;
          if (HX_RARE(msk & (1u << HXM_CO2_CON))) yc.co2 = HX_GCD(buf.mseries[HXM_CO2_CON])[o];
          if constexpr (hx_nbp<CON>()) {
          if (HX_RARE(msk & (1u << HXM_NBP_CON))) {
            yc.nbp_hi = HX_GCD(buf.mseries[HXM_NBP_CON])[o];
            yc.nbp_lo = HX_GCD(buf.mseries[HXM_NBP_CON])[o - buf.npad];
          }
          }
        }
        yc.t_half = year - 0.5;
      }
      solve_year<B, false, CON>(m, args->kc, year - 1.0, year, yc);
      cost_steps += m.nsteps; cost_stash += m.nstash;
    }
    HX_FENCE();
    HX_STAMP(m, 10);    // rest of the solver (loop control, lanes idling through others' segments)
    // ======================= phase C ========================================
    HX_MASKS_LOCAL();
    {
      const HxBuffers &buf = args->buf;
      const HxConst &kc = args->kc;
      const int ns = kc.ns;
      hx_ccd sh = HX_CCD(buf.shared) + (size_t)iy * HXSH_STRIDE;
      if (blk0 < 0 || iy >= blk_end) {
        // new DOECLIM block: this lane's partial sums over its SST history
        blk_end = iy + (blk0 < 0 ? blk_len0 : HX_DBLK);
        blk0 = iy;
        if constexpr (B == 1) {
          // (the block's SST tile starts out as zeros: an in-block term of a year that has not been
          // computed yet then enters the sums as 0 * Ker by itself, without a select per entry)
#pragma unroll
          for (int r = 0; r < HX_DBLK; ++r) s_tblk[r][lane] = 0.0;
        }
#if HX_HAS_MFMA
        if constexpr (!KERPM)
          doeclim_pass_mfma<HF, hx_w2<B>()>(buf.out[HXO_SST], buf.ker, const_cast<double *>(buf.dpart),
                                const_cast<double *>(buf.dpart2), ns, buf.npad, blk0, mem, blk0);
        else
#endif
        doeclim_pass_dev<KERPM, HF, (hx_w2<B>() ? 8 : 16)>(buf.out[HXO_SST], buf.ker, const_cast<double *>(buf.dpart),
                                    const_cast<double *>(buf.dpart2), ns, buf.npad, blk0, mem, blk0);
        HX_FENCE();
        if constexpr (hx_rowio<B>()) pf_dpart = hx_ldm(HX_GCD(buf.dpart), m.moff);
        else pf_dpart = HX_GCD(buf.dpart)[mem];
        HX_STAMP(m, 11);  // DOECLIM history pass (once per HX_DBLK years)
      }
      // (every kernel without the LDS tile of in-block SSTs -- the two-wavefront flavour and the
      // multi-biome kernels -- requests the block's SSTs from the output array all at once:
      // 65 536 x 4 biomes 10.16 -> 10.08 ms; -DHX_TALL_W2_ONLY: chunk after chunk for the others)
#ifndef HX_TALL_W2_ONLY
      constexpr bool TALL = (B != 1);
#else
      constexpr bool TALL = hx_w2<B>();
#endif
      // (ahead of the forcing arithmetic; rows from this year on -- stale values of an earlier
      // run -- are replaced by row 0, the SST anomaly of startDate = 0 for every member, like in
      // the history pass: a scalar row choice, the terms enter as 0 * Ker with no vector select)
      [[maybe_unused]] double Tall[HX_DBLK];
      if constexpr (TALL) {
        // (round 6: requesting only the chunks of eight the year's sum will use -- a wave-uniform test
        // per chunk -- makes every Tall entry a merge of "loaded" and "not": the two-wavefront flavour
        // then needs 256 registers + 44 spilled to scratch instead of 237; profiles/r06_variant_log.md)
#pragma unroll
        for (int r = 0; r < HX_DBLK; ++r) {
          const int i = blk0 + r;
          Tall[r] = hx_ldm<hx_w2<B>()>(HX_GCD(buf.out[HXO_SST]) + (size_t)(i < iy ? i : 0) * buf.npad, m.moff);
        }
      }
      if constexpr (YAL && !hx_w2<B>()) {
        yc4[0] = sh[HXSH_SQRT_N2O]; yc4[1] = sh[HXSH_RF_OTHER]; yc4[2] = sh[HXSH_RF_AERO]; yc4[3] = sh[HXSH_RF_VOL];
      }
      if constexpr (hx_w2<B>()) {  // what the other kernels request ahead of the solver
        pf_dpart = hx_ldm<true>(HX_GCD(buf.dpart) + (size_t)(iy - blk0) * buf.npad, m.moff);
        yc4[0] = sh[HXSH_SQRT_N2O]; yc4[1] = sh[HXSH_RF_OTHER]; yc4[2] = sh[HXSH_RF_AERO]; yc4[3] = sh[HXSH_RF_VOL];
        ch4 = PKM(m, PK_CH4);
        o3 = ((5 * PKM(m, PK_LN_CH4) + sh[HXSH_O3_NOX]) + sh[HXSH_O3_CO]) + sh[HXSH_O3_NMVOC];
      }
      // every HBM value this phase needs, issued back to back
      const double tland = PKM(m, PK_TLAND), sst = PKM(m, PK_SST);
      const double f_prev = PKM(m, PK_F_PREV);
      const double base_tot = PKM(m, PK_BASE_TOT), base_co2 = PKM(m, PK_BASE_CO2);
      const double tl_m1 = PKM(m, PK_TL_M1);
      double p_aero, p_vol;
      if constexpr (hx_w2<B>()) {
        HX_W2_LOCAL(m);
        if (buf.uni_avc) {   // (scalar loads when every member shares them)
          hx_ccd u = HX_CCD(buf.uparams);
          p_aero = u[HXP_AERO]; p_vol = u[HXP_VOL]; m.C0 = u[HXP_C0];
        } else {
          p_aero = w2_ld<true>(buf.params, m.npad, HXP_AERO, m.moff); p_vol = w2_ld<true>(buf.params, m.npad, HXP_VOL, m.moff);
          m.C0 = w2_ld<true>(buf.params, m.npad, HXP_C0, m.moff);
        }
      } else {
        if constexpr (hx_tbl<B>()) hx_tbl_local<B>(m);
        p_aero = PKM(m, PK_AERO); p_vol = PKM(m, PK_VOL);
      }
#define HXDK(row) dconst<B>(m, (row))
      const double dA0 = HXDK(HXD_A0), dA1 = HXDK(HXD_A1), dA2 = HXDK(HXD_A2), dA3 = HXDK(HXD_A3),
                   dIB0 = HXDK(HXD_IB0), dIB1 = HXDK(HXD_IB1), dIB2 = HXDK(HXD_IB2),
                   dIB3 = HXDK(HXD_IB3), dQC1 = HXDK(HXD_QC1), dQC2 = HXDK(HXD_QC2),
                   dDQ1 = HXDK(HXD_DQ1), dDQ2 = HXDK(HXD_DQ2), dDPS = HXDK(HXD_DPSCALE),
                   dHFS = want_hf ? HXDK(HXD_HFSCALE) : 0.0;
#undef HXDK
      const int jb = iy - blk0;
      double dpast = pf_dpart;
      double hint = 0.0;
      if (want_hf) {
        if constexpr (hx_rowio<B>()) hint = hx_ldm<hx_w2<B>()>(HX_GCD(buf.dpart2) + (size_t)jb * buf.npad, m.moff);
        else hint = HX_GCD(buf.dpart2)[(size_t)jb * buf.npad + mem];
      }
      // ---- forcing ----
      const double co2c = m.atmos * PGC2PPM;
      const double ln_co2r = hx_log(hx_div(co2c, m.C0), kc.mtab);
      PKM(m, PK_LN_CO2R) = ln_co2r;
      double rf_tot = 0, rf_co2 = 0;
      if (iy >= kc.baseyear_idx) {
        const double a1 = -2.4785e-7, b1 = 7.5906e-4, c1 = -2.1492e-3, d1 = 5.2488;
        const double a2 = -3.4197e-4, b2 = 2.5455e-4, c2 = -2.4357e-4, d2 = 0.12173;
        const double a3 = -8.9603e-5, b3 = -1.2462e-4, d3 = 0.045194;
        double sqN = yc4[0], sqN0 = kc.sqrtN0, rf_other = yc4[1];
        if constexpr (hx_cons<CON>()) {  // N2O / halocarbon parameters that differ between members
          if (HX_RARE(msk & (1u << HXM_N2O))) {
            sqN = hx_sqrt(HX_GCD(buf.mseries[HXM_N2O])[(size_t)iy * buf.npad + mem]);
            sqN0 = hx_sqrt(HX_GCD(buf.mseries[HXM_N2O])[mem]);
            rf_other = HX_GCD(buf.mseries[HXM_RF_OTHER])[(size_t)iy * buf.npad + mem];
          }
        }
        const double sqM = hx_sqrt(ch4), sqC = hx_sqrt(co2c);
        const double C_alpha_max = m.C0 - (b1 / (2 * a1));
        double alpha_prime;
        if (co2c > C_alpha_max) alpha_prime = d1 - ((b1 * b1) / (4 * a1));
        else if (m.C0 < co2c && co2c < C_alpha_max)
          alpha_prime = d1 + a1 * ((co2c - m.C0) * (co2c - m.C0)) + b1 * (co2c - m.C0);
        else alpha_prime = d1;
        const double sarf_co2 = (alpha_prime + c1 * sqN) * ln_co2r;
        const double fco2 = (sarf_co2 * kc.delta_co2) + sarf_co2;
        const double sarf_n2o = (a2 * sqC + b2 * sqN + c2 * sqM + d2) * (sqN - sqN0);
        const double fn2o = (kc.delta_n2o * sarf_n2o) + sarf_n2o;
        const double sarf_ch4 = (a3 * sqM + b3 * sqN + d3) * (sqM - kc.sqrtM0);
        const double fch4 = (kc.delta_ch4 * sarf_ch4) + sarf_ch4;
        const double fh2o = 0.0485 * ((ch4 - kc.M0f) * kc.inv_h2o_span);
        const double fo3 = kc.o3_rf * o3;  // (0 with [ozone] enabled=0, forcing_component.cpp:392)
        double ftot = ((((((fco2 + fn2o) + fch4) + fh2o) + fo3) + rf_other) +
                       p_aero * yc4[2]) +
                      p_vol * yc4[3];
        if constexpr (hx_cons<CON>()) {  // forcing_component.cpp:498-505
          if (HX_RARE(cmk & HXC_FTOT)) {
            double c = sh[HXSH_FTOT_CON];
            if (HX_RARE(msk & (1u << HXM_FTOT_CON))) c = HX_GCD(buf.mseries[HXM_FTOT_CON])[(size_t)iy * buf.npad + mem];
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
      HX_STAMP(m, 12);    // phase C loads + forcing
      // ---- DOECLIM: history before the block (doeclim_pass_dev) + in-block terms ----
      double tl_new, sst_new, heatflux = 0, tgav, flux_mixed = 0, flux_interior = 0;
      {
        const int j = jb;
        // Ker is stored with HX_KPAD zeros in front: entry k lives at k + HX_KPAD
        const int kq = ns - iy - 1 + HX_KPAD;
        auto ldk = [&](int idx) -> double {
          if constexpr (KERPM && hx_rowio<B>()) return hx_ldm<hx_w2<B>()>(HX_GCD(buf.ker) + (size_t)idx * buf.npad, m.moff);
          else if constexpr (KERPM) return HX_GCD(buf.ker)[(size_t)idx * buf.npad + mem];
          else return HX_CCD(buf.ker)[idx];
        };
        // in chunks of 8 so that the loads of a chunk are in flight together (one exposed
        // latency per chunk instead of per entry); entries from this year on enter as 0 * Ker,
        // which leaves the sums bit for bit what the entry-by-entry loop gives
        const int nchunk = (jb + 7) >> 3;
        if constexpr (TALL) {
          // (the block's SSTs were requested at the head of this phase: Tall)
#pragma unroll
          for (int c = 0; c < HX_DBLK / 8; ++c) {
            if (c < nchunk) {
              const int i0 = blk0 + 8 * c;
              double K[8], K2[8];
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                if constexpr (KPF) {
                  K[r] = kpf[8 * c + r];
                  K2[r] = want_hf ? kpf[8 * c + r + 1] : 0.0;
                } else {
                K[r] = ldk(kq + i0 + r);
                K2[r] = want_hf ? ldk(kq + i0 + r + 1) : 0.0;
                }
              }
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                dpast += Tall[8 * c + r] * K[r];
                if (want_hf) hint += Tall[8 * c + r] * K2[r];
              }
            }
          }
        } else if constexpr (KPF) {   // (one biome: the SSTs from the LDS tile, the entries from kpf)
#pragma unroll
          for (int c = 0; c < HX_DBLK / 8; ++c) {
            if (c < nchunk) {
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const double t = s_tblk[8 * c + r][lane];   // (zeros from this year on)
                dpast += t * kpf[8 * c + r];
                if (want_hf) hint += t * kpf[8 * c + r + 1];
              }
            }
          }
        } else
        for (int c = 0; c < nchunk; ++c) {
          const int i0 = blk0 + 8 * c;
          double T[8], K[8], K2[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int i = i0 + r;
            if constexpr (B == 1) T[r] = s_tblk[8 * c + r][lane];
            else T[r] = HX_GCD(buf.out[HXO_SST])[(size_t)(i < ns ? i : ns - 1) * buf.npad + mem];
            K[r] = ldk(kq + i);
            K2[r] = want_hf ? ldk(kq + i + 1) : 0.0;
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            // (B == 1: the tile holds zeros from this year on, see the block start)
            const double t = (B == 1 || i0 + r < iy) ? T[r] : 0.0;
            dpast += t * K[r];
            if (want_hf) hint += t * K2[r];
          }
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
        if constexpr (hx_cons<CON>()) {  // user-supplied temperature :510-525
          if (HX_RARE(cmk & HXC_TAS)) {
            double c = sh[HXSH_TAS_CON];
            if (HX_RARE(msk & (1u << HXM_TAS_CON))) c = HX_GCD(buf.mseries[HXM_TAS_CON])[(size_t)iy * buf.npad + mem];
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
      if constexpr (hx_cons<CON>()) {
        if (HX_RARE(cmk & HXC_LO)) {   // (selects, like at the year start)
          const double lo = ldp(buf, HXP_LO_RATIO, mem);
          const bool on = lo != 0;
          const double tg0 = D_flnd * tland + (1.0 - D_flnd) * D_bsi * sst;
          tl_seen = (on && iy > 1) ? (tg0 / ((lo * D_flnd) + (1 - D_flnd))) * lo : tl_seen;
          const double toa = tgav / ((lo * D_flnd) + (1 - D_flnd));
          tl_rep = on ? toa * lo : tl_rep;
          sst_rep = on ? toa / D_bsi : sst_rep;
        }
      }
      HX_STAMP(m, 13);    // DOECLIM in-block sum + year step
      PKM(m, PK_F_PREV) = rf_tot;
      PKM(m, PK_TL_M2) = tl_m1;  // Tland of years iy-2, iy-1
      PKM(m, PK_TL_M1) = tl_seen;  // for the next year
      PKM(m, PK_TLAND) = tl_new;
      PKM(m, PK_SST) = sst_new;
      if constexpr (W2K) {   // next year's Q10-window entry, ahead of the fit (see w2k_* above)
        const int iold1 = iy + 1 - 203;
        pf_tl_old = hx_ldm<true>(HX_GCD(buf.out[HXO_TLAND]) + (size_t)(iold1 >= 1 ? iold1 : 0) * buf.npad, m.moff);
      }
#ifdef HX_FIT_CARRIED
      if constexpr (FITC) chem_constants_fit(sst_new + 18 + (-16.4), sst_new + 18 + 2.9, kc.kfit, fitc);
#endif
      // ---- outputs ----
      HX_MASKS_LOCAL();
      const size_t o = (size_t)iy * buf.npad + mem;
      if constexpr (hx_rowio<B>()) {  // (wave-uniform row address + the lane's 32-bit offset)
        const size_t orow = (size_t)iy * buf.npad;
        hx_stm<hx_w2<B>()>(HX_GD(buf.out[HXO_SST]) + orow, m.moff, sst_new);
        hx_stm<hx_w2<B>()>(HX_GD(buf.out[HXO_TLAND]) + orow, m.moff, tl_rep);
        if constexpr (hx_cons<CON>()) { if (HX_RARE(omk & (1ull << HXO_SST_LO))) hx_stm<hx_w2<B>()>(HX_GD(buf.out[HXO_SST_LO]) + orow, m.moff, sst_rep); }
        if (buf.out[HXO_CO2]) hx_stm<hx_w2<B>()>(HX_GD(buf.out[HXO_CO2]) + orow, m.moff, co2c);
        if (buf.out[HXO_TGAV]) hx_stm<hx_w2<B>()>(HX_GD(buf.out[HXO_TGAV]) + orow, m.moff, tgav);
      } else {
      sto_(buf, HXO_SST, o, sst_new);
      sto_(buf, HXO_TLAND, o, tl_rep);
      if constexpr (hx_cons<CON>()) { if (HX_RARE(omk & (1ull << HXO_SST_LO))) sto_(buf, HXO_SST_LO, o, sst_rep); }
      if (buf.out[HXO_CO2]) sto_(buf, HXO_CO2, o, co2c);
      if (buf.out[HXO_TGAV]) sto_(buf, HXO_TGAV, o, tgav);
      }
      if (HX_RARE(buf.out_rare)) {  // (one test instead of ~25 pointer loads and branches a year)
      // (and inside: a bit of HxBuffers::out_mask0 per output instead of its pointer)
      // (in groups: a run that records one diagnostic skips the others a group at a time)
      unsigned long long om = buf.out_mask0;
      if constexpr (CON) om = omk;
      constexpr unsigned long long OM_G1 =
          (1ull << HXO_RF_TOT) | (1ull << HXO_RF_CO2) | (1ull << HXO_OCEAN_C) | (1ull << HXO_HL_PH) |
          (1ull << HXO_LL_PH) | (1ull << HXO_ATMOS_C) | (1ull << HXO_HEATFLUX) | (1ull << HXO_CH4) |
          (1ull << HXO_O3) | (1ull << HXO_EARTH_C) | (1ull << HXO_NBP) | (1ull << HXO_OCEAN_UPTAKE) |
          (1ull << HXO_NSTASH) | (1ull << HXO_NSTEPS);
      constexpr unsigned long long OM_G3 =
          (1ull << HXO_GMST) | (1ull << HXO_FLUX_MIXED) | (1ull << HXO_FLUX_INTERIOR) | (1ull << HXO_C_HL) |
          (1ull << HXO_C_LL) | (1ull << HXO_C_IO) | (1ull << HXO_C_DO) | (1ull << HXO_PCO2_HL) |
          (1ull << HXO_PCO2_LL);
      if (HX_RARE(om & OM_G1)) {
      if (HX_RARE(om & (1ull << HXO_RF_TOT))) sto_(buf, HXO_RF_TOT, o, rf_tot);
      if (HX_RARE(om & (1ull << HXO_RF_CO2))) sto_(buf, HXO_RF_CO2, o, rf_co2);
      if (HX_RARE(om & (1ull << HXO_OCEAN_C))) sto_(buf, HXO_OCEAN_C, o, m.cDO + m.cIO + m.cLL + m.cHL);
      if (HX_RARE(om & (1ull << HXO_HL_PH))) sto_(buf, HXO_HL_PH, o, -log10(m.hH));
      if (HX_RARE(om & (1ull << HXO_LL_PH))) sto_(buf, HXO_LL_PH, o, -log10(m.hL));
      if (HX_RARE(om & (1ull << HXO_ATMOS_C))) sto_(buf, HXO_ATMOS_C, o, m.atmos);
      if (HX_RARE(om & (1ull << HXO_HEATFLUX))) sto_(buf, HXO_HEATFLUX, o, heatflux);
      if (HX_RARE(om & (1ull << HXO_CH4))) sto_(buf, HXO_CH4, o, ch4);
      if (HX_RARE(om & (1ull << HXO_O3))) sto_(buf, HXO_O3, o, o3);
      if (HX_RARE(om & (1ull << HXO_EARTH_C))) sto_(buf, HXO_EARTH_C, o, m.earth);
      if (HX_RARE(om & (1ull << HXO_NBP))) sto_(buf, HXO_NBP, o, m.nbp);
      if (HX_RARE(om & (1ull << HXO_OCEAN_UPTAKE))) sto_(buf, HXO_OCEAN_UPTAKE, o, m.annualflux_sum);
      if (HX_RARE(om & (1ull << HXO_NSTASH))) sto_(buf, HXO_NSTASH, o, (double)m.nstash);
      if (HX_RARE(om & (1ull << HXO_NSTEPS))) sto_(buf, HXO_NSTEPS, o, (double)m.nsteps);
      }
      if (HX_RARE(om & ((1ull << HXO_PERMAFROST_C) | (1ull << HXO_VEG_C) | (1ull << HXO_DET_C) |
                        (1ull << HXO_SOIL_C) | (1ull << HXO_THAWED_C)))) {
        double v = 0, d = 0, s = 0, p = 0, th = 0;
#pragma unroll hx_ur<B>()
        for (int b = 0; b < nbio<B>(m); ++b) { v += m.veg[b]; d += m.det[b]; s += m.soil[b];
                                       p += m.pf[b]; th += m.thawed[b]; }
        if (HX_RARE(om & (1ull << HXO_PERMAFROST_C))) sto_(buf, HXO_PERMAFROST_C, o, p);
        if (HX_RARE(om & (1ull << HXO_VEG_C))) sto_(buf, HXO_VEG_C, o, v);
        if (HX_RARE(om & (1ull << HXO_DET_C))) sto_(buf, HXO_DET_C, o, d);
        if (HX_RARE(om & (1ull << HXO_SOIL_C))) sto_(buf, HXO_SOIL_C, o, s);
        if (HX_RARE(om & (1ull << HXO_THAWED_C))) sto_(buf, HXO_THAWED_C, o, th);
      }
      if constexpr (CON) {  // diagnostics of the extended kernel
      if (HX_RARE(om & OM_G3)) {
      if (HX_RARE(om & (1ull << HXO_GMST))) sto_(buf, HXO_GMST, o, D_flnd * tl_new + (1.0 - D_flnd) * sst_new);
      if (HX_RARE(om & (1ull << HXO_FLUX_MIXED))) sto_(buf, HXO_FLUX_MIXED, o, flux_mixed);
      if (HX_RARE(om & (1ull << HXO_FLUX_INTERIOR))) sto_(buf, HXO_FLUX_INTERIOR, o, flux_interior);
      if (HX_RARE(om & (1ull << HXO_C_HL))) sto_(buf, HXO_C_HL, o, m.cHL);
      if (HX_RARE(om & (1ull << HXO_C_LL))) sto_(buf, HXO_C_LL, o, m.cLL);
      if (HX_RARE(om & (1ull << HXO_C_IO))) sto_(buf, HXO_C_IO, o, m.cIO);
      if (HX_RARE(om & (1ull << HXO_C_DO))) sto_(buf, HXO_C_DO, o, m.cDO);
      if (HX_RARE(om & (1ull << HXO_PCO2_HL))) sto_(buf, HXO_PCO2_HL, o, m.pco2H);
      if (HX_RARE(om & (1ull << HXO_PCO2_LL))) sto_(buf, HXO_PCO2_LL, o, m.pco2L);
      }
      if (HX_RARE(om & (1ull << HX_OM_BIOME_ANY))) {  // "<biome>.veg_c" ...: pools and factors of each biome
        LandK<B> lkb;
        load_landk<B>(m, lkb);
#pragma unroll hx_ur<B>()
        for (int b = 0; b < nbio<B>(m); ++b) {
          auto putb = [&](int k, double v) { if (buf.out[HXO_B(k, b)]) sto_(buf, HXO_B(k, b), o, v); };
          putb(HXOB_VEG, m.veg[b]); putb(HXOB_DET, m.det[b]); putb(HXOB_SOIL, m.soil[b]);
          putb(HXOB_PF, m.pf[b]); putb(HXOB_THAWED, m.thawed[b]);
          putb(HXOB_RH_CH4, m_rh_tp_ch4(m, lkb, b));
          putb(HXOB_F_FROZEN, ffrozen_of<B>(m, b));
          putb(HXOB_TEMPFERTD, m.tempfertd[b]); putb(HXOB_TEMPFERTS, m.tempferts[b]);
        }
      }
      if (HX_RARE(om & ((1ull << HXO_RH_CH4) | (1ull << HXO_F_FROZEN)))) {
        // record_state: RH_ch4 = rh_ftpa_ch4 of the year-end pools (simpleNbox.cpp:800-812);
        // f_frozen: permafrost-weighted mean over biomes, 1 without permafrost (:492-514)
        LandK<B> lk;
        load_landk<B>(m, lk);
        double rch4 = 0, ptot = 0, ff = 0;
#pragma unroll hx_ur<B>()
        for (int b = 0; b < nbio<B>(m); ++b) { rch4 += m_rh_tp_ch4(m, lk, b); ptot += m.pf[b]; }
        if (ptot > 0.0) {
#pragma unroll hx_ur<B>()
          for (int b = 0; b < nbio<B>(m); ++b) ff += (m.pf[b] / ptot) * ffrozen_of<B>(m, b);
        } else ff = 1.0;
        if (HX_RARE(om & (1ull << HXO_RH_CH4))) sto_(buf, HXO_RH_CH4, o, rch4);
        if (HX_RARE(om & (1ull << HXO_F_FROZEN))) sto_(buf, HXO_F_FROZEN, o, ff);
      }
      }
      }  // out_rare
      if constexpr (CON >= 2) {  // CSVFluxPoolVisitor: the pools, once a year (their origins -- the
        // year's matrix and masks -- are where the year's last stash left them, hx_dev_track.h)
        if (buf.track_out_f && kc.trk_iy >= 0 && iy >= kc.trk_iy) {
          const int nbt = nbio<B>(m);
          const size_t slot = (size_t)(iy - kc.trk_iy) + 1;
          hx_gd ov = HX_GD(buf.track_out_v) +
                     (((size_t)blockIdx.x * buf.trk_slots + slot) * (size_t)hx_trk_vrows(nbt) * 64 + m.lane);
          ov[TKP_ATM * 64] = m.atmos; ov[TKP_EARTH * 64] = m.earth;
#pragma unroll 1
          for (int b = 0; b < nbt; ++b) {
            hx_gd ob = ov + (2 + 5 * b) * 64;
            ob[0] = m.veg[b]; ob[64] = m.det[b]; ob[128] = m.soil[b]; ob[192] = m.pf[b];
            ob[256] = m.thawed[b];
          }
          hx_gd oo = ov + (2 + 5 * nbt) * 64;
          oo[0] = m.cHL; oo[64] = m.cLL; oo[128] = m.cIO; oo[192] = m.cDO;
          if constexpr (CON == 3) track_post(s_trk_cmd, TRKC_YEAR, iy);  // the companion writes the maps
        }
      }
      HX_STAMP(m, 14);    // outputs
      if (HX_RARE(buf.hist)) {  // Core::reset(date) needs every component's state of every year
        double *slab = buf.hist + (size_t)iy * (size_t)HX_NSTATE(nbio<B>(m)) * buf.npad;
        store_state<B>(buf, mem, m, slab);
        store_park_state<B>(buf, mem, m, slab);
        HX_GU(buf.hist_status)[(size_t)iy * buf.npad + mem] = m.status;
      }
    }
  }
  HX_FENCE();
  if constexpr (CON == 3) track_post(s_trk_cmd, TRKC_DONE, 0);
  store_state<B>(args->buf, mem, m);
  store_park_state<B>(args->buf, mem, m);
  // (a dopri5 pass of the wavefront costs ~2.7k cycles, a stash ~3.4k: tools/prof/phase_clock.py)
  if (args->buf.cost) HX_GD(args->buf.cost)[mem] += (double)(4 * cost_steps + 5 * cost_stash);
  hx_wave_stamp(args->buf, blockIdx.x, 1, lane);
#ifdef HX_PHASE_CLOCK
  if (args->buf.out[HXO_TGAV])
    for (int k = 0; k < HX_NCLK; ++k)
      HX_GD(args->buf.out[HXO_TGAV])[(size_t)k * args->buf.npad + mem] = (double)hx_s_clk[k];
#endif
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
// status |= the member's own derive-time flags (HXD_FLAG row), after the spinup
__global__ void hx_or_flags_kernel(unsigned *status, const double *flag_row, int npad) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  if (mem >= npad) return;
  status[mem] |= (unsigned)flag_row[mem];
}

// ===========================================================================
// N2OComponent::run (src/n2o_component.cpp:152-191) and HalocarbonComponent::run
// (src/halocarbon_component.cpp:181-229) per member, for ensembles whose N2O or halocarbon
// parameters differ between members (otherwise the host runs them once while it builds the
// per-year table).  Neither depends on the carbon-climate state, so they run ahead of the year
// loop, one member per thread: N2O[t] and the year's halocarbon + albedo + misc forcing.
//   par [3 + 3 nh][npad]: N0, TN2O0, UC_N2O, then tau, rho, delta of each gas
//   ser [4 + 2 nh][ns]:   N2O emissions (anthropogenic + natural), N2O constraint (NaN = none),
//                         RF_albedo, RF_misc, then per gas the year's concentration increment
//                         per unit lifetime (emissions / molar mass / 0.18) and its constraint
//   h0 [nh]: preindustrial concentrations.  Gases in the order their forcings are summed.
// ===========================================================================
__global__ __launch_bounds__(64) void hx_gas_kernel(const double *par, const double *ser,
                                                    const double *h0, int nh, int ns, int npad,
                                                    double *n2o_out, double *rf_other_out) {
  const int mem = blockIdx.x * blockDim.x + threadIdx.x;
  if (mem >= npad) return;
  const size_t np = (size_t)npad;
  const double N0 = par[mem], TN2O0 = par[np + mem], UC = par[2 * np + mem];
  const double *em = ser, *ncon = ser + ns, *alb = ser + 2 * (size_t)ns, *misc = ser + 3 * (size_t)ns;
  constexpr int MAXH = 40;
  double conc[MAXH], expfac[MAXH];
  for (int h = 0; h < nh && h < MAXH; ++h) {
    conc[h] = h0[h];
    expfac[h] = exp(-(1 / par[(size_t)(3 + 3 * h) * np + mem]));
  }
  const double N0f = isnan(ncon[0]) ? N0 : ncon[0];  // n2o_component.cpp:137-145
  double n2o = N0f;
  n2o_out[mem] = n2o;
  rf_other_out[mem] = (0.0 + alb[0]) + misc[0];
  for (int iy = 1; iy < ns; ++iy) {
    const double tau_n = TN2O0 * pow(n2o / N0f, -0.05);
    n2o = n2o + (em[iy] / UC - n2o / tau_n);
    if (!isnan(ncon[iy])) n2o = ncon[iy];
    double rf_h = 0.0;
    for (int h = 0; h < nh && h < MAXH; ++h) {
      const double tau = par[(size_t)(3 + 3 * h) * np + mem], rho = par[(size_t)(4 + 3 * h) * np + mem],
                   delta = par[(size_t)(5 + 3 * h) * np + mem];
      const double dconc = ser[(size_t)(4 + 2 * h) * ns + iy], hc = ser[(size_t)(5 + 2 * h) * ns + iy];
      conc[h] = conc[h] * expfac[h] + dconc * tau * (1.0 - expfac[h]);
      if (!isnan(hc)) conc[h] = hc;
      const double rf_un = rho * conc[h];
      rf_h = rf_h + (rf_un + delta * rf_un);
    }
    n2o_out[(size_t)iy * np + mem] = n2o;
    rf_other_out[(size_t)iy * np + mem] = (rf_h + alb[iy]) + misc[iy];
  }
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
// The last step of the multi-GPU statistics (hx_fleet.cpp): slots[rank][row][5] holds every
// rank's {count, sum, sum of squares, min, max} of a (variable, year) row, gathered by ONE
// ncclAllGather; every rank folds them in rank order, so the result is bit-identical everywhere.
// ===========================================================================
__global__ __launch_bounds__(256) void hx_combine_stats_kernel(const double *slots, int world,
                                                               int rows, double *out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const double *p = slots + (size_t)r * 5;
  double cnt = p[0], s = p[1], s2 = p[2], mn = p[3], mx = p[4];
  for (int k = 1; k < world; ++k) {
    p += (size_t)rows * 5;
    cnt += p[0]; s += p[1]; s2 += p[2]; mn = fmin(mn, p[3]); mx = fmax(mx, p[4]);
  }
  double *o = out + (size_t)r * 5;
  o[0] = cnt; o[1] = s; o[2] = s2; o[3] = mn; o[4] = mx;
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
#ifndef HX_HOST_EMULATION
  __builtin_amdgcn_s_setprio(3);   // (see hx_spinup_kernel)
#endif
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
  double sqN = sh[HXSH_SQRT_N2O], sqN0 = a.sqrtN0;
  if (a.n2o_members) { sqN = sqrt(a.n2o_members[o]); sqN0 = sqrt(a.n2o_members[mem]); }
  if (kind == HXG_RF_O3) return 0.042 * a.o3[o];
  const double ch4 = a.ch4[o], sqM = sqrt(ch4);
  if (kind == HXG_RF_H2O) return 0.0485 * ((ch4 - a.M0f) / (1831 - a.M0f));
  if (kind == HXG_RF_CH4) {
    const double sarf = (a3 * sqM + b3 * sqN + d3) * (sqM - a.sqrtM0);
    return (a.delta_ch4 * sarf) + sarf;
  }
  const double sqC = sqrt(a.co2[o]);
  const double sarf = (a2 * sqC + b2 * sqN + c2 * sqM + d2) * (sqN - sqN0);
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
// dynamic LDS of a looped kernel for nb biomes (hx_dev_member.h: hx_npark_dyn)
static size_t hx_dyn_lds_bytes(int nb) { return (size_t)hx_npark_dyn(nb) * 64 * sizeof(double); }
static void hx_allow_dynamic_lds(const void *kernel, size_t bytes) {
#if HX_HAS_MFMA   // (the host-emulation build has no such attribute)
  if (bytes > 64 * 1024)   // beyond the default window a kernel has to be told
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
#else
  (void)kernel; (void)bytes;
#endif
}

// experiment builds (tools/prof/build_variant.sh <name> -DHX_ONLY_B=<biomes>, with HX_MINIMAL_BUILD):
// nothing but that biome count's kernels (0: the looped ones) -- a compile of seconds
#ifdef HX_ONLY_B
#define HX_ONLY_B_IS(b) (HX_ONLY_B == (b))
#else
#define HX_ONLY_B_IS(b) 1
#endif
// biome counts from this one on take the looped kernels (default: 9; HECTOR_AMD_LOOPED_BIOMES_FROM=5
// sends 5-8 there as well -- the tests hold the unrolled and the looped kernels against each other)
static int hx_looped_from() {   // (read at every launch: a test switches it between two cores)
  const char *e = getenv("HECTOR_AMD_LOOPED_BIOMES_FROM");
  const int x = e ? atoi(e) : 9;
  return x < 5 ? 5 : x;
}

hipError_t hx_launch_spinup(int B, const HxArgs *d_args, int nmem_launch, int *d_steps,
                            hipStream_t st) {
  const int blocks = (nmem_launch + 63) / 64;
#ifdef HX_W2_ONLY
  if (B != 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hx_spinup_kernel<1>, dim3(blocks), dim3(64), 0, st, d_args, d_steps);
  return hipGetLastError();
#else
  switch (B >= hx_looped_from() ? HX_BDYN + 1 : B) {
#if HX_ONLY_B_IS(1)
    case 1: hipLaunchKernelGGL(hx_spinup_kernel<1>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
#endif
#ifndef HX_MINIMAL_BUILD
    case 2: hipLaunchKernelGGL(hx_spinup_kernel<2>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
    case 3: hipLaunchKernelGGL(hx_spinup_kernel<3>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
#endif
#if HX_ONLY_B_IS(4)
    case 4: hipLaunchKernelGGL(hx_spinup_kernel<4>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
#endif
#if HX_ONLY_B_IS(5)
    case 5: hipLaunchKernelGGL(hx_spinup_kernel<5>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
#endif
#if HX_ONLY_B_IS(6)
    case 6: hipLaunchKernelGGL(hx_spinup_kernel<6>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
#endif
#if HX_ONLY_B_IS(7)
    case 7: hipLaunchKernelGGL(hx_spinup_kernel<7>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
#endif
#if HX_ONLY_B_IS(8)
    case 8: hipLaunchKernelGGL(hx_spinup_kernel<8>, dim3(blocks), dim3(64), 0, st, d_args, d_steps); break;
#endif
#if HX_ONLY_B_IS(0)
    default:
      if (B < 1 || B > HX_BDYN) return hipErrorInvalidValue;
      {
        const size_t lds = hx_dyn_lds_bytes(B);
        hx_allow_dynamic_lds(reinterpret_cast<const void *>(&hx_spinup_kernel<HX_DYN>), lds);
        hipLaunchKernelGGL(hx_spinup_kernel<HX_DYN>, dim3(blocks), dim3(64), lds, st, d_args, d_steps);
      }
#else
    default: return hipErrorInvalidValue;
#endif
  }
  return hipGetLastError();
#endif
}

hipError_t hx_launch_alk(const HxArgs *d_args, int nmem_launch, hipStream_t st) {
  hipLaunchKernelGGL(hx_alk_kernel, dim3((nmem_launch + 63) / 64), dim3(64), 0, st, d_args,
                     nmem_launch);
  return hipGetLastError();
}

// biome counts whose carbon tracking runs on companion wavefronts (tools/prof/tracking_times.py)
// (8 192 members tracked 1750-2300: one biome 11.4 ms on companions against 29.9 ms inside the
//  stash, two biomes 23.5 / 32.1 ms, three 70.9 / 60.3 ms, four 215 / 82 ms -- their 147 and 234
//  fractions a wavefront spill; a block of four 512-register wavefronts owns a CU, so two biomes
//  take the companions only while every block gets a CU of its own: 32 768 members 49.8 / 41.3 ms)
#ifndef HX_TRK_COMPANION_MAXB
#define HX_TRK_COMPANION_MAXB 2
#endif
template <int B>
static void launch_run_b(const HxArgs *d_args, int npad, bool hf, bool kpm, int con,
                         int iy_from, int iy_to, hipStream_t st, int nb = B, bool two_wave = false,
                         int cus = 256) {
  const int blocks = npad / 64;
  const size_t lds = (B == HX_DYN) ? hx_dyn_lds_bytes(nb) : 0;
  if constexpr (B == HX_DYN) {
    for (const void *k : {reinterpret_cast<const void *>(&hx_run_kernel<B, false, false, 0>),
#ifndef HX_MINIMAL_BUILD
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, false, 0>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, false, true, 0>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, true, 0>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, false, 1>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, true, 1>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, false, -1>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, false, false, -1>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, true, -1>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, false, 2>),
                          reinterpret_cast<const void *>(&hx_run_kernel<B, true, true, 2>),
#endif
                          static_cast<const void *>(nullptr)})
      if (k)
      hx_allow_dynamic_lds(k, lds);
  }
  // CON = -2 (the plain kernel + the extended one's diagnostics, hx_dev_solver.h) is built for one
  // to four biomes and for the two-wavefront flavour; other biome counts take CON = -1 for it
  if (con == -2 && !(B >= 1 && B <= 4)) con = -1;
  if constexpr (B == 1) {
    // the flavour built for two resident wavefronts per SIMD (EnsembleCore::run decides)
    if (two_wave && con == -2) {
      if (kpm) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, true, -2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else if (hf) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, false, -2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, false, false, -2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      return;
    }
    if (two_wave && con == 1) {
      if (kpm) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, true, 1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, false, 1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      return;
    }
    if (two_wave && con == -1) {   // (extended, no NBP constraint: hx_dev_solver.h, hx_nbp)
      if (kpm) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, true, -1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else if (hf) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, false, -1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, false, false, -1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      return;
    }
    if (two_wave && !con) {
      if (hf && kpm) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, true, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else if (kpm) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, false, true, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else if (hf) hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, true, false, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else hipLaunchKernelGGL((hx_run_kernel<HX_B1W2, false, false, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      return;
    }
  }
#ifdef HX_MINIMAL_BUILD  // experiment builds (tools/prof): the plain kernel only
  (void)hf; (void)kpm; (void)con;
#ifdef HX_MINIMAL_TRACK  // ... and the carbon-tracking ones
  if constexpr (B == 1) {
    if (con == 2 && !getenv("HECTOR_AMD_TRACK_INLINE")) {
      hipLaunchKernelGGL((hx_run_kernel<1, true, false, 3>), dim3(blocks), dim3(64 * (1 + trk_waves<1>())), lds, st, d_args, iy_from, iy_to);
      return;
    }
  }
  if (con == 2) {
    hipLaunchKernelGGL((hx_run_kernel<B, true, false, 2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
    return;
  }
#endif
#ifdef HX_MINIMAL_EXT    // ... and the extended ones without the NBP machinery / second history sum
  if constexpr (B >= 1 && B <= 4) {
    if (con == -2) {
      hipLaunchKernelGGL((hx_run_kernel<B, false, false, -2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      return;
    }
  }
  if (con == -1) {
    hipLaunchKernelGGL((hx_run_kernel<B, false, false, -1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
    return;
  }
#endif
  hipLaunchKernelGGL((hx_run_kernel<B, false, false, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  return;
#endif
#ifndef HX_W2_ONLY   // (HX_W2_ONLY, with HX_MINIMAL_BUILD: nothing but the plain one-biome kernels)
#ifndef HX_HOST_EMULATION   // (the host build runs a block's threads one after the other)
  if constexpr (B >= 1 && B <= HX_TRK_COMPANION_MAXB) {  // the maps live on companion wavefronts
    // (read at every launch, like hx_looped_from(): a test switches it between two cores; the
    // CU count is the launching core's device's, handed down by EnsembleCore::run)
    const bool inline_maps = getenv("HECTOR_AMD_TRACK_INLINE") != nullptr;
    if (con == 2 && !inline_maps && (B == 1 || blocks <= cus)) {
      constexpr int threads = 64 * (1 + trk_waves<B>());
      if (kpm) hipLaunchKernelGGL((hx_run_kernel<B, true, true, 3>), dim3(blocks), dim3(threads), lds, st, d_args, iy_from, iy_to);
      else hipLaunchKernelGGL((hx_run_kernel<B, true, false, 3>), dim3(blocks), dim3(threads), lds, st, d_args, iy_from, iy_to);
      return;
    }
  }
#endif
  if constexpr (B <= 4) {  // (HX_DYN = 0 included; hx_launch_run sends 5 and 6 biomes with tracking there)
  if (con == 2 && kpm) {
    hipLaunchKernelGGL((hx_run_kernel<B, true, true, 2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
    return;
  }
  if (con == 2) {
    hipLaunchKernelGGL((hx_run_kernel<B, true, false, 2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
    return;
  }
  }
  if constexpr (B >= 1 && B <= 4) {
    if (con == -2) {   // diagnostics only: no constraint, warming ratio or per-member series anywhere
      if (kpm) hipLaunchKernelGGL((hx_run_kernel<B, true, true, -2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else if (hf) hipLaunchKernelGGL((hx_run_kernel<B, true, false, -2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      else hipLaunchKernelGGL((hx_run_kernel<B, false, false, -2>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
      return;
    }
  }
  if (con == -1 && kpm)
    hipLaunchKernelGGL((hx_run_kernel<B, true, true, -1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (con == -1 && hf)
    hipLaunchKernelGGL((hx_run_kernel<B, true, false, -1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (con == -1)   // (shared diffusivity, no heat-flux output: without the second history sum)
    hipLaunchKernelGGL((hx_run_kernel<B, false, false, -1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (con && kpm)
    hipLaunchKernelGGL((hx_run_kernel<B, true, true, 1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (con)
    hipLaunchKernelGGL((hx_run_kernel<B, true, false, 1>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (hf && kpm)
    hipLaunchKernelGGL((hx_run_kernel<B, true, true, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (hf)
    hipLaunchKernelGGL((hx_run_kernel<B, true, false, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else if (kpm)
    hipLaunchKernelGGL((hx_run_kernel<B, false, true, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
  else
    hipLaunchKernelGGL((hx_run_kernel<B, false, false, 0>), dim3(blocks), dim3(64), lds, st, d_args, iy_from, iy_to);
#endif
}
#if HX_HAS_MFMA && ((!defined(HX_W2_ONLY) && !defined(HX_ONLY_B)) || defined(HX_WITH_PAIR))
#define HX_HAS_PAIR 1
#include "hx_dev_pair.h"
#else
#define HX_HAS_PAIR 0
#endif
// the small-ensemble kernel (two wavefronts per 64 members, hx_dev_pair.h): one biome, no
// constraints, the outputs listed in EnsembleCore::run; kpm: per-member DOECLIM kernel tables
int hx_pair_available() { return HX_HAS_PAIR; }
hipError_t hx_launch_run_pair(const HxArgs *d_args, int npad, bool heatflux, bool kpm, int iy_from,
                              int iy_to, hipStream_t st, bool cons, int nbiome) {
#if HX_HAS_PAIR
  const dim3 g(npad / 64), b(128);
  // (two to four biomes: shared diffusivity, no heat-flux sum -- plain or with scenario-wide constraints;
  // the host sends every other split ensemble to the run kernels)
  if (nbiome > 1) {
    if (heatflux || kpm || nbiome > 4) return hipErrorInvalidValue;
    if (cons) {
      if (nbiome == 2) hipLaunchKernelGGL((hx_pair_kernel<false, false, true, 2>), g, b, 0, st, d_args, iy_from, iy_to);
      else if (nbiome == 3) hipLaunchKernelGGL((hx_pair_kernel<false, false, true, 3>), g, b, 0, st, d_args, iy_from, iy_to);
      else hipLaunchKernelGGL((hx_pair_kernel<false, false, true, 4>), g, b, 0, st, d_args, iy_from, iy_to);
      return hipGetLastError();
    }
    if (nbiome == 2) hipLaunchKernelGGL((hx_pair_kernel<false, false, false, 2>), g, b, 0, st, d_args, iy_from, iy_to);
    else if (nbiome == 3) hipLaunchKernelGGL((hx_pair_kernel<false, false, false, 3>), g, b, 0, st, d_args, iy_from, iy_to);
    else hipLaunchKernelGGL((hx_pair_kernel<false, false, false, 4>), g, b, 0, st, d_args, iy_from, iy_to);
    return hipGetLastError();
  }
  // (with a CO2 / tas / RF_tot / CH4 constraint: the shared-diffusivity instantiations; the host
  // sends per-member diffusivity with constraints to the run kernel)
  if (cons && kpm) return hipErrorInvalidValue;
  if (cons && heatflux) hipLaunchKernelGGL((hx_pair_kernel<false, true, true>), g, b, 0, st, d_args, iy_from, iy_to);
  else if (cons) hipLaunchKernelGGL((hx_pair_kernel<false, false, true>), g, b, 0, st, d_args, iy_from, iy_to);
  else if (kpm && heatflux) hipLaunchKernelGGL((hx_pair_kernel<true, true>), g, b, 0, st, d_args, iy_from, iy_to);
  else if (kpm) hipLaunchKernelGGL((hx_pair_kernel<true, false>), g, b, 0, st, d_args, iy_from, iy_to);
  else if (heatflux) hipLaunchKernelGGL((hx_pair_kernel<false, true>), g, b, 0, st, d_args, iy_from, iy_to);
  else hipLaunchKernelGGL((hx_pair_kernel<false, false>), g, b, 0, st, d_args, iy_from, iy_to);
  return hipGetLastError();
#else
  (void)d_args; (void)npad; (void)heatflux; (void)kpm; (void)iy_from; (void)iy_to; (void)st; (void)cons; (void)nbiome;
  return hipErrorInvalidValue;
#endif
}
hipError_t hx_launch_run(int B, const HxArgs *d_args, int npad, bool heatflux, bool kpm, int con,
                         int iy_from, int iy_to, hipStream_t st, bool two_wave, int cus) {
#ifdef HX_W2_ONLY
  if (B != 1) return hipErrorInvalidValue;
  launch_run_b<1>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, 1, two_wave, cus);
  return hipGetLastError();
#else
  switch (B >= hx_looped_from() ? HX_BDYN + 1 : B) {
#if HX_ONLY_B_IS(1)
    case 1: launch_run_b<1>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, 1, two_wave, cus); break;
#endif
#ifndef HX_MINIMAL_BUILD
    case 2: launch_run_b<2>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, 2, false, cus); break;
    case 3: launch_run_b<3>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break;
#endif
#if HX_ONLY_B_IS(4)
    case 4: launch_run_b<4>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break;
#endif
    // five to eight biomes: unrolled like 1-4 (lean park, hx_dev_member.h); carbon tracking on the
    // looped kernels' 8-column chunks
#if HX_ONLY_B_IS(5)
    case 5: if (con != 2) { launch_run_b<5>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break; }
#if HX_ONLY_B_IS(0)
            launch_run_b<HX_DYN>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, B); break;
#else
            return hipErrorInvalidValue;
#endif
#endif
#if HX_ONLY_B_IS(6)
    case 6: if (con != 2) { launch_run_b<6>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break; }
#if HX_ONLY_B_IS(0)
            launch_run_b<HX_DYN>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, B); break;
#else
            return hipErrorInvalidValue;
#endif
#endif
#if HX_ONLY_B_IS(7)
    case 7: if (con != 2) { launch_run_b<7>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break; }
#if HX_ONLY_B_IS(0)
            launch_run_b<HX_DYN>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, B); break;
#else
            return hipErrorInvalidValue;
#endif
#endif
#if HX_ONLY_B_IS(8)
    case 8: if (con != 2) { launch_run_b<8>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st); break; }
#if HX_ONLY_B_IS(0)
            launch_run_b<HX_DYN>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, B); break;
#else
            return hipErrorInvalidValue;
#endif
#endif
    default:
#if HX_ONLY_B_IS(0)
      if (B < 1 || B > HX_BDYN) return hipErrorInvalidValue;
      launch_run_b<HX_DYN>(d_args, npad, heatflux, kpm, con, iy_from, iy_to, st, B);
#else
      return hipErrorInvalidValue;
#endif
  }
  return hipGetLastError();
#endif
}

int hx_doeclim_block_years() { return HX_DBLK; }
void hx_fill_chem_table_host(double *t) { hx_fill_chem_table(t); }
int hx_track_value_rows(int B) { return hx_trk_vrows(B); }
int hx_doeclim_kernel_pad() { return HX_KPAD; }
hipError_t hx_launch_unit_csys(int n, const double *Tc, const double *carbon, const double *alk,
                               double inv_vol, double *out, hipStream_t st) {
  hipLaunchKernelGGL(hx_unit_csys_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, Tc, carbon, alk,
                     inv_vol, out);
  return hipGetLastError();
}
// ---- the chip's clocks ahead of a one-shot run ---------------------------------------------------
// A run kernel launched after an idle gap -- the first run of a fresh core: the host has just spent
// 10-15 ms uploading and spinning up -- executes at ramping clocks: +6 % at 65 536 members, +11-15 %
// on the two-wavefront kernel, and ~12 ms of full-chip work right before it removes most of that
// (tools/prof/prewarm_curve.py, profiles/r06_prewarm_curve.txt).  This kernel is that work: one
// small wavefront on some of the SIMDs (never all: a kernel that needs a whole SIMD's registers
// must find one) multiplies and adds until the host raises `stop` (a memset on the core's stream,
// right ahead of the run kernel) or its own deadline on the constant 100 MHz clock passes --
// whichever comes first, so it cannot outlive its budget whatever the host does.
__global__ __launch_bounds__(64) void hx_prewarm_kernel(const int *stop, long long max_ticks, double *sink,
                                                         const double *mem, unsigned long long n_mem) {
#ifndef HX_HOST_EMULATION
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  double a = 1.0 + 1e-9 * threadIdx.x, b = 0.999999, c = 1e-7, d = a, e = a + c, f = a - c, acc = 0.0;
  unsigned long long pos = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * 64;
  for (;;) {
#pragma unroll
    for (int k = 0; k < 32; ++k) { a = fma(a, b, c); d = fma(d, b, c); e = fma(e, b, c); f = fma(f, b, c); }
    if (n_mem) {   // (... and the memory system: a sweep over the arrays the run will write)
#pragma unroll
      for (int k = 0; k < 4; ++k) { acc += __builtin_nontemporal_load(mem + pos % n_mem); pos += stride; }
    }
    if (__builtin_nontemporal_load(stop) != 0) break;
    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) break;
  }
  if (((a + d) + (e + f)) + acc == 12345.678) sink[blockIdx.x & 1023] = a;   // (keeps the arithmetic)
#endif
}
hipError_t hx_launch_prewarm(const int *d_stop, long long max_ticks, double *d_sink, int waves,
                             const double *mem, unsigned long long n_mem, hipStream_t st) {
  hipLaunchKernelGGL(hx_prewarm_kernel, dim3(waves), dim3(64), 0, st, d_stop, max_ticks, d_sink, mem, n_mem);
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
hipError_t hx_launch_gas(const double *par, const double *ser, const double *h0, int nh, int ns,
                         int npad, double *n2o_out, double *rf_other_out, hipStream_t st) {
  if (nh > 40) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hx_gas_kernel, dim3((npad + 63) / 64), dim3(64), 0, st, par, ser, h0, nh, ns,
                     npad, n2o_out, rf_other_out);
  return hipGetLastError();
}
hipError_t hx_launch_or_flags(unsigned *status, const double *flag_row, int npad, hipStream_t st) {
  hipLaunchKernelGGL(hx_or_flags_kernel, dim3((npad + 255) / 256), dim3(256), 0, st, status,
                     flag_row, npad);
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
hipError_t hx_launch_combine_stats(const double *slots, int world, int rows, double *out,
                                   hipStream_t st) {
  hipLaunchKernelGGL(hx_combine_stats_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, slots,
                     world, rows, out);
  return hipGetLastError();
}
hipError_t hx_launch_stats(const double *var, int n, int npad, int iy0, int nyears,
                           double *stats, hipStream_t st) {
  hipLaunchKernelGGL(hx_stats_kernel, dim3(nyears), dim3(256), 0, st, var, n, npad, iy0, stats);
  return hipGetLastError();
}
}
