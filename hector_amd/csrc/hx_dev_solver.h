// hx_dev_solver.h -- CarbonCycleSolver::run for one lane: interval constants, RHS, stash, dopri5 + controller + retry logic
// Part of the device code of hx_kernels.hip (one translation unit; see its header for the
// reference file:line map).
#pragma once

#define HX_MAX_STEPS_PER_YEAR 20000

namespace {

// rhs constants that only change at a stash (pools frozen in between,
// src/simpleNbox-runtime.cpp:809-840)
struct Interval {
  double Pn;              // (P - npp) + rh: everything in d(atmos)/dt but the air-sea flux
  double v1, d2, s3, k4, k5, k7;
  double totC;
  // air-sea flux of the interval as a function of the solver's c[] (calc_annual_surface_flux for
  // both boxes, ocean_component.cpp:606-616, ocean_csys.cpp:375-396):
  //   ao = (co2 - pH scale) gH + (co2 - pL scale) gL,  co2 = c0 / 2.13,
  //   scale = (surf + (c4 - totC)) / surf = 1 + (c4 - totC) / surf
  //      = c0 aoA - (pG + (c4 - totC) aoB),  aoA = (gH + gL) / 2.13,  pG = pH gH + pL gL,
  //        aoB = pG / surf
  // three operations per evaluation instead of seven; c4 - totC stays an exact small difference
  double aoA, aoB, pG;
  // d(veg + detritus + soil)/dt: every land-use loss r * y_i adds up to luc_e, so the three
  // pools' sum moves at this constant rate within the interval
  double dtot;
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
// :707-711: rh_ftpa_co2 / (1 - rh_ch4_frac) * rh_ch4_frac -- the same product as rh_ftpa_co2 with
// rh_ch4_frac in the place of (1 - rh_ch4_frac), so no division (an ulp of a flux that is zero
// until permafrost thaws; the reference divides what it has just multiplied)
template <int B> __device__ __forceinline__ double m_rh_tp_ch4(const Member<B> &m, const LandK<B> &k, int b) {
  return ((m.thawed[b] * (1 - k.fpf_static[b])) * 0.02) * m.tempferts[b] * k.rh_ch4_frac[b];
}

// CON = -1: the extended kernel for a scenario WITHOUT an NBP constraint (EnsembleCore::run picks
// it from con_mask and the per-member series): diagnostics, the other constraints and the warming
// ratio like CON = 1, but the solver as in the plain kernel -- five variables (the thawed pool's
// derivative is constant within an interval), one set of interval constants, the land-use rates of
// an attempt's stages side by side.  hx_nbp<CON>(): the instantiations that carry the NBP machinery.
template <int CON> constexpr bool hx_nbp() { return CON >= 1; }
// CON = -2: the plain kernel plus the DIAGNOSTICS of the extended one (NPP / RH / ocean-uptake /
// box outputs ...) and nothing else: no constraint, no land-ocean warming ratio, no per-member
// series -- what a run that merely records such an output needs.  EnsembleCore::run picks it when
// the scenario and the members hold none of those.  hx_cons<CON>(): the instantiations that carry
// constraints, the warming ratio and per-member series.
#ifdef HX_EXT_NOCONS   // (experiment builds: CON = -1 without them, to time what their code costs)
template <int CON> constexpr bool hx_cons() { return CON != 0 && CON != -2 && CON != -1; }
#else
template <int CON> constexpr bool hx_cons() { return CON != 0 && CON != -2; }
#endif
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
  K.Pn = ((((m.ffi - m.daccs) + m.luc_e) - m.luc_u) - F.npp) + F.rh;
  K.v1 = F.fav - F.litter;
  K.d2 = ((F.fad + F.lfvd) - F.detsoil) - F.fda;
  K.s3 = ((F.fas + F.lfvs) + F.detsoil) - F.fsa;
  K.k4 = -F.thaw + F.refr;
  K.k5 = ((F.thaw - F.refr) - F.tpm) - F.tpc;
  K.k7 = -m.ffi + m.daccs;
  K.totC = m.cDO + m.cIO + m.cLL + m.cHL;  // ocean_component.cpp:325-328
  K.pG = m.pco2H * m.kH.g + m.pco2L * m.kL.g;
  K.aoA = PGC2PPM * (m.kH.g + m.kL.g);
  K.aoB = K.pG * hx_recip(m.cLL + m.cHL);
  K.dtot = (((K.v1 + m.luc_u) + K.d2) + K.s3) - m.luc_e;
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
#pragma unroll hx_ur<B>()
  for (int b = 0; b < nbio<B>(m); ++b) {
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

// ---- looped kernels (9 to HX_BDYN biomes): the per-biome loops in chunks ------------------------
// Their per-biome values live in LDS (four pools), in HBM rows (the thawed pool, tempferts,
// co2fert, tempfertd, f_new_thaw) and in the parameter table.  Written as one loop per sum, every
// biome of every loop waited out its own scalar and HBM loads -- a stash's land half 1.6 k
// cycles PER BIOME, the interval constants 1.1 k (section clock, profiles/r04_phase_clock_8192x9
// .json): a store to one per-biome array may alias the next biome's loads as far as the compiler
// can tell.  Here a chunk's values are all requested first (nothing is stored in between), then
// the chunk is worked through in biome order -- the same operations and the same order of every
// sum as the unrolled kernels' loops.
#ifndef HX_DYN_CHUNK
#define HX_DYN_CHUNK 4
#endif
struct BioIn {
  double veg, det, soil, pf, thw, tfs, tfd, co2f, fnt, npp0, f_nppv, f_nppd, f_litterd, fpf, rch4;
};
// FLOWS: also what only the flows of an interval need (vegetation, f_new_thaw, the allocation
// fractions); POOLS: the pools themselves (false: the caller has a chunk's NEW pools in hand)
template <bool FLOWS, bool POOLS = true>
__device__ __forceinline__ void load_bio(const Member<HX_DYN> &m, const LandK<HX_DYN> &lk, int b, BioIn &v) {
  if constexpr (POOLS) {
    v.det = m.det[b]; v.soil = m.soil[b]; v.pf = m.pf[b]; v.thw = m.thawed[b];
    if constexpr (FLOWS) v.veg = m.veg[b];
  }
  v.tfs = m.tempferts[b]; v.tfd = m.tempfertd[b]; v.co2f = m.co2fert[b];
  v.npp0 = lk.npp0[b]; v.fpf = lk.fpf_static[b]; v.rch4 = lk.rh_ch4_frac[b];
  if constexpr (FLOWS) {
    v.fnt = m.f_new_thaw[b];
    v.f_nppv = lk.f_nppv[b]; v.f_nppd = lk.f_nppd[b]; v.f_litterd = lk.f_litterd[b];
  }
}
// A per-biome loop in chunks: load(b0, v) requests chunk b0's values into v, work(b0, v) uses them
// (and may store).  (One chunk's loads AHEAD of the chunk being worked on, two register sets in
// turn, was measured and lost: 240 more live registers, 512 + scratch; 65 536 members x 9 / 12 /
// 16 biomes 30.8 / 33.6 / 42.5 -> 34.2 / 36.6 / 45.3 ms.  Chunks of two that way: 29.2 / 35.3 /
// 44.2.  Chunks of eight: 36.2 / 39.0 / 42.8.  profiles/r04_looped_kernel_variants.txt)
template <class T, class L, class W>
__device__ __forceinline__ void chunk_loop(int nb, L load, W work) {
  for (int b0 = 0; b0 < nb; b0 += HX_DYN_CHUNK) {
    T v[HX_DYN_CHUNK];
    load(b0, v);
    work(b0, v);
  }
}
// Round 5: the TABLE part of a chunk's values one chunk ahead.  A chunk's values come from three
// places -- the LDS park (the pools: ~64 clocks), scalar loads (the biome constants every member
// shares) and HBM rows (the thawed pool, tempferts, tempfertd, co2fert, f_new_thaw; the biome
// constants when they differ between members) -- and a chunk's work ends in stores to such rows,
// behind which the next chunk's loads queue (one counter, in order): the section clock put a
// chunk of any of these loops at ~4.5 k ticks whatever it computed, two HBM round trips.  Only the
// HBM part (5 to 11 doubles a biome, not the 15 of the attempt recorded above) is requested a
// chunk ahead -- before the current chunk's stores -- and copied over when its turn comes; a
// chunk's biomes are never the ones the previous chunk stores to (the tail's clamp stays inside
// the last chunk).  -DHX_DYN_NO_AHEAD: the old form.
struct BioTab {
  double thw, tfs, tfd, co2f, fnt, k[6];
};
template <bool FLOWS>
__device__ __forceinline__ void load_bio_tab(const Member<HX_DYN> &m, const LandK<HX_DYN> &lk, int b, BioTab &h) {
  h.thw = m.thawed[b]; h.tfs = m.tempferts[b]; h.tfd = m.tempfertd[b]; h.co2f = m.co2fert[b];
  if constexpr (FLOWS) h.fnt = m.f_new_thaw[b];
  // (a zero the compiler takes for a vector value: left undefined in the uniform case, the slots --
  // carried from one chunk to the next -- were assigned scalar registers, "illegal VGPR to SGPR copy")
  double z = 0.0;
#ifndef HX_HOST_EMULATION
  asm("" : "+v"(z));
#endif
  if (!m.upar) {   // (the biome constants differ between members: rows of the parameter table)
    // (the member's rows themselves, not ParamCol's choice between them and the uniform table)
    const ParamCol &c = lk.npp0;
    const int r = HXP_NGLOBAL + b * HXPB_N;
    h.k[0] = w2_ld(c.par, c.npad, r + HXPB_NPP0, c.moff);
    h.k[1] = w2_ld(c.par, c.npad, r + HXPB_FPF_STATIC, c.moff);
    h.k[2] = w2_ld(c.par, c.npad, r + HXPB_RH_CH4_FRAC, c.moff);
    if constexpr (FLOWS) {
      h.k[3] = w2_ld(c.par, c.npad, r + HXPB_F_NPPV, c.moff);
      h.k[4] = w2_ld(c.par, c.npad, r + HXPB_F_NPPD, c.moff);
      h.k[5] = w2_ld(c.par, c.npad, r + HXPB_F_LITTERD, c.moff);
    }
  } else {
    h.k[0] = h.k[1] = h.k[2] = z;
    if constexpr (FLOWS) h.k[3] = h.k[4] = h.k[5] = z;
  }
  if constexpr (!FLOWS) { h.fnt = z; h.k[3] = h.k[4] = h.k[5] = z; }
}
template <bool FLOWS>
__device__ __forceinline__ void load_bio_rest(const Member<HX_DYN> &m, const LandK<HX_DYN> &lk, int b,
                                              const BioTab &h, BioIn &v) {
  v.det = m.det[b]; v.soil = m.soil[b]; v.pf = m.pf[b]; v.thw = h.thw;
  if constexpr (FLOWS) v.veg = m.veg[b];
  v.tfs = h.tfs; v.tfd = h.tfd; v.co2f = h.co2f;
  if constexpr (FLOWS) v.fnt = h.fnt;
  if (m.upar) {   // (scalar loads from the uniform table)
    hx_ccd u = m.upar + (HXP_NGLOBAL + b * HXPB_N);
    v.npp0 = u[HXPB_NPP0]; v.fpf = u[HXPB_FPF_STATIC]; v.rch4 = u[HXPB_RH_CH4_FRAC];
    if constexpr (FLOWS) { v.f_nppv = u[HXPB_F_NPPV]; v.f_nppd = u[HXPB_F_NPPD]; v.f_litterd = u[HXPB_F_LITTERD]; }
  } else {
    v.npp0 = h.k[0]; v.fpf = h.k[1]; v.rch4 = h.k[2];
    if constexpr (FLOWS) { v.f_nppv = h.k[3]; v.f_nppd = h.k[4]; v.f_litterd = h.k[5]; }
  }
}
template <bool FLOWS, class W>
__device__ __forceinline__ void chunk_loop_bio(const Member<HX_DYN> &m, const LandK<HX_DYN> &lk, int nb, W work) {
#ifdef HX_DYN_NO_AHEAD
  chunk_loop<BioIn>(
      nb,
      [&](int b0, BioIn (&v)[HX_DYN_CHUNK]) {
#pragma unroll
        for (int j = 0; j < HX_DYN_CHUNK; ++j) load_bio<FLOWS>(m, lk, min(b0 + j, nb - 1), v[j]);
      },
      work);
#else
  BioTab nxt[HX_DYN_CHUNK];
#pragma unroll
  for (int j = 0; j < HX_DYN_CHUNK; ++j) load_bio_tab<FLOWS>(m, lk, min(j, nb - 1), nxt[j]);
  for (int b0 = 0; b0 < nb; b0 += HX_DYN_CHUNK) {
    BioTab cur[HX_DYN_CHUNK];
#pragma unroll
    for (int j = 0; j < HX_DYN_CHUNK; ++j) cur[j] = nxt[j];
    if (b0 + HX_DYN_CHUNK < nb) {
#pragma unroll
      for (int j = 0; j < HX_DYN_CHUNK; ++j)
        load_bio_tab<FLOWS>(m, lk, min(b0 + HX_DYN_CHUNK + j, nb - 1), nxt[j]);
    }
    BioIn v[HX_DYN_CHUNK];
#pragma unroll
    for (int j = 0; j < HX_DYN_CHUNK; ++j) load_bio_rest<FLOWS>(m, lk, min(b0 + j, nb - 1), cur[j], v[j]);
    work(b0, v);
  }
#endif
}
// the fluxes of one biome from its values (m_npp ... m_rh_tp_ch4 above, operation by operation)
__device__ __forceinline__ double bio_npp(const BioIn &v, double adj) { return (v.npp0 * v.co2f) * adj; }
__device__ __forceinline__ double bio_fda(const BioIn &v) { return (v.det * 0.25) * v.tfd; }
__device__ __forceinline__ double bio_fsa(const BioIn &v) { return (v.soil * 0.02) * v.tfs; }
__device__ __forceinline__ double bio_tpc(const BioIn &v) {
  return ((v.thw * (1 - v.fpf)) * 0.02) * v.tfs * (1.0 - v.rch4);
}
__device__ __forceinline__ double bio_tpm(const BioIn &v) {
  return ((v.thw * (1 - v.fpf)) * 0.02) * v.tfs * v.rch4;
}
// one biome's terms of the land flows (the body of compute_flows' loop)
template <bool SPIN>
__device__ __forceinline__ void flows_add(const BioIn &v, double adj, Flows &F) {
  const double n = bio_npp(v, adj);
  F.npp += n;
  F.fav += n * v.f_nppv;
  F.fad += n * v.f_nppd;
  F.fas += n * (1 - v.f_nppv - v.f_nppd);
  F.fda += bio_fda(v);
  F.fsa += bio_fsa(v);
  const double co2 = bio_tpc(v), ch4 = bio_tpm(v);
  F.tpc += co2;
  F.tpm += ch4;
  const double vv = v.veg * 0.035;
  F.litter += vv;
  F.lfvd += vv * v.f_litterd;
  F.lfvs += vv * (1 - v.f_litterd);
  F.detsoil += v.det * 0.6;
  if (!SPIN) {
    double c_thaw = v.pf * v.fnt;
    double r_tp = 0.0;
    if (c_thaw < 0) {
      const double want = -c_thaw;
      c_thaw = 0.0;
      r_tp = fmin(want, v.thw - co2 - ch4);
    }
    F.thaw += c_thaw;
    F.refr += r_tp;
  }
}
__device__ __forceinline__ void flows_zero(Flows &F) {
  F.npp = F.rh = F.fav = F.fad = F.fas = F.fda = F.fsa = F.tpc = F.tpm = 0;
  F.litter = F.lfvd = F.lfvs = F.detsoil = F.thaw = F.refr = 0;
}
template <bool SPIN>
__device__ __forceinline__ void compute_flows_chunked(const Member<HX_DYN> &m, const LandK<HX_DYN> &lk,
                                                      Flows &F) {
  flows_zero(F);
  const int nb = m.nb;
  chunk_loop_bio<true>(
      m, lk, nb,
      [&](int b0, const BioIn (&v)[HX_DYN_CHUNK]) {
#pragma unroll
        for (int j = 0; j < HX_DYN_CHUNK; ++j)
          if (b0 + j < nb) flows_add<SPIN>(v[j], m.npp_luc_adjust, F);
      });
  F.rh = F.fda + F.fsa + F.tpc;
}

// the interval's constants from its land flows
template <int B, bool SPIN, int CON = 0>
__device__ __forceinline__ void finish_interval(const Member<B> &m, const Flows &F, Interval &K,
                                                Interval &K2, const YearCon &yc) {
  if constexpr (hx_nbp<CON>() && !SPIN) {
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

// K: the interval's constants; K2 (CON kernels): the same for the second half of the year,
// where round(t) picks the next date's NBP constraint
template <int B, bool SPIN, int CON = 0>
__device__ __forceinline__ void prep_interval(const Member<B> &m, const LandK<B> &lk,
                                              Interval &K, Interval &K2, const YearCon &yc) {
  Flows F;
  if constexpr (B == HX_DYN) {
    compute_flows_chunked<SPIN>(m, lk, F);
    finish_interval<B, SPIN, CON>(m, F, K, K2, yc);
    return;
  } else
  compute_flows<B, SPIN>(m, lk, F);
  if constexpr (hx_nbp<CON>() && !SPIN) {
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
// pools whose derivative depends on c[] (atmos, veg, det, soil, ocean).
// r: the land-use loss rate luc_e / (veg + det + soil) at this stage, if the caller knows it
// (plain kernels: the sum of the three pools is linear in time within an interval and
// Runge-Kutta stages preserve linear invariants, so the rates of all stages of an attempt are
// computed side by side before it instead of one division at the head of every stage's
// dependency chain -- see solve_year); NaN = divide here.
template <int B, bool SPIN, int CON = 0, bool RATE = false>
__device__ __forceinline__ void rhs(const Member<B> &m, const Interval &K1, const Interval &K2,
                                    const YearCon &yc, double t, const double *y, double *d,
                                    double rate = 0.0) {
  // CON kernels carry the thawed-permafrost pool as a sixth solver variable: with an NBP
  // constraint its derivative changes where round(t) does, so it is no longer constant
  const Interval &K = (hx_nbp<CON>() && !SPIN && t >= yc.t_half) ? K2 : K1;
  if constexpr (hx_nbp<CON>()) d[5] = K.k5;
  double r;
  if constexpr (RATE) {
    r = rate;
  } else {
    const double total = y[1] + y[2] + y[3];
    r = hx_div1(m.luc_e, total);  // 2e-15 on a term that is itself ~1e-3 of the flux
  }
  double ao;
  if (SPIN) {
    ao = 0.0;  // preindustrial fluxes +1 / -1 PgC/yr  ocean_component.cpp:343-345
  } else {
    ao = fma(y[0], K.aoA, -fma(y[4] - K.totC, K.aoB, K.pG));
  }
  d[0] = K.Pn - ao;
  d[1] = (K.v1 - r * y[1]) + m.luc_u;
  d[2] = K.d2 - r * y[2];
  d[3] = K.s3 - r * y[3];
  d[4] = ao;
}

// The land half of rhs() alone: vegetation, detritus, soil (the same operations in the same order).
template <int B>
__device__ __forceinline__ void rhs_land(const Member<B> &m, const Interval &K, const double *y, double *d,
                                         double r) {
  d[1] = (K.v1 - r * y[1]) + m.luc_u;
  d[2] = K.d2 - r * y[2];
  d[3] = K.s3 - r * y[3];
}

// ---- the atmosphere-ocean pair of a dopri5 attempt as ONE chain -----------------------------------
// Within a stash interval (constants K) the two pools enter the right-hand side only through the
// air-sea flux  z = ao(c0, c4) = c0 aoA - ((c4 - totC) aoB + pG),  d c0/dt = Pn - z,  d c4/dt = z,
// and z obeys its own scalar linear equation  dz/dt = aoA (Pn - z) - aoB z = alp - lam z.  A
// Runge-Kutta stage is an affine combination, and affine maps commute with it: the stage values of
// z computed by the dopri5 recursion on z alone ARE ao(stage c0, stage c4) of the recursion on the
// pair (to rounding).  So the attempt carries z through its stages (one variable instead of two),
// and the pair follows from the stage fluxes at the end:
//   c4' = c4 + Z,  c0' = (c0 + h Pn) - Z,   Z = h sum_l c_l z_l      (sum_l c_l = 1)
//   error estimates  xe_0 = -E, xe_4 = +E,  E = h sum_l dc_l z_l      (sum_l dc_l = 0),
// whose two quotients share the numerator: max(|E| / d_0, |E| / d_4) = |E| / min(d_0, d_4).
// 39 operations and one division where the pair took 80 and two; same scheme, same decisions,
// results within rounding of the pairwise form (-DHX_NO_ZCHAIN: the pairwise form, experiments).
#ifndef HX_NO_ZCHAIN
template <int CON, bool SPIN> constexpr bool hx_zchain() { return !hx_nbp<CON>() && !SPIN; }
#else
template <int CON, bool SPIN> constexpr bool hx_zchain() { return false; }
#endif

}  // namespace
#include "hx_dev_track.h"
namespace {

// OceanComponent::stashCValues + SimpleNbox::stashCValues for one lane
template <int B, bool SPIN, int CON = 0>
__device__ __forceinline__ void stash(Member<B> &m, double t, const double *y,
                                      double c4, double c5, double c7, Interval &K,
                                      Interval &K2, const YearCon &yc, bool more) {
  LandK<B> lk;
  load_landk<B>(m, lk);
  double kHD, kLH, kLI, kIL, kIH, kID, kDI;
  bool k_uniform = false;
  if constexpr (hx_w2<B>()) k_uniform = m.bufp->uni_k != 0;
  if (k_uniform) {   // (two-wavefront flavour: scalar loads when every member shares them)
    hx_ccd u = HX_CCD(m.bufp->uderived);
    kHD = u[HXD_KHD]; kLH = u[HXD_KLH]; kLI = u[HXD_KLI]; kIL = u[HXD_KIL]; kIH = u[HXD_KIH];
    kID = u[HXD_KID]; kDI = u[HXD_KDI];
  } else {
    kHD = dconst<B>(m, HXD_KHD); kLH = dconst<B>(m, HXD_KLH); kLI = dconst<B>(m, HXD_KLI);
    kIL = dconst<B>(m, HXD_KIL); kIH = dconst<B>(m, HXD_KIH); kID = dconst<B>(m, HXD_KID);
    kDI = dconst<B>(m, HXD_KDI);
  }
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
      HX_CHEM_SOLVE2(m.kH, m.kL, m.cHL, m.cLL, m.alkH, m.alkL, m.hH, m.hL, m.pco2H, m.pco2L,
                  m.status);
    m.chem_fresh = false;
    aH = ((co2 - m.pco2H) * m.kH.g) * yf;
    aL = ((co2 - m.pco2L) * m.kL.g) * yf;
  }
  HX_STAMP(m, 7);   // stash: constants from the park + carbonate solve
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
  // which of the stash's diagnostics are recorded: bits of HxBuffers::out_mask0 (one scalar load
  // a stash instead of a pointer load + branch per diagnostic)
  [[maybe_unused]] unsigned long long omk = 0;
  constexpr unsigned long long OM_STASH =
      (1ull << HXO_NPP) | (1ull << HXO_RH) | (1ull << HXO_RH_DET) | (1ull << HXO_RH_SOIL) |
      (1ull << HXO_HL_UPTAKE) | (1ull << HXO_LL_UPTAKE) | (1ull << HXO_HL_DO) |
      (1ull << HXO_CA_RESIDUAL) | (1ull << HX_OM_BIOME_FLUX);
  if constexpr (CON && !SPIN) {
    omk = m.omk;
#ifndef HX_HOST_EMULATION
    asm volatile("" : "+s"(omk));   // (the tests below stay here: see HX_MASKS_LOCAL in hx_run_kernel)
#endif
    omk &= OM_STASH;
    diag = omk != 0;
    if (HX_RARE(diag)) {  // annualflux_sumHL/LL, annual_box_fluxes[HL->DO]: sums over the year's stashes
      const HxBuffers &buf = *m.bufp;
      dgo = (size_t)m.iy * buf.npad + (blockIdx.x * 64 + m.lane);
      if (HX_RARE(omk & (1ull << HXO_HL_UPTAKE))) HX_GD(buf.out[HXO_HL_UPTAKE])[dgo] += aH;
      if (HX_RARE(omk & (1ull << HXO_LL_UPTAKE))) HX_GD(buf.out[HXO_LL_UPTAKE])[dgo] += aL;
      if (HX_RARE(omk & (1ull << HXO_HL_DO))) HX_GD(buf.out[HXO_HL_DO])[dgo] += lHD;
    }
  }
  [[maybe_unused]] TrkStashIn tk;
  if constexpr (CON >= 2 && !SPIN) {
    tk.yf = yf;
    tk.pre[0] = m.cHL; tk.pre[1] = m.cLL; tk.pre[2] = m.cIO; tk.pre[3] = m.cDO;
    tk.closs[0] = lHD; tk.closs[1] = lLH; tk.closs[2] = lLI; tk.closs[3] = lIL;
    tk.closs[4] = lIH; tk.closs[5] = lID; tk.closs[6] = lDI;
    tk.aH = aH; tk.aL = aL;
  }
  const double lastflux = aL + aH;
  m.annualflux_sum += lastflux;
  m.lastflux_ann = lastflux * inv_yf;
  // update_state: carbon + additions + ao - oa - subtractions (oceanbox.cpp:297-303)
  m.cHL = ((m.cHL + (lLH + lIH)) + aH) - lHD;
  m.cLL = ((m.cLL + lIL) + aL) - (lLH + lLI);
  m.cIO = (m.cIO + (lLI + lDI)) - ((lIL + lIH) + lID);
  m.cDO = (m.cDO + (lHD + lID)) - lDI;
  if constexpr (CON >= 2 && !SPIN) {
    tk.post[0] = m.cHL; tk.post[1] = m.cLL; tk.post[2] = m.cIO; tk.post[3] = m.cDO;
  }

  // ---- land: simpleNbox-runtime.cpp:270-609 --------------------------------
  double npp_t = 0, rh_t = 0, pf_t = 0;
  const int NB = nbio<B>(m);
  [[maybe_unused]] double sp_rd = 0, sp_rs = 0, sp_rc = 0;  // spinup record: final_rh_detritus / _soil, thawed part
  if constexpr (B == HX_DYN) {   // (looped kernels: a chunk's values requested together, see load_bio)
    chunk_loop_bio<false>(
        m, lk, NB,
        [&](int b0, const BioIn (&v)[HX_DYN_CHUNK]) {
#pragma unroll
          for (int j = 0; j < HX_DYN_CHUNK; ++j) {
            if (b0 + j < NB) {
              npp_t += bio_npp(v[j], m.npp_luc_adjust);
              rh_t += (bio_fda(v[j]) + bio_fsa(v[j])) + bio_tpc(v[j]);
              pf_t += v[j].pf;
              if constexpr (SPIN) { sp_rd += bio_fda(v[j]); sp_rs += bio_fsa(v[j]); sp_rc += bio_tpc(v[j]); }
            }
          }
        });
  } else {
#pragma unroll hx_ur<B>()
  for (int b = 0; b < NB; ++b) npp_t += m_npp(m, lk, b);
#pragma unroll hx_ur<B>()
  for (int b = 0; b < NB; ++b)
    rh_t += (m_rh_fda(m, b) + m_rh_fsa(m, b)) + m_rh_tp_co2(m, lk, b);
#pragma unroll hx_ur<B>()
  for (int b = 0; b < NB; ++b) pf_t += m.pf[b];
  if constexpr (SPIN) {
    if (m.spin_row) {
#pragma unroll hx_ur<B>()
      for (int b = 0; b < NB; ++b) { sp_rd += m_rh_fda(m, b); sp_rs += m_rh_fsa(m, b); sp_rc += m_rh_tp_co2(m, lk, b); }
    }
  }
  }
  double alf = ((npp_t - rh_t) - m.luc_e) + m.luc_u;
  const double npp_rh = npp_t + rh_t;
  double tpf = c5;
  if (fabs(tpf) < 1e-10) tpf = 0.0;  // :337-341
  if (y[0] < 0 || y[1] < 0 || y[2] < 0 || y[3] < 0 || c4 < 0 || tpf < 0)
    m.status |= HX_ERR_NEGPOOL;
  double nveg = y[1], ndet = y[2], nsoil = y[3];
  double rh_adj = 1.0;
  double npp_fin_total = npp_t;  // npp_total after any NBP constraint (final_npp weights it)
  if constexpr (hx_nbp<CON>() && !SPIN) {
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
  // (hx_div_cr: the biome weights as correctly rounded quotients -- an equal split into 2 or 4
  // identical biomes then gets weights of exactly 1/2, 1/4 and reproduces the single-biome run
  // bit for bit, the reference's own property (test_biome.R))
  const double inv_nr = hx_recip(npp_rh);
  const double inv_pf = (pf_t > 0) ? hx_recip(pf_t) : 0.0;
  if constexpr (CON >= 2 && !SPIN) {
    // the origin maps, from the pools and fluxes as they are BEFORE the new pools are written
    if (m.bufp->track_out_f && m.trk_iy >= 0 && m.iy >= m.trk_iy) {
      tk.npp_total = npp_fin_total; tk.rh_adj = rh_adj; tk.total = total;
      tk.npp_rh = npp_rh; tk.inv_nr = inv_nr;
      if constexpr (CON == 3) track_post_stash<B>(m, lk, tk);  // (one biome: the companion wavefront mixes)
      else track_stash<B>(m, lk, tk);
    }
  }
  [[maybe_unused]] Flows Fn;   // looped kernels: the next interval's land flows, from the new pools
  if constexpr (B == HX_DYN) {
    flows_zero(Fn);
    chunk_loop_bio<true>(
        m, lk, NB,
        [&](int b0, const BioIn (&v)[HX_DYN_CHUNK]) {
#pragma unroll
      for (int j = 0; j < HX_DYN_CHUNK; ++j) {
        if (b0 + j < NB) {
          const int b = b0 + j;
          const double fda = bio_fda(v[j]), fsa = bio_fsa(v[j]), tpc = bio_tpc(v[j]), tpm = bio_tpm(v[j]);
          const double wt = hx_div_cr(bio_npp(v[j], m.npp_luc_adjust) + ((fda + fsa) + tpc), npp_rh, inv_nr);
          const double wt_pf = hx_div_cr(v[j].pf, pf_t, inv_pf);
          if (HX_RARE(diag)) {  // final_npp / final_rh / final_rh_detritus / final_rh_soil :420-440
            const double a = fda * rh_adj, bb = fsa * rh_adj, cc = tpc * rh_adj, dd = tpm * rh_adj;
            fin_npp += npp_fin_total * wt;
            fin_rh += ((a + bb) + cc) + dd;
            if (HX_RARE(omk & (1ull << HX_OM_BIOME_FLUX))) {
            const HxBuffers &buf = *m.bufp;
            if (buf.out[HXO_B(HXOB_NPP, b)]) HX_GD(buf.out[HXO_B(HXOB_NPP, b)])[dgo] = npp_fin_total * wt;
            if (buf.out[HXO_B(HXOB_RH, b)]) HX_GD(buf.out[HXO_B(HXOB_RH, b)])[dgo] = ((a + bb) + cc) + dd;
            }
            fin_det += a;
            fin_soil += bb;
          }
          if constexpr (CON) m.cum_pf_ch4 += (tpm * rh_adj) * yf;
          else m.cum_pf_ch4 += tpm * yf;  // :481
          BioIn w = v[j];   // the biome after the stash
          w.veg = nveg * wt; w.det = ndet * wt; w.soil = nsoil * wt; w.pf = c4 * wt_pf; w.thw = tpf * wt_pf;
          m.veg[b] = w.veg; m.det[b] = w.det; m.soil[b] = w.soil; m.pf[b] = w.pf; m.thawed[b] = w.thw;
          if (more) flows_add<SPIN>(w, m.npp_luc_adjust, Fn);
        }
      }
        });
    Fn.rh = Fn.fda + Fn.fsa + Fn.tpc;
  } else
#pragma unroll hx_ur<B>()
  for (int b = 0; b < NB; ++b) {
    const double wt = hx_one<B>() ? 1.0
        : hx_div_cr(m_npp(m, lk, b) + ((m_rh_fda(m, b) + m_rh_fsa(m, b)) + m_rh_tp_co2(m, lk, b)),
                    npp_rh, inv_nr);
    const double wt_pf = hx_one<B>() ? ((pf_t > 0) ? 1.0 : 0.0) : hx_div_cr(m.pf[b], pf_t, inv_pf);
    if (HX_RARE(diag)) {  // final_npp / final_rh / final_rh_detritus / final_rh_soil :420-440
      const double a = m_rh_fda(m, b) * rh_adj, bb = m_rh_fsa(m, b) * rh_adj;
      const double cc = m_rh_tp_co2(m, lk, b) * rh_adj, dd = m_rh_tp_ch4(m, lk, b) * rh_adj;
      fin_npp += npp_fin_total * wt;
      fin_rh += ((a + bb) + cc) + dd;
      if (HX_RARE(omk & (1ull << HX_OM_BIOME_FLUX))) {
      const HxBuffers &buf = *m.bufp;  // "<biome>.NPP", "<biome>.RH"
      if (buf.out[HXO_B(HXOB_NPP, b)]) HX_GD(buf.out[HXO_B(HXOB_NPP, b)])[dgo] = npp_fin_total * wt;
      if (buf.out[HXO_B(HXOB_RH, b)]) HX_GD(buf.out[HXO_B(HXOB_RH, b)])[dgo] = ((a + bb) + cc) + dd;
      }
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
  // (written so that a NaN state raises the flag too: the reference would carry it on silently)
  if (m.masstot > 0.0 && !(fabs(sum - m.masstot) <= 0.001)) m.status |= HX_ERR_MASS;
  m.masstot = sum;
  double ca_residual = 0.0;
  if (SPIN) {  // pin the atmosphere to C0, residual to the deep box :567-603
    const double match = m.C0 / PGC2PPM;
    const double residual = m.atmos - match;
    m.cDO = residual + m.cDO;
    m.atmos = m.atmos - residual;
    if (m.spin_row) {  // the output stream's view of this spinup step (hx_enable_spinup_record)
      hx_gd r = HX_GD(m.spin_row);
      const size_t np = (size_t)m.npad;
      r[HXSR_NBP * np] = alf;
      r[HXSR_NPP * np] = npp_t;   // (final_npp: the weights sum to one)
      r[HXSR_RH * np] = (sp_rd + sp_rs) + sp_rc;   // (no CH4 from thawed permafrost in the spinup)
      r[HXSR_RH_DET * np] = sp_rd; r[HXSR_RH_SOIL * np] = sp_rs;
      r[HXSR_CA_RESIDUAL * np] = residual;
      r[HXSR_HL_UPTAKE * np] += aH; r[HXSR_LL_UPTAKE * np] += aL; r[HXSR_HL_DO * np] += lHD;
    }
  } else if constexpr (hx_cons<CON>()) {
    // user-supplied [CO2] at this date: same transfer (:567-603); only whole dates exist
    if (HX_RARE((yc.mask & HXC_CO2) && !in_partial_year && !isnan(yc.co2))) {
      const double match = yc.co2 / PGC2PPM;
      const double residual = m.atmos - match;
      ca_residual = residual;
      m.cDO = residual + m.cDO;
      m.atmos = m.atmos - residual;
    }
  }
  if constexpr (CON && !SPIN) {
    if (HX_RARE(diag)) {  // the last stash of the year is the one that stays
      const HxBuffers &buf = *m.bufp;
      if (HX_RARE(omk & (1ull << HXO_NPP))) HX_GD(buf.out[HXO_NPP])[dgo] = fin_npp;
      if (HX_RARE(omk & (1ull << HXO_RH))) HX_GD(buf.out[HXO_RH])[dgo] = fin_rh;
      if (HX_RARE(omk & (1ull << HXO_RH_DET))) HX_GD(buf.out[HXO_RH_DET])[dgo] = fin_det;
      if (HX_RARE(omk & (1ull << HXO_RH_SOIL))) HX_GD(buf.out[HXO_RH_SOIL])[dgo] = fin_soil;
      if (HX_RARE(omk & (1ull << HXO_CA_RESIDUAL))) HX_GD(buf.out[HXO_CA_RESIDUAL])[dgo] = ca_residual;
    }
  }
  m.ode_start = t;
  HX_STAMP(m, 8);   // stash: ocean boxes + land pools
  if (more) {  // constants of the next segment
    if constexpr (B == HX_DYN) finish_interval<B, SPIN, CON>(m, Fn, K, K2, yc);
    else prep_interval<B, SPIN, CON>(m, lk, K, K2, yc);
  }
  HX_STAMP(m, 9);   // stash: next segment's interval constants
}

// x^(-1/3) for the step-shrink rule (x = err > 1): single-precision seed, two Newton steps on
// y^-3 = x in fp64 (relative error e -> 2e^2).  A wavefront takes the rejection branch whenever
// one of its 64 lanes rejects -- most passes of the step loop -- and exp(log(x) / -3) through the
// device library was ~180 instructions of it.
__device__ __forceinline__ double pow_m13(double x) {
  x = fmin(x, 1e30);  // (a larger error shrinks by the cap of 0.2 anyway)
  double y = (double)HX_EXP2F(-0.33333334f * HX_LOG2F((float)x));  // (x in [1, 1e30])
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double y3 = y * y * y;
    y = y + y * ((1.0 - x * y3) * (1.0 / 3.0));
  }
  return y;
}

// x^(-1/5) for the step-growth rule.  The fp64 log/exp pair it replaces is a ~75-instruction
// dependent chain; the result sits on the step loop's critical path (error -> growth -> dt ->
// the next attempt's first stage) and a single resident wavefront cannot hide it.
// Single-precision seed y (v_log_f32 / v_exp_f32: ~2e-7), then ONE step of third order: with
// d = 1 - x y^5 the root is y (1 - d)^(-1/5) = y (1 + d/5 + 3 d^2 / 25 + O(d^3)) exactly, d ~ 1e-6,
// so the two terms leave 0.09 d^3 ~ 1e-19; d itself is one fused operation on x y and y^4 (its
// rounding, 3e-16, enters the result by a fifth).  Seven dependent operations behind the seed
// where two Newton steps (round 1-4) were twelve.  (-DHX_POW_NEWTON: those, experiments.)
__device__ __forceinline__ double pow_m15(double x) {
  double y = (double)HX_EXP2F(-0.2f * HX_LOG2F((float)x));  // (x in [3.2e-4, 1]: normal range)
#ifndef HX_POW_NEWTON
  const double y2 = y * y, xy = x * y;
  const double y4 = y2 * y2;
  const double d = fma(-xy, y4, 1.0);
  return fma(y * d, fma(0.12, d, 0.2), y);
#else
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double y2 = y * y;
    const double y5 = y2 * y2 * y;
    y = y + y * ((1.0 - x * y5) * 0.2);
  }
  return y;
#endif
}

// CarbonCycleSolver::run for one model year t0 -> tnew (carbon-cycle-solver.cpp:
// 222-303).  The 64 lanes walk the reference's control flow in lock-step over
// SEGMENTS (one stash interval each): inner loop = dopri5 attempts until every
// lane has reached its own t_target (retries only move t_target), then ONE stash
// block for all lanes.  Lanes in reduced-timestep mode take up to 4 segments a
// year, the others idle through the extra ones; the expensive step and stash
// blocks are never interleaved lane by lane.
template <int B, bool SPIN, int CON = 0>
__device__ __forceinline__ void solve_year(Member<B> &m, const HxConst &kc,
                                           double t0, double tnew, const YearCon &yc) {
  constexpr int NP = hx_nbp<CON>() ? 6 : 5;  // solver variables (see rhs)
  // dopri5 tableau (odeint runge_kutta_dopri5)
#ifndef HX_TAB_LITERALS
  // The tableau as DATA (HxConst::tab, in the order a step uses it) behind wide scalar loads:
  // as literals its 30 constants cost 60 s_mov in every pass of the step loop (a 64-bit literal
  // is two s_mov_b32, and with machine LICM off -- see the Makefile -- they are materialised where
  // they are used), 12 % of the pass's instructions; five s_load_dwordx8/x16 a pass instead.
  // (-DHX_TAB_LITERALS: the old form, for experiment builds.)
  const double *T = kc.tab;
#if defined(HX_TAB_HEAD_AHEAD) && !defined(HX_HOST_EMULATION) && !defined(HX_TOP_TESTED_LOOPS)
  // (experiment: the tableau's first eight entries -- stages 2 to 4 -- requested at the END of the
  // previous pass, behind the error norm, and carried over the back edge in 16 scalar registers: the
  // head of a pass then does not wait for its first scalar load)
  double T0[8];
#define HX_TAB_HEAD() do { int o_ = 0; asm volatile("" : "+s"(o_)); const double *S_ = kc.tab + o_; \
    _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) T0[k_] = S_[k_]; } while (0)
#define b21 T0[0]
#define f2 T0[1]
#define b31 T0[2]
#define b32 T0[3]
#define f3 T0[4]
#define b41 T0[5]
#define b42 T0[6]
#define b43 T0[7]
#else
#define HX_TAB_HEAD() do { } while (0)
#define b21 T[0]
#define f2 T[1]
#define b31 T[2]
#define b32 T[3]
#define f3 T[4]
#define b41 T[5]
#define b42 T[6]
#define b43 T[7]
#endif
#define f4 T[8]
#define b51 T[9]
#define b52 T[10]
#define b53 T[11]
#define b54 T[12]
#define f5 T[13]
#define b61 T[14]
#define b62 T[15]
#define b63 T[16]
#define b64 T[17]
#define b65 T[18]
#define c1 T[19]
#define c3 T[20]
#define c4 T[21]
#define c5 T[22]
#define c6 T[23]
#define dc1 T[24]
#define dc3 T[25]
#define dc4 T[26]
#define dc5 T[27]
#define dc6 T[28]
#define dc7 T[29]
#else
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
  constexpr double f2 = 1.0 / 5, f3 = 3.0 / 10, f4 = 4.0 / 5, f5 = 8.0 / 9;
#endif
  constexpr double EPS = 2.220446049250313e-16;
  // The three land pools of an attempt as TWO coefficient chains (round 6).  Within a stash
  // interval vegetation, detritus and soil obey  dx_i/dt = c_i - r(t) x_i  with ONE loss rate r(t)
  // = luc_e / (their sum), known at every stage time (see rhs); the equations do not couple.  A
  // Runge-Kutta stage is an affine map of (x_i, c_i), the same map for the three pools: stage
  // value  xt_l,i = a_l x_i + g_l (h c_i),  scaled derivative  H_l,i = -(h r_l) xt_l,i + h c_i.  So
  // the dopri5 recursion runs ONCE on the coefficient pairs (a_l, g_l) instead of three times on
  // the pools, and each pool takes its candidate, its error estimate and its first derivative
  // from the final coefficients: x_i' = x_i + Pd x_i + Q (h c_i), xe_i = Ex x_i + Ec (h c_i).
  // The g chain works on deviations Gd_l = -(h r_l) g_l from the rate-free scheme (whose stage
  // coefficients are the nodes f_l and whose weights sum to 1 / 0 exactly), so nothing cancels.
  // 90 operations where the three pool chains took 118; the pass carries the rate r(t) from
  // attempt to attempt instead of three derivatives.  Same scheme, same step-size decisions;
  // results within rounding of the pool-wise form (-DHX_NO_LAND2: that form, experiments).
#if !defined(HX_NO_LAND2) && !defined(HX_NO_SCALED_STAGES) && !defined(HX_NO_RATE)
  constexpr bool LAND2 = hx_zchain<CON, SPIN>();
#else
  constexpr bool LAND2 = false;
#endif

  Interval K, K2s;
  Interval &K2 = hx_nbp<CON>() ? K2s : K;
  if constexpr (hx_w2<B>()) HX_W2_LOCAL(m);
  {
    LandK<B> lk;
    load_landk<B>(m, lk);
    prep_interval<B, SPIN, CON>(m, lk, K, K2, yc);
  }
  HX_STAMP(m, 5);   // interval constants of the year's first segment
  // getCValues  simpleNbox-runtime.cpp:247-258
  double y[NP], l4, l5, l7;
  auto load_pools = [&](bool in_loop) {
    if constexpr (hx_w2<B>()) {
      // inside the step loop (a retry) the surface boxes' carbon is in its parking slots
      if (in_loop) {
        y[0] = m.atmos; y[1] = m.veg[0]; y[2] = m.det[0]; y[3] = m.soil[0];
        y[4] = m.cDO + m.cIO + PKM(m, w2_slot(23)) + PKM(m, w2_slot(22));
        l4 = m.pf[0]; l5 = m.thawed[0]; l7 = m.earth;
        if constexpr (hx_nbp<CON>()) y[5] = m.thawed[0];
        return;
      }
    }
    double v = 0, d = 0, s = 0, p = 0, th = 0;
#pragma unroll hx_ur<B>()
    for (int b = 0; b < nbio<B>(m); ++b) { v += m.veg[b]; d += m.det[b]; s += m.soil[b];
                                            p += m.pf[b]; th += m.thawed[b]; }
    y[0] = m.atmos; y[1] = v; y[2] = d; y[3] = s; y[4] = m.cDO + m.cIO + m.cLL + m.cHL;
    l4 = p; l5 = th; l7 = m.earth;
    if constexpr (hx_nbp<CON>()) y[5] = th;
  };
  load_pools(false);
  m.ode_start = t0;
  double t = t0;   // time reached by accepted steps
  int retry = 0;
  // A member that has raised an error is not integrated any further: the reference aborts the
  // run at that point (h_exception), and a state that has left the model's domain (negative
  // pool, runaway CO2) can drive the step size towards zero -- a lane that never finishes its
  // year would hang the whole launch.
  bool alive = m.status == 0;
#ifndef HX_TOP_TESTED_LOOPS   // (bottom-tested like the step loop below, for the same reason)
  for (bool go_seg = __any(alive && t < tnew); go_seg; go_seg = __any(alive && t < tnew)) {
#else
  while (__any(alive && t < tnew)) {
#endif
    const bool seg = alive && t < tnew;
    // fresh integrate_adaptive call: by-value dt, fresh controlled stepper
    const double t_start = t;
    double t_target = tnew, dtl = m.sdt;
    double dxdt[NP];
    // A freshly constructed stepper evaluates the RHS once before its first step (FSAL starts
    // empty): here, ahead of the loop, for every lane -- only the lanes of this segment use it.
    rhs<B, SPIN, CON>(m, K, K2, yc, t, y, dxdt);
    if constexpr (LAND2) dxdt[1] = hx_div1(m.luc_e, (y[1] + y[2]) + y[3]);   // (the loss rate itself: see LAND2)
    int fails = 0;
    bool stepping = seg;
    HX_TAB_HEAD();
    if constexpr (hx_w2<B>()) w2_park_out<B>(m);
    // One pass = one dopri5 attempt of every stepping lane.  The attempt itself is straight-line
    // code that ALL lanes execute (a lane that has reached its target computes on stale values
    // and discards the result: ~2 % of the lanes, against the predication and register shuffling
    // of a divergent region around 330 instructions); per-lane control is the clip of dt, the
    // rare retry block, and ONE masked region at the end where an accepted step is committed.
#ifndef HX_TOP_TESTED_LOOPS
    // (the wavefront's vote at the BOTTOM of the loop: with `while (__any(stepping))` the vote is
    // the loop header, which the compiler may not duplicate (a convergent operation), so the loop
    // stays top-tested and every value that leaves it -- t, dt, the pools, their derivatives -- is
    // copied to its exit register at the top of EVERY pass)
    for (bool go_ = __any(stepping); go_; go_ = __any(stepping)) {
#else
    while (__any(stepping)) {
#endif
      HX_STAMP(m, 10);
      HX_COUNT(m, 16);  // step-loop iterations
#if !defined(HX_TAB_LITERALS) && !defined(HX_HOST_EMULATION) && !defined(HX_TOP_TESTED_LOOPS)
      {  // The tableau is read HERE, in every pass: an offset the optimiser cannot see through keeps
        // it from requesting all 30 entries ahead of the (bottom-tested) loop, where 60 scalar
        // registers held across the loop mean as many spilled ones around it.
        int toff = 0;
        asm volatile("" : "+s"(toff));
        T = kc.tab + toff;
      }
#endif
      if (stepping && ((t + dtl) - t_target) > EPS) dtl = t_target - t;
      // Every dopri5 stage time is <= t+dtl, and the model refuses any RHS
      // evaluation beyond max_timestep (ocean_component.cpp:621-625), so the
      // attempt throws CARBON_CYCLE_RETRY iff its last stage does.  A retry
      // (carbon-cycle-solver.cpp:266-276) is bookkeeping -- target halved, pools reloaded, a
      // fresh stepper -- and the attempt towards the new target follows in the same pass.
      bool need = stepping && ((t + dtl) - m.ode_start) > m.max_ts;
      if (__builtin_expect(__any(need), 0)) {
        bool reload = false;
        while (need) {
          ++retry;
          t_target = t_start + (t_target - t_start) / 2.0;
          t = t_start;
          m.sdt = t_target - t;
          dtl = m.sdt;
          reload = true;
          fails = 0;
          if (retry >= 8) { m.status |= HX_ERR_RETRIES; alive = false; stepping = false; }
          need = stepping && ((t + dtl) - m.ode_start) > m.max_ts;
        }
        if (reload) {
          load_pools(true); rhs<B, SPIN, CON>(m, K, K2, yc, t, y, dxdt);
          if constexpr (LAND2) dxdt[1] = hx_div1(m.luc_e, (y[1] + y[2]) + y[3]);
        }
      }
      double k2[NP], k3[NP], k4[NP], k5[NP], k6[NP], xt[NP], xn[NP], dn[NP];
      // the land-use loss rates of the attempt's stage times (see rhs): five independent
      // reciprocals up front.  Not in the CON kernels: an NBP constraint switches the interval
      // constants in mid-year, so the pools' sum is only piecewise linear there.
#ifdef HX_NO_RATE   // (experiment builds: the division at the head of every stage)
      constexpr bool RT = false;
#else
      constexpr bool RT = !hx_nbp<CON>();
#endif
      constexpr bool ZCH = hx_zchain<CON, SPIN>() && RT;
      double rr[5] = {0, 0, 0, 0, 0};
      if constexpr (RT) {
        const double tot0 = (y[1] + y[2]) + y[3];
        const double hC = dtl * K.dtot;
        rr[0] = hx_div1(m.luc_e, fma(hC, f2, tot0));
        rr[1] = hx_div1(m.luc_e, fma(hC, f3, tot0));
        rr[2] = hx_div1(m.luc_e, fma(hC, f4, tot0));
        rr[3] = hx_div1(m.luc_e, fma(hC, f5, tot0));
        rr[4] = hx_div1(m.luc_e, tot0 + hC);
      }
      double err = 0.0;
      if constexpr (ZCH) {
#ifndef HX_NO_SCALED_STAGES
        // atmosphere + ocean as the flux chain z (see hx_zchain above); vegetation, detritus and
        // soil through the stages.  dxdt[4] carries z at (t, y) from pass to pass.
        // The stages work on SCALED derivatives H = h k: the right-hand sides are affine,
        // h (c - r x) = (h c) - (h r) x, so scaling their constants once per pass (h lam, h alp, the
        // three h c_i, h luc_e inside the loss rates) replaces the 26 products h b_jl / h c_l / h dc_l
        // of the tableau and the h |k| of the error scales: the stage combinations then take the
        // tableau entries straight from their scalar registers.  13 multiplications by h a pass
        // where there were 31.  (-DHX_NO_SCALED_STAGES: the unscaled form, experiments.)
        const double h = dtl;
        const double hl = h * (K.aoA + K.aoB), ha = h * (K.aoA * K.Pn);
        const double cs[4] = {0.0, h * (K.v1 + m.luc_u), h * K.d2, h * K.s3};
        double hr[5], r5;
        {
          const double tot0 = (y[1] + y[2]) + y[3];
          const double hC = h * K.dtot, hle = h * m.luc_e;
          hr[0] = hx_div1(hle, fma(hC, f2, tot0));
          hr[1] = hx_div1(hle, fma(hC, f3, tot0));
          hr[2] = hx_div1(hle, fma(hC, f4, tot0));
          hr[3] = hx_div1(hle, fma(hC, f5, tot0));
          double inv = HX_RCP(tot0 + hC);
          inv = fma(fma(-(tot0 + hC), inv, 1.0), inv, inv);   // (hx_div1's reciprocal)
          hr[4] = hle * inv;
          r5 = m.luc_e * inv;
        }
        auto hland = [&](const double *x, double *H, double hrj) {   // h times rhs_land
#pragma unroll
          for (int i = 1; i <= 3; ++i) H[i] = fma(-hrj, x[i], cs[i]);
        };
        const double z1 = dxdt[4];
        double H1[4], H7[4];
        [[maybe_unused]] double l2_Ex = 0, l2_Ec = 0;
        [[maybe_unused]] double z3 = 0, z4 = 0, z5 = 0, z6 = 0;
        if constexpr (LAND2) {
          // STAGE by stage -- the flux chain z, chain a (coefficient of x_i: a_1 = 1, A_l = -(h r_l)
          // a_l) and chain g (coefficient of h c_i: g_1 = 0, G_l = 1 + Gd_l, Gd_l = -(h r_l) g_l; the
          // rate-free parts of its sums are the nodes f_l, Gd_1 = 0) side by side, so that every row
          // of the tableau is used in ONE place (chain by chain its 30 scalars were wanted three
          // times over: 54 spilled scalars against 37)
          const double hr0 = h * dxdt[1];      // dxdt[1] carries the loss rate at (t, y)
          const double Hz1 = fma(-hl, z1, ha);
          const double A1 = -hr0;
          // stage 2
          const double z2 = z1 + b21 * Hz1;
          const double a2 = fma(b21, A1, 1.0);
          const double Hz2 = fma(-hl, z2, ha);
          const double A2 = -hr[0] * a2;
          const double G2 = -hr[0] * f2;
          // stage 3
          z3 = z1 + b31 * Hz1 + b32 * Hz2;
          const double a3 = 1.0 + b31 * A1 + b32 * A2;
          const double g3 = fma(b32, G2, f3);
          const double Hz3 = fma(-hl, z3, ha);
          const double A3 = -hr[1] * a3;
          const double G3 = -hr[1] * g3;
          // stage 4
          z4 = z1 + b41 * Hz1 + b42 * Hz2 + b43 * Hz3;
          const double a4 = 1.0 + b41 * A1 + b42 * A2 + b43 * A3;
          const double g4 = f4 + b42 * G2 + b43 * G3;
          const double Hz4 = fma(-hl, z4, ha);
          const double A4 = -hr[2] * a4;
          const double G4 = -hr[2] * g4;
          // stage 5
          z5 = z1 + b51 * Hz1 + b52 * Hz2 + b53 * Hz3 + b54 * Hz4;
          const double a5 = 1.0 + b51 * A1 + b52 * A2 + b53 * A3 + b54 * A4;
          const double g5 = f5 + b52 * G2 + b53 * G3 + b54 * G4;
          const double Hz5 = fma(-hl, z5, ha);
          const double A5 = -hr[3] * a5;
          const double G5 = -hr[3] * g5;
          // stage 6
          z6 = z1 + b61 * Hz1 + b62 * Hz2 + b63 * Hz3 + b64 * Hz4 + b65 * Hz5;
          const double a6 = 1.0 + b61 * A1 + b62 * A2 + b63 * A3 + b64 * A4 + b65 * A5;
          const double g6 = 1.0 + b62 * G2 + b63 * G3 + b64 * G4 + b65 * G5;
          const double A6 = -hr[4] * a6;
          const double G6 = -hr[4] * g6;
          // the candidate: a_7 - 1, g_7
          const double Pd = c1 * A1 + c3 * A3 + c4 * A4 + c5 * A5 + c6 * A6;
          const double Q = 1.0 + (c3 * G3 + c4 * G4 + c5 * G5 + c6 * G6);
          const double A7 = -hr[4] * (1.0 + Pd);
          const double G7 = -hr[4] * Q;
          l2_Ex = dc1 * A1 + dc3 * A3 + dc4 * A4 + dc5 * A5 + dc6 * A6 + dc7 * A7;
          l2_Ec = dc3 * G3 + dc4 * G4 + dc5 * G5 + dc6 * G6 + dc7 * G7;
#pragma unroll
          for (int i = 1; i <= 3; ++i) {
            xn[i] = fma(Pd, y[i], fma(Q, cs[i], y[i]));
            H1[i] = fma(-hr0, y[i], cs[i]);      // the first stage's scaled derivative (error scale)
          }
          dn[1] = r5; dn[2] = 0.0; dn[3] = 0.0;  // (the rate at the candidate, for the next attempt)
        } else {
        const double Hz1 = fma(-hl, z1, ha);
        const double z2 = z1 + b21 * Hz1;
        const double Hz2 = fma(-hl, z2, ha);
        z3 = z1 + b31 * Hz1 + b32 * Hz2;
        const double Hz3 = fma(-hl, z3, ha);
        z4 = z1 + b41 * Hz1 + b42 * Hz2 + b43 * Hz3;
        const double Hz4 = fma(-hl, z4, ha);
        z5 = z1 + b51 * Hz1 + b52 * Hz2 + b53 * Hz3 + b54 * Hz4;
        const double Hz5 = fma(-hl, z5, ha);
        z6 = z1 + b61 * Hz1 + b62 * Hz2 + b63 * Hz3 + b64 * Hz4 + b65 * Hz5;
#pragma unroll
        for (int i = 1; i <= 3; ++i) H1[i] = h * dxdt[i];
#pragma unroll
        for (int i = 1; i <= 3; ++i) xt[i] = y[i] + b21 * H1[i];
        hland(xt, k2, hr[0]);
#pragma unroll
        for (int i = 1; i <= 3; ++i) xt[i] = y[i] + b31 * H1[i] + b32 * k2[i];
        hland(xt, k3, hr[1]);
#pragma unroll
        for (int i = 1; i <= 3; ++i) xt[i] = y[i] + b41 * H1[i] + b42 * k2[i] + b43 * k3[i];
        hland(xt, k4, hr[2]);
#pragma unroll
        for (int i = 1; i <= 3; ++i) xt[i] = y[i] + b51 * H1[i] + b52 * k2[i] + b53 * k3[i] + b54 * k4[i];
        hland(xt, k5, hr[3]);
#pragma unroll
        for (int i = 1; i <= 3; ++i)
          xt[i] = y[i] + b61 * H1[i] + b62 * k2[i] + b63 * k3[i] + b64 * k4[i] + b65 * k5[i];
        hland(xt, k6, hr[4]);
#pragma unroll
        for (int i = 1; i <= 3; ++i)
          xn[i] = y[i] + c1 * H1[i] + c3 * k3[i] + c4 * k4[i] + c5 * k5[i] + c6 * k6[i];
        rhs_land<B>(m, K, xn, dn, r5);          // (unscaled: the next attempt's first derivative)
        hland(xn, H7, hr[4]);
        }
        // the pair from the stage fluxes
        const double Z = h * (c1 * z1 + c3 * z3 + c4 * z4 + c5 * z5 + c6 * z6);
        xn[0] = fma(h, K.Pn, y[0]) - Z;
        xn[4] = y[4] + Z;
        const double z7 = fma(xn[0], K.aoA, -fma(xn[4] - K.totC, K.aoB, K.pG));   // = rhs()'s ao at the candidate
        dn[0] = K.Pn - z7;
        dn[4] = z7;
        const double E = h * (dc1 * z1 + dc3 * z3 + dc4 * z4 + dc5 * z5 + dc6 * z6 + dc7 * z7);
        const double d0 = kc.eps_abs + kc.eps_rel * (fabs(y[0]) + h * fabs(K.Pn - z1));
        const double d4 = kc.eps_abs + kc.eps_rel * (fabs(y[4]) + h * fabs(z1));
        const double qn = fabs(E), qd = fmin(d0, d4);
        err = hx_div(qn, qd);
        // The land pools' quotients: their equations are nearly linear in time (the loss rate r is
        // ~1e-3 / yr), so their error estimates sit orders of magnitude below the flux chain's and
        // the maximum is err as it stands (not once in 3e6 attempts of the bench ensemble).
        // Whether that holds is decided without a division -- |xe_i| / d_i <= qn / qd  <=>
        // |xe_i| qd <= qn d_i -- and the three divisions are only made when some lane of the
        // wavefront needs them; a lane takes their maximum only if ITS OWN test failed, so its
        // result does not depend on its neighbours.  (-DHX_LAND_QUOTIENTS: always divide.)
        double xe3[4], d3[4];
        bool small = true;
#if !defined(HX_LAND_QUOTIENTS) && !defined(HX_NO_LAND_BOUND)
        constexpr bool LBND = LAND2;
#else
        constexpr bool LBND = false;
#endif
        if constexpr (LBND) {
          // (round 6) ... and with the coefficient chains the common case does not look at the pools
          // one by one at all: |xe_i| = |Ex x_i + Ec (h c_i)| <= |Ex| max|x| + |Ec| max|h c|, and
          // d_i >= eps_abs + eps_rel min|x|, so ONE comparison with a factor of two in hand (it
          // covers every rounding of the two sides) implies the three exact ones -- thirteen
          // operations instead of 24.  The embedded pair's order conditions make Ex and Ec
          // rounding-sized (1e-17), the bound sits ten orders of magnitude below the flux chain's
          // quotient; a lane it does not clear takes the exact tests below, so every lane's result
          // is what the exact tests alone would give.
          const double ymax = fmax(fmax(fabs(y[1]), fabs(y[2])), fabs(y[3]));
          const double ymin = fmin(fmin(fabs(y[1]), fabs(y[2])), fabs(y[3]));
          const double cmax = fmax(fmax(fabs(cs[1]), fabs(cs[2])), fabs(cs[3]));
          const double bnd = fma(fabs(l2_Ex), ymax, fabs(l2_Ec) * cmax);
          const double dlow = fma(kc.eps_rel, ymin, kc.eps_abs);
          small = (bnd + bnd) * qd <= qn * dlow;
          if (__builtin_expect(__any(!small), 0)) {
            bool sm = true;
#pragma unroll
            for (int i = 1; i <= 3; ++i) {
              xe3[i] = fabs(fma(l2_Ex, y[i], l2_Ec * cs[i]));
              d3[i] = kc.eps_abs + kc.eps_rel * (fabs(y[i]) + fabs(H1[i]));
              sm = sm && (xe3[i] * qd <= qn * d3[i]);
            }
            double el = err;
#pragma unroll
            for (int i = 1; i <= 3; ++i) el = fmax(el, hx_div(xe3[i], d3[i]));
            err = (small || sm) ? err : el;
          }
        } else {
#pragma unroll
        for (int i = 1; i <= 3; ++i) {
          if constexpr (LAND2) xe3[i] = fabs(fma(l2_Ex, y[i], l2_Ec * cs[i]));
          else xe3[i] = fabs(dc1 * H1[i] + dc3 * k3[i] + dc4 * k4[i] + dc5 * k5[i] + dc6 * k6[i] + dc7 * H7[i]);
          d3[i] = kc.eps_abs + kc.eps_rel * (fabs(y[i]) + fabs(H1[i]));
          small = small && (xe3[i] * qd <= qn * d3[i]);
        }
#ifndef HX_LAND_QUOTIENTS
        if (__builtin_expect(__any(!small), 0))
#endif
        {
          double el = err;
#pragma unroll
          for (int i = 1; i <= 3; ++i) el = fmax(el, hx_div(xe3[i], d3[i]));
#ifndef HX_LAND_QUOTIENTS
          err = small ? err : el;
#else
          err = el;
#endif
        }
        }
#else
        // atmosphere + ocean as the flux chain z (see hx_zchain above); vegetation, detritus and
        // soil through the stages as before.  dxdt[4] carries z at (t, y) from pass to pass.
        const double lam = K.aoA + K.aoB, alp = K.aoA * K.Pn;
        const double z1 = dxdt[4];
        const double kz1 = fma(-lam, z1, alp);
        const double z2 = z1 + dtl * b21 * kz1;
#pragma unroll
        for (int i = 1; i <= 3; ++i) xt[i] = y[i] + dtl * b21 * dxdt[i];
        rhs_land<B>(m, K, xt, k2, rr[0]);
        const double kz2 = fma(-lam, z2, alp);
        const double z3 = z1 + dtl * b31 * kz1 + dtl * b32 * kz2;
#pragma unroll
        for (int i = 1; i <= 3; ++i) xt[i] = y[i] + dtl * b31 * dxdt[i] + dtl * b32 * k2[i];
        rhs_land<B>(m, K, xt, k3, rr[1]);
        const double kz3 = fma(-lam, z3, alp);
        const double z4 = z1 + dtl * b41 * kz1 + dtl * b42 * kz2 + dtl * b43 * kz3;
#pragma unroll
        for (int i = 1; i <= 3; ++i)
          xt[i] = y[i] + dtl * b41 * dxdt[i] + dtl * b42 * k2[i] + dtl * b43 * k3[i];
        rhs_land<B>(m, K, xt, k4, rr[2]);
        const double kz4 = fma(-lam, z4, alp);
        const double z5 = z1 + dtl * b51 * kz1 + dtl * b52 * kz2 + dtl * b53 * kz3 + dtl * b54 * kz4;
#pragma unroll
        for (int i = 1; i <= 3; ++i)
          xt[i] = y[i] + dtl * b51 * dxdt[i] + dtl * b52 * k2[i] + dtl * b53 * k3[i] + dtl * b54 * k4[i];
        rhs_land<B>(m, K, xt, k5, rr[3]);
        const double kz5 = fma(-lam, z5, alp);
        const double z6 = z1 + dtl * b61 * kz1 + dtl * b62 * kz2 + dtl * b63 * kz3 + dtl * b64 * kz4 +
                          dtl * b65 * kz5;
#pragma unroll
        for (int i = 1; i <= 3; ++i)
          xt[i] = y[i] + dtl * b61 * dxdt[i] + dtl * b62 * k2[i] + dtl * b63 * k3[i] + dtl * b64 * k4[i] +
                  dtl * b65 * k5[i];
        rhs_land<B>(m, K, xt, k6, rr[4]);
#pragma unroll
        for (int i = 1; i <= 3; ++i)
          xn[i] = y[i] + dtl * c1 * dxdt[i] + dtl * c3 * k3[i] + dtl * c4 * k4[i] + dtl * c5 * k5[i] +
                  dtl * c6 * k6[i];
        rhs_land<B>(m, K, xn, dn, rr[4]);
        // the pair from the stage fluxes
        const double Z = dtl * c1 * z1 + dtl * c3 * z3 + dtl * c4 * z4 + dtl * c5 * z5 + dtl * c6 * z6;
        xn[0] = fma(dtl, K.Pn, y[0]) - Z;
        xn[4] = y[4] + Z;
        const double z7 = fma(xn[0], K.aoA, -fma(xn[4] - K.totC, K.aoB, K.pG));   // = rhs()'s ao at the candidate
        dn[0] = K.Pn - z7;
        dn[4] = z7;
        const double E = dtl * dc1 * z1 + dtl * dc3 * z3 + dtl * dc4 * z4 + dtl * dc5 * z5 + dtl * dc6 * z6 +
                         dtl * dc7 * z7;
        const double d0 = kc.eps_abs + kc.eps_rel * (fabs(y[0]) + dtl * fabs(K.Pn - z1));
        const double d4 = kc.eps_abs + kc.eps_rel * (fabs(y[4]) + dtl * fabs(z1));
        const double qn = fabs(E), qd = fmin(d0, d4);
        err = hx_div(qn, qd);
        // The land pools' quotients: their equations are nearly linear in time (the loss rate r is
        // ~1e-3 / yr), so their error estimates sit orders of magnitude below the flux chain's and
        // the maximum is err as it stands.  Whether that holds is decided without a division --
        // |xe_i| / d_i <= qn / qd  <=>  |xe_i| qd <= qn d_i -- and the three divisions are only made
        // when some lane of the wavefront needs them; a lane takes their maximum only if ITS OWN
        // test failed, so its result does not depend on its neighbours.  (-DHX_LAND_QUOTIENTS:
        // always divide, experiments.)
        double xe3[4], d3[4];
        bool small = true;
#pragma unroll
        for (int i = 1; i <= 3; ++i) {
          xe3[i] = fabs(dtl * dc1 * dxdt[i] + dtl * dc3 * k3[i] + dtl * dc4 * k4[i] + dtl * dc5 * k5[i] +
                        dtl * dc6 * k6[i] + dtl * dc7 * dn[i]);
          d3[i] = kc.eps_abs + kc.eps_rel * (fabs(y[i]) + dtl * fabs(dxdt[i]));
          small = small && (xe3[i] * qd <= qn * d3[i]);
        }
#ifndef HX_LAND_QUOTIENTS
        if (__builtin_expect(__any(!small), 0))
#endif
        {
          double el = err;
#pragma unroll
          for (int i = 1; i <= 3; ++i) el = fmax(el, hx_div(xe3[i], d3[i]));
#ifndef HX_LAND_QUOTIENTS
          err = small ? err : el;
#else
          err = el;
#endif
        }
#endif
      } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) xt[i] = y[i] + dtl * b21 * dxdt[i];
      rhs<B, SPIN, CON, RT>(m, K, K2, yc, t + dtl * (1.0 / 5), xt, k2, rr[0]);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        xt[i] = y[i] + dtl * b31 * dxdt[i] + dtl * b32 * k2[i];
      rhs<B, SPIN, CON, RT>(m, K, K2, yc, t + dtl * (3.0 / 10), xt, k3, rr[1]);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        xt[i] = y[i] + dtl * b41 * dxdt[i] + dtl * b42 * k2[i] + dtl * b43 * k3[i];
      rhs<B, SPIN, CON, RT>(m, K, K2, yc, t + dtl * (4.0 / 5), xt, k4, rr[2]);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        xt[i] = y[i] + dtl * b51 * dxdt[i] + dtl * b52 * k2[i] +
                dtl * b53 * k3[i] + dtl * b54 * k4[i];
      rhs<B, SPIN, CON, RT>(m, K, K2, yc, t + dtl * (8.0 / 9), xt, k5, rr[3]);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        xt[i] = y[i] + dtl * b61 * dxdt[i] + dtl * b62 * k2[i] +
                dtl * b63 * k3[i] + dtl * b64 * k4[i] + dtl * b65 * k5[i];
      rhs<B, SPIN, CON, RT>(m, K, K2, yc, t + dtl, xt, k6, rr[4]);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        xn[i] = y[i] + dtl * c1 * dxdt[i] + dtl * c3 * k3[i] + dtl * c4 * k4[i] +
                dtl * c5 * k5[i] + dtl * c6 * k6[i];
      rhs<B, SPIN, CON, RT>(m, K, K2, yc, t + dtl, xn, dn, rr[4]);
      // default_error_checker: err = max_i |xe_i| / (eps_abs + eps_rel (|y_i| + dt |dy_i|)):
      // the quotients side by side (five independent reciprocals), then their maximum -- odeint's
      // own order of operations, and no chain of compare-and-select from one variable to the next.
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const double xe = dtl * dc1 * dxdt[i] + dtl * dc3 * k3[i] +
                          dtl * dc4 * k4[i] + dtl * dc5 * k5[i] +
                          dtl * dc6 * k6[i] + dtl * dc7 * dn[i];
        const double d = kc.eps_abs + kc.eps_rel * (fabs(y[i]) + dtl * fabs(dxdt[i]));
        err = fmax(err, hx_div(fabs(xe), d));
      }
      }
      HX_TAB_HEAD();   // (experiment builds: the next pass's first rows)
      // increase_step: err < 0.5 -> dt *= 0.9 * max(err, 5^-5)^(-1/5)
      const double grow = 0.9 * pow_m15(fmax(0.00032, err));
      if (stepping) {
        if (__builtin_expect(err > 1.0, 0)) {  // reject (rare): default_step_adjuster::decrease_step
          dtl *= fmax(0.9 * pow_m13(err), 0.2);
          if (++fails > 500) { m.status |= HX_ERR_STEPFAIL; alive = false; stepping = false; }
        } else {          // accept
          // pools with a constant derivative over the interval advance exactly
          l4 += dtl * K.k4; l7 += dtl * K.k7;
          if constexpr (!hx_nbp<CON>()) l5 += dtl * K.k5;
          t += dtl;
          if (err < 0.5) dtl *= grow;
#pragma unroll
          for (int i = 0; i < NP; ++i) { y[i] = xn[i]; dxdt[i] = dn[i]; }
          fails = 0;
          m.nsteps++;
          if (!((t_target - t) > EPS)) stepping = false;  // integrate_adaptive done
          // odeint puts no limit on accepted steps; a launch needs one (a member takes 3-8
          // steps a year, a stiff one a few hundred)
          if (m.nsteps > HX_MAX_STEPS_PER_YEAR) {
            m.status |= HX_ERR_STEPFAIL; alive = false; stepping = false;
          }
        }
      }
      HX_STAMP(m, 6);   // dopri5 attempts (+ retries)
    }
    if constexpr (hx_w2<B>()) w2_park_in<B>(m);
    HX_STAMP(m, 10);
    HX_COUNT(m, 17);    // segments
    if (seg && alive) {
      // the solver keeps integrating its own c[] afterwards (no getCValues,
      // carbon-cycle-solver.cpp:282-287); only the frozen-pool constants move
      retry = 0;
      stash<B, SPIN, CON>(m, t, y, l4, hx_nbp<CON>() ? y[NP - 1] : l5, l7, K, K2, yc, t < tnew);
      if (m.status != 0) alive = false;
    }
  }
}

}  // namespace
#ifndef HX_TAB_LITERALS
#undef b21
#undef f2
#undef b31
#undef b32
#undef f3
#undef b41
#undef b42
#undef b43
#undef f4
#undef b51
#undef b52
#undef b53
#undef b54
#undef f5
#undef b61
#undef b62
#undef b63
#undef b64
#undef b65
#undef c1
#undef c3
#undef c4
#undef c5
#undef c6
#undef dc1
#undef dc3
#undef dc4
#undef dc5
#undef dc6
#undef dc7
#endif
