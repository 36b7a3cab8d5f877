// hx_dev_track.h -- carbon tracking: where the carbon of every pool originated.
// Part of the device code of hx_kernels.hip (one translation unit).
//
// The reference carries, with every pool and flux, a map source pool -> fraction
// (fluxpool, inst/include/fluxpool.hpp:166-298): pool + flux mixes the two maps by value,
// pool - flux and pool * k keep the map.  From Core::trackingDate on, SimpleNbox::stashCValues
// (src/simpleNbox-runtime.cpp:289-540) and the ocean boxes (src/oceanbox.cpp:240-303) therefore
// move origins around with every flux of every stash; CSVFluxPoolVisitor prints the maps once a
// year (src/csv_tracking_visitor.cpp:60-120).
//
// Here: one TP x TP matrix of fractions per member and tracked year in HBM -- the yearly record
// itself (HxBuffers::track_out_f, tiled [year][64-member block][row][64]; TP = 2 + 5 B + 4 pools: atmos_c, earth_c, per biome veg/
// detritus/soil/permafrost/thawedp, ocean HL/LL/intermediate/deep).  The first stash of a year
// reads last year's matrix and writes this year's, later stashes of the year update it in place;
// the ocean's copy of the atmosphere's origins (SimpleNbox::run hands it over once a year,
// simpleNbox-runtime.cpp:215-227) is last year's atmosphere row.  No second table, no yearly copy.
//
// Every mixing operation acts on each source column separately (only the rare "a zero total is
// shared equally among the sources of the map" needs the whole map, and that is a bit mask), so a
// stash walks the matrix in chunks of 4 (looped kernels: 8) source columns: the pools' values, the reciprocals of
// the totals and the masks are recomputed for every chunk, the fractions of a chunk live in
// registers (~14 maps x 4 or 8 doubles) whatever the number of pools.  The biome count may
// therefore be the looped kernels' run-time value (up to 16 biomes: 86 pools, two mask words).
// Only the tracking instantiations of the run kernel (CON == 2) contain this code.
#pragma once

namespace {

// source columns per pass: 4 in the unrolled kernels (8 spill: 8 192 x 4 biomes 83 -> 114 ms), 8 in
// the looped ones (1 024 x 8 biomes 288 -> 224 ms, 256 x 16 966 -> 671 ms)
template <int B> constexpr int trk_c() { return B == 0 ? 8 : 4; }
enum { TKP_ATM = 0, TKP_EARTH = 1 };
__host__ __device__ constexpr int hx_trk_pools(int nb) { return 2 + 5 * nb + 4; }
__host__ __device__ constexpr int hx_trk_mask_words(int nb) { return hx_trk_pools(nb) > 64 ? 2 : 1; }
// rows of one year of HxBuffers::track_out_v: TP pool values, then TP mask words (bit patterns,
// low word), then -- more than 64 pools -- the TP high words
__host__ __device__ constexpr int hx_trk_vrows(int nb) {
  return hx_trk_pools(nb) * (1 + hx_trk_mask_words(nb));
}

template <int W> struct TMask { unsigned long long w[W]; };
template <int W> __device__ __forceinline__ TMask<W> tm_or(const TMask<W> &a, const TMask<W> &b) {
  TMask<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.w[i] = a.w[i] | b.w[i];
  return r;
}
template <int W> __device__ __forceinline__ TMask<W> tm_bit(int s) {
  TMask<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.w[i] = (s >> 6) == i ? 1ull << (s & 63) : 0ull;
  return r;
}
template <int W> __device__ __forceinline__ int tm_count(const TMask<W> &a) {
  int n = 0;
#pragma unroll
  for (int i = 0; i < W; ++i) n += __popcll(a.w[i]);
  return n;
}
// bits s0 .. s0 + C - 1 of a mask
template <int W, int C> __device__ __forceinline__ unsigned tm_window(const TMask<W> &a, int s0) {
  unsigned long long x = a.w[0];
  if constexpr (W == 2) x = (s0 >= 64) ? a.w[1] : x;  // (chunks are aligned: never across words)
  return (unsigned)(x >> (s0 & 63)) & ((1u << C) - 1u);
}

template <int W, int C>
struct TV {  // a pool or a flux with its origins, columns s0 .. s0 + C - 1 of its map
  double val;
  double f[C];
  TMask<W> mask;  // which sources are in the map (all of them)
  unsigned win;   // ... and the bits of this chunk's columns
};

template <int W, int C>
__device__ __forceinline__ TV<W, C> tv_self(int self, double val, int s0) {
  TV<W, C> r;
  r.val = val;
#pragma unroll
  for (int c = 0; c < C; ++c) r.f[c] = (s0 + c == self) ? 1.0 : 0.0;
  r.mask = tm_bit<W>(self);
  r.win = tm_window<W, C>(r.mask, s0);
  return r;
}
template <int W, int C>
__device__ __forceinline__ TV<W, C> tv_from(const TV<W, C> &pool, double val) {  // flux_from_*
  TV<W, C> r = pool;
  r.val = val;
  return r;
}
template <int W, int C>
__device__ __forceinline__ TV<W, C> tv_add(const TV<W, C> &a, const TV<W, C> &b) {  // operator+
  TV<W, C> r;
  r.val = a.val + b.val;
  r.mask = tm_or<W>(a.mask, b.mask);
  r.win = a.win | b.win;
  // ONE reciprocal of the new total for the fractions (the reference divides source by source,
  // fluxpool.hpp:197-257; a fraction then differs from the quotient in its last place).  A source
  // outside both maps has fraction 0 on both sides and stays 0.
  const double inv = hx_recip(r.val);
  const double wa = a.val * inv, wb = b.val * inv;  // (the same for every column: computed once a stash)
#pragma unroll
  for (int c = 0; c < C; ++c) r.f[c] = a.f[c] * wa + b.f[c] * wb;
  if (__builtin_expect(__any(r.val == 0.0), 0)) {  // a zero total: equal shares, :243-251
    const double share = 1.0 / (double)tm_count<W>(r.mask);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const double v = (r.win >> c & 1u) ? share : 0.0;
      r.f[c] = (r.val == 0.0) ? v : r.f[c];
    }
  }
  return r;
}
template <int W, int C>
__device__ __forceinline__ TV<W, C> tv_sub(TV<W, C> a, const TV<W, C> &b) { a.val = a.val - b.val; return a; }
template <int W, int C>
__device__ __forceinline__ TV<W, C> tv_mul(TV<W, C> a, double k) { a.val = a.val * k; return a; }

// what the stash hands over: values it has computed anyway (the per-biome ones are read from the
// member where they are used -- the stash calls this BEFORE it writes the new pools)
struct TrkStashIn {
  double yf;
  double pre[4];       // box carbon before update_state: HL, LL, IO, DO
  double closs[7];     // HL->DO, LL->HL, LL->IO, IO->LL, IO->HL, IO->DO, DO->IO
  double aH, aL;       // final atmosphere_flux of the surface boxes (after the solver adjustment)
  double post[4];      // box carbon after update_state
  double npp_total, rh_adj, total;   // NPP after any NBP constraint, RH factor, c[veg+det+soil]
  double npp_rh, inv_nr;  // what the biome weights are made of
};

// one member's view of the record for one stash.  The record is tiled by wavefront,
// [block of 64 members][slot][row][64] (slot 0: the identity every pool starts from at the
// tracking date, slot 1 + k: tracked year k): what a stash touches -- last year's matrix and
// this year's -- is one contiguous piece of HBM (2 x 62 KB for one biome), not TP x TP rows that
// lie npad doubles apart (a TLB miss and a DRAM page for every row), and every address is a
// wave-uniform base plus the lane's 32-bit offset.
template <int W, int C>
struct TrkIo {
  hx_gd blk_f, blk_v;            // the block's slot of LAST year; this year's follows it
  unsigned src_f, dst_f, prv_f;  // the lane's offsets from blk_f: read / written by this stash, last year
  unsigned src_v, dst_v, prv_v;  // ... and from blk_v (masks)
  int TP, s0;
  bool last;  // last chunk: the masks are stored

  __device__ __forceinline__ TV<W, C> load_from(unsigned off_f, unsigned off_v, int p, double val, int s0) const {
    TV<W, C> r;
    r.val = val;
    hx_gd row = blk_f + (size_t)(p * TP + s0) * 64;  // (uniform)
    // (columns past the last pool read the next row's first entries -- finite numbers, the record
    //  is padded by a row group at its end -- and are never stored)
#pragma unroll
    for (int c = 0; c < C; ++c) r.f[c] = row[off_f + c * 64];
    hx_gd mrow = blk_v + (size_t)(TP + p) * 64;
#pragma unroll
    for (int i = 0; i < W; ++i)  // (the looped kernels carry two words; the record has the high
      r.mask.w[i] = (i == 0 || TP > 64)  // ones only when there are more than 64 pools)
          ? (unsigned long long)__double_as_longlong(mrow[off_v + (unsigned)(i * TP * 64)]) : 0ull;
    r.win = tm_window<W, C>(r.mask, s0);
    return r;
  }
  // (s: the first column of the chunk -- the current one, or the next one's for a prefetch)
  __device__ __forceinline__ TV<W, C> load(int p, double val, int s) const { return load_from(src_f, src_v, p, val, s); }
  __device__ __forceinline__ TV<W, C> load_prev(int p, int s) const { return load_from(prv_f, prv_v, p, 0.0, s); }
  __device__ __forceinline__ void store(int p, const TV<W, C> &t) const {
    hx_gd row = blk_f + (size_t)(p * TP + s0) * 64;
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (s0 + c < TP) row[dst_f + c * 64] = t.f[c];
    if (last) {
      hx_gd mrow = blk_v + (size_t)(TP + p) * 64;
#pragma unroll
      for (int i = 0; i < W; ++i)
        if (i == 0 || TP > 64) mrow[dst_v + (unsigned)(i * TP * 64)] = __longlong_as_double((long long)t.mask.w[i]);
    }
  }
};

// start_tracking() at Core::trackingDate: every pool is 100 % itself (slot 0 of the record)
template <int B>
__device__ void track_start(const Member<B> &m) {
  const HxBuffers &buf = *m.bufp;
  const int nb = nbio<B>(m);
  const int TP = hx_trk_pools(nb), vrows = hx_trk_vrows(nb);
  hx_gd f = HX_GD(buf.track_out_f) + ((size_t)blockIdx.x * buf.trk_slots * (size_t)(TP * TP) * 64 + m.lane);
  hx_gd v = HX_GD(buf.track_out_v) + ((size_t)blockIdx.x * buf.trk_slots * (size_t)vrows * 64 + m.lane);
#pragma unroll 1
  for (int p = 0; p < TP; ++p) {
#pragma unroll 1
    for (int s = 0; s < TP; ++s) f[(size_t)(p * TP + s) * 64] = (p == s) ? 1.0 : 0.0;
    v[(size_t)(TP + p) * 64] = __longlong_as_double(p < 64 ? (long long)(1ull << p) : 0ll);
    if (TP > 64) v[(size_t)(2 * TP + p) * 64] = __longlong_as_double(p >= 64 ? (long long)(1ull << (p - 64)) : 0ll);
  }
}

template <int B>
__device__ void track_stash(const Member<B> &m, const LandK<B> &lk, const TrkStashIn &in) {
  constexpr int W = (B == HX_DYN) ? 2 : (hx_trk_pools(B) > 64 ? 2 : 1);
  constexpr int C = trk_c<B>();
  using T = TV<W, C>;
  const HxBuffers &buf = *m.bufp;
  const int nb = nbio<B>(m);
  const int TP = hx_trk_pools(nb), O0 = 2 + 5 * nb;
  const int vrows = hx_trk_vrows(nb);
  const int k = m.iy - m.trk_iy;
  const bool first = m.nstash == 1;  // this lane's first stash of the year: from last year's matrix
  TrkIo<W, C> io;
  io.TP = TP;
  io.blk_f = HX_GD(buf.track_out_f) + ((size_t)blockIdx.x * buf.trk_slots + k) * (size_t)(TP * TP) * 64;
  io.blk_v = HX_GD(buf.track_out_v) + ((size_t)blockIdx.x * buf.trk_slots + k) * (size_t)vrows * 64;
  io.prv_f = (unsigned)m.lane;
  io.prv_v = (unsigned)m.lane;
  io.dst_f = (unsigned)m.lane + (unsigned)(TP * TP * 64);
  io.dst_v = (unsigned)m.lane + (unsigned)(vrows * 64);
  io.src_f = first ? io.prv_f : io.dst_f;
  io.src_v = first ? io.prv_v : io.dst_v;
  const double yf = in.yf;
  HX_STAMP(m, 8);

  struct Land5 { T veg, det, soil, pf, tp; };
  auto load_biome = [&](int b, int s) {
    Land5 r;
    r.veg = io.load(2 + 5 * b + 0, m.veg[b], s); r.det = io.load(2 + 5 * b + 1, m.det[b], s);
    r.soil = io.load(2 + 5 * b + 2, m.soil[b], s); r.pf = io.load(2 + 5 * b + 3, m.pf[b], s);
    r.tp = io.load(2 + 5 * b + 4, m.thawed[b], s);
    return r;
  };
#pragma unroll 1
  for (int s0 = 0; s0 < TP; s0 += C) {
    HX_STAMP(m, 20);  // tracking: set-up, everything that does not depend on the column
    io.s0 = s0;
    io.last = s0 + C >= TP;
    T atm, earth;
    Land5 cur;
    // ---------------- ocean: oceanbox.cpp:240-257, 262-271, 297-303 ----------------
    T oa_flux, ao_flux;
    {
      T box[4], addn[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        box[b] = io.load(O0 + b, in.pre[b], s0);
        addn[b] = tv_self<W, C>(O0 + b, 0.0, s0);
      }
      const T atm_copy = io.load_prev(TKP_ATM, s0);  // OceanComponent::atmosphere_cpool
      // (everything else the land part starts with is requested now: the stores below may alias
      // any later load as far as the compiler can tell, and with one wavefront per SIMD every
      // load issued after them is a fully exposed HBM round trip.  Requesting the NEXT chunk's
      // rows here as well was tried: 120 more live registers, 422 spills, twice the time.)
      atm = io.load(TKP_ATM, m.atmos, s0);
      earth = io.load(TKP_EARTH, m.earth, s0);
      cur = load_biome(0, s0);
      constexpr int from_[7] = {0, 1, 1, 2, 2, 2, 3}, to_[7] = {3, 0, 2, 1, 0, 3, 2};
#pragma unroll
      for (int i = 0; i < 7; ++i)
        addn[to_[i]] = tv_add<W, C>(addn[to_[i]], tv_from<W, C>(box[from_[i]], in.closs[i]));
      T ao[2], oa[2];
      const double af[2] = {in.aH, in.aL};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        ao[b] = tv_from<W, C>(atm_copy, (af[b] > 0) ? af[b] : 0.0);
        oa[b] = tv_from<W, C>(box[b], (af[b] > 0) ? 0.0 : -af[b]);
      }
      oa_flux = tv_add<W, C>(oa[1], oa[0]);  // get_oaflux: LL + HL
      ao_flux = tv_add<W, C>(ao[1], ao[0]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        T c = tv_add<W, C>(box[b], addn[b]);
        if (b < 2) { c = tv_add<W, C>(c, ao[b]); c = tv_sub<W, C>(c, oa[b]); }
        else c = tv_add<W, C>(c, tv_from<W, C>(atm_copy, 0.0));
        io.store(O0 + b, c);
      }
    }

    HX_STAMP(m, 21);  // tracking: ocean boxes of a chunk (with the chunk's loads)
    // ---------------- land: simpleNbox-runtime.cpp:289-540 ----------------
    const T ccs_flux = tv_from<W, C>(atm, m.daccs);
    auto biome = [&](int b, const Land5 &pools) {
      T veg = pools.veg, det = pools.det, soil = pools.soil, pf = pools.pf, tp = pools.tp;
      const double rfda = m_rh_fda(m, b), rfsa = m_rh_fsa(m, b), rco2 = m_rh_tp_co2(m, lk, b),
                   rch4 = m_rh_tp_ch4(m, lk, b);
      // the weights of SimpleNbox::stashCValues, as the stash computes them
      const double wt = (B == 1) ? 1.0 : hx_div_cr(m_npp(m, lk, b) + ((rfda + rfsa) + rco2), in.npp_rh, in.inv_nr);
      const double veg_frac = veg.val / in.total, det_frac = det.val / in.total,
                   soil_frac = soil.val / in.total;
      const T luc_fva = tv_mul<W, C>(tv_from<W, C>(veg, m.luc_e * veg_frac), yf);
      const T luc_fda = tv_mul<W, C>(tv_from<W, C>(det, m.luc_e * det_frac), yf);
      const T luc_fsa = tv_mul<W, C>(tv_from<W, C>(soil, m.luc_e * soil_frac), yf);
      const T luc_fav = tv_mul<W, C>(tv_from<W, C>(atm, m.luc_u), yf);
      const double npp_biome = in.npp_total * wt;
      const double fv = lk.f_nppv[b], fd = lk.f_nppd[b], fl = lk.f_litterd[b];
      const T npp_fav = tv_mul<W, C>(tv_from<W, C>(atm, npp_biome * fv), yf);
      const T npp_fad = tv_mul<W, C>(tv_from<W, C>(atm, npp_biome * fd), yf);
      const T npp_fas = tv_mul<W, C>(tv_from<W, C>(atm, npp_biome * (1 - fv - fd)), yf);
      const double rh_co2_adj = rco2 * in.rh_adj, rh_ch4_adj = rch4 * in.rh_adj;
      const T rh_fda_flux = tv_mul<W, C>(tv_from<W, C>(det, rfda * in.rh_adj), yf);
      const T rh_fsa_flux = tv_mul<W, C>(tv_from<W, C>(soil, rfsa * in.rh_adj), yf);
      const T rh_fpa_co2 = tv_mul<W, C>(tv_from<W, C>(tp, rh_co2_adj), yf);
      const T rh_fpa_ch4 = tv_mul<W, C>(tv_from<W, C>(tp, rh_ch4_adj), yf);
      atm = tv_add<W, C>(tv_add<W, C>(tv_sub<W, C>(tv_add<W, C>(atm, luc_fva), luc_fav), luc_fda), luc_fsa);
      veg = tv_sub<W, C>(tv_add<W, C>(veg, luc_fav), luc_fva);
      soil = tv_sub<W, C>(soil, luc_fsa);  // (the reference's detritus line has no effect, :458)
      veg = tv_add<W, C>(veg, npp_fav);
      det = tv_add<W, C>(det, npp_fad);
      soil = tv_add<W, C>(soil, npp_fas);
      atm = tv_sub<W, C>(tv_sub<W, C>(tv_sub<W, C>(atm, npp_fav), npp_fad), npp_fas);
      atm = tv_add<W, C>(tv_add<W, C>(tv_add<W, C>(atm, rh_fda_flux), rh_fsa_flux), rh_fpa_co2);
      det = tv_sub<W, C>(det, rh_fda_flux);
      soil = tv_sub<W, C>(soil, rh_fsa_flux);
      tp = tv_sub<W, C>(tv_sub<W, C>(tp, rh_fpa_co2), rh_fpa_ch4);
      {  // compute_pf_thaw_refreeze :744-772 on the pools as they are now
        double x = pf.val * m.f_new_thaw[b], y = 0.0;
        if (x < 0) {
          const double want = -x;
          x = 0.0;
          const double remaining = tp.val - rh_co2_adj - rh_ch4_adj;
          y = (remaining < want) ? remaining : want;
        }
        const T pf_thaw = tv_mul<W, C>(tv_from<W, C>(pf, x), yf);
        const T pf_refreeze_tp = tv_mul<W, C>(tv_from<W, C>(tp, y), yf);
        const T pf_refreeze_soil = tv_mul<W, C>(tv_from<W, C>(soil, 0.0), yf);
        pf = tv_add<W, C>(tv_add<W, C>(tv_sub<W, C>(pf, pf_thaw), pf_refreeze_tp), pf_refreeze_soil);
        tp = tv_sub<W, C>(tv_add<W, C>(tp, pf_thaw), pf_refreeze_tp);
        soil = tv_sub<W, C>(soil, pf_refreeze_soil);
      }
      const T litter = tv_mul<W, C>(veg, 0.035 * yf);
      det = tv_add<W, C>(det, tv_mul<W, C>(litter, fl));
      soil = tv_add<W, C>(soil, tv_mul<W, C>(litter, 1 - fl));
      veg = tv_sub<W, C>(veg, litter);
      const T detsoil = tv_mul<W, C>(det, 0.6 * yf);
      soil = tv_add<W, C>(soil, detsoil);
      det = tv_sub<W, C>(det, detsoil);
      io.store(2 + 5 * b + 0, veg); io.store(2 + 5 * b + 1, det); io.store(2 + 5 * b + 2, soil);
      io.store(2 + 5 * b + 3, pf); io.store(2 + 5 * b + 4, tp);
    };
    if constexpr (B == HX_DYN) {
#pragma unroll 1
      for (int b = 0; b < nb; ++b) {  // the next biome's rows are on their way while this one mixes
        const Land5 nxb = load_biome(b + 1 < nb ? b + 1 : b, s0);
        biome(b, cur);
        cur = nxb;
      }
    } else {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        Land5 nxb;
        if (b + 1 < B) nxb = load_biome(b + 1, s0);
        biome(b, cur);
        if (b + 1 < B) cur = nxb;
      }
    }
    const T ffi_flux = tv_from<W, C>(earth, m.ffi);
    earth = tv_add<W, C>(tv_sub<W, C>(earth, ffi_flux), ccs_flux);
    atm = tv_sub<W, C>(tv_add<W, C>(atm, ffi_flux), ccs_flux);
    atm = tv_sub<W, C>(tv_add<W, C>(atm, oa_flux), ao_flux);
    io.store(TKP_EARTH, earth);
    io.store(TKP_ATM, atm);
    HX_STAMP(m, 22);  // tracking: land pools of a chunk
  }
}

// ===========================================================================================
// One biome: the companion wavefronts (hx_run_kernel<1, HF, KERPM, 3>, 192 threads a block).
//
// The eleven pools' maps -- 121 fractions, 11 masks -- fit the register files of two wavefronts that
// do nothing else (source columns 0-5 and 6-10: every mixing operation acts on each column
// separately).  Wave 0 runs the model and, at every stash, hands the values the maps move with
// (36 doubles a lane) over through LDS; waves 1 and 2 keep their half of the matrix in registers
// for the launch, mix while wave 0 is already integrating the next segment, and write a year's
// matrix to the record once, at the year's end.  Nothing of the matrix is re-read from HBM, the
// model wavefront's registers hold no tracking state, and the three overlap.
// ===========================================================================================
enum { TRKR_YF = 0, TRKR_PRE = 1, TRKR_CLOSS = 5, TRKR_AH = 12, TRKR_AL, TRKR_NPP, TRKR_RHADJ, TRKR_TOTAL,
       TRKR_ATMOS, TRKR_EARTH, TRKR_DACCS, TRKR_FFI, TRKR_LUCE, TRKR_LUCU, TRKR_ACTIVE, TRKR_B0 };
// ... and per biome
enum { TRKB_VEG = 0, TRKB_DET, TRKB_SOIL, TRKB_PF, TRKB_TP, TRKB_FNT, TRKB_RFDA, TRKB_RFSA, TRKB_RCO2,
       TRKB_RCH4, TRKB_FV, TRKB_FD, TRKB_FL, TRKB_WT, TRKB_N };
constexpr int trkr_n(int B) { return TRKR_B0 + TRKB_N * B; }  // hand-over slots of one set
// companions of a block and source columns each: 11 pools 2 x 6, 16 pools 3 x 6, 21 pools 3 x 7,
// 26 pools 3 x 9 (a block is at most four wavefronts of 512 registers: the CU's register file)
template <int B> constexpr int trk_waves() { return B == 1 ? 2 : 3; }
template <int B> constexpr int trk_nc() { return (hx_trk_pools(B) + trk_waves<B>() - 1) / trk_waves<B>(); }
enum { TRKC_STASH = 1, TRKC_YEAR = 2, TRKC_DONE = 3 };

// The companions' additions are kept in program order: left alone, the scheduler starts every
// reciprocal and weight of a stash as early as it can (they depend on the handed-over values only)
// and ~100 more doubles are live than the maps need -- into AGPRs and scratch again.
#ifdef HX_HOST_EMULATION
#define HX_TRK_ORDER()
#else
#define HX_TRK_ORDER() __builtin_amdgcn_sched_barrier(0)
#endif
template <int NC>
struct TVR {  // a pool or a flux with columns c0 .. c0 + NC - 1 of its map, in registers
  double val;
  double f[NC];
  unsigned long long mask;  // (the whole map's names)
};
template <int TP> __device__ __forceinline__ TVR<TP> tvr_self(int self, double val, int c0) {
  TVR<TP> r;
  r.val = val;
#pragma unroll
  for (int s = 0; s < TP; ++s) r.f[s] = (c0 + s == self) ? 1.0 : 0.0;
  r.mask = 1ull << self;
  return r;
}
template <int TP> __device__ __forceinline__ TVR<TP> tvr_from(const TVR<TP> &pool, double val) {
  TVR<TP> r = pool;
  r.val = val;
  return r;
}
template <int TP> __device__ __forceinline__ TVR<TP> tvr_add(const TVR<TP> &a, const TVR<TP> &b, int c0) {
  HX_TRK_ORDER();
  TVR<TP> r;
  r.val = a.val + b.val;
  r.mask = a.mask | b.mask;
  const double inv = hx_recip(r.val);
  const double wa = a.val * inv, wb = b.val * inv;
#pragma unroll
  for (int s = 0; s < TP; ++s) r.f[s] = a.f[s] * wa + b.f[s] * wb;
  if (__builtin_expect(__any(r.val == 0.0), 0)) {  // a zero total: equal shares, fluxpool.hpp:243-251
    const double share = 1.0 / (double)__popcll(r.mask);
#pragma unroll
    for (int s = 0; s < TP; ++s) {
      const double v = (r.mask >> (c0 + s) & 1ull) ? share : 0.0;
      r.f[s] = (r.val == 0.0) ? v : r.f[s];
    }
  }
  return r;
}
// the same with a structurally empty left side (a fresh sum of inflows: value 0, the pool's own
// name in the map): the flux's fractions as they are instead of times value / value
template <int TP> __device__ __forceinline__ TVR<TP> tvr_add_to_empty(int self, const TVR<TP> &b, double bval, int c0) {
  TVR<TP> r = b;
  r.val = 0.0 + bval;
  r.mask = b.mask | 1ull << self;
  if (__builtin_expect(__any(r.val == 0.0), 0)) {
    const double share = 1.0 / (double)__popcll(r.mask);
#pragma unroll
    for (int s = 0; s < TP; ++s) {
      const double v = (r.mask >> (c0 + s) & 1ull) ? share : 0.0;
      r.f[s] = (r.val == 0.0) ? v : r.f[s];
    }
  }
  return r;
}
// ... and with a flux of value 0 (the reference adds a few of those: only the names arrive)
template <int TP> __device__ __forceinline__ TVR<TP> tvr_add_names(TVR<TP> a, unsigned long long names, int c0) {
  a.mask |= names;
  if (__builtin_expect(__any(a.val == 0.0), 0)) {
    const double share = 1.0 / (double)__popcll(a.mask);
#pragma unroll
    for (int s = 0; s < TP; ++s) {
      const double v = (a.mask >> (c0 + s) & 1ull) ? share : 0.0;
      a.f[s] = (a.val == 0.0) ? v : a.f[s];
    }
  }
  return a;
}
template <int TP> __device__ __forceinline__ TVR<TP> tvr_sub(TVR<TP> a, const TVR<TP> &b) { a.val = a.val - b.val; return a; }
template <int TP> __device__ __forceinline__ TVR<TP> tvr_mul(TVR<TP> a, double k) { a.val = a.val * k; return a; }

// wave 0, inside the stash: publish what the maps move with.  Two sets of slots: event n goes to
// set n & 1 and is followed by ONE barrier, which both wavefronts pass once per stash of the
// WAVEFRONT (the lanes that are not at a segment end stay inactive, TRKR_ACTIVE tells the
// companion which ones are).  Wave 0 can be one event ahead: it waits only when the companion has
// not finished the event before the last.
template <int B>
__device__ __forceinline__ void track_post_stash(const Member<B> &m, const LandK<B> &lk,
                                                 const TrkStashIn &in) {
  const int ev = m.trk_cmd[4];
  double (*r)[64] = m.trk_rec + (ev & 1) * trkr_n(B);
  const int l = m.lane;
  r[TRKR_YF][l] = in.yf;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[TRKR_PRE + i][l] = in.pre[i];
#pragma unroll
  for (int i = 0; i < 7; ++i) r[TRKR_CLOSS + i][l] = in.closs[i];
  r[TRKR_AH][l] = in.aH; r[TRKR_AL][l] = in.aL;
  r[TRKR_NPP][l] = in.npp_total; r[TRKR_RHADJ][l] = in.rh_adj; r[TRKR_TOTAL][l] = in.total;
  r[TRKR_ATMOS][l] = m.atmos; r[TRKR_EARTH][l] = m.earth; r[TRKR_DACCS][l] = m.daccs;
  r[TRKR_FFI][l] = m.ffi; r[TRKR_LUCE][l] = m.luc_e; r[TRKR_LUCU][l] = m.luc_u;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    double (*rb)[64] = r + TRKR_B0 + TRKB_N * b;
    const double rfda = m_rh_fda(m, b), rfsa = m_rh_fsa(m, b), rco2 = m_rh_tp_co2(m, lk, b);
    rb[TRKB_VEG][l] = m.veg[b]; rb[TRKB_DET][l] = m.det[b]; rb[TRKB_SOIL][l] = m.soil[b];
    rb[TRKB_PF][l] = m.pf[b]; rb[TRKB_TP][l] = m.thawed[b]; rb[TRKB_FNT][l] = m.f_new_thaw[b];
    rb[TRKB_RFDA][l] = rfda; rb[TRKB_RFSA][l] = rfsa; rb[TRKB_RCO2][l] = rco2;
    rb[TRKB_RCH4][l] = m_rh_tp_ch4(m, lk, b);
    rb[TRKB_FV][l] = lk.f_nppv[b]; rb[TRKB_FD][l] = lk.f_nppd[b]; rb[TRKB_FL][l] = lk.f_litterd[b];
    // the biome's weight, as SimpleNbox::stashCValues computes it
    rb[TRKB_WT][l] = (B == 1) ? 1.0 : hx_div_cr(m_npp(m, lk, b) + ((rfda + rfsa) + rco2), in.npp_rh, in.inv_nr);
  }
  r[TRKR_ACTIVE][l] = (double)(ev + 1);  // (this lane takes part in event ev: every companion reads it)
  m.trk_cmd[(ev & 1) * 2] = TRKC_STASH;
  m.trk_cmd[4] = ev + 1;
  __syncthreads();
}
// wave 0, uniform code: the year's matrix goes to the record / the launch is over
__device__ __forceinline__ void track_post(int *cmd, int what, int iy) {
  const int ev = cmd[4];
  cmd[(ev & 1) * 2] = what; cmd[(ev & 1) * 2 + 1] = iy;
  cmd[4] = ev + 1;
  __syncthreads();
}

// waves 1 .. trk_waves<B>(): columns c0 .. c0 + trk_nc<B>() - 1 of every map (one biome: sources 0-5
// and 6-10 -- 66 fractions each, not 121: with the whole matrix on one wavefront the maps spilled
// into AGPRs and every addition paid ~40 register moves for its 22 multiply-adds)
template <int B>
__device__ void track_companion(const HxArgs *__restrict__ args, int iy_from, int lane, int c0,
                                double (*rec)[64], int *cmd) {
  constexpr int TP = hx_trk_pools(B), O0 = 2 + 5 * B;
  constexpr int NC = trk_nc<B>();
  using T = TVR<NC>;
  const HxBuffers &buf = args->buf;
  const int trk_iy = args->kc.trk_iy;
  hx_gd rf = HX_GD(buf.track_out_f) + ((size_t)blockIdx.x * buf.trk_slots * (size_t)(TP * TP) * 64 + lane);
  hx_gd rv = HX_GD(buf.track_out_v) + ((size_t)blockIdx.x * buf.trk_slots * (size_t)hx_trk_vrows(B) * 64 + lane);
  T P[TP];  // (rows: every pool; columns: this wavefront's)
  // start_tracking(): every pool 100 % itself -- or, when a run resumes past the tracking date
  // (run() again, reset(date)), the maps of the end of last year from the record
  const bool resume = iy_from + 1 > trk_iy;
  const int slot0 = resume ? iy_from - trk_iy + 1 : 0;
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    P[p] = tvr_self<NC>(p, 0.0, c0);
    if (resume) {
#pragma unroll
      for (int s = 0; s < NC; ++s)  // (past the last source: the next row's first entries, never stored)
        P[p].f[s] = rf[((size_t)slot0 * TP * TP + p * TP + c0 + s) * 64];
      P[p].mask = (unsigned long long)__double_as_longlong(rv[((size_t)slot0 * hx_trk_vrows(B) + TP + p) * 64]);
    }
  }
  T atm_copy = P[TKP_ATM];  // OceanComponent::atmosphere_cpool: the atmosphere as of SimpleNbox::run
  for (int ev = 0;; ++ev) {
    __syncthreads();  // wave 0 has published event ev (and may go on to prepare the next one)
    const int what = cmd[(ev & 1) * 2], iy = cmd[(ev & 1) * 2 + 1];
    if (what == TRKC_DONE) break;
    if (what == TRKC_YEAR) {
      // CSVFluxPoolVisitor: the year's maps (the pool values are written by wave 0)
      const size_t slot = (size_t)(iy - trk_iy) + 1;
      hx_gd of = rf + slot * (size_t)(TP * TP) * 64;
      hx_gd ov = rv + slot * (size_t)hx_trk_vrows(B) * 64;
#pragma unroll
      for (int p = 0; p < TP; ++p) {
#pragma unroll
        for (int s = 0; s < NC; ++s)
          if (c0 + s < TP) of[(size_t)(p * TP + c0 + s) * 64] = P[p].f[s];
        if (c0 == 0) ov[(size_t)(TP + p) * 64] = __longlong_as_double((long long)P[p].mask);
      }
      atm_copy = P[TKP_ATM];
      continue;
    }
    // ---- a stash: the values stay in their LDS slots (wave 0 writes the other set next) ----
    double (*in_)[64] = rec + (ev & 1) * trkr_n(B);
#define in(i) in_[(i)][lane]
    const bool active = in(TRKR_ACTIVE) == (double)(ev + 1);
    if (active) {
      const double yf = in(TRKR_YF);
      // ---------------- ocean: oceanbox.cpp:240-257, 262-271, 297-303 ----------------
#pragma unroll
      for (int b = 0; b < 4; ++b) P[O0 + b].val = in(TRKR_PRE + b);
      constexpr int from_[7] = {0, 1, 1, 2, 2, 2, 3}, to_[7] = {3, 0, 2, 1, 0, 3, 2};
      T addn[4];  // (each box's first inflow is transfers 0..3, its second 4..6)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        addn[to_[i]] = tvr_add_to_empty<NC>(O0 + to_[i], P[O0 + from_[i]], in(TRKR_CLOSS + i), c0);
#pragma unroll
      for (int i = 4; i < 7; ++i)
        addn[to_[i]] = tvr_add<NC>(addn[to_[i]], tvr_from<NC>(P[O0 + from_[i]], in(TRKR_CLOSS + i)), c0);
      T ao[2], oa[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const double af = in(TRKR_AH + b);
        ao[b] = tvr_from<NC>(atm_copy, (af > 0) ? af : 0.0);
        oa[b] = tvr_from<NC>(P[O0 + b], (af > 0) ? 0.0 : -af);
      }
      const T oa_flux = tvr_add<NC>(oa[1], oa[0], c0);  // get_oaflux: LL + HL
      const T ao_flux = tvr_add<NC>(ao[1], ao[0], c0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        T c = tvr_add<NC>(P[O0 + b], addn[b], c0);
        if (b < 2) { c = tvr_add<NC>(c, ao[b], c0); c = tvr_sub<NC>(c, oa[b]); }
        else c = tvr_add_names<NC>(c, atm_copy.mask, c0);
        P[O0 + b] = c;
      }
      // ---------------- land: simpleNbox-runtime.cpp:289-540 ----------------
      T &atm = P[TKP_ATM];
      atm.val = in(TRKR_ATMOS); P[TKP_EARTH].val = in(TRKR_EARTH);
      const T ffi_flux = tvr_from<NC>(P[TKP_EARTH], in(TRKR_FFI));
      const T ccs_flux = tvr_from<NC>(atm, in(TRKR_DACCS));
      const double total = in(TRKR_TOTAL), luc_e = in(TRKR_LUCE), rh_adj = in(TRKR_RHADJ);
#pragma unroll
      for (int b = 0; b < B; ++b) {
#define inb(i) in_[TRKR_B0 + TRKB_N * b + (i)][lane]
      T &veg = P[2 + 5 * b], &det = P[3 + 5 * b], &soil = P[4 + 5 * b], &pf = P[5 + 5 * b], &tp = P[6 + 5 * b];
      veg.val = inb(TRKB_VEG); det.val = inb(TRKB_DET); soil.val = inb(TRKB_SOIL); pf.val = inb(TRKB_PF);
      tp.val = inb(TRKB_TP);
      const double veg_frac = veg.val / total, det_frac = det.val / total, soil_frac = soil.val / total;
      const T luc_fva = tvr_mul<NC>(tvr_from<NC>(veg, luc_e * veg_frac), yf);
      const T luc_fda = tvr_mul<NC>(tvr_from<NC>(det, luc_e * det_frac), yf);
      const T luc_fsa = tvr_mul<NC>(tvr_from<NC>(soil, luc_e * soil_frac), yf);
      const T luc_fav = tvr_mul<NC>(tvr_from<NC>(atm, in(TRKR_LUCU)), yf);
      const double npp_biome = in(TRKR_NPP) * inb(TRKB_WT);
      const double fv = inb(TRKB_FV), fd = inb(TRKB_FD), fl = inb(TRKB_FL);
      const T npp_fav = tvr_mul<NC>(tvr_from<NC>(atm, npp_biome * fv), yf);
      const T npp_fad = tvr_mul<NC>(tvr_from<NC>(atm, npp_biome * fd), yf);
      const T npp_fas = tvr_mul<NC>(tvr_from<NC>(atm, npp_biome * (1 - fv - fd)), yf);
      const double rh_co2_adj = inb(TRKB_RCO2) * rh_adj, rh_ch4_adj = inb(TRKB_RCH4) * rh_adj;
      const T rh_fda_flux = tvr_mul<NC>(tvr_from<NC>(det, inb(TRKB_RFDA) * rh_adj), yf);
      const T rh_fsa_flux = tvr_mul<NC>(tvr_from<NC>(soil, inb(TRKB_RFSA) * rh_adj), yf);
      const T rh_fpa_co2 = tvr_mul<NC>(tvr_from<NC>(tp, rh_co2_adj), yf);
      const T rh_fpa_ch4 = tvr_mul<NC>(tvr_from<NC>(tp, rh_ch4_adj), yf);
      atm = tvr_add<NC>(tvr_add<NC>(tvr_sub<NC>(tvr_add<NC>(atm, luc_fva, c0), luc_fav), luc_fda, c0), luc_fsa, c0);
      veg = tvr_sub<NC>(tvr_add<NC>(veg, luc_fav, c0), luc_fva);
      soil = tvr_sub<NC>(soil, luc_fsa);  // (the reference's detritus line has no effect, :458)
      veg = tvr_add<NC>(veg, npp_fav, c0);
      det = tvr_add<NC>(det, npp_fad, c0);
      soil = tvr_add<NC>(soil, npp_fas, c0);
      atm = tvr_sub<NC>(tvr_sub<NC>(tvr_sub<NC>(atm, npp_fav), npp_fad), npp_fas);
      atm = tvr_add<NC>(tvr_add<NC>(tvr_add<NC>(atm, rh_fda_flux, c0), rh_fsa_flux, c0), rh_fpa_co2, c0);
      det = tvr_sub<NC>(det, rh_fda_flux);
      soil = tvr_sub<NC>(soil, rh_fsa_flux);
      tp = tvr_sub<NC>(tvr_sub<NC>(tp, rh_fpa_co2), rh_fpa_ch4);
      {  // compute_pf_thaw_refreeze :744-772 on the pools as they are now
        double x = pf.val * inb(TRKB_FNT), y = 0.0;
        if (x < 0) {
          const double want = -x;
          x = 0.0;
          const double remaining = tp.val - rh_co2_adj - rh_ch4_adj;
          y = (remaining < want) ? remaining : want;
        }
        const T pf_thaw = tvr_mul<NC>(tvr_from<NC>(pf, x), yf);
        const T pf_refreeze_tp = tvr_mul<NC>(tvr_from<NC>(tp, y), yf);
        // (pf_refreeze_soil: a flux of 0 from the soil)
        pf = tvr_add_names<NC>(tvr_add<NC>(tvr_sub<NC>(pf, pf_thaw), pf_refreeze_tp, c0), soil.mask, c0);
        tp = tvr_sub<NC>(tvr_add<NC>(tp, pf_thaw, c0), pf_refreeze_tp);
      }
      const T litter = tvr_mul<NC>(veg, 0.035 * yf);
      det = tvr_add<NC>(det, tvr_mul<NC>(litter, fl), c0);
      soil = tvr_add<NC>(soil, tvr_mul<NC>(litter, 1 - fl), c0);
      veg = tvr_sub<NC>(veg, litter);
      const T detsoil = tvr_mul<NC>(det, 0.6 * yf);
      soil = tvr_add<NC>(soil, detsoil, c0);
      det = tvr_sub<NC>(det, detsoil);
#undef inb
      }
      P[TKP_EARTH] = tvr_add<NC>(tvr_sub<NC>(P[TKP_EARTH], ffi_flux), ccs_flux, c0);
      atm = tvr_sub<NC>(tvr_add<NC>(atm, ffi_flux, c0), ccs_flux);
      atm = tvr_sub<NC>(tvr_add<NC>(atm, oa_flux, c0), ao_flux);
    }
#undef in
  }
}

}  // namespace
