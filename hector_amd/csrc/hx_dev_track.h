// hx_dev_track.h -- carbon tracking: where the carbon of every pool originated.
// Part of the device code of hx_kernels.hip (one translation unit).
//
// The reference carries, with every pool and flux, a map source pool -> fraction
// (fluxpool, inst/include/fluxpool.hpp:166-298): pool + flux mixes the two maps by value,
// pool - flux and pool * k keep the map.  From Core::trackingDate on, SimpleNbox::stashCValues
// (src/simpleNbox-runtime.cpp:289-540) and the ocean boxes (src/oceanbox.cpp:240-303) therefore
// move origins around with every flux of every stash; CSVFluxPoolVisitor prints the maps once a
// year (src/csv_tracking_visitor.cpp:60-120).  Here: one TP x TP matrix of fractions per member
// in HBM (TP = 2 + 5 B + 4 pools: atmos_c, earth_c, per biome veg/detritus/soil/permafrost/
// thawedp, ocean HL/LL/intermediate/deep), loaded into lane-private arrays for the stash,
// mixed in the reference's order, stored back.  Only the tracking instantiation of the run
// kernel (CON == 2) contains this code.
#pragma once

namespace {

template <int B> constexpr int hx_tp() { return 2 + 5 * B + 4; }
enum { TKP_ATM = 0, TKP_EARTH = 1 };
template <int B> constexpr int tkp_land(int b, int k) { return 2 + 5 * b + k; }
template <int B> constexpr int tkp_ocean(int box) { return 2 + 5 * B + box; }  // 0 HL 1 LL 2 IO 3 DO

// rows of the per-member tracking table (HxBuffers::track, [hx_trk_rows<B>()][npad])
template <int B> constexpr int hx_trk_rows() { return hx_tp<B>() * hx_tp<B>() + 2 * hx_tp<B>() + 1; }
template <int B> constexpr int trk_row_f(int pool, int src) { return pool * hx_tp<B>() + src; }
template <int B> constexpr int trk_row_mask(int pool) { return hx_tp<B>() * hx_tp<B>() + pool; }
template <int B> constexpr int trk_row_atmcopy(int src) {
  return hx_tp<B>() * hx_tp<B>() + hx_tp<B>() + src;
}
template <int B> constexpr int trk_row_atmcopy_mask() { return hx_tp<B>() * hx_tp<B>() + 2 * hx_tp<B>(); }

template <int TP>
struct TV {  // a pool or a flux with its origins
  double val;
  double f[TP];
  unsigned long long mask;  // which sources are in the map
};

template <int TP>
__device__ __forceinline__ TV<TP> tv_self(int self, double val) {
  TV<TP> r;
  r.val = val;
#pragma unroll
  for (int s = 0; s < TP; ++s) r.f[s] = (s == self) ? 1.0 : 0.0;
  r.mask = 1ull << self;
  return r;
}
template <int TP>
__device__ __forceinline__ TV<TP> tv_from(const TV<TP> &pool, double val) {  // flux_from_*
  TV<TP> r = pool;
  r.val = val;
  return r;
}
template <int TP>
__device__ __forceinline__ TV<TP> tv_add(const TV<TP> &a, const TV<TP> &b) {  // operator+
  TV<TP> r;
  r.val = a.val + b.val;
  r.mask = a.mask | b.mask;
  const double share = 1.0 / (double)__popcll(r.mask);
  // ONE reciprocal of the new total for the TP fractions (the reference divides source by source,
  // fluxpool.hpp:197-257: an IEEE division each -- ~30 instructions on this machine, TP of them
  // in each of the ~35 additions of a stash were most of the tracking kernels' time); a
  // fraction then differs from the quotient in its last place
  const double inv = hx_recip(r.val);
#pragma unroll
  for (int s = 0; s < TP; ++s) {
    const double pool_s = a.val * a.f[s] + b.val * b.f[s];
    double v = (r.val != 0.0) ? pool_s * inv : share;
    r.f[s] = (r.mask >> s & 1ull) ? v : 0.0;
  }
  return r;
}
template <int TP>
__device__ __forceinline__ TV<TP> tv_sub(TV<TP> a, const TV<TP> &b) { a.val = a.val - b.val; return a; }
template <int TP>
__device__ __forceinline__ TV<TP> tv_mul(TV<TP> a, double k) { a.val = a.val * k; return a; }

// what the stash hands over: everything is a value it has computed anyway
template <int B>
struct TrkStashIn {
  double yf;
  double pre[4];       // box carbon before update_state: HL, LL, IO, DO
  double closs[7];     // HL->DO, LL->HL, LL->IO, IO->LL, IO->HL, IO->DO, DO->IO
  double aH, aL;       // final atmosphere_flux of the surface boxes (after the solver adjustment)
  double post[4];      // box carbon after update_state
  double veg[B], det[B], soil[B], pf[B], tp[B];  // land pools before the stash
  double atmos, earth;                          // ... and atmosphere / earth
  double npp_total, rh_adj, total;              // NPP after any NBP constraint, RH factor, c[veg+det+soil]
  double wt[B], wt_pf[B];
  double nveg, ndet, nsoil, npf, ntp, natm, nearth;  // the solver's end values
  double ffi, daccs, luc_e, luc_u;
  double f_new_thaw[B];
  double rh_fda[B], rh_fsa[B], rh_co2[B], rh_ch4[B];  // per year, before rh_adj
};

template <int B>
__device__ void track_stash(const HxBuffers &buf, int mem, const LandK<B> &lk,
                            const TrkStashIn<B> &in) {
  constexpr int TP = hx_tp<B>();
  using T = TV<TP>;
  hx_gd tr = HX_GD(buf.track) + mem;
  const size_t np = (size_t)buf.npad;
  T P[TP];
#pragma unroll 1
  for (int p = 0; p < TP; ++p) {
    for (int s = 0; s < TP; ++s) P[p].f[s] = tr[(size_t)trk_row_f<B>(p, s) * np];
    P[p].mask = (unsigned long long)tr[(size_t)trk_row_mask<B>(p) * np];
    P[p].val = 0.0;
  }
  T atm_copy;  // OceanComponent::atmosphere_cpool: the atmosphere as of SimpleNbox::run
  for (int s = 0; s < TP; ++s) atm_copy.f[s] = tr[(size_t)trk_row_atmcopy<B>(s) * np];
  atm_copy.mask = (unsigned long long)tr[(size_t)trk_row_atmcopy_mask<B>() * np];
  atm_copy.val = 0.0;

  // ---------------- ocean: oceanbox.cpp:240-257, 262-271, 297-303 ----------------
  constexpr int O0 = tkp_ocean<B>(0);
  for (int b = 0; b < 4; ++b) P[O0 + b].val = in.pre[b];
  const int from_[7] = {0, 1, 1, 2, 2, 2, 3}, to_[7] = {3, 0, 2, 1, 0, 3, 2};
  T addn[4];
  double subn[4] = {0, 0, 0, 0};
  for (int b = 0; b < 4; ++b) addn[b] = tv_self<TP>(O0 + b, 0.0);
#pragma unroll 1
  for (int i = 0; i < 7; ++i) {
    const T closs = tv_from<TP>(P[O0 + from_[i]], in.closs[i]);
    addn[to_[i]] = tv_add<TP>(addn[to_[i]], closs);
    subn[from_[i]] = subn[from_[i]] + closs.val;
  }
  T ao[2], oa[2];
  const double af[2] = {in.aH, in.aL};
  for (int b = 0; b < 2; ++b) {
    if (af[b] > 0) { ao[b] = tv_from<TP>(atm_copy, af[b]); oa[b] = tv_from<TP>(P[O0 + b], 0.0); }
    else { ao[b] = tv_from<TP>(atm_copy, 0.0); oa[b] = tv_from<TP>(P[O0 + b], -af[b]); }
  }
#pragma unroll 1
  for (int b = 0; b < 4; ++b) {
    T c = tv_add<TP>(P[O0 + b], addn[b]);
    if (b < 2) { c = tv_add<TP>(c, ao[b]); c = tv_sub<TP>(c, oa[b]); }
    else c = tv_add<TP>(c, tv_from<TP>(atm_copy, 0.0));
    c.val = in.post[b];
    P[O0 + b] = c;
  }

  // ---------------- land: simpleNbox-runtime.cpp:289-540 ----------------
  P[TKP_ATM].val = in.atmos; P[TKP_EARTH].val = in.earth;
  const T ffi_flux = tv_from<TP>(P[TKP_EARTH], in.ffi);
  const T ccs_flux = tv_from<TP>(P[TKP_ATM], in.daccs);
  const T oa_flux = tv_add<TP>(oa[1], oa[0]);  // get_oaflux: LL + HL
  const T ao_flux = tv_add<TP>(ao[1], ao[0]);
#pragma unroll 1
  for (int b = 0; b < B; ++b) {
    T &veg = P[tkp_land<B>(b, 0)], &det = P[tkp_land<B>(b, 1)], &soil = P[tkp_land<B>(b, 2)],
      &pf = P[tkp_land<B>(b, 3)], &tp = P[tkp_land<B>(b, 4)], &atm = P[TKP_ATM];
    veg.val = in.veg[b]; det.val = in.det[b]; soil.val = in.soil[b]; pf.val = in.pf[b];
    tp.val = in.tp[b];
    const double yf = in.yf;
    const double veg_frac = veg.val / in.total, det_frac = det.val / in.total,
                 soil_frac = soil.val / in.total;
    const T luc_fva = tv_mul<TP>(tv_from<TP>(veg, in.luc_e * veg_frac), yf);
    const T luc_fda = tv_mul<TP>(tv_from<TP>(det, in.luc_e * det_frac), yf);
    const T luc_fsa = tv_mul<TP>(tv_from<TP>(soil, in.luc_e * soil_frac), yf);
    const T luc_fav = tv_mul<TP>(tv_from<TP>(atm, in.luc_u), yf);
    const double npp_biome = in.npp_total * in.wt[b];
    const T npp_fav = tv_mul<TP>(tv_from<TP>(atm, npp_biome * lk.f_nppv[b]), yf);
    const T npp_fad = tv_mul<TP>(tv_from<TP>(atm, npp_biome * lk.f_nppd[b]), yf);
    const T npp_fas = tv_mul<TP>(tv_from<TP>(atm, npp_biome * (1 - lk.f_nppv[b] - lk.f_nppd[b])), yf);
    const double rh_co2_adj = in.rh_co2[b] * in.rh_adj, rh_ch4_adj = in.rh_ch4[b] * in.rh_adj;
    const T rh_fda_flux = tv_mul<TP>(tv_from<TP>(det, in.rh_fda[b] * in.rh_adj), yf);
    const T rh_fsa_flux = tv_mul<TP>(tv_from<TP>(soil, in.rh_fsa[b] * in.rh_adj), yf);
    const T rh_fpa_co2 = tv_mul<TP>(tv_from<TP>(tp, rh_co2_adj), yf);
    const T rh_fpa_ch4 = tv_mul<TP>(tv_from<TP>(tp, rh_ch4_adj), yf);
    atm = tv_add<TP>(tv_add<TP>(tv_sub<TP>(tv_add<TP>(atm, luc_fva), luc_fav), luc_fda), luc_fsa);
    veg = tv_sub<TP>(tv_add<TP>(veg, luc_fav), luc_fva);
    soil = tv_sub<TP>(soil, luc_fsa);  // (the reference's detritus line has no effect, :458)
    veg = tv_add<TP>(veg, npp_fav);
    det = tv_add<TP>(det, npp_fad);
    soil = tv_add<TP>(soil, npp_fas);
    atm = tv_sub<TP>(tv_sub<TP>(tv_sub<TP>(atm, npp_fav), npp_fad), npp_fas);
    atm = tv_add<TP>(tv_add<TP>(tv_add<TP>(atm, rh_fda_flux), rh_fsa_flux), rh_fpa_co2);
    det = tv_sub<TP>(det, rh_fda_flux);
    soil = tv_sub<TP>(soil, rh_fsa_flux);
    tp = tv_sub<TP>(tv_sub<TP>(tp, rh_fpa_co2), rh_fpa_ch4);
    {  // compute_pf_thaw_refreeze :744-772 on the pools as they are now
      double x = pf.val * in.f_new_thaw[b], y = 0.0;
      if (x < 0) {
        const double want = -x;
        x = 0.0;
        const double remaining = tp.val - rh_co2_adj - rh_ch4_adj;
        y = (remaining < want) ? remaining : want;
      }
      const T pf_thaw = tv_mul<TP>(tv_from<TP>(pf, x), yf);
      const T pf_refreeze_tp = tv_mul<TP>(tv_from<TP>(tp, y), yf);
      const T pf_refreeze_soil = tv_mul<TP>(tv_from<TP>(soil, 0.0), yf);
      pf = tv_add<TP>(tv_add<TP>(tv_sub<TP>(pf, pf_thaw), pf_refreeze_tp), pf_refreeze_soil);
      tp = tv_sub<TP>(tv_add<TP>(tp, pf_thaw), pf_refreeze_tp);
      soil = tv_sub<TP>(soil, pf_refreeze_soil);
    }
    const T litter = tv_mul<TP>(veg, 0.035 * yf);
    det = tv_add<TP>(det, tv_mul<TP>(litter, lk.f_litterd[b]));
    soil = tv_add<TP>(soil, tv_mul<TP>(litter, 1 - lk.f_litterd[b]));
    veg = tv_sub<TP>(veg, litter);
    const T detsoil = tv_mul<TP>(det, 0.6 * yf);
    soil = tv_add<TP>(soil, detsoil);
    det = tv_sub<TP>(det, detsoil);
    veg.val = in.nveg * in.wt[b]; det.val = in.ndet * in.wt[b]; soil.val = in.nsoil * in.wt[b];
    pf.val = in.npf * in.wt_pf[b]; tp.val = in.ntp * in.wt_pf[b];
  }
  P[TKP_EARTH] = tv_add<TP>(tv_sub<TP>(P[TKP_EARTH], ffi_flux), ccs_flux);
  P[TKP_ATM] = tv_sub<TP>(tv_add<TP>(P[TKP_ATM], ffi_flux), ccs_flux);
  P[TKP_ATM] = tv_sub<TP>(tv_add<TP>(P[TKP_ATM], oa_flux), ao_flux);

#pragma unroll 1
  for (int p = 0; p < TP; ++p) {
    for (int s = 0; s < TP; ++s) tr[(size_t)trk_row_f<B>(p, s) * np] = P[p].f[s];
    tr[(size_t)trk_row_mask<B>(p) * np] = (double)P[p].mask;
  }
}

}  // namespace
