// hx_dev_member.h -- what a lane holds: Member, the LDS park, biome constants, SoA load/store helpers
// Part of the device code of hx_kernels.hip (one translation unit; see its header for the
// reference file:line map).
#pragma once

// the looped kernels' park (dynamic LDS: hx_npark_dyn(nbiome) x 512 B, set at launch)
extern __shared__ double hx_dyn_park[][64];

namespace {

// ---------------------------------------------------------------------------
// A compiler-only fence: values cached from memory may not be carried across it.
// The year loop is split into phases by these so that per-member constants that a
// phase needs are (re)loaded from HBM / L2 inside the phase instead of being kept
// in registers across the whole solver -- register pressure, not bandwidth, is what
// limits this kernel (DESIGN.md "registers").
#define HX_FENCE() asm volatile("" ::: "memory")

// Table accesses of the form row address + lane offset (hx_ldm / hx_stm / w2_ld below):
// (HX_SROW: the row address passes through an empty asm as a scalar pair -- left alone the
// optimiser folds base + lane offset into one 64-bit VECTOR address per lane and adds each row's
// scalar offset to it with a v_lshl_add_u64, 32 of them a year for the block's SSTs alone; as an
// opaque scalar the row offset is added on the scalar unit and the access takes the
// `global_load v, v_off, s[base:base+1]` form.  Only where control flow is wave-uniform.)
#if !defined(HX_HOST_EMULATION) && !defined(HX_NO_SROW)
#define HX_SROW(p) asm("" : "+s"(p))
#else
#define HX_SROW(p)
#endif
// (HX_VOFF: and the lane offset through one as a 32-bit vector register -- its zero extension to
// 64 bits is then made next to the access, where instruction selection can see it and fold it
// into the instruction; hoisted out of the block as a 64-bit pair, it cannot.)
#if !defined(HX_HOST_EMULATION) && !defined(HX_NO_VOFF)
#define HX_VOFF(o) asm("" : "+v"(o))
#else
#define HX_VOFF(o)
#endif

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
// Template tag of the LOOPED kernels: the biome count is the core's (HxBuffers::nbiome, up to
// HX_BDYN), every per-biome loop runs to it, the per-biome arrays are sized for HX_BDYN.  The
// reference creates any number of biomes (simpleNbox.cpp:864-1124); 1-4 have unrolled kernels.
constexpr int HX_DYN = 0;
// Template tag of the one-biome kernel built for TWO resident wavefronts per SIMD (hx_run_kernel
// <HX_B1W2, ...>: at most 256 registers and 20 KB of LDS a wavefront).  Ensembles of more
// wavefronts than the GPU has SIMDs take it: the second wavefront fills the issue slots that a
// dependent fp64 chain leaves empty.  Same model code as B = 1; what differs
// is where a member's values live between their uses:
//  * the LDS park holds 11 year-level slots and f_frozen; its other 28 slots take what only the
//    carbonate solve of a stash needs (both boxes' constants, surface-box carbon, [H+], pCO2)
//    while the dopri5 step loop runs (w2_park_out / w2_park_in): 56 registers;
//  * constants (DOECLIM matrices, ocean exchange, biome parameters, aerosol scalings, the
//    alkalinities) are read from the parameter / derived / state tables where they are used, the
//    block's SSTs from the output array, and nothing is requested a phase ahead (what waits in
//    registers through the solver is what the second wavefront is there to hide instead).
constexpr int HX_B1W2 = 101;
template <int B> constexpr bool hx_w2() { return B == HX_B1W2; }
// compile-time biome count of the unrolled kernels (0: the looped ones)
template <int B> constexpr int hx_nbc() { return B == HX_B1W2 ? 1 : B; }
template <int B> constexpr bool hx_one() { return hx_nbc<B>() == 1; }
// unroll factor of the per-biome loops: the biome count, or (looped kernels, runtime count) four --
// rolled, every iteration waits out its own LDS / table latencies; in fours the loads of a chunk
// are in flight together
#ifndef HX_DYN_UNROLL
#define HX_DYN_UNROLL 4
#endif
template <int B> constexpr int hx_ur() { return B == HX_DYN ? HX_DYN_UNROLL : hx_nbc<B>(); }
template <int B> constexpr int hx_bmax() { return B == HX_DYN ? HX_BDYN : hx_nbc<B>(); }
// Five to eight biomes have unrolled kernels too (round 3: the looped kernel waits out a memory
// latency per biome and loop -- 22.8 ms against 10.9 for four biomes at 8 192 members; unrolled,
// five biomes take 12.1 ms, eight 16.8); like the looped ones they keep the 21 DOECLIM /
// ocean-exchange constants OUT of the park: 14 + 10 B slots, 32 KB and 37 KB for five and six
// biomes, 35 KB and 39 KB for seven and eight with two more values per biome left in HBM
// (hx_slim_park): four wavefronts a CU for all of them.
template <int B> constexpr bool hx_lean_park() { return B == HX_DYN || hx_nbc<B>() > 4 || hx_w2<B>(); }
// The looped kernels size their park for the core's biome count at launch (dynamic LDS) and do
// NOT park the 21 DOECLIM / ocean-exchange constants (read from the derived table where they are
// used: three loads a year against a model year of ~100k cycles): 14 + 10 nb slots -- 33 KB for
// five biomes, 49 KB for eight -- so that up to four wavefronts share a CU where the fixed
// 97.5 KB allowed one.
// Kernels that read a member's constants from the tables where they are used (the lean park):
// every such access is a wave-uniform row address in scalar registers plus the lane's 32-bit byte
// offset (hx_ldm / w2_ld below).  As `table[row * npad + mem]` with a 64-bit per-lane index the
// optimiser computed each row's per-lane ADDRESS once, ahead of the year loop -- ~45 register
// pairs in the eight-biome kernel, kept in AGPRs and scratch; every use then waited for its own
// scratch reload before the load itself could be issued (13 of them in a row in phase C):
// 65 536 members x 7 / 8 biomes 16.9 / 18.9 -> 15.1 / 16.5 ms, no scratch left.  (Five and six
// biomes have the registers for the hoisted addresses and lose 2 % to the scalar form's spill
// lanes: they keep the indexed form; the looped kernels are indifferent.)
template <int B> constexpr bool hx_b78() { return B == 7 || B == 8; }
template <int B> constexpr bool hx_tbl() { return hx_w2<B>() || B == HX_DYN || hx_b78<B>(); }
template <int B> constexpr int pk_ff0() {
  return hx_w2<B>() ? (int)PK_AERO : hx_lean_park<B>() ? (int)PK_D0 : (int)PK_FFROZEN0;
}
// Seven and eight biomes: the two per-biome values that are touched least -- f_frozen (read and
// written once a year; it IS a row of the state table) and f_new_thaw (written once a year, read
// by the three or so flow computations) -- stay in HBM: 14 + 8 B slots = 35 / 39 KB, four
// wavefronts a CU where 42 / 47 KB allowed three (65 536 members then take one round, not two).
template <int B> constexpr bool hx_slim_park() { return B == 7 || B == 8 || B == HX_DYN; }
// The looped kernels (9 to HX_BDYN biomes) go further: only four POOLS of a biome -- vegetation,
// detritus, soil, permafrost: what a stash reads, mixes and writes back -- live in the park; the
// thawed-permafrost pool and tempferts (rows of the state table anyway), co2fert, tempfertd and
// f_new_thaw (written once a year, read by the stashes: three scratch rows per biome) stay in HBM.
// 14 + 4 nb slots = 25 ... 39 KB: four wavefronts a CU up to 16 biomes, 65 536 members in ONE
// round where 53 ... 89 KB took two to four: 9 / 13 / 16 biomes 59.9 / 84.3 / 181 -> 39.7 / 54.9 /
// 64.8 ms.  (Measured alternatives, profiles/r04_variant_log.md: the thawed pool parked too --
// 37.6 / 50.9 ms but two rounds from 14 biomes on, 16 biomes 102 ms; parked or not by a
// wave-uniform branch per access -- 47 / 67 / 76 ms, the branches keep a chunk's loads from being
// in flight together.  An ensemble of one wavefront per CU pays the HBM trips without needing the
// occupancy: 8 192 members x 9 biomes 31.2 -> 34.3 ms.)
template <int B> constexpr bool hx_pools_only_park() { return B == HX_DYN; }
template <int B> constexpr int hx_npark_arr() { return hx_slim_park<B>() ? HX_NBIOME_ARR - 1 : HX_NBIOME_ARR; }
template <int B> constexpr int hx_nff() { return hx_slim_park<B>() ? 0 : hx_bmax<B>(); }  // parked f_frozen slots
// two-wavefront flavour: what the step loop does not touch waits in these slots (w2_park_out)
// (27 slots behind f_frozen, and PK_EOS: the two-wavefront flavour reads end_of_spinup_vegc from
// its state row once a year)
constexpr int PK_W2_0 = PK_AERO + 1, PK_W2_N = 28;
constexpr int w2_slot(int i) { return i == PK_W2_N - 1 ? (int)PK_EOS : PK_W2_0 + i; }
template <int B> constexpr int hx_npark() {
  if (hx_w2<B>()) return PK_W2_0 + PK_W2_N - 1;   // 40 slots = 20 KB: eight wavefronts a CU
  return pk_ff0<B>() + hx_nff<B>() + (B == 1 ? (int)PKB_N : hx_npark_arr<B>() * hx_bmax<B>());
}
__host__ __device__ inline int hx_npark_dyn(int nb) { return PK_D0 + 4 * nb; }
// rows of HxBuffers::bscratch a core of nb biomes needs
__host__ __device__ inline int hx_bscratch_rows(int nb) { return 3 * nb; }
template <int B> constexpr int hx_pkb1() { return pk_ff0<B>() + hx_nff<B>(); }

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
template <> struct BiomeArr<HX_B1W2> { using type = RegArr1; };
// a per-biome array in HBM: element b of this lane in row row0 + b * rstride of a [rows][npad]
// table (wave-uniform row address + the lane's byte offset, see hx_tbl)
struct TblRef {
  char HX_GLOBAL *row;
  unsigned moff;
  __device__ __forceinline__ operator double() const { return *(hx_gcd)(row + moff); }
  __device__ __forceinline__ void operator=(double v) const { *(hx_gd)(row + moff) = v; }
};
struct GlobArr {
  double *tbl;
  int row0, rstride;
  int npad;   // (a copy of the member's: hx_tbl_local refreshes it with the region's)
  unsigned moff;
  __device__ __forceinline__ TblRef operator[](int b) const {
    return TblRef{(char HX_GLOBAL *)HX_GD(tbl) + (size_t)((unsigned)(row0 + b * rstride) * ((unsigned)npad * 8u)), moff};
  }
};
template <int B> struct BiomeArrThaw { using type = typename BiomeArr<B>::type; };
template <> struct BiomeArrThaw<7> { using type = GlobArr; };
template <> struct BiomeArrThaw<8> { using type = GlobArr; };
template <> struct BiomeArrThaw<HX_DYN> { using type = GlobArr; };
// tempferts, co2fert, tempfertd: the park, or (looped kernels) HBM rows
template <int B> struct BiomeArrYear { using type = typename BiomeArr<B>::type; };
template <> struct BiomeArrYear<HX_DYN> { using type = GlobArr; };


// What stays in registers through the carbon-cycle solver of one year.
template <int B>
struct Member {
  double C0;
  // state
  double cHL, cLL, cIO, cDO, atmos, earth;
  typename BiomeArr<B>::type veg, det, soil, pf;
  typename BiomeArrYear<B>::type thawed, tempferts;
  double cum_luc_va, cum_pf_ch4, masstot;
  double max_ts, lastflux_ann, sdt;
  int ts_timeout;
  double alkH, alkL, hH, hL;
  unsigned status;
  // per-year
  typename BiomeArrYear<B>::type co2fert, tempfertd;
  typename BiomeArrThaw<B>::type f_new_thaw;
  GlobArr ffz;   // slim park: f_frozen of this lane in the state table
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
  unsigned moff;     // two-wavefront flavour: this lane's byte offset within a table row (8 mem)
  double (*pk)[64];  // LDS park
  int lane;
  hx_ccd upar;            // multi-biome kernels: the uniform-row table, or null if LandK rows vary
  const HxBuffers *bufp;  // the core's tables (bind_member)
  unsigned long long omk; // extended kernels: HxBuffers::out_mask0 (wave-uniform, read once a launch)
  int iy;                 // year index being integrated
  int trk_iy;             // first tracked year index (tracking kernels)
  int nb;                 // biome count (looped kernels)
  double *spin_row;       // spinup kernel: this lane's column of the current step's record, or null
  double (*trk_rec)[64];  // CON == 3: hand-over slots to the tracking companion wavefront
  int *trk_cmd;
};
// trip count of the per-biome loops
template <int B> __device__ __forceinline__ int nbio(const Member<B> &m) {
  if constexpr (B == HX_DYN) return m.nb; else return hx_nbc<B>();
}
#define PKM(m, slot) ((m).pk[(slot)][(m).lane])
// f_frozen of biome b: a park slot, or (slim park) the state table's row itself
template <int B> __device__ __forceinline__ double ffrozen_of(const Member<B> &m, int b) {
  if constexpr (hx_slim_park<B>()) return m.ffz[b];
  else return PKM(m, pk_ff0<B>() + b);
}
template <int B> __device__ __forceinline__ void set_ffrozen(const Member<B> &m, int b, double v) {
  if constexpr (hx_slim_park<B>()) m.ffz[b] = v;
  else PKM(m, pk_ff0<B>() + b) = v;
}
// a member's DOECLIM / ocean-exchange constant (row HXD_A0.. / HXD_KLH.. of the derived table):
// from the park, or -- looped kernels -- from the table itself
// (two-wavefront flavour) Row offsets row * npad * 8 are scalar products of a launch constant;
// left alone the optimiser computes all ~40 of them ahead of the year loop and the register
// allocator then keeps them in the lanes of a spill VGPR (a v_readlane where each is used: a
// vector-ALU slot).  Passing npad through an empty asm at the head of a region makes the products
// that region's own: one s_mul_i32 each, on the scalar unit, next to the other wavefront's
// vector instructions.  (Only where the wavefront's control flow is uniform.)
#ifndef HX_HOST_EMULATION
#define HX_W2_LOCAL(m) asm volatile("" : "+s"((m).npad))
#else
#define HX_W2_LOCAL(m)
#endif
// (two-wavefront flavour) element of row `row` of a [rows][npad] table for this lane: a
// wave-uniform row address plus the lane's 32-bit byte offset -- the form global_load takes an
// SGPR base and one VGPR for, no 64-bit vector address arithmetic.  Tables of up to 4 GB.
// the same for a row whose (wave-uniform, 64-bit) address the caller has: any table size
// S: the two-wavefront flavour, whose limit is the vector unit's issue slots -- there the scalar
// form pays (131 072 members 10.29 -> 10.21 ms, 262 144 19.59 -> 19.24); one wavefront per SIMD
// waits for the five dependent scalar instructions of a row address instead (9 / 16 biomes
// 39.7 / 64.3 -> 45.9 / 85.2 ms, four biomes 10.06 -> 10.19): those keep the optimiser's choice.
template <bool S = false>
__device__ __forceinline__ double hx_ldm(hx_gcd row, unsigned moff) {
  if constexpr (S) { HX_SROW(row); HX_VOFF(moff); }
  return *(hx_gcd)((const char HX_GLOBAL *)row + moff);
}
template <bool S = false>
__device__ __forceinline__ void hx_stm(hx_gd row, unsigned moff, double v) {
  if constexpr (S) { HX_SROW(row); HX_VOFF(moff); }
  *(hx_gd)((char HX_GLOBAL *)row + moff) = v;
}
template <bool S = false>
__device__ __forceinline__ double w2_ld(const double *tbl, int npad, int row, unsigned moff) {
  const char HX_GLOBAL *r = (const char HX_GLOBAL *)HX_GCD(tbl) + (size_t)((unsigned)row * ((unsigned)npad * 8u));
  if constexpr (S) HX_VOFF(moff);
  return *(hx_gcd)(r + moff);
}
template <bool S = false>
__device__ __forceinline__ void w2_st(double *tbl, int npad, int row, unsigned moff, double v) {
  char HX_GLOBAL *r = (char HX_GLOBAL *)HX_GD(tbl) + (size_t)((unsigned)row * ((unsigned)npad * 8u));
  if constexpr (S) HX_VOFF(moff);
  *(hx_gd)(r + moff) = v;
}
template <int B> __device__ __forceinline__ double dconst(const Member<B> &m, int row) {
  if constexpr (hx_tbl<B>()) return w2_ld<hx_w2<B>()>(m.bufp->derived, m.npad, row, m.moff);
  else if constexpr (hx_lean_park<B>()) return m.der[(size_t)row * m.npad];
  else return PKM(m, row >= HXD_KLH && row < HXD_KLH + 7 ? PK_K0 + (row - HXD_KLH) : PK_D0 + (row - HXD_A0));
}

// head of a region with uniform control flow in a lean-park kernel: the row products are this
// region's own (HX_W2_LOCAL)
template <int B> __device__ __forceinline__ void hx_tbl_local(Member<B> &m) {
  HX_W2_LOCAL(m);
  if constexpr (hx_slim_park<B>()) { m.ffz.npad = m.npad; m.f_new_thaw.npad = m.npad; }
  if constexpr (hx_pools_only_park<B>()) {
    m.thawed.npad = m.npad; m.tempferts.npad = m.npad; m.co2fert.npad = m.npad; m.tempfertd.npad = m.npad;
  }
}

// biome constants of the land model, fetched where they are used
template <int B>
struct LandK {
  static constexpr int N = hx_nbc<B>();
  double npp0[N], f_nppv[N], f_nppd[N], f_litterd[N], rh_ch4_frac[N], fpf_static[N];
};
// looped kernels: a column of the parameter table read where it is used (scalar load from the
// uniform table when every member shares the biome constants, else the member's row)
struct ParamCol {
  hx_ccd upar;
  const double *par;   // the parameter table
  int npad;
  unsigned moff;
  int col;
  __device__ __forceinline__ double operator[](int b) const {
    const int row = HXP_NGLOBAL + b * HXPB_N + col;
    return upar ? upar[row] : w2_ld(par, npad, row, moff);
  }
};
template <>
struct LandK<HX_DYN> {
  ParamCol npp0, f_nppv, f_nppd, f_litterd, rh_ch4_frac, fpf_static;
};
template <int B>
__device__ __forceinline__ void load_landk(const Member<B> &m, LandK<B> &k) {
  HX_FENCE();
  if constexpr (B == HX_DYN) {
    const ParamCol c{m.upar, m.bufp->params, m.npad, m.moff, 0};
    k.npp0 = c; k.npp0.col = HXPB_NPP0; k.f_nppv = c; k.f_nppv.col = HXPB_F_NPPV;
    k.f_nppd = c; k.f_nppd.col = HXPB_F_NPPD; k.f_litterd = c; k.f_litterd.col = HXPB_F_LITTERD;
    k.rh_ch4_frac = c; k.rh_ch4_frac.col = HXPB_RH_CH4_FRAC;
    k.fpf_static = c; k.fpf_static.col = HXPB_FPF_STATIC;
  } else if constexpr (hx_w2<B>()) {
    // the biome's constants where they live: scalar loads when every member shares them (the
    // usual case: ensembles perturb Q10, beta, warming factors), else the member's rows
    const int r = HXP_NGLOBAL;
    if (m.upar) {
      hx_ccd u = m.upar + r;
      k.npp0[0] = u[HXPB_NPP0]; k.f_nppv[0] = u[HXPB_F_NPPV]; k.f_nppd[0] = u[HXPB_F_NPPD];
      k.f_litterd[0] = u[HXPB_F_LITTERD]; k.rh_ch4_frac[0] = u[HXPB_RH_CH4_FRAC];
      k.fpf_static[0] = u[HXPB_FPF_STATIC];
    } else {
      const double *p = m.bufp->params;
      k.npp0[0] = w2_ld<true>(p, m.npad, r + HXPB_NPP0, m.moff);
      k.f_nppv[0] = w2_ld<true>(p, m.npad, r + HXPB_F_NPPV, m.moff);
      k.f_nppd[0] = w2_ld<true>(p, m.npad, r + HXPB_F_NPPD, m.moff);
      k.f_litterd[0] = w2_ld<true>(p, m.npad, r + HXPB_F_LITTERD, m.moff);
      k.rh_ch4_frac[0] = w2_ld<true>(p, m.npad, r + HXPB_RH_CH4_FRAC, m.moff);
      k.fpf_static[0] = w2_ld<true>(p, m.npad, r + HXPB_FPF_STATIC, m.moff);
    }
  } else if constexpr (B == 1) {
    constexpr int o = hx_pkb1<B>();
    k.npp0[0] = PKM(m, o + PKB_NPP0); k.f_nppv[0] = PKM(m, o + PKB_F_NPPV);
    k.f_nppd[0] = PKM(m, o + PKB_F_NPPD); k.f_litterd[0] = PKM(m, o + PKB_F_LITTERD);
    k.rh_ch4_frac[0] = PKM(m, o + PKB_RH_CH4_FRAC); k.fpf_static[0] = PKM(m, o + PKB_FPF_STATIC);
  } else {
    if (m.upar) {
      // every member has the same biome constants (the usual case: ensembles perturb Q10, beta,
      // warming factors): wave-uniform scalar loads instead of 6 B vector loads from HBM
#pragma unroll
      for (int b = 0; b < hx_nbc<B>(); ++b) {
        hx_ccd r = m.upar + (HXP_NGLOBAL + b * HXPB_N);
        k.npp0[b] = r[HXPB_NPP0]; k.f_nppv[b] = r[HXPB_F_NPPV]; k.f_nppd[b] = r[HXPB_F_NPPD];
        k.f_litterd[b] = r[HXPB_F_LITTERD]; k.rh_ch4_frac[b] = r[HXPB_RH_CH4_FRAC];
        k.fpf_static[b] = r[HXPB_FPF_STATIC];
      }
      return;
    }
#pragma unroll
    for (int b = 0; b < hx_nbc<B>(); ++b) {
      const int r0 = HXP_NGLOBAL + b * HXPB_N;
      if constexpr (hx_tbl<B>()) {
        const double *p = m.bufp->params;
        k.npp0[b] = w2_ld(p, m.npad, r0 + HXPB_NPP0, m.moff);
        k.f_nppv[b] = w2_ld(p, m.npad, r0 + HXPB_F_NPPV, m.moff);
        k.f_nppd[b] = w2_ld(p, m.npad, r0 + HXPB_F_NPPD, m.moff);
        k.f_litterd[b] = w2_ld(p, m.npad, r0 + HXPB_F_LITTERD, m.moff);
        k.rh_ch4_frac[b] = w2_ld(p, m.npad, r0 + HXPB_RH_CH4_FRAC, m.moff);
        k.fpf_static[b] = w2_ld(p, m.npad, r0 + HXPB_FPF_STATIC, m.moff);
      } else {
      hx_gcd r = m.par + (size_t)r0 * m.npad;
      k.npp0[b] = r[(size_t)HXPB_NPP0 * m.npad];
      k.f_nppv[b] = r[(size_t)HXPB_F_NPPV * m.npad];
      k.f_nppd[b] = r[(size_t)HXPB_F_NPPD * m.npad];
      k.f_litterd[b] = r[(size_t)HXPB_F_LITTERD * m.npad];
      k.rh_ch4_frac[b] = r[(size_t)HXPB_RH_CH4_FRAC * m.npad];
      k.fpf_static[b] = r[(size_t)HXPB_FPF_STATIC * m.npad];
      }
    }
  }
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
  m.bufp = &buf;
  m.par = HX_GCD(buf.params) + mem;
  m.der = HX_GCD(buf.derived) + mem;
  m.npad = buf.npad;
  m.moff = (unsigned)mem * 8u;
  m.pk = park;
  m.lane = lane;
  m.upar = (B != 1 && buf.uni_landk) ? HX_CCD(buf.uparams) : nullptr;
  m.nb = buf.nbiome;
  if constexpr (!hx_one<B>()) {
    const int bm = (B == HX_DYN) ? buf.nbiome : hx_bmax<B>();   // slots per array
    const int o = pk_ff0<B>() + (hx_slim_park<B>() ? 0 : bm);
    if constexpr (hx_pools_only_park<B>()) {
      ParkArr *arr[4] = {&m.veg, &m.det, &m.soil, &m.pf};
#pragma unroll
      for (int k = 0; k < 4; ++k) { arr[k]->base = park + o + k * bm; arr[k]->lane = lane; }
      m.thawed = GlobArr{buf.state, HXS_NGLOBAL + HXSB_THAWED, HXSB_N, m.npad, m.moff};
      m.tempferts = GlobArr{buf.state, HXS_NGLOBAL + HXSB_TEMPFERTS, HXSB_N, m.npad, m.moff};
      m.co2fert = GlobArr{buf.bscratch, 0, 1, m.npad, m.moff};
      m.tempfertd = GlobArr{buf.bscratch, bm, 1, m.npad, m.moff};
      m.f_new_thaw = GlobArr{buf.bscratch, 2 * bm, 1, m.npad, m.moff};
      m.ffz = GlobArr{buf.state, HXS_NGLOBAL + HXSB_F_FROZEN, HXSB_N, m.npad, m.moff};
    } else if constexpr (hx_slim_park<B>()) {
      ParkArr *arr[HX_NBIOME_ARR - 1] = {&m.veg, &m.det, &m.soil, &m.pf, &m.thawed, &m.tempferts,
                                         &m.co2fert, &m.tempfertd};
#pragma unroll
      for (int k = 0; k < HX_NBIOME_ARR - 1; ++k) { arr[k]->base = park + o + k * bm; arr[k]->lane = lane; }
      m.f_new_thaw = GlobArr{buf.bscratch, 0, 1, m.npad, m.moff};
      m.ffz = GlobArr{buf.state, HXS_NGLOBAL + HXSB_F_FROZEN, HXSB_N, m.npad, m.moff};
    } else {
    ParkArr *arr[HX_NBIOME_ARR] = {&m.veg, &m.det, &m.soil, &m.pf, &m.thawed, &m.tempferts,
                                   &m.co2fert, &m.tempfertd, &m.f_new_thaw};
#pragma unroll
    for (int k = 0; k < HX_NBIOME_ARR; ++k) {
      arr[k]->base = park + o + k * bm;
      arr[k]->lane = lane;
    }
    }
  }
  m.C0 = ldp(buf, HXP_C0, mem);
  // constants -> park
  if constexpr (!hx_w2<B>()) {
  PKM(m, PK_AERO) = ldp(buf, HXP_AERO, mem);
  PKM(m, PK_VOL) = ldp(buf, HXP_VOL, mem);
  }
  if constexpr (!hx_lean_park<B>()) {
#pragma unroll
    for (int k = 0; k < 14; ++k) PKM(m, PK_D0 + k) = ldd(buf, HXD_A0 + k, mem);
#pragma unroll
    for (int k = 0; k < 7; ++k) PKM(m, PK_K0 + k) = ldd(buf, HXD_KLH + k, mem);
  }
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
// The parked chemistry (HX_B1W2): while the dopri5 step loop runs, what only a stash touches --
// both surface boxes' carbonate constants, their carbon, [H+] and pCO2 -- waits in LDS.
template <int B>
__device__ __forceinline__ void w2_park_out(const Member<B> &m) {
  const ChemK *k[2] = {&m.kH, &m.kL};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int o = 11 * b;
    PKM(m, w2_slot(o + 0)) = k[b]->A2; PKM(m, w2_slot(o + 1)) = k[b]->A1; PKM(m, w2_slot(o + 2)) = k[b]->C4;
    PKM(m, w2_slot(o + 3)) = k[b]->C3; PKM(m, w2_slot(o + 4)) = k[b]->C2; PKM(m, w2_slot(o + 5)) = k[b]->C1;
    PKM(m, w2_slot(o + 6)) = k[b]->C0; PKM(m, w2_slot(o + 7)) = k[b]->K1; PKM(m, w2_slot(o + 8)) = k[b]->K1K2;
    PKM(m, w2_slot(o + 9)) = k[b]->rKh; PKM(m, w2_slot(o + 10)) = k[b]->g;
  }
  PKM(m, w2_slot(22)) = m.cHL; PKM(m, w2_slot(23)) = m.cLL; PKM(m, w2_slot(24)) = m.hH;
  PKM(m, w2_slot(25)) = m.hL; PKM(m, w2_slot(26)) = m.pco2H; PKM(m, w2_slot(27)) = m.pco2L;
  HX_FENCE();
}
template <int B>
__device__ __forceinline__ void w2_park_in(Member<B> &m) {
  HX_FENCE();
  ChemK *k[2] = {&m.kH, &m.kL};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int o = 11 * b;
    k[b]->A2 = PKM(m, w2_slot(o + 0)); k[b]->A1 = PKM(m, w2_slot(o + 1)); k[b]->C4 = PKM(m, w2_slot(o + 2));
    k[b]->C3 = PKM(m, w2_slot(o + 3)); k[b]->C2 = PKM(m, w2_slot(o + 4)); k[b]->C1 = PKM(m, w2_slot(o + 5));
    k[b]->C0 = PKM(m, w2_slot(o + 6)); k[b]->K1 = PKM(m, w2_slot(o + 7)); k[b]->K1K2 = PKM(m, w2_slot(o + 8));
    k[b]->rKh = PKM(m, w2_slot(o + 9)); k[b]->g = PKM(m, w2_slot(o + 10));
  }
  m.cHL = PKM(m, w2_slot(22)); m.cLL = PKM(m, w2_slot(23)); m.hH = PKM(m, w2_slot(24));
  m.hL = PKM(m, w2_slot(25)); m.pco2H = PKM(m, w2_slot(26)); m.pco2L = PKM(m, w2_slot(27));
}

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
  // (two-wavefront flavour: the alkalinities never change during a run and only the year start
  // needs them: read from their rows there instead of held in registers through the year)
  if constexpr (!hx_w2<B>()) { m.alkH = lds_(buf, HXS_ALK_HL, mem); m.alkL = lds_(buf, HXS_ALK_LL, mem); }
  m.hH = lds_(buf, HXS_H_HL, mem); m.hL = lds_(buf, HXS_H_LL, mem);
#pragma unroll hx_ur<B>()
  for (int b = 0; b < nbio<B>(m); ++b) {
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
  if constexpr (hx_w2<B>()) {   // (not held in registers: the live rows are the values)
    if (base) {
      sts_(buf, HXS_ALK_HL, mem, lds_(buf_, HXS_ALK_HL, mem));
      sts_(buf, HXS_ALK_LL, mem, lds_(buf_, HXS_ALK_LL, mem));
    }
  } else {
  sts_(buf, HXS_ALK_HL, mem, m.alkH); sts_(buf, HXS_ALK_LL, mem, m.alkL);
  }
  sts_(buf, HXS_H_HL, mem, m.hH); sts_(buf, HXS_H_LL, mem, m.hL);
#pragma unroll hx_ur<B>()
  for (int b = 0; b < nbio<B>(m); ++b) {
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
  if constexpr (hx_w2<B>()) {  // (PK_EOS is a parking slot there; the live row is the value)
    if (base) sts_(buf, HXS_EOS_VEGC, mem, lds_(buf_, HXS_EOS_VEGC, mem));
  } else {
  if (base) sts_(buf, HXS_EOS_VEGC, mem, PKM(m, PK_EOS));
  }
  if (hx_slim_park<B>() && !base) return;  // (the live table's rows are where f_frozen lives)
#pragma unroll hx_ur<B>()
  for (int b = 0; b < nbio<B>(m); ++b)
    sts_(buf, HXS_NGLOBAL + b * HXSB_N + HXSB_F_FROZEN, mem, ffrozen_of<B>(m, b));
}

}  // namespace
