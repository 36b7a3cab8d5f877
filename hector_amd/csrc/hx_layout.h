// hx_layout.h -- HBM data layout shared by the host runtime and the HIP kernels.
//
// Everything per-member is structure-of-arrays: row r of a table lives at
// base[r * npad + member], npad = members rounded up to 64, so lane i of a wave
// reads member i of a row with one coalesced 512-byte transaction.
#pragma once
#include <stdint.h>

#define HX_MAXB 4          // biome counts with a templated (fully unrolled) kernel: B = 1..4
#define HX_BDYN 32         // more biomes, up to this many, run the looped kernels (template tag 0):
                           // per-biome arrays in the LDS park, loops over the core's biome count
#define HX_WAVE 64

// ---- per-member parameter rows (read-only during a run) -------------------
enum HxParamRow {
  HXP_S = 0,        // ECS                     temperature_component.cpp:165
  HXP_DIFF,         // ocean heat diffusivity
  HXP_QCO2,
  HXP_AERO,         // aero_scalar (alpha)     forcing_component.cpp:430-470
  HXP_VOL,          // vol_scalar
  HXP_C0,           // preindustrial CO2 ppmv
  HXP_TT, HXP_TU, HXP_TWI, HXP_TID,           // ocean transports m3/s
  HXP_PRE_SURF, HXP_PRE_ID,                   // preindustrial ocean C
  HXP_LO_RATIO,     // land-ocean warming ratio, 0 = off  temperature_component.cpp:722-739
  HXP_NGLOBAL
};
enum HxBiomeParam {  // row = HXP_NGLOBAL + biome * HXPB_N + k
  HXPB_BETA = 0, HXPB_Q10, HXPB_WF, HXPB_NPP0,
  HXPB_VEG0, HXPB_DET0, HXPB_SOIL0, HXPB_PF0,
  HXPB_F_NPPV, HXPB_F_NPPD, HXPB_F_LITTERD,
  HXPB_RH_CH4_FRAC, HXPB_PF_MU, HXPB_PF_SIGMA, HXPB_FPF_STATIC,
  HXPB_N
};
#define HX_NPARAM(B) (HXP_NGLOBAL + (B) * HXPB_N)

// ---- per-member state rows (carried across years / run() calls) -----------
enum HxStateRow {
  HXS_C_HL = 0, HXS_C_LL, HXS_C_IO, HXS_C_DO,   // ocean box carbon, PgC
  HXS_ATMOS, HXS_EARTH,
  HXS_CUM_LUC_VA, HXS_CUM_PF_CH4, HXS_MASSTOT, HXS_EOS_VEGC,
  HXS_MAX_TS, HXS_TS_TIMEOUT, HXS_LASTFLUX_ANN,  // ocean timestep controller
  HXS_SOLVER_DT,
  HXS_CH4, HXS_ALK_HL, HXS_ALK_LL, HXS_H_HL, HXS_H_LL,
  HXS_TLAND, HXS_SST, HXS_F_PREV, HXS_BASE_TOT, HXS_BASE_CO2,
  HXS_TL_M1, HXS_TL_M2, HXS_TWIN,                // Q10 window bookkeeping
  HXS_NGLOBAL
};
enum HxBiomeState {  // row = HXS_NGLOBAL + biome * HXSB_N + k
  HXSB_VEG = 0, HXSB_DET, HXSB_SOIL, HXSB_PF, HXSB_THAWED,
  HXSB_TEMPFERTS, HXSB_F_FROZEN,
  HXSB_N
};
#define HX_NSTATE(B) (HXS_NGLOBAL + (B) * HXSB_N)

// ---- per-member derived constants (computed once per parameter upload) -----
enum HxDerivedRow {
  HXD_A0 = 0, HXD_A1, HXD_A2, HXD_A3,          // DOECLIM predecessor matrix
  HXD_IB0, HXD_IB1, HXD_IB2, HXD_IB3,          // inverse successor matrix
  HXD_QC1, HXD_QC2, HXD_DQ1, HXD_DQ2, HXD_DPSCALE, HXD_HFSCALE,
  HXD_KLH, HXD_KLI, HXD_KHD, HXD_KIL, HXD_KIH, HXD_KID, HXD_KDI,  // ocean exchange, 1/yr
  HXD_FLAG,                                     // status bits found while deriving
  HXD_NGLOBAL                                   // then ln(q10_rh) per biome
};
#define HX_NDERIVED(B) (HXD_NGLOBAL + (B))

// ---- outputs: one [ns][npad] array per variable ---------------------------
enum HxOutVar {
  HXO_SST = 0,      // always on: it is DOECLIM's history
  HXO_TLAND,        // always on: Q10 window history
  HXO_CO2,          // CO2_concentration, ppmv
  HXO_TGAV,         // global_tas
  HXO_RF_TOT, HXO_RF_CO2, HXO_OCEAN_C, HXO_HL_PH, HXO_ATMOS_C,
  HXO_PERMAFROST_C, HXO_HEATFLUX,
  HXO_CH4, HXO_O3, HXO_VEG_C, HXO_DET_C, HXO_SOIL_C, HXO_THAWED_C, HXO_EARTH_C,
  HXO_NBP, HXO_OCEAN_UPTAKE, HXO_NSTASH, HXO_NSTEPS, HXO_LL_PH,
  HXO_SST_LO,       // D_SST as reported when a land-ocean warming ratio is set (HXO_SST stays
                    // DOECLIM's own history); allocated only then
  // diagnostics the reference's output stream writes every year (csv_outputstream_visitor.cpp)
  HXO_NPP, HXO_RH, HXO_RH_DET, HXO_RH_SOIL,    // final_* of the year's last stash
  HXO_HL_UPTAKE, HXO_LL_UPTAKE, HXO_HL_DO,     // accumulated over the year's stashes
  HXO_CA_RESIDUAL,
  HXO_RH_CH4, HXO_F_FROZEN, HXO_GMST, HXO_FLUX_MIXED, HXO_FLUX_INTERIOR,
  HXO_C_HL, HXO_C_LL, HXO_C_IO, HXO_C_DO, HXO_PCO2_HL, HXO_PCO2_LL, HXO_TAU_OH,
  // per-biome "<biome>.veg_c" ...: index HXO_BIOME0 + k * HX_BDYN + biome, k = HxBiomeOut
  HXO_BIOME0,
  HXO_NVAR = HXO_BIOME0 + 11 * HX_BDYN
};

enum HxBiomeOut { HXOB_VEG = 0, HXOB_DET, HXOB_SOIL, HXOB_PF, HXOB_THAWED, HXOB_NPP, HXOB_RH,
                   HXOB_RH_CH4, HXOB_F_FROZEN, HXOB_TEMPFERTD, HXOB_TEMPFERTS, HXOB_N };
#define HXO_B(k, b) (HXO_BIOME0 + (k) * HX_BDYN + (b))
#define HX_OM_BIOME_FLUX 62   // HxBuffers::out_mask0: some "<biome>.NPP" or "<biome>.RH" is recorded
#define HX_OM_BIOME_ANY 61    // ... some "<biome>.<variable>" output at all (= HxBuffers::biome_diag)
static_assert(HXO_BIOME0 <= HX_OM_BIOME_ANY, "HxBuffers::out_mask0 holds one bit per global output");

// ---- shared per-year scenario table: row iy = year - startDate ------------
enum HxSharedCol {
  HXSH_FFI = 0, HXSH_DACCS, HXSH_LUC_E, HXSH_LUC_U,  // of date year-1 (slowparameval t)
  HXSH_OH_B, HXSH_OH_C, HXSH_OH_D,                   // OH lifetime terms of year
  HXSH_CH4_EM, HXSH_CH4N,
  HXSH_O3_NOX, HXSH_O3_CO, HXSH_O3_NMVOC,
  HXSH_N2O, HXSH_SQRT_N2O,
  HXSH_RF_OTHER,   // halocarbons + albedo + misc (member independent)
  HXSH_RF_AERO,    // BC+OC+SO2+NH3+aci for aero_scalar = 1
  HXSH_RF_VOL,     // SV
  // constraints of the year, NaN = none (runs use them only if HxConst::con_mask says so)
  HXSH_CO2_CON, HXSH_NBP_CON, HXSH_TAS_CON, HXSH_FTOT_CON, HXSH_CH4_CON,
  HXSH_NCOL,
  HXSH_STRIDE = 24
};

// ---- optional per-member input series [ns][npad], overriding the shared column ----------
enum HxMemberSeries {
  HXM_FFI = 0, HXM_DACCS, HXM_LUC_E, HXM_LUC_U, HXM_CH4_EM,
  // a constraint series per member (NaN = none for that member and year); the HXC_* bit of
  // HxConst::con_mask is set when the scenario OR some member holds such a constraint
  HXM_CO2_CON, HXM_NBP_CON, HXM_TAS_CON, HXM_FTOT_CON, HXM_CH4_CON,
  // computed on the device (hx_gas_kernel) when N2O / halocarbon parameters differ between
  // members: the N2O concentration and the sum of halocarbon + albedo + misc forcings of every
  // year (row 0 of the N2O series is the member's preindustrial value)
  HXM_N2O, HXM_RF_OTHER,
  HXM_N
};
static_assert(HXM_N <= 32, "HxBuffers::ms_mask holds one bit per member series");

// ---- status bits (per member) ---------------------------------------------
#define HX_ERR_MASS 1u       // mass balance > 1e-3 PgC   simpleNbox-runtime.cpp:553-563
#define HX_ERR_RETRIES 2u    // > 8 solver retries        carbon-cycle-solver.cpp:242-294
#define HX_ERR_NEGPOOL 4u    // negative pool (fluxpool assert)
#define HX_ERR_SPINUP 8u     // not spun up in max_spinup steps
#define HX_ERR_SINGULAR 16u  // DOECLIM 2x2 singular
#define HX_ERR_ROOT 32u      // carbonate root not found
#define HX_ERR_STEPFAIL 64u  // > 500 rejected steps

// constraint kinds present in the scenario (HxConst::con_mask)
#define HXC_CO2 1
#define HXC_NBP 2
#define HXC_TAS 4
#define HXC_FTOT 8
#define HXC_CH4 16
#define HXC_LO 32   // some member has a land-ocean warming ratio

// scenario scalars every lane needs (kernel argument, lives in SGPRs)
struct HxConst {
  int start_year, ns, baseyear_idx, max_spinup, spinup_chem;
  double eps_abs, eps_rel, dt0, eps_spinup;
  double M0, lnM0, Tsoil, Tstrat, UC_CH4, TOH0, CCH4;  // M0: the INI value (OH component)
  double M0f, sqrtM0;  // preindustrial CH4 as the forcing sees it (CH4 constraint at startDate)
  double inv_UC_CH4, inv_Tsoil, inv_Tstrat, inv_h2o_span;  // reciprocals of uniform divisors
  int con_mask;        // HXC_* bits: which constraint columns hold values
  int trk_iy;          // year index of Core::trackingDate, -1 = no tracking
  double N0, sqrtN0;
  double delta_co2, delta_ch4, delta_n2o;
  double o3_rf;        // 0.042 W/m2 per DU, or 0 when the ozone component is disabled
  // dopri5 tableau in the order a step uses it (hx_fill_tableau), for builds that read it as data
  double tab[32];
  // coefficients of the year loop's exp / log batches, the same way (hx_fill_math_table)
  double mtab[32];
  // ... and the constants of the equilibrium-constant formulas (hx_fill_chem_table, hx_dev_chem.h)
  double ctab[40];
  // The surface boxes' six T-only equilibrium constants as polynomials in the box temperature
  // (hx_chem_fit.inc, generated by tools/make_chem_fit.py; chem_constants_fit in hx_dev_chem.h):
  // [power, descending][box * 6 + {K0, Kw, 1/Kh, K1, K2, Kb}]
  double kfit[14 * 12];
};
#include "hx_chem_fit.inc"
static_assert(HX_CHEM_FIT_DEGREE == 13, "HxConst::kfit is sized for degree 13");
inline void hx_fill_chem_fit(double *t) {
  for (int i = 0; i < (HX_CHEM_FIT_DEGREE + 1) * 12; ++i) t[i] = hx_chem_fit_table[i];
}
// [0..11] exp Taylor 1/13! .. 1/2!, [12] log2 e, [13] -ln2_hi, [14] -ln2_lo; [16..22] Lg1..Lg7,
// [23] ln2_hi, [24] ln2_lo, [25] sqrt(1/2)   (hx_dev_math.h)
inline void hx_fill_math_table(double *t) {
  const double v[26] = {1.6059043836821613e-10, 2.0876756987868099e-09, 2.5052108385441719e-08,
                        2.7557319223985893e-07, 2.7557319223985888e-06, 2.4801587301587302e-05,
                        1.9841269841269841e-04, 1.3888888888888889e-03, 8.3333333333333332e-03,
                        4.1666666666666664e-02, 1.6666666666666666e-01, 0.5,
                        1.4426950408889634074, -6.93147180369123816490e-01, -1.90821492927058770002e-10, 0.0,
                        6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
                        2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
                        1.479819860511658591e-01, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
                        0.70710678118654752440};
  for (int i = 0; i < 26; ++i) t[i] = v[i];
  for (int i = 26; i < 32; ++i) t[i] = 0.0;
}
// stage 2: b21, 1/5 | 3: b31 b32 3/10 | 4: b41 b42 b43 4/5 | 5: b51..b54 8/9 | 6: b61..b65 |
// candidate: c1 c3 c4 c5 c6 | error: dc1 dc3 dc4 dc5 dc6 dc7
inline void hx_fill_tableau(double *t) {
  const double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
  const double v[30] = {1.0 / 5, 1.0 / 5,
                        3.0 / 40, 9.0 / 40, 3.0 / 10,
                        44.0 / 45, -56.0 / 15, 32.0 / 9, 4.0 / 5,
                        19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 8.0 / 9,
                        9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656,
                        c1, c3, c4, c5, c6,
                        c1 - 5179.0 / 57600, c3 - 7571.0 / 16695, c4 - 393.0 / 640,
                        c5 - (-92097.0 / 339200), c6 - 187.0 / 2100, -1.0 / 40};
  for (int i = 0; i < 30; ++i) t[i] = v[i];
  t[30] = t[31] = 0.0;
}

// pointers handed to the kernels
struct HxBuffers {
  const double *params;  // [HX_NPARAM(B)][npad]
  double *state;         // [HX_NSTATE(B)][npad]
  unsigned *status;      // [npad]
  const double *derived; // [HX_NDERIVED(B)][npad]
  const double *shared;  // [ns][HXSH_STRIDE]
  const double *ker;     // zero-padded DOECLIM kernel (32 in front, 64 behind): [ns+96] (shared diffusivity) or [ns+96][npad]
  const double *dpart;   // [HX_DBLK][npad] history partial sums of the current block
  const double *dpart2;  // same for the heat-flux diagnostic
  double *hist;          // optional per-year state history [ns][HX_NSTATE(B)][npad] for reset(date)
  unsigned *hist_status; // ... and the members' status bits of every year [ns][npad]
  int n, npad, ker_per_member;
  int nbiome;              // the core's biome count (the looped kernels' trip count)
  const double *mseries[HXM_N];  // per-member series (row iy as in the shared table) or nullptr
  const double *uparams;   // [HX_NPARAM(B)] one value per parameter row (member 0): rows that are
  int uni_landk, uni_bio;  // uniform over members are read through scalar loads (multi-biome kernels)
  // two-wavefront flavour (HX_B1W2): lane 0's derived constants [HX_NDERIVED(B)], and which groups
  // of rows every member shares -- the ocean exchange coefficients (HXD_KLH..HXD_KDI), aerosol /
  // volcanic scaling and C0, the biomes' warming factors
  const double *uderived;
  int uni_k, uni_avc, uni_wf;
  // carbon tracking (CON == 2 kernels): the yearly record of every member's origin matrix; the
  // current year's matrix is updated in place by every stash (hx_dev_track.h)
  // both tiled by wavefront, [npad/64][trk_slots][rows][64]: slot 0 the identity of the tracking
  // date, slot 1 + k tracked year k
  double *track_out_f;   // rows: TP*TP fractions
  double *track_out_v;   // rows: TP pool values, then W mask words per pool (bit patterns)
  int trk_slots;         // ns - trk_iy + 1
  int out_rare;            // some output besides sst, land_tas, CO2_concentration, global_tas is
                           // recorded: the year's output block tests the others only then
  int biome_diag;          // some "<biome>.<variable>" output is recorded
  int stash_diag;        // some of HXO_NPP..HXO_CA_RESIDUAL are recorded (written inside the stash)
  // The same information bit by bit: bit v of out_mask0 = out[v] is recorded (v < HXO_BIOME0; bit
  // HX_OM_BIOME_FLUX = some "<biome>.NPP" / "<biome>.RH"), bit k of ms_mask = mseries[k] is set.
  // The extended kernels test a bit of one scalar register where they would load and test a
  // pointer -- ~10 dependent scalar loads + branches per stash, ~45 a year otherwise.
  unsigned long long out_mask0;
  unsigned ms_mask;
  double *cost;          // [npad] what the launches since the last reset cost each lane (4 per dopri5
                         // step + 5 per stash): the host orders the lanes by it (EnsembleCore::assign_lanes)
  // optional (hx_enable_spinup_record): what the reference's output stream sees after every spinup
  // step -- its "spinup = 1" rows (csv_outputstream_visitor.cpp:86-95) -- [max_spinup][HXSR_N][npad]
  double *spin_rec;
  // per-biome values of the year that wait in HBM instead of the park (hx_dev_member.h): f_new_thaw
  // of every biome for the seven- and eight-biome kernels; co2fert, tempfertd and f_new_thaw for the
  // looped ones -- [3 nbiome][npad]
  double *bscratch;
  // [2 * npad / 64][2] start and end of every wavefront of the last year-loop launch, in ticks of
  // the constant 100 MHz clock (s_memrealtime): what the launch's tail looks like -- the launch
  // lasts as long as its last wavefront (hx_wave_clock)
  long long *wave_clk;
  // (last: 396 pointers, see HxArgs)
  double *out[HXO_NVAR]; // each [ns][npad] or nullptr
};
// rows of HxBuffers::spin_rec: the carbon-cycle variables of the stream, which are the ones that
// move during the spinup
enum { HXSR_NBP = 0, HXSR_NPP, HXSR_RH, HXSR_RH_DET, HXSR_RH_SOIL, HXSR_ATMOS_C, HXSR_CA_RESIDUAL,
       HXSR_VEG_C, HXSR_DET_C, HXSR_SOIL_C, HXSR_PERMAFROST_C, HXSR_THAWED_C, HXSR_EARTH_C,
       HXSR_HL_UPTAKE, HXSR_LL_UPTAKE, HXSR_C_DO, HXSR_C_HL, HXSR_C_IO, HXSR_C_LL, HXSR_HL_DO,
       HXSR_OCEAN_UPTAKE, HXSR_N };

// ---- diagnostics derived on the device from recorded outputs (hx_diag_kernel) ----
enum HxDiagKind {
  HXG_TEMP = 0, HXG_DIC, HXG_CO3, HXG_OMEGA_AR, HXG_OMEGA_CA, HXG_REVELLE, HXG_OCEAN_TAS,
  HXG_RF_N2O, HXG_RF_CH4, HXG_RF_H2O, HXG_RF_O3
};

struct HxDiagArgs {
  const double *sst, *ph, *pco2, *carbon;  // recorded outputs [ns][npad] (null if unused)
  const double *co2, *ch4, *o3, *tgav;
  const double *lo_ratio;                  // parameter row (ocean_tas)
  const double *shared;                    // per-year table
  const double *n2o_members;               // per-member N2O series [ns][npad] or null (RF_N2O)
  double deltaT, inv_vol;                  // of the box
  double sqrtN0, sqrtM0, M0f, delta_n2o, delta_ch4;
  int base_idx, npad, iy0, ny;
};


// kernel arguments live in device memory (one copy per core) and are read through
// wave-uniform scalar loads on demand -- passing them by value would pin ~110 SGPRs
// (kc FIRST, and HxBuffers' big pointer tables last: a scalar load off a base the optimiser
//  cannot see through -- the tableau, the batches' coefficient tables, the polynomial fits, all
//  read through laundered offsets -- folds its constant offset into the instruction only below
//  4 KB; with the per-biome output pointers of 32 biomes in front of them these tables sat beyond
//  that, every such load got a 64-bit scalar add of its own and the plain kernel 70 more spilled
//  scalars: 65 536 members 5.90 -> 6.04 ms.  In this order it has fewer of both than with 16.)
struct HxArgs {
  HxConst kc;
  HxBuffers buf;
};
