// ensemble_core.cpp -- see ensemble_core.hpp.
#include "ensemble_core.hpp"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <fstream>
#include <sstream>
#include <string>
#include <stdexcept>
#include <map>
#include <mutex>

// launchers defined in hx_kernels.hip
hipError_t hx_launch_spinup(int B, const HxArgs *d_args, int nmem_launch, int *d_steps,
                            hipStream_t st);
int hx_track_value_rows(int B);
int hx_pair_available();
hipError_t hx_launch_run_pair(const HxArgs *d_args, int npad, bool heatflux, bool kpm, int iy_from,
                              int iy_to, hipStream_t st, bool cons, int nbiome);
hipError_t hx_launch_prewarm(const int *d_stop, long long max_ticks, double *d_sink, int waves,
                             const double *mem, unsigned long long n_mem, hipStream_t st);
hipError_t hx_launch_run(int B, const HxArgs *d_args, int npad, bool heatflux, bool kpm, int con,
                         int iy_from, int iy_to, hipStream_t st, bool two_wave, int cus);
int hx_doeclim_block_years();
void hx_fill_chem_table_host(double *t);
hipError_t hx_launch_broadcast(double *table, int nrows, int npad, hipStream_t st);
hipError_t hx_launch_broadcast_u32(unsigned *v, int npad, hipStream_t st);
hipError_t hx_launch_alk(const HxArgs *d_args, int nmem_launch, hipStream_t st);
hipError_t hx_launch_or_flags(unsigned *status, const double *flag_row, int npad, hipStream_t st);
hipError_t hx_launch_gas(const double *par, const double *ser, const double *h0, int nh, int ns,
                         int npad, double *n2o_out, double *rf_other_out, hipStream_t st);
hipError_t hx_launch_diag(int kind, const HxDiagArgs &a, double *out, hipStream_t st);
hipError_t hx_launch_slr(const double *tgav, int npad, int start_year, int iy_to, double *out,
                         size_t var_stride, hipStream_t st);
hipError_t hx_launch_gather(const double *src, const int *lane_of_member, double *dst, int n,
                            int npad, int nyears, hipStream_t st);
hipError_t hx_launch_stats(const double *var, int n, int npad, int iy0, int nyears,
                           double *stats, hipStream_t st);
hipError_t hx_launch_derive(const double *params, double *derived, const double *ker,
                            int ker_per_member, int ns, int nbiome, int npad, hipStream_t st);
hipError_t hx_launch_doeclim_kernel(const double *diff_row, double *ker, int ns, int count,
                                    int stride, hipStream_t st);

namespace hx {
namespace {

struct ParamDef {
  const char *name; int row; bool per_biome; const char *units; bool spinup;
};
// capability strings: inst/include/component_data.hpp; units: src/unitval.cpp
const ParamDef kParams[] = {
    {"S", HXP_S, false, "degC", false},
    {"diff", HXP_DIFF, false, "cm2/s", false},
    {"qco2", HXP_QCO2, false, "W/m2", false},
    {"aero_scalar", HXP_AERO, false, "(unitless)", false},
    {"vol_scalar", HXP_VOL, false, "(unitless)", false},
    {"C0", HXP_C0, false, "ppmv CO2", true},
    {"tt", HXP_TT, false, "m3/s", true},
    {"tu", HXP_TU, false, "m3/s", true},
    {"twi", HXP_TWI, false, "m3/s", true},
    {"tid", HXP_TID, false, "m3/s", true},
    {"preind_surface_c", HXP_PRE_SURF, false, "Pg C", true},
    {"preind_interdeep_c", HXP_PRE_ID, false, "Pg C", true},
    {"lo_warming_ratio", HXP_LO_RATIO, false, "(unitless)", false},
    {"beta", HXPB_BETA, true, "(unitless)", false},
    {"q10_rh", HXPB_Q10, true, "(unitless)", false},
    {"warmingfactor", HXPB_WF, true, "(unitless)", false},
    {"npp_flux0", HXPB_NPP0, true, "Pg C/yr", true},
    {"veg_c", HXPB_VEG0, true, "Pg C", true},
    {"detritus_c", HXPB_DET0, true, "Pg C", true},
    {"soil_c", HXPB_SOIL0, true, "Pg C", true},
    {"permafrost_c", HXPB_PF0, true, "Pg C", true},
    {"f_nppv", HXPB_F_NPPV, true, "(unitless)", true},
    {"f_nppd", HXPB_F_NPPD, true, "(unitless)", true},
    {"f_litterd", HXPB_F_LITTERD, true, "(unitless)", true},
    {"rh_ch4_frac", HXPB_RH_CH4_FRAC, true, "(unitless)", false},
    {"pf_mu", HXPB_PF_MU, true, "degC", false},
    {"pf_sigma", HXPB_PF_SIGMA, true, "degC", false},
    {"fpf_static", HXPB_FPF_STATIC, true, "(unitless)", false},
};

struct OutDef { const char *name; int idx; };
const OutDef kOutputs[] = {
    {"sst", HXO_SST}, {"land_tas", HXO_TLAND}, {"CO2_concentration", HXO_CO2},
    {"global_tas", HXO_TGAV}, {"RF_tot", HXO_RF_TOT}, {"RF_CO2", HXO_RF_CO2},
    {"ocean_c", HXO_OCEAN_C}, {"HL_pH", HXO_HL_PH}, {"atmos_co2", HXO_ATMOS_C},
    {"permafrost_c", HXO_PERMAFROST_C}, {"heatflux", HXO_HEATFLUX},
    {"CH4_concentration", HXO_CH4}, {"O3_concentration", HXO_O3}, {"veg_c", HXO_VEG_C},
    {"detritus_c", HXO_DET_C}, {"soil_c", HXO_SOIL_C}, {"thawedp_c", HXO_THAWED_C},
    {"earth_c", HXO_EARTH_C}, {"NBP", HXO_NBP}, {"ocean_uptake", HXO_OCEAN_UPTAKE},
    {"timesteps", HXO_NSTASH}, {"solver_steps", HXO_NSTEPS}, {"LL_pH", HXO_LL_PH},
    {"sst_reported", HXO_SST_LO},  // internal: see out_index()
    {"NPP", HXO_NPP}, {"RH", HXO_RH}, {"rh_det", HXO_RH_DET}, {"rh_soil", HXO_RH_SOIL},
    {"HL_ocean_uptake", HXO_HL_UPTAKE}, {"LL_ocean_uptake", HXO_LL_UPTAKE},
    {"HL_downwelling", HXO_HL_DO}, {"atmos_c_residual", HXO_CA_RESIDUAL},
    {"rh_ch4", HXO_RH_CH4}, {"f_frozen", HXO_F_FROZEN}, {"gmst", HXO_GMST},
    {"heatflux_mixed", HXO_FLUX_MIXED}, {"heatflux_interior", HXO_FLUX_INTERIOR},
    {"HL_ocean_c", HXO_C_HL}, {"LL_ocean_c", HXO_C_LL}, {"IO_ocean_c", HXO_C_IO},
    {"DO_ocean_c", HXO_C_DO}, {"HL_PCO2", HXO_PCO2_HL}, {"LL_PCO2", HXO_PCO2_LL},
    {"TAU_OH", HXO_TAU_OH},
};
const char *kOutputNames[HXO_NVAR];

// diagnostics computed on request from recorded outputs (hx_diag_kernel / hx_slr_kernel)
struct DerivedDef { const char *name; int kind; int box; const char *deps[4]; };
enum { DK_SLR = 100, DK_SL_RC, DK_SLR_NI, DK_SL_RC_NI };
const DerivedDef kDerived[] = {
    {"HL_sst", HXG_TEMP, 0, {"sst"}}, {"LL_sst", HXG_TEMP, 1, {"sst"}},
    {"HL_DIC", HXG_DIC, 0, {"HL_ocean_c"}}, {"LL_DIC", HXG_DIC, 1, {"LL_ocean_c"}},
    {"HL_CO3", HXG_CO3, 0, {"sst", "HL_pH", "HL_PCO2"}},
    {"LL_CO3", HXG_CO3, 1, {"sst", "LL_pH", "LL_PCO2"}},
    {"HL_OmegaAr", HXG_OMEGA_AR, 0, {"sst", "HL_pH", "HL_PCO2"}},
    {"LL_OmegaAr", HXG_OMEGA_AR, 1, {"sst", "LL_pH", "LL_PCO2"}},
    {"HL_OmegaCa", HXG_OMEGA_CA, 0, {"sst", "HL_pH", "HL_PCO2"}},
    {"LL_OmegaCa", HXG_OMEGA_CA, 1, {"sst", "LL_pH", "LL_PCO2"}},
    {"HL_Revelle", HXG_REVELLE, 0, {"sst", "HL_pH", "HL_PCO2", "HL_ocean_c"}},
    {"LL_Revelle", HXG_REVELLE, 1, {"sst", "LL_pH", "LL_PCO2", "LL_ocean_c"}},
    {"ocean_tas", HXG_OCEAN_TAS, -1, {"global_tas"}},
    {"RF_N2O", HXG_RF_N2O, -1, {"CO2_concentration", "CH4_concentration"}},
    {"RF_CH4", HXG_RF_CH4, -1, {"CH4_concentration"}},
    {"RF_H2O_strat", HXG_RF_H2O, -1, {"CH4_concentration"}},
    {"RF_O3_trop", HXG_RF_O3, -1, {"O3_concentration"}},
    {"slr", DK_SLR, -1, {"global_tas"}}, {"sl_rc", DK_SL_RC, -1, {"global_tas"}},
    {"slr_no_ice", DK_SLR_NI, -1, {"global_tas"}}, {"sl_rc_no_ice", DK_SL_RC_NI, -1, {"global_tas"}},
};
const DerivedDef *derived_of(const std::string &name) {
  for (const DerivedDef &d : kDerived) if (name == d.name) return &d;
  return nullptr;
}

}  // namespace

const char *const *EnsembleCore::output_capabilities(int *count) {
  for (auto &o : kOutputs) kOutputNames[o.idx] = o.name;
  if (count) *count = HXO_BIOME0;  // (per-biome pools "<biome>.veg_c" ... are named by the core's biomes)
  return kOutputNames;
}

void EnsembleCore::check(hipError_t e, const char *what) const {
  if (e != hipSuccess)
    throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

EnsembleCore::EnsembleCore(const std::string &scenario_path, int n_members, int device)
    : scen_(Scenario::load(scenario_path)), n_(n_members), B_(1), device_(device) {
  if (n_members <= 0) throw std::runtime_error("n_members must be > 0");
  npad_ = (n_ + HX_WAVE - 1) / HX_WAVE * HX_WAVE;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    throw std::runtime_error(
        "hector_amd: no HIP device available -- the ensemble integrator has no CPU path");
  if (device < 0 || device >= ndev) throw std::runtime_error("invalid device index");
  check(hipSetDevice(device_), "hipSetDevice");
#ifndef HX_HOST_EMULATION
  {  // (ahead of the stream: nothing to release if the query fails)
    hipDeviceProp_t prop;
    check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties");
    simds_ = 4 * prop.multiProcessorCount;
  }
#endif
  check(hipStreamCreate(&stream_), "hipStreamCreate");
  if (const char *e2 = std::getenv("HECTOR_AMD_PAIR_MAX_MEMBERS")) pair_max_members_ = std::atoi(e2);
  if (const char *e3 = std::getenv("HECTOR_AMD_TWO_WAVE_FROM")) two_wave_from_ = std::atoi(e3);
  if (const char *e4 = std::getenv("HECTOR_AMD_PAIR_ORDER")) pair_costly_with_cheap_ = std::atoi(e4) != 0;
  if (const char *e5 = std::getenv("HECTOR_AMD_KEY_ORDER")) key_order_mode_ = std::atoi(e5);
  if (const char *e6 = std::getenv("HECTOR_AMD_COST_MODEL")) cost_model_ = std::atoi(e6) != 0;
  // (tests: the lane-order logic of a GPU with fewer SIMDs, so that small ensembles exercise it)
  if (const char *e7 = std::getenv("HECTOR_AMD_SIMDS")) simds_ = std::max(1, std::atoi(e7));
  if (const char *e8 = std::getenv("HECTOR_AMD_PREWARM_MS")) prewarm_ms_ = std::max(0, std::atoi(e8));
  try {  // a constructor that throws gets no destructor: release the stream and events here
    check(hipEventCreate(&ev0_), "hipEventCreate");
    check(hipEventCreate(&ev1_), "hipEventCreate");
    init_from_scenario();
  } catch (...) {
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    (void)hipStreamDestroy(stream_);
    throw;
  }
}

// parameter rows, biomes and the per-year table from the scenario (INI) values
void EnsembleCore::init_from_scenario() {
  for (int v = 0; v < HXO_NVAR; ++v) { d_out_[v] = nullptr; out_enabled_[v] = false; }
  out_enabled_[HXO_SST] = out_enabled_[HXO_TLAND] = true;
  out_enabled_[HXO_CO2] = out_enabled_[HXO_TGAV] = true;
  biome_names_ = {"global"};  // SNBOX_DEFAULT_BIOME
  // parameter rows from the scenario (INI) values
  params_.assign(HX_NPARAM(1), std::vector<double>((size_t)npad_, 0.0));
  row_uniform_.assign(HX_NPARAM(1), true);
  auto setrow = [&](int row, double v) { std::fill(params_[row].begin(), params_[row].end(), v); };
  const Scenario &s = scen_;
  tracking_year_ = (int)s.scalar("core", "trackingDate", 0.0);  // [core] trackingDate
  if (tracking_year_ >= 9999 || tracking_year_ < 0) tracking_year_ = 0;
  setrow(HXP_S, s.scalar("temperature", "S"));
  setrow(HXP_DIFF, s.scalar("temperature", "diff"));
  setrow(HXP_QCO2, s.scalar("temperature", "qco2"));
  setrow(HXP_AERO, s.scalar("forcing", "aero_scalar", 1.0));
  setrow(HXP_VOL, s.scalar("forcing", "vol_scalar", 1.0));
  setrow(HXP_C0, s.scalar("simpleNbox", "C0"));
  setrow(HXP_TT, s.scalar("ocean", "tt"));
  setrow(HXP_TU, s.scalar("ocean", "tu"));
  setrow(HXP_TWI, s.scalar("ocean", "twi"));
  setrow(HXP_TID, s.scalar("ocean", "tid"));
  setrow(HXP_PRE_SURF, s.scalar("ocean", "preind_surface_c", 900));
  setrow(HXP_PRE_ID, s.scalar("ocean", "preind_interdeep_c", 37100));
  setrow(HXP_LO_RATIO, s.scalar("temperature", "lo_warming_ratio", 0.0));
  out_enabled_[HXO_SST_LO] = s.scalar("temperature", "lo_warming_ratio", 0.0) != 0.0;
  const int r = HXP_NGLOBAL;
  setrow(r + HXPB_BETA, s.scalar("simpleNbox", "beta", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_Q10, s.scalar("simpleNbox", "q10_rh", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_WF, s.scalar("simpleNbox", "warmingfactor", 1.0));
  setrow(r + HXPB_NPP0, s.scalar("simpleNbox", "npp_flux0", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_VEG0, s.scalar("simpleNbox", "veg_c", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_DET0, s.scalar("simpleNbox", "detritus_c", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_SOIL0, s.scalar("simpleNbox", "soil_c", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_PF0, s.scalar("simpleNbox", "permafrost_c", 0.0));
  setrow(r + HXPB_F_NPPV, s.scalar("simpleNbox", "f_nppv", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_F_NPPD, s.scalar("simpleNbox", "f_nppd", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_F_LITTERD, s.scalar("simpleNbox", "f_litterd", 0.0));  // 0 if only biomes define it
  setrow(r + HXPB_RH_CH4_FRAC, s.scalar("simpleNbox", "rh_ch4_frac", 0.023));
  setrow(r + HXPB_PF_MU, s.scalar("simpleNbox", "pf_mu", 1.67));
  setrow(r + HXPB_PF_SIGMA, s.scalar("simpleNbox", "pf_sigma", 0.986));
  setrow(r + HXPB_FPF_STATIC, s.scalar("simpleNbox", "fpf_static", 0.74));
  // Biomes defined in the INI file: "<biome>.<variable>" keys of [simpleNbox]
  // (src/simpleNbox.cpp:190-330; the checks of simpleNbox-runtime.cpp:64-135)
  {
    std::vector<std::string> biomes;
    for (const std::string &k : s.scalar_keys("simpleNbox")) {
      const size_t dot = k.find('.');
      if (dot == std::string::npos) continue;
      const std::string b = k.substr(0, dot);
      if (std::find(biomes.begin(), biomes.end(), b) == biomes.end()) biomes.push_back(b);
    }
    if (!biomes.empty()) {
      if ((int)biomes.size() > HX_BDYN)
        throw std::runtime_error("at most " + std::to_string(HX_BDYN) + " biomes are supported"
                                 " (the reference creates any number, simpleNbox.cpp:864-1124; here the per-biome pools of a member live in one wavefront's share of the LDS and the per-biome outputs in a table sized at build time: HX_BDYN in hx_layout.h)");
      B_ = (int)biomes.size();
      biome_names_ = biomes;
      const std::vector<std::vector<double>> global = params_;
      params_.assign(HX_NPARAM(B_), std::vector<double>((size_t)npad_, 0.0));
      row_uniform_.assign(HX_NPARAM(B_), true);
      for (int g = 0; g < HXP_NGLOBAL; ++g) params_[g] = global[g];
      for (int b = 0; b < B_; ++b)
        for (const ParamDef &d : kParams) {
          if (!d.per_biome) continue;
          const std::string key = biomes[(size_t)b] + "." + d.name;
          double v;
          if (s.has_scalar("simpleNbox", key)) v = s.scalar("simpleNbox", key);
          else if (d.row == HXPB_WF) v = 1.0;
          else if (d.row == HXPB_RH_CH4_FRAC) v = 0.023;
          else if (d.row == HXPB_PF_MU) v = 1.67;
          else if (d.row == HXPB_PF_SIGMA) v = 0.986;
          else if (d.row == HXPB_FPF_STATIC) v = 0.74;
          else
            throw std::runtime_error(std::string(d.name) + " and veg_c not same size: no " + d.name +
                                     " data for " + biomes[(size_t)b]);
          setrow(HXP_NGLOBAL + b * HXPB_N + d.row, v);
        }
    }
  }
  // "enabled=0" in a component's section removes it from the model (core.cpp:251-256).  The
  // forcing component copes with missing halocarbons, aerosols, volcanic SO2 and ozone
  // (forcing_component.cpp:392-484); sea-level rise has no dependants.  The components of the
  // year loop itself cannot be taken out of this integrator.
  for (const char *sec : {"simpleNbox", "ocean", "temperature", "forcing", "carbon-cycle-solver"})
    if (component_disabled(sec))
      throw std::runtime_error(std::string("[") + sec + "] enabled=0 is not supported: the component "
                               "is part of the GPU year loop (the reference's run fails on the "
                               "capability it no longer finds, core.cpp:743)");
  // The gas components: N2O has no dependants but the forcing component, which then skips the
  // major greenhouse gases altogether (CO2, N2O, CH4, stratospheric H2O:
  // forcing_component.cpp:315-389 asks for all three concentrations or none).  CH4 can only go
  // together with the two components that ask for its concentration every year (OH:
  // oh_component.cpp:150, ozone: o3_component.cpp:134), OH only with CH4 (ch4_component.cpp:164
  // asks for the OH lifetime): the reference aborts its first year otherwise (core.cpp:743).
  {
    const bool no_ch4 = component_disabled("CH4"), no_oh = component_disabled("OH"),
               no_o3 = component_disabled("ozone");
    if (no_ch4 && !(no_oh && no_o3))
      throw std::runtime_error("[CH4] enabled=0: [" + std::string(no_oh ? "ozone" : "OH") +
                               "] requests the CH4 concentration every year (Capability "
                               "CH4_concentration not found, core.cpp:743): disable [OH] and [ozone] too");
    if (no_oh && !no_ch4)
      throw std::runtime_error("[OH] enabled=0: [CH4] requests the OH lifetime every year "
                               "(Capability TAU_OH not found, core.cpp:743): disable [CH4] (and [ozone]) too");
  }
  if (s.scalar("ocean", "spinup_chem", 0) != 0)
    // With spinup_chem = 1 the reference never tunes -- never even sets -- the surface boxes'
    // alkalinity (oceanbox::chem_equilibrate runs only "if (!spinup_chem ...)",
    // ocean_component.cpp:392-400; oceancsys starts with alk = 0, ocean_csys.cpp:88-91): the first
    // carbonate solve of the spinup finds no root and the run aborts (the oracle's restatement
    // stops at spinup step 1 with its root-not-found flag).  Nothing to integrate: say so.
    throw std::runtime_error("ocean.spinup_chem=1: the reference aborts in its first spinup step "
                             "with this setting (the surface boxes' alkalinity is never set, so the "
                             "carbonate system has no root); every shipped scenario uses 0");
  build_shared();
}

bool EnsembleCore::component_disabled(const std::string &section) const {
  return scen_.scalar(section, "enabled", 1.0) <= 0;
}

EnsembleCore::~EnsembleCore() {
  if (prewarm_on_) (void)hipMemsetAsync(d_prewarm_, 1, 4, stream_);
  if (aux_stream_) { (void)hipStreamSynchronize(aux_stream_); (void)hipStreamDestroy(aux_stream_); }
  if (d_prewarm_) (void)hipFree(d_prewarm_);
  free_device();
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// Member-independent per-year series: N2O, 26 halocarbons, aerosols, albedo,
// volcanic, misc; OH/O3 emission terms; the carbon cycle's current-year fluxes.
void EnsembleCore::build_shared() {
  const Scenario &s = scen_;
  const int ns = s.ns();
  shared_.assign((size_t)ns * HXSH_STRIDE, 0.0);
  auto ser = [&](const char *sec, const char *key) -> const std::vector<double> & {
    return s.series(sec, key);
  };
  auto ser0 = [&](const char *sec, const char *key) {
    return s.has_series(sec, key) ? s.series(sec, key) : std::vector<double>((size_t)ns, 0.0);
  };
  const auto &ffi = ser("simpleNbox", "ffi_emissions");
  const auto daccs = ser0("simpleNbox", "daccs_uptake");
  const auto &luce = ser("simpleNbox", "luc_emissions");
  const auto lucu = ser0("simpleNbox", "luc_uptake");
  // If no albedo data, assume constant -0.2 (simpleNbox-runtime.cpp:162-167)
  const auto albedo = s.has_series("simpleNbox", "RF_albedo")
                          ? s.series("simpleNbox", "RF_albedo")
                          : std::vector<double>((size_t)ns, -0.2);
  const auto &so2 = ser("so2", "SO2_emissions");
  const auto sv = ser0("so2", "SV");
  const auto &ch4n = ser("CH4", "CH4N");
  const auto &ch4em = ser("CH4", "CH4_emissions");
  const auto &nox_oh = ser("OH", "NOX_emissions");
  const auto &co_oh = ser("OH", "CO_emissions");
  const auto &nmvoc_oh = ser("OH", "NMVOC_emissions");
  const auto &nox_o3 = ser("ozone", "NOX_emissions");
  const auto &co_o3 = ser("ozone", "CO_emissions");
  const auto &nmvoc_o3 = ser("ozone", "NMVOC_emissions");
  const auto &n2o_nat = ser("N2O", "N2O_natural_emissions");
  const auto &n2o_em = ser("N2O", "N2O_emissions");
  const auto misc = ser0("forcing", "RF_misc");
  const auto &bc = ser("bc", "BC_emissions");
  const auto &oc = ser("oc", "OC_emissions");
  const auto &nh3 = ser("nh3", "NH3_emissions");
  const double CNOX = s.scalar("OH", "CNOX"), CCO = s.scalar("OH", "CCO"),
               CNMVOC = s.scalar("OH", "CNMVOC");
  const double N0 = s.scalar("N2O", "N0"), UC_N2O = s.scalar("N2O", "UC_N2O"),
               TN2O0 = s.scalar("N2O", "TN2O0");
  const double rho_bc = s.scalar("forcing", "rho_bc"), rho_oc = s.scalar("forcing", "rho_oc"),
               rho_so2 = s.scalar("forcing", "rho_so2"), rho_nh3 = s.scalar("forcing", "rho_nh3");
  // forcing_component.hpp:120-131
  const double aci_beta = 2.279759, s_BCOC = 111.05064063;
  const double s_SO2 = (260.34644166 * 1000) * (32.065 / 64.066);
  const bool so2_off = component_disabled("so2");
  const bool aerosols_off = so2_off || component_disabled("bc") || component_disabled("oc") ||
                            component_disabled("nh3");
  std::vector<double> hconc(s.halocarbons.size());
  for (size_t h = 0; h < hconc.size(); ++h) hconc[h] = s.halocarbons[h].H0;
  // halocarbon forcings enter the total in std::map key order ("RF_<gas>")
  std::vector<size_t> horder(hconc.size());
  for (size_t h = 0; h < horder.size(); ++h) horder[h] = h;
  std::sort(horder.begin(), horder.end(), [&](size_t a, size_t b) {
    return s.halocarbons[a].name < s.halocarbons[b].name;
  });
  // constraints: dense series, NaN where the reference's tseries has no value
  const std::vector<double> none;
  auto con = [&](const std::string &sec, const std::string &key) -> const std::vector<double> & {
    return s.has_series(sec, key) ? s.series(sec, key) : none;
  };
  auto has = [](const std::vector<double> &c, int iy) { return !c.empty() && !std::isnan(c[(size_t)iy]); };
  auto any = [](const std::vector<double> &c) {
    for (double v : c) if (!std::isnan(v)) return true;
    return false;
  };
  const auto &co2_con = con("simpleNbox", "CO2_constrain");
  const auto &nbp_con = con("simpleNbox", "NBP_constrain");
  const auto &tas_con = con("temperature", "tas_constrain");
  const auto &ftot_con = con("forcing", "RF_tot_constrain");
  const auto &ch4_con = con("CH4", "CH4_constrain");
  const auto &n2o_con = con("N2O", "N2O_constrain");
  std::vector<const std::vector<double> *> hcon(hconc.size());
  for (size_t h = 0; h < hconc.size(); ++h)
    hcon[h] = &con(s.halocarbons[h].name + "_halocarbon", s.halocarbons[h].name + "_constrain");
  // a constraint at startDate replaces the preindustrial value (n2o_component.cpp:137-145,
  // ch4_component.cpp:137-147); the OH component read M0 before that (its prepareToRun runs
  // first: CH4 depends on the OH lifetime)
  const double N0f = has(n2o_con, 0) ? n2o_con[0] : N0;
  double n2o = N0f;
  halo_conc_.assign(hconc.size(), std::vector<double>((size_t)ns, 0.0));
  halo_names_.clear();
  for (auto &H : s.halocarbons) halo_names_.push_back(H.name);
  for (int iy = 0; iy < ns; ++iy) {
    double *row = &shared_[(size_t)iy * HXSH_STRIDE];
    row[HXSH_CO2_CON] = has(co2_con, iy) ? co2_con[(size_t)iy] : std::nan("");
    row[HXSH_NBP_CON] = has(nbp_con, iy) ? nbp_con[(size_t)iy] : std::nan("");
    row[HXSH_TAS_CON] = has(tas_con, iy) ? tas_con[(size_t)iy] : std::nan("");
    row[HXSH_FTOT_CON] = has(ftot_con, iy) ? ftot_con[(size_t)iy] : std::nan("");
    row[HXSH_CH4_CON] = has(ch4_con, iy) ? ch4_con[(size_t)iy] : std::nan("");
    if (iy >= 1) {  // slowparameval(t = year-1): emissions of date t (runtime.cpp:951-955)
      row[HXSH_FFI] = ffi[iy - 1]; row[HXSH_DACCS] = daccs[iy - 1];
      row[HXSH_LUC_E] = luce[iy - 1]; row[HXSH_LUC_U] = lucu[iy - 1];
    }
    // oh_component.cpp:157-170
    row[HXSH_OH_B] = CNOX * ((1.0 * nox_oh[iy]) - nox_oh[0]);
    row[HXSH_OH_C] = CCO * ((1.0 * co_oh[iy]) - co_oh[0]);
    row[HXSH_OH_D] = CNMVOC * ((1.0 * nmvoc_oh[iy]) - nmvoc_oh[0]);
    row[HXSH_CH4_EM] = ch4em[iy];
    row[HXSH_CH4N] = ch4n[iy];
    // o3_component.cpp:136-139
    row[HXSH_O3_NOX] = 0.125 * nox_o3[iy];
    row[HXSH_O3_CO] = 0.0011 * co_o3[iy];
    row[HXSH_O3_NMVOC] = 0.0033 * nmvoc_o3[iy];
    double rf_h = 0.0;
    if (iy >= 1) {
      // n2o_component.cpp:152-191
      const double tau = TN2O0 * std::pow(n2o / N0f, -0.05);
      const double em = n2o_em[iy] + n2o_nat[iy];
      n2o = n2o + (em / UC_N2O - n2o / tau);
      if (has(n2o_con, iy)) n2o = n2o_con[(size_t)iy];  // n2o_component.cpp:157-158
      // halocarbon_component.cpp:181-229
      for (size_t h = 0; h < hconc.size(); ++h) {
        const Halocarbon &H = s.halocarbons[h];
        const double alpha = 1 / H.tau;
        const double emissMol = H.emissions[iy] / H.molarMass * 1.0;
        const double dconc = emissMol / (0.1 * 1.8);
        const double expfac = std::exp(-alpha);
        hconc[h] = hconc[h] * expfac + dconc * H.tau * (1.0 - expfac);
        if (has(*hcon[h], iy)) hconc[h] = (*hcon[h])[(size_t)iy];  // halocarbon_component.cpp:189
      }
      for (size_t k = 0; k < horder.size(); ++k) {
        const Halocarbon &H = s.halocarbons[horder[k]];
        if (component_disabled(H.name + "_halocarbon")) continue;  // forcing_component.cpp:413-419
        const double rf_un = H.rho * hconc[horder[k]];
        rf_h = rf_h + (rf_un + H.delta * rf_un);
      }
    }
    for (size_t h = 0; h < hconc.size(); ++h) halo_conc_[h][(size_t)iy] = hconc[h];
    row[HXSH_N2O] = n2o;
    row[HXSH_SQRT_N2O] = std::sqrt(n2o);
    row[HXSH_RF_OTHER] = (rf_h + albedo[iy]) + misc[iy];
    // forcing_component.cpp:430-470 for aero_scalar = 1
    row[HXSH_RF_AERO] = (((rho_bc * bc[iy] + rho_oc * oc[iy]) + rho_so2 * so2[iy]) +
                         rho_nh3 * nh3[iy]) +
                        (-1 * aci_beta *
                         std::log(1 + (so2[iy] / s_SO2) + ((bc[iy] + oc[iy]) / s_BCOC)));
    row[HXSH_RF_VOL] = sv[iy];
    // aerosol forcings need all four emission components, the volcanic one the SO2 component
    // (forcing_component.cpp:422-425, 478)
    if (aerosols_off) row[HXSH_RF_AERO] = 0.0;
    if (so2_off) row[HXSH_RF_VOL] = 0.0;
  }
  HxConst &k = kc_;
  k.start_year = s.start; k.ns = ns;
  {  // ForcingComponent::prepareToRun (forcing_component.cpp:278-289): default startDate + 1
    double by = s.scalar("forcing", "baseyear", 0.0);
    if (by == 0.0) by = s.start + 1;
    if (!(by > s.start)) throw std::runtime_error("Base year must be >= model start date");
    k.baseyear_idx = (int)by - s.start;
  }
  k.max_spinup = (int)s.scalar("core", "max_spinup", 2000);
  if (s.scalar("core", "do_spinup", 1) == 0) k.max_spinup = 1;  // core.cpp:378-384: no spinup steps
  k.spinup_chem = 0;
  k.eps_abs = s.scalar("carbon-cycle-solver", "eps_abs", 1e-6);
  k.eps_rel = s.scalar("carbon-cycle-solver", "eps_rel", 1e-6);
  k.dt0 = s.scalar("carbon-cycle-solver", "dt", 0.3);
  k.eps_spinup = s.scalar("carbon-cycle-solver", "eps_spinup");
  k.M0 = s.scalar("CH4", "M0"); k.lnM0 = std::log(k.M0);
  const double M0f_old = k.M0f;
  k.M0f = has(ch4_con, 0) ? ch4_con[0] : k.M0;
  k.sqrtM0 = std::sqrt(k.M0f);
  k.inv_h2o_span = 1.0 / (1831 - k.M0f);
  if (k.M0f != M0f_old) { need_spinup_ = true; spin_valid_ = false; }  // the post-spinup state holds CH4(startDate)
  k.con_mask = (any(co2_con) ? HXC_CO2 : 0) | (any(nbp_con) ? HXC_NBP : 0) |
               (any(tas_con) ? HXC_TAS : 0) | (any(ftot_con) ? HXC_FTOT : 0) |
               (any(ch4_con) ? HXC_CH4 : 0);
  k.Tsoil = s.scalar("CH4", "Tsoil"); k.Tstrat = s.scalar("CH4", "Tstrat");
  k.UC_CH4 = s.scalar("CH4", "UC_CH4");
  k.inv_UC_CH4 = 1.0 / k.UC_CH4; k.inv_Tsoil = 1.0 / k.Tsoil; k.inv_Tstrat = 1.0 / k.Tstrat;
  k.TOH0 = s.scalar("OH", "TOH0"); k.CCH4 = s.scalar("OH", "CCH4");
  k.N0 = N0f; k.sqrtN0 = std::sqrt(N0f);
  k.delta_co2 = s.scalar("forcing", "delta_co2"); k.delta_ch4 = s.scalar("forcing", "delta_ch4");
  k.delta_n2o = s.scalar("forcing", "delta_n2o");
  if (component_disabled("CH4") || component_disabled("N2O")) {
    // the forcing component leaves out CO2, N2O, CH4 and stratospheric H2O (see the constructor):
    // sarf * (-1) + sarf = 0 exactly, and (CH4 - M0) * 0 = 0
    k.delta_co2 = k.delta_ch4 = k.delta_n2o = -1.0;
    k.inv_h2o_span = 0.0;
  }
  k.o3_rf = component_disabled("ozone") ? 0.0 : 0.042;
  hx_fill_tableau(k.tab);
  hx_fill_math_table(k.mtab);
  hx_fill_chem_table_host(k.ctab);
  hx_fill_chem_fit(k.kfit);
}

void EnsembleCore::free_device() {
  auto fr = [](void *p) { if (p) (void)hipFree(p); };
  fr(d_params_); fr(d_state_); fr(d_shared_); fr(d_ker_); fr(d_status_); fr(d_spin_steps_);
  fr(d_uparams_); d_uparams_ = nullptr;
  fr(d_track_out_f_); fr(d_track_out_v_);
  d_track_out_f_ = d_track_out_v_ = nullptr;
  fr(d_args_); fr(d_derived_); fr(d_dpart_); fr(d_gather_); fr(d_lane_of_member_); fr(d_hist_);
  fr(d_hist_status_);
  fr(d_gas_par_); fr(d_gas_ser_); d_gas_par_ = d_gas_ser_ = nullptr;
  fr(d_cost_); d_cost_ = nullptr; cost_from_iy_ = -1;
  fr(d_wave_clk_); d_wave_clk_ = nullptr;
  fr(d_bscratch_); d_bscratch_ = nullptr;
  fr(d_spin_rec_); d_spin_rec_ = nullptr;
  d_hist_ = nullptr; d_hist_status_ = nullptr;
  for (int k = 0; k < HXM_N; ++k) { fr(d_mseries_[k]); d_mseries_[k] = nullptr; if (!member_series_[k].empty()) mseries_dirty_ = true; }
  fr(d_diag_); fr(d_slr_); d_diag_ = d_slr_ = nullptr; diag_cap_ = 0; slr_valid_to_ = -1;
  d_derived_ = nullptr; d_dpart_ = nullptr; d_gather_ = nullptr; d_lane_of_member_ = nullptr;
  gather_cap_ = 0;
  d_params_ = d_state_ = d_shared_ = d_ker_ = nullptr; d_status_ = nullptr; d_spin_steps_ = nullptr;
  d_args_ = nullptr;
  for (int v = 0; v < HXO_NVAR; ++v) { fr(d_out_[v]); d_out_[v] = nullptr; }
}

void EnsembleCore::alloc_device() {
  check(hipSetDevice(device_), "hipSetDevice");
  free_device();
  const size_t np = (size_t)npad_, ns = (size_t)scen_.ns();
  {  // The kernels that read their tables as row address + lane offset (hx_dev_member.h: w2_ld,
    // GlobArr) form the row's byte offset row * npad * 8 in 32 bits: a table must stay below 4 GB
    // (253 parameter rows of a 16-biome core reach that at 2.1 million members).
    const size_t rows = std::max({(size_t)HX_NPARAM(B_), (size_t)HX_NSTATE(B_), (size_t)HX_NDERIVED(B_),
                                  (size_t)(3 * B_)});
    if (rows * np * sizeof(double) >= (size_t(1) << 32))
      throw std::runtime_error("an ensemble of " + std::to_string(n_) + " members with " + std::to_string(B_) +
                               " biome(s) needs a per-member table of " + std::to_string(rows) +
                               " rows >= 4 GB, beyond the kernels' 32-bit row offsets: split it over "
                               "several cores (hx_newcore_devices)");
  }
  check(hipMalloc(&d_params_, sizeof(double) * np * HX_NPARAM(B_)), "hipMalloc params");
  check(hipMalloc(&d_uparams_, sizeof(double) * (HX_NPARAM(B_) + HX_NDERIVED(B_))), "hipMalloc uniform params");
  if (trk_iy() >= 0) {  // carbon tracking: the yearly record of the origin matrices from the tracking
    // date on (the kernels update the current year's matrix in place, hx_dev_track.h)
    const size_t TP = (size_t)(2 + 5 * B_ + 4), nyt = ns - (size_t)trk_iy();
    if (TP > 128)   // (the origin masks are two 64-bit words a pool: hx_trk_mask_words)
      throw std::runtime_error("carbon tracking: at most 24 biomes (" + std::to_string(TP) +
                               " pools; the kernels keep a pool's origin mask in two 64-bit words)");
    const size_t vr = (size_t)hx_track_value_rows(B_);
    // (slot 0: the identity of the tracking date; 4 rows of padding: the last chunk of source
    //  columns is read whole)
    const size_t bytes_f = sizeof(double) * ((nyt + 1) * TP * TP * np + 8 * 64),
                 bytes_v = sizeof(double) * (nyt + 1) * vr * np;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes_f + bytes_v > free_b)
      throw std::runtime_error("carbon tracking: the record of " + std::to_string(nyt) + " years x " +
                               std::to_string(TP) + " x " + std::to_string(TP) + " fractions x " +
                               std::to_string(npad_) + " members needs " +
                               std::to_string((bytes_f + bytes_v) >> 20) + " MiB of device memory, " +
                               std::to_string(free_b >> 20) + " MiB are free: track fewer members or "
                               "set a later trackingDate");
    check(hipMalloc(&d_track_out_f_, bytes_f), "hipMalloc tracking record");
    check(hipMalloc(&d_track_out_v_, bytes_v), "hipMalloc tracking record");
    check(hipMemsetAsync(d_track_out_f_, 0, bytes_f, stream_), "zero");
    check(hipMemsetAsync(d_track_out_v_, 0, bytes_v, stream_), "zero");
  }
  check(hipMalloc(&d_state_, sizeof(double) * np * HX_NSTATE(B_) * 2), "hipMalloc state");
  check(hipMalloc(&d_shared_, sizeof(double) * shared_.size()), "hipMalloc shared");
  // DOECLIM kernel table, zero-padded by HX_KPAD (= 32) entries in front and 64 behind (the
  // history pass runs up to three slices of four years past the end)
  check(hipMalloc(&d_ker_, sizeof(double) * (ns + 96) * np), "hipMalloc ker");
  // status: live | post-spinup snapshot (with the derive-time flags) | the spinup's own result
  check(hipMalloc(&d_status_, sizeof(unsigned) * np * 3), "hipMalloc status");
  check(hipMalloc(&d_spin_steps_, sizeof(int) * np), "hipMalloc spin");
  check(hipMalloc(&d_args_, sizeof(HxArgs)), "hipMalloc args");
  check(hipMalloc(&d_derived_, sizeof(double) * np * HX_NDERIVED(B_)), "hipMalloc derived");
  check(hipMalloc(&d_lane_of_member_, sizeof(int) * np), "hipMalloc lane map");
  if (history_) {
    check(hipMalloc(&d_hist_, sizeof(double) * ns * np * HX_NSTATE(B_)), "hipMalloc state history");
    check(hipMalloc(&d_hist_status_, sizeof(unsigned) * ns * np), "hipMalloc status history");
  }
  if (spin_record_) {
    const size_t bytes = sizeof(double) * (size_t)kc_.max_spinup * HXSR_N * np;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b)
      throw std::runtime_error("spinup record: " + std::to_string(kc_.max_spinup) + " steps x " +
                               std::to_string((int)HXSR_N) + " variables x " + std::to_string(npad_) +
                               " members need " + std::to_string(bytes >> 20) + " MiB of device memory, " +
                               std::to_string(free_b >> 20) + " MiB are free");
    check(hipMalloc(&d_spin_rec_, bytes), "hipMalloc spinup record");
    check(hipMemsetAsync(d_spin_rec_, 0, bytes, stream_), "zero");
  }
  hist_valid_to_ = 0;
  check(hipMalloc(&d_dpart_, sizeof(double) * np * 2 * (size_t)hx_doeclim_block_years()),
        "hipMalloc doeclim partial sums");
  check(hipMalloc(&d_cost_, sizeof(double) * np), "hipMalloc lane cost");
  check(hipMalloc(&d_wave_clk_, sizeof(long long) * 4 * (np / HX_WAVE)), "hipMalloc wave clock");
  check(hipMemsetAsync(d_wave_clk_, 0, sizeof(long long) * 4 * (np / HX_WAVE), stream_), "zero");
  check(hipMalloc(&d_bscratch_, sizeof(double) * np * (size_t)(3 * B_)), "hipMalloc biome scratch");
  check(hipMemsetAsync(d_bscratch_, 0, sizeof(double) * np * (size_t)(3 * B_), stream_), "zero");
  cost_from_iy_ = -1;
  for (int v = 0; v < HXO_NVAR; ++v)
    if (out_enabled_[v]) {
      check(hipMalloc(&d_out_[v], sizeof(double) * ns * np), "hipMalloc out");
      check(hipMemsetAsync(d_out_[v], 0, sizeof(double) * ns * np, stream_), "memset out");
    }
  check(hipMemcpyAsync(d_shared_, shared_.data(), sizeof(double) * shared_.size(),
                       hipMemcpyHostToDevice, stream_), "upload shared");
  layout_dirty_ = false;
  params_dirty_ = true;
  rows_all_dirty_ = true;   // fresh buffers: every row of the parameter table is sent again
  need_spinup_ = true; spin_valid_ = false;
}

HxBuffers EnsembleCore::buffers() const {
  HxBuffers b;
  b.params = d_params_; b.derived = d_derived_; b.state = d_state_; b.status = d_status_; b.shared = d_shared_;
  b.ker = d_ker_;
  b.dpart = d_dpart_;
  b.dpart2 = d_dpart_ ? d_dpart_ + (size_t)npad_ * hx_doeclim_block_years() : nullptr;
  for (int v = 0; v < HXO_NVAR; ++v) b.out[v] = d_out_[v];
  b.hist = d_hist_; b.hist_status = d_hist_status_;
  for (int k = 0; k < HXM_N; ++k) b.mseries[k] = d_mseries_[k];
  b.uparams = d_uparams_;
  b.uderived = d_uparams_ ? d_uparams_ + HX_NPARAM(B_) : nullptr;
  b.uni_k = (row_uniform_[HXP_TT] && row_uniform_[HXP_TU] && row_uniform_[HXP_TWI] && row_uniform_[HXP_TID]) ? 1 : 0;
  b.uni_avc = (row_uniform_[HXP_AERO] && row_uniform_[HXP_VOL] && row_uniform_[HXP_C0]) ? 1 : 0;
  b.uni_wf = 1;
  for (int bb = 0; bb < B_; ++bb) if (!row_uniform_[HXP_NGLOBAL + bb * HXPB_N + HXPB_WF]) b.uni_wf = 0;
  b.track_out_f = d_track_out_f_; b.track_out_v = d_track_out_v_;
  b.trk_slots = trk_iy() >= 0 ? scen_.ns() - trk_iy() + 1 : 0;
  b.uni_landk = b.uni_bio = 1;
  for (int bb = 0; bb < B_; ++bb) {
    const int r = HXP_NGLOBAL + bb * HXPB_N;
    for (int k : {HXPB_NPP0, HXPB_F_NPPV, HXPB_F_NPPD, HXPB_F_LITTERD, HXPB_RH_CH4_FRAC, HXPB_FPF_STATIC})
      if (!row_uniform_[r + k]) b.uni_landk = 0;
    for (int k : {HXPB_BETA, HXPB_PF_MU, HXPB_PF_SIGMA})
      if (!row_uniform_[r + k]) b.uni_bio = 0;
  }
  b.stash_diag = b.biome_diag = b.out_rare = 0;
  for (int v = 0; v < HXO_NVAR; ++v)
    if (d_out_[v] && v != HXO_SST && v != HXO_TLAND && v != HXO_CO2 && v != HXO_TGAV) b.out_rare = 1;
  for (int v = HXO_NPP; v <= HXO_CA_RESIDUAL; ++v) if (d_out_[v]) b.stash_diag = 1;
  for (int v = HXO_BIOME0; v < HXO_NVAR; ++v) if (d_out_[v]) b.biome_diag = 1;
  for (int bb = 0; bb < HX_BDYN; ++bb)
    if (d_out_[HXO_B(HXOB_NPP, bb)] || d_out_[HXO_B(HXOB_RH, bb)]) b.stash_diag = 1;
  b.out_mask0 = 0; b.ms_mask = 0;
  for (int v = 0; v < HXO_BIOME0; ++v) if (d_out_[v]) b.out_mask0 |= 1ull << v;
  for (int bb = 0; bb < HX_BDYN; ++bb)
    if (d_out_[HXO_B(HXOB_NPP, bb)] || d_out_[HXO_B(HXOB_RH, bb)]) b.out_mask0 |= 1ull << HX_OM_BIOME_FLUX;
  if (b.biome_diag) b.out_mask0 |= 1ull << HX_OM_BIOME_ANY;
  for (int k = 0; k < HXM_N; ++k) if (d_mseries_[k]) b.ms_mask |= 1u << k;
  b.n = n_; b.npad = npad_; b.ker_per_member = ker_per_member_ ? 1 : 0;
  b.nbiome = B_;
  b.cost = d_cost_;
  b.spin_rec = d_spin_rec_;
  b.bscratch = d_bscratch_;
  b.wave_clk = d_wave_clk_;
  return b;
}

int EnsembleCore::resolve_param(const std::string &capability, const ParamRef **) const {
  std::string biome, var = capability;
  size_t dot = capability.find('.');
  if (dot != std::string::npos) { biome = capability.substr(0, dot); var = capability.substr(dot + 1); }
  for (const ParamDef &d : kParams) {
    if (var != d.name) continue;
    if (!d.per_biome) {
      if (!biome.empty()) throw std::runtime_error("variable " + var + " takes no biome prefix");
      return d.row;
    }
    int b = -1;
    if (biome.empty()) biome = "global";
    for (int i = 0; i < B_; ++i) if (biome_names_[i] == biome) b = i;
    if (b < 0)
      throw std::runtime_error("Biome '" + biome + "' missing from biome list. Hit this error "
                               "while trying to retrieve variable: '" + capability + "'.");
    return HXP_NGLOBAL + b * HXPB_N + d.row;
  }
  throw std::runtime_error("Unknown variable name while parsing: " + capability);
}

static const ParamDef &def_of(const std::string &capability) {
  std::string var = capability;
  size_t dot = capability.find('.');
  if (dot != std::string::npos) var = capability.substr(dot + 1);
  for (const ParamDef &d : kParams) if (var == d.name) return d;
  throw std::runtime_error("Unknown variable name while parsing: " + capability);
}

namespace {
// Parameters of the member-independent components (forcing efficiencies, CH4/N2O/O3
// preindustrial values and lifetimes, halocarbon rho/delta/tau): one value for the whole core.
struct SharedParamDef { const char *name, *section, *units; };
const SharedParamDef kSharedParams[] = {
    {"M0", "CH4", "ppbv CH4"}, {"Tsoil", "CH4", "Years"}, {"Tstrat", "CH4", "Years"},
    {"N0", "N2O", "ppbv N2O"}, {"PO3", "ozone", "DU O3"}, {"TOH0", "OH", "Years"},
    {"delta_co2", "forcing", "(unitless)"}, {"delta_ch4", "forcing", "(unitless)"},
    {"delta_n2o", "forcing", "(unitless)"}, {"rho_bc", "forcing", "W/m2/Tg"},
    {"rho_oc", "forcing", "W/m2/Tg"}, {"rho_so2", "forcing", "W/m2/Gg"},
    {"rho_nh3", "forcing", "W/m2/Tg"},
    // N2O lifetime and conversion (component_data.hpp:274-275; INI keys of [N2O])
    {"TN2O0", "N2O", "Years"}, {"UC_N2O", "N2O", "Tg/ppbv"},
};
}  // namespace

// -> section of a shared scalar parameter ("" if `capability` is not one); *units gets its unit,
// *key the INI key it is stored under ("tau_<gas>" is the key "tau" of [<gas>_halocarbon])
static std::string shared_param_section(const Scenario &scen, const std::string &capability,
                                        std::string *units, std::string *key = nullptr) {
  if (key) *key = capability;
  for (const SharedParamDef &d : kSharedParams)
    if (capability == d.name) { if (units) *units = d.units; return d.section; }
  for (const Halocarbon &h : scen.halocarbons) {
    if (capability == "rho_" + h.name) { if (units) *units = "W/m2/pptv"; return h.name + "_halocarbon"; }
    if (capability == "delta_" + h.name) { if (units) *units = "(unitless)"; return h.name + "_halocarbon"; }
    if (capability == "tau_" + h.name) {
      if (units) *units = "Years";
      if (key) *key = "tau";
      return h.name + "_halocarbon";
    }
  }
  return "";
}

// The parameters of the N2O and halocarbon components that may differ between members: their
// recurrences then run per member on the device (hx_gas_kernel) instead of once on the host.
static bool gas_member_capable(const Scenario &scen, const std::string &capability) {
  if (capability == "N0" || capability == "TN2O0" || capability == "UC_N2O") return true;
  for (const Halocarbon &h : scen.halocarbons)
    if (capability == "rho_" + h.name || capability == "delta_" + h.name || capability == "tau_" + h.name)
      return true;
  return false;
}

void EnsembleCore::setvar(const std::string &capability, const double *values, int nvalues,
                          const char *units) {
  // R/messages.R (setvar): "<biome>.<variable>" at most
  if (std::count(capability.begin(), capability.end(), '.') > 1)
    throw std::runtime_error("Invalid input variable: '" + capability + "'");
  if (capability == "trackingDate") {  // Core::setData  core.cpp:230-236
    set_tracking_date((int)values[0]);
    return;
  }
  {
    std::string expect, key;
    const std::string sec = shared_param_section(scen_, capability, &expect, &key);
    if (!sec.empty()) {
      if (units && units[0] && expect != units)
        throw std::runtime_error("Units: " + std::string(units) + " do not match expected: " +
                                 expect + " for " + capability);
      bool uniform = true;
      for (int i = 1; i < nvalues; ++i) if (values[i] != values[0]) uniform = false;
      if (!uniform) {
        if (nvalues != n_) throw std::runtime_error("setvar: need 1 or n_members values");
        if (!gas_member_capable(scen_, capability))
          throw std::runtime_error(capability + " belongs to a member-independent component: one "
                                   "value for the whole core");
        gas_member_[capability].assign(values, values + n_);
      } else {
        gas_member_.erase(capability);
      }
      scen_.set_scalar(sec, key, values[0]);  // (member 0's value where members differ)
      gas_dirty_ = true;
      shared_dirty_ = true;
      last_iy_ = 0;
      need_spinup_ = true; spin_valid_ = false;
      return;
    }
  }
  const int row = resolve_param(capability, nullptr);
  const ParamDef &d = def_of(capability);
  if (units && units[0] && std::string(units) != d.units)
    throw std::runtime_error("Units: " + std::string(units) + " do not match expected: " +
                             d.units + " for " + capability);
  if (nvalues != 1 && nvalues != n_)
    throw std::runtime_error("setvar: need 1 or n_members values");
  std::vector<double> &r = params_[row];
  bool uniform = true;
  for (int i = 0; i < npad_; ++i) {
    const int src = (nvalues == 1) ? 0 : std::min(i, n_ - 1);  // pad with the last member
    r[(size_t)i] = values[src];
    if (values[src] != values[0]) uniform = false;
  }
  row_uniform_[row] = uniform;
  params_dirty_ = true;
  if (row_dirty_.size() == params_.size()) row_dirty_[(size_t)row] = 1; else rows_all_dirty_ = true;
  if (d.spinup) spin_valid_ = false;  // (any other parameter leaves the post-spinup state as it is)
  lane_cost_.clear();  // measured with other parameters: back to the parameter key
  // R/messages.R:107-140: a parameter change invalidates the run from date 0
  last_iy_ = 0;
  need_spinup_ = true;
  if (row == HXP_LO_RATIO) {  // the reported SST needs its own output array
    bool any = false;
    for (double v : r) if (v != 0.0) any = true;
    if (any != out_enabled_[HXO_SST_LO]) { out_enabled_[HXO_SST_LO] = any; layout_dirty_ = true; }
  }
}

void EnsembleCore::getvar(const std::string &capability, double *out) const {
  if (capability == "trackingDate") {  // 9999 = never, like Core's default (core.cpp:60)
    std::fill(out, out + n_, tracking_year_ > 0 ? (double)tracking_year_ : 9999.0);
    return;
  }
  {
    std::string key;
    const std::string sec = shared_param_section(scen_, capability, nullptr, &key);
    if (!sec.empty()) {
      auto it = gas_member_.find(capability);
      if (it != gas_member_.end()) std::copy(it->second.begin(), it->second.end(), out);
      else std::fill(out, out + n_, scen_.scalar(sec, key));
      return;
    }
  }
  const int row = resolve_param(capability, nullptr);
  std::memcpy(out, params_[row].data(), sizeof(double) * (size_t)n_);
}

void EnsembleCore::split_biome(const std::vector<std::string> &names, const double *fveg,
                               const double *fdet, const double *fsoil, const double *fpf,
                               const double *fnpp) {
  if (B_ != 1) throw std::runtime_error("split_biome: name the biome to split (several exist)");
  split_biome_of(biome_names_[0], names, fveg, fdet, fsoil, fpf, fnpp);
}

// split_biome(core, old_biome, new_biomes, ...)  R/biome.R:61-130: the new biomes are appended
// to the list (create_biome), the old one is deleted.  Pools and npp_flux0 are the old biome's
// times the fractions; warmingfactor, beta, q10_rh, f_nppv, f_nppd, f_litterd the old biome's;
// what create_biome leaves alone (rh_ch4_frac, permafrost parameters) comes from the last biome
// of the list, as in SimpleNbox::createBiome.
void EnsembleCore::split_biome_of(const std::string &old_biome,
                                  const std::vector<std::string> &names, const double *fveg,
                                  const double *fdet, const double *fsoil, const double *fpf,
                                  const double *fnpp) {
  const int nb = (int)names.size(), ob = biome_index(old_biome);
  if (ob < 0) throw std::runtime_error("Biome '" + old_biome + "' missing from biome list.");
  if (nb < 1 || B_ - 1 + nb > HX_BDYN)
    throw std::runtime_error("split_biome: at most " + std::to_string(HX_BDYN) + " biomes supported"
                             " (the reference creates any number, simpleNbox.cpp:864-1124; here the per-biome pools of a member live in one wavefront's share of the LDS and the per-biome outputs in a table sized at build time: HX_BDYN in hx_layout.h)");
  for (int a = 0; a < nb; ++a) {
    if (names[(size_t)a].empty() || names[(size_t)a].find('.') != std::string::npos)
      throw std::runtime_error("split_biome: bad biome name '" + names[(size_t)a] + "'");
    if (biome_index(names[(size_t)a]) >= 0 && names[(size_t)a] != old_biome)
      throw std::runtime_error("Biome '" + names[(size_t)a] + "' is already in `biome_list`.");
    for (int c = 0; c < a; ++c)
      if (names[(size_t)a] == names[(size_t)c])
        throw std::runtime_error("Biome '" + names[(size_t)a] + "' is already in `biome_list`.");
  }
  std::vector<double> eq((size_t)nb, 1.0 / nb);
  if (!fveg) fveg = eq.data();
  if (!fdet) fdet = fveg;
  if (!fsoil) fsoil = fveg;
  if (!fpf) fpf = fveg;
  if (!fnpp) fnpp = fveg;
  const int nB = B_ - 1 + nb;
  std::vector<std::vector<double>> np(HX_NPARAM(nB), std::vector<double>((size_t)npad_, 0.0));
  std::vector<bool> nu(HX_NPARAM(nB), true);
  std::vector<std::string> nn;
  for (int r = 0; r < HXP_NGLOBAL; ++r) { np[r] = params_[r]; nu[r] = row_uniform_[r]; }
  int dstb = 0;
  for (int b = 0; b < B_; ++b) {  // the biomes that stay, in order
    if (b == ob) continue;
    for (int k = 0; k < HXPB_N; ++k) {
      np[HXP_NGLOBAL + dstb * HXPB_N + k] = params_[HXP_NGLOBAL + b * HXPB_N + k];
      nu[HXP_NGLOBAL + dstb * HXPB_N + k] = row_uniform_[HXP_NGLOBAL + b * HXPB_N + k];
    }
    nn.push_back(biome_names_[(size_t)b]);
    ++dstb;
  }
  const int lastb = B_ - 1;  // createBiome copies the untouched parameters from the list's last biome
  for (int a = 0; a < nb; ++a, ++dstb) {
    for (int k = 0; k < HXPB_N; ++k) {
      const int dst = HXP_NGLOBAL + dstb * HXPB_N + k;
      int src = HXP_NGLOBAL + ob * HXPB_N + k;
      double f = 1.0;
      bool scaled = true;
      switch (k) {
        case HXPB_VEG0: f = fveg[a]; break;
        case HXPB_DET0: f = fdet[a]; break;
        case HXPB_SOIL0: f = fsoil[a]; break;
        case HXPB_PF0: f = fpf[a]; break;
        case HXPB_NPP0: f = fnpp[a]; break;
        case HXPB_RH_CH4_FRAC: case HXPB_PF_MU: case HXPB_PF_SIGMA: case HXPB_FPF_STATIC:
          scaled = false; src = HXP_NGLOBAL + lastb * HXPB_N + k; break;
        default: scaled = false;
      }
      for (int i = 0; i < npad_; ++i)
        np[dst][(size_t)i] = scaled ? params_[src][(size_t)i] * f : params_[src][(size_t)i];
      nu[dst] = row_uniform_[src];
    }
    nn.push_back(names[(size_t)a]);
  }
  {
    std::vector<int> old_of_new;
    for (int b = 0; b < B_; ++b) if (b != ob) old_of_new.push_back(b);
    for (int a = 0; a < nb; ++a) old_of_new.push_back(-1);
    remap_biome_outputs(old_of_new);
  }
  params_.swap(np);
  row_uniform_.swap(nu);
  B_ = nB;
  biome_names_ = nn;
  layout_dirty_ = true;
  params_dirty_ = true;
  need_spinup_ = true; spin_valid_ = false;
  last_iy_ = 0;
}

// "<biome>.<variable>" outputs are enabled by biome index: keep them with their biome when the
// list is renumbered (old_of_new[b] = previous index of new biome b, -1 = a new biome)
void EnsembleCore::remap_biome_outputs(const std::vector<int> &old_of_new) {
  bool was[HXOB_N][HX_BDYN];
  for (int k = 0; k < HXOB_N; ++k)
    for (int b = 0; b < HX_BDYN; ++b) { was[k][b] = out_enabled_[HXO_B(k, b)]; out_enabled_[HXO_B(k, b)] = false; }
  for (size_t b = 0; b < old_of_new.size() && b < (size_t)HX_BDYN; ++b)
    if (old_of_new[b] >= 0)
      for (int k = 0; k < HXOB_N; ++k) out_enabled_[HXO_B(k, (int)b)] = was[k][old_of_new[b]];
}

int EnsembleCore::biome_index(const std::string &biome) const {
  for (int b = 0; b < B_; ++b) if (biome_names_[(size_t)b] == biome) return b;
  return -1;
}

// SimpleNbox::createBiome (simpleNbox.cpp:864-932): empty pools, zero npp_flux0, every other
// parameter like the most recent biome; appended to the biome list.
void EnsembleCore::create_biome(const std::string &biome) {
  if (biome_index(biome) >= 0)
    throw std::runtime_error("Biome '" + biome + "' is already in `biome_list`.");
  if (biome.empty() || biome.find('.') != std::string::npos)
    throw std::runtime_error("create_biome: bad biome name '" + biome + "'");
  if (B_ >= HX_BDYN)
    throw std::runtime_error("create_biome: at most " + std::to_string(HX_BDYN) + " biomes supported"
                             " (the reference creates any number, simpleNbox.cpp:864-1124; here the per-biome pools of a member live in one wavefront's share of the LDS and the per-biome outputs in a table sized at build time: HX_BDYN in hx_layout.h)");
  const int last = HXP_NGLOBAL + (B_ - 1) * HXPB_N;
  for (int k = 0; k < HXPB_N; ++k) {
    const bool pool = k == HXPB_VEG0 || k == HXPB_DET0 || k == HXPB_SOIL0 || k == HXPB_PF0 ||
                      k == HXPB_NPP0;
    params_.push_back(pool ? std::vector<double>((size_t)npad_, 0.0) : params_[(size_t)(last + k)]);
    row_uniform_.push_back(pool ? true : (bool)row_uniform_[(size_t)(last + k)]);
  }
  ++B_;
  biome_names_.push_back(biome);
  layout_dirty_ = params_dirty_ = need_spinup_ = true; spin_valid_ = false;
  last_iy_ = 0;
}

// SimpleNbox::deleteBiome (simpleNbox.cpp:934-990): the biome and everything it holds go away
void EnsembleCore::delete_biome(const std::string &biome) {
  const int b = biome_index(biome);
  if (b < 0) throw std::runtime_error("Biome '" + biome + "' missing from biome list.");
  if (B_ == 1) throw std::runtime_error("delete_biome: the core needs at least one biome");
  const auto first = HXP_NGLOBAL + b * HXPB_N;
  params_.erase(params_.begin() + first, params_.begin() + first + HXPB_N);
  row_uniform_.erase(row_uniform_.begin() + first, row_uniform_.begin() + first + HXPB_N);
  biome_names_.erase(biome_names_.begin() + b);
  {
    std::vector<int> old_of_new;
    for (int k = 0; k < B_; ++k) if (k != b) old_of_new.push_back(k);
    remap_biome_outputs(old_of_new);
  }
  --B_;
  layout_dirty_ = params_dirty_ = need_spinup_ = true; spin_valid_ = false;
  last_iy_ = 0;
}

// SimpleNbox::renameBiome (simpleNbox.cpp:992-1060)
void EnsembleCore::rename_biome(const std::string &oldname, const std::string &newname) {
  const int b = biome_index(oldname);
  if (b < 0) throw std::runtime_error("Biome '" + oldname + "' missing from biome list.");
  if (biome_index(newname) >= 0)
    throw std::runtime_error("Biome '" + newname + "' already exists in biome list.");
  if (newname.empty() || newname.find('.') != std::string::npos)
    throw std::runtime_error("rename_biome: bad biome name '" + newname + "'");
  biome_names_[(size_t)b] = newname;
}

int EnsembleCore::out_index(const std::string &capability) const {
  // with a land-ocean warming ratio D_SST is not DOECLIM's own SST (temperature_component.cpp:
  // 614-625); the kernel records the reported one separately
  if (capability == "sst" && out_enabled_[HXO_SST_LO]) return HXO_SST_LO;
  if (capability == "ocean_timesteps") return HXO_NSTASH;  // D_TIMESTEPS, component_data.hpp:337
  const size_t dot = capability.find('.');
  if (dot != std::string::npos) {  // "<biome>.<pool>"  (SNBOX_PARSECHAR, simpleNbox.cpp:527-540)
    static const char *const pools[HXOB_N] = {"veg_c", "detritus_c", "soil_c", "permafrost_c",
                                               "thawedp_c", "NPP", "RH", "rh_ch4", "f_frozen",
                                               "detritus_tempfert", "soil_tempfert"};
    const std::string biome = capability.substr(0, dot), var = capability.substr(dot + 1);
    // "global.<pool>" of a core without a biome of that name is the total over the biomes
    // (SimpleNbox::sum_fluxpool_biome_ts, simpleNbox.cpp:463-485)
    if (biome == "global" && biome_index(biome) < 0) return out_index(var);
    for (int b = 0; b < B_; ++b)
      if (biome_names_[(size_t)b] == biome)
        for (int k = 0; k < HXOB_N; ++k)
          if (var == pools[k]) return HXO_B(k, b);
    throw std::runtime_error("Biome '" + biome + "' missing from biome list. Hit this error while "
                             "trying to retrieve variable: '" + capability + "'.");
  }
  for (auto &o : kOutputs) if (capability == o.name) return o.idx;
  throw std::runtime_error("Caller is requesting unknown variable: " + capability);
}

void EnsembleCore::set_outputs(const std::vector<std::string> &caps) {
  bool want[HXO_NVAR];
  for (int v = 0; v < HXO_NVAR; ++v) want[v] = false;
  want[HXO_SST] = want[HXO_TLAND] = true;
  want[HXO_SST_LO] = out_enabled_[HXO_SST_LO];
  for (auto &c : caps) {
    if (const DerivedDef *d = derived_of(c)) {  // record what the diagnostic is derived from
      for (const char *dep : d->deps) if (dep) want[out_index(dep)] = true;
      continue;
    }
    if (c == "N2O_concentration") continue;  // the N2O component's own series (shared or per member)
    if (host_output(c)) continue;  // answered from the scenario / the shared gas cycles
    if (c == "pH" || c == "PCO2" || c == "DIC" || c == "CO3" || c == "ML_ocean_c") {
      const std::string base = c == "ML_ocean_c" ? "ocean_c" : c;
      for (const char *box : {"LL_", "HL_"}) {
        const std::string part = box + base;
        if (const DerivedDef *d = derived_of(part)) { for (const char *dep : d->deps) if (dep) want[out_index(dep)] = true; }
        else want[out_index(part)] = true;
      }
      continue;
    }
    want[out_index(c)] = true;
  }
  bool changed = false;
  for (int v = 0; v < HXO_NVAR; ++v) if (want[v] != out_enabled_[v]) changed = true;
  if (!changed) return;
  for (int v = 0; v < HXO_NVAR; ++v) out_enabled_[v] = want[v];
  layout_dirty_ = true;
  need_spinup_ = true; spin_valid_ = false;
  last_iy_ = 0;
}

void EnsembleCore::enable_spinup_record(bool on) {
  if (on == spin_record_) return;
  spin_record_ = on;
  layout_dirty_ = true;
  need_spinup_ = true; spin_valid_ = false;
  last_iy_ = 0;
}

const std::vector<std::string> &EnsembleCore::spinup_record_vars() {
  // capability names, in the order of the HXSR_* rows
  static const std::vector<std::string> v = {
      "NBP", "NPP", "RH", "rh_det", "rh_soil", "atmos_co2", "atmos_c_residual", "veg_c", "detritus_c",
      "soil_c", "permafrost_c", "thawedp_c", "earth_c", "HL_ocean_uptake", "LL_ocean_uptake",
      "DO_ocean_c", "HL_ocean_c", "IO_ocean_c", "LL_ocean_c", "HL_downwelling", "ocean_uptake"};
  static_assert(HXSR_N == 21, "spinup_record_vars lists the HXSR_* rows");
  return v;
}

int EnsembleCore::spinup_record(int member, double *values, int max_steps) {
  if (!spin_record_) throw std::runtime_error("the spinup record is off (hx_enable_spinup_record first)");
  if (member < 0 || member >= n_) throw std::runtime_error("spinup_record: bad member index");
  const int steps = spinup_steps(member);  // (prepares: runs the spinup if it is due)
  if (steps > max_steps) throw std::runtime_error("spinup_record: room for " + std::to_string(max_steps) +
                                                  " steps, the spinup took " + std::to_string(steps));
  // a shared spinup (no member differs in a parameter the spinup sees) ran on the first wavefront
  const size_t lane = spin_uniform_ ? 0 : (size_t)lane_of_member_[(size_t)member];
  if (steps > 0)
    check(hipMemcpy2D(values, sizeof(double), d_spin_rec_ + lane, (size_t)npad_ * sizeof(double),
                      sizeof(double), (size_t)steps * HXSR_N, hipMemcpyDeviceToHost), "spinup record");
  return steps;
}

void EnsembleCore::enable_history(bool on) {
  if (on == history_) return;
  history_ = on;
  layout_dirty_ = true;
  need_spinup_ = true; spin_valid_ = false;
  last_iy_ = 0;
}

namespace {
struct DatedDef { const char *name; const char *sections[2]; const char *units; };
// inputs with dates: capability -> INI section(s) holding the series (component_data.hpp;
// NOX/CO/NMVOC are read by both the OH and the ozone component)
const DatedDef kDated[] = {
    {"ffi_emissions", {"simpleNbox", nullptr}, "Pg C/yr"},
    {"luc_emissions", {"simpleNbox", nullptr}, "Pg C/yr"},
    {"daccs_uptake", {"simpleNbox", nullptr}, "Pg C/yr"},
    {"luc_uptake", {"simpleNbox", nullptr}, "Pg C/yr"},
    {"RF_albedo", {"simpleNbox", nullptr}, "W/m2"},
    {"SO2_emissions", {"so2", nullptr}, "Gg S"},
    {"SV", {"so2", nullptr}, "W/m2"},
    {"CH4_emissions", {"CH4", nullptr}, "Tg CH4"},
    {"CH4N", {"CH4", nullptr}, "Tg CH4"},
    {"NOX_emissions", {"OH", "ozone"}, "Tg N"},
    {"CO_emissions", {"OH", "ozone"}, "Tg CO"},
    {"NMVOC_emissions", {"OH", "ozone"}, "Tg NMVOC"},
    {"N2O_emissions", {"N2O", nullptr}, "Tg N"},
    {"N2O_natural_emissions", {"N2O", nullptr}, "Tg N"},
    {"RF_misc", {"forcing", nullptr}, "W/m2"},
    {"BC_emissions", {"bc", nullptr}, "Tg"},
    {"OC_emissions", {"oc", nullptr}, "Tg"},
    {"NH3_emissions", {"nh3", nullptr}, "Tg"},
    // constraints (component_data.hpp:46,263,273,379-381)
    {"CO2_constrain", {"simpleNbox", nullptr}, "ppmv CO2"},
    {"NBP_constrain", {"simpleNbox", nullptr}, "Pg C/yr"},
    {"tas_constrain", {"temperature", nullptr}, "degC"},
    {"RF_tot_constrain", {"forcing", nullptr}, "W/m2"},
    {"CH4_constrain", {"CH4", nullptr}, "ppbv CH4"},
    {"N2O_constrain", {"N2O", nullptr}, "ppbv N2O"},
};
}  // namespace

namespace {
struct MemberSeriesDef { const char *name; int k; const char *section; const char *units; int lag; };
// lag: the shared table holds the value of date year-1 for what slowparameval reads
const MemberSeriesDef kMemberSeries[] = {
    {"ffi_emissions", HXM_FFI, "simpleNbox", "Pg C/yr", 1},
    {"daccs_uptake", HXM_DACCS, "simpleNbox", "Pg C/yr", 1},
    {"luc_emissions", HXM_LUC_E, "simpleNbox", "Pg C/yr", 1},
    {"luc_uptake", HXM_LUC_U, "simpleNbox", "Pg C/yr", 1},
    {"CH4_emissions", HXM_CH4_EM, "CH4", "Tg CH4", 0},
    // constraints that differ between members (NaN = no constraint for that member and year)
    {"CO2_constrain", HXM_CO2_CON, "simpleNbox", "ppmv CO2", 0},
    {"NBP_constrain", HXM_NBP_CON, "simpleNbox", "Pg C/yr", 0},
    {"tas_constrain", HXM_TAS_CON, "temperature", "degC", 0},
    {"RF_tot_constrain", HXM_FTOT_CON, "forcing", "W/m2", 0},
    {"CH4_constrain", HXM_CH4_CON, "CH4", "ppbv CH4", 0},
};
bool is_constraint_series(int k) { return k >= HXM_CO2_CON && k <= HXM_CH4_CON; }
bool interpolated_constraint(int k) { return k == HXM_TAS_CON || k == HXM_FTOT_CON; }
int constraint_bit(int k) {
  switch (k) {
    case HXM_CO2_CON: return HXC_CO2; case HXM_NBP_CON: return HXC_NBP;
    case HXM_TAS_CON: return HXC_TAS; case HXM_FTOT_CON: return HXC_FTOT;
    case HXM_CH4_CON: return HXC_CH4; default: return 0;
  }
}
}  // namespace

void EnsembleCore::setvar_dated(const std::string &capability, const int *years,
                                const double *values, int n, const char *units) {
  std::string sections[2];
  std::string expect;
  for (const DatedDef &d : kDated)
    if (capability == d.name) {
      sections[0] = d.sections[0];
      if (d.sections[1]) sections[1] = d.sections[1];
      expect = d.units;
    }
  const std::string suf = "_emissions";
  if (sections[0].empty() && capability.size() > suf.size() &&
      capability.compare(capability.size() - suf.size(), suf.size(), suf) == 0) {
    const std::string gas = capability.substr(0, capability.size() - suf.size());
    for (auto &h : scen_.halocarbons)
      if (h.name == gas) { sections[0] = gas + "_halocarbon"; expect = "Gg"; }
  }
  if (sections[0].empty() && Scenario::is_constraint(capability)) {
    const std::string gas = capability.substr(0, capability.size() - 10);
    for (auto &h : scen_.halocarbons)
      if (h.name == gas) { sections[0] = gas + "_halocarbon"; expect = "pptv"; }
  }
  if (sections[0].empty())
    throw std::runtime_error("Unknown variable name while parsing: " + capability +
                             " (dated inputs)");
  if (units && units[0] && expect != units)
    throw std::runtime_error("Units: " + std::string(units) + " do not match expected: " + expect +
                             " for " + capability);
  int miny = scen_.end;
  for (int i = 0; i < n; ++i) {
    for (auto &sec : sections)
      if (!sec.empty()) scen_.set_series_value(sec, capability, years[i], values[i]);
    miny = std::min(miny, years[i]);
  }
  // once a variable has per-member series (setvar_dated_members) the kernels and fetchvars read
  // those: a value "for every member" goes into every member's row as well
  for (const MemberSeriesDef &d : kMemberSeries) {
    const int k = d.k;
    if (capability != d.name || member_series_[k].empty()) continue;
    if (interpolated_constraint(k)) {  // the interpolated years move with the points
      for (int i = 0; i < n; ++i) member_points_[k][years[i]].assign((size_t)n_, values[i]);
      densify_member_constraint(k, capability);
    } else {
      for (int i = 0; i < n; ++i)
        std::fill(member_series_[k].begin() + (size_t)(years[i] - scen_.start) * n_,
                  member_series_[k].begin() + (size_t)(years[i] - scen_.start + 1) * n_, values[i]);
    }
    mseries_dirty_ = true;
  }
  shared_dirty_ = true;
  // R/messages.R:125-133: reset_date = min(date) - 1
  const int target = std::max(0, miny - 1 - scen_.start);
  if (target < last_iy_) dirty_from_iy_ = (dirty_from_iy_ < 0) ? target : std::min(dirty_from_iy_, target);
}


void EnsembleCore::setvar_dated_members(const std::string &capability, const int *years,
                                        const double *values, int nyears, const char *units) {
  const MemberSeriesDef *d = nullptr;
  for (const MemberSeriesDef &m : kMemberSeries) if (capability == m.name) d = &m;
  if (!d)
    throw std::runtime_error("per-member dated input not supported for " + capability +
                             " (ffi_emissions, luc_emissions, daccs_uptake, luc_uptake, "
                             "CH4_emissions, CO2_constrain, NBP_constrain, tas_constrain, "
                             "RF_tot_constrain, CH4_constrain are)");
  if (units && units[0] && std::string(units) != d->units)
    throw std::runtime_error("Units: " + std::string(units) + " do not match expected: " +
                             d->units + " for " + capability);
  const int ns = scen_.ns();
  std::vector<double> &ms = member_series_[d->k];
  if (ms.empty()) {  // start from the scenario's series, the same for every member
    ms.resize((size_t)ns * n_);
    const std::vector<double> base =
        scen_.has_series(d->section, capability)
            ? scen_.series(d->section, capability)
            : std::vector<double>((size_t)ns, is_constraint_series(d->k) ? std::nan("") : 0.0);
    for (int iy = 0; iy < ns; ++iy)
      std::fill(ms.begin() + (size_t)iy * n_, ms.begin() + (size_t)(iy + 1) * n_, base[(size_t)iy]);
    if (interpolated_constraint(d->k)) {  // ... and from the shared series' points
      member_points_[d->k].clear();
      for (auto &pt : scen_.constraint_points(d->section, capability))
        member_points_[d->k][pt.first].assign((size_t)n_, pt.second);
    }
  }
  int miny = scen_.end;
  for (int i = 0; i < nyears; ++i) {
    if (years[i] < scen_.start || years[i] > scen_.end)
      throw std::runtime_error("date outside startDate..endDate");
    // a CH4 constraint at startDate replaces the preindustrial concentration of the whole core
    // (ch4_component.cpp:137-147): that one cannot differ between members
    if (d->k == HXM_CH4_CON && years[i] == scen_.start)
      throw std::runtime_error("CH4_constrain at startDate: one value for the whole core (setvar_dated)");
    if (interpolated_constraint(d->k))
      member_points_[d->k][years[i]].assign(values + (size_t)i * n_, values + (size_t)(i + 1) * n_);
    else
      std::copy(values + (size_t)i * n_, values + (size_t)(i + 1) * n_,
                ms.begin() + (size_t)(years[i] - scen_.start) * n_);
    miny = std::min(miny, years[i]);
  }
  if (interpolated_constraint(d->k)) densify_member_constraint(d->k, capability);
  mseries_dirty_ = true;
  const int target = std::max(0, miny - 1 - scen_.start);
  if (target < last_iy_) dirty_from_iy_ = (dirty_from_iy_ < 0) ? target : std::min(dirty_from_iy_, target);
}

// member_points_[k] -> member_series_[k]: every member's points densified by the rule of the
// shared series (Scenario::densify_points: interpolation between a member's first and last date,
// RF_tot_constrain flat before its first one; NaN points are no points)
void EnsembleCore::densify_member_constraint(int k, const std::string &capability) {
  std::vector<double> &ms = member_series_[k];
  for (auto it = member_points_[k].begin(); it != member_points_[k].end();) {  // dates nobody holds
    bool any = false;
    for (double v : it->second) if (!std::isnan(v)) { any = true; break; }
    it = any ? std::next(it) : member_points_[k].erase(it);
  }
  std::map<int, double> pts;
  for (int m = 0; m < n_; ++m) {
    pts.clear();
    for (auto &pt : member_points_[k])
      if (!std::isnan(pt.second[(size_t)m])) pts.emplace_hint(pts.end(), pt.first, pt.second[(size_t)m]);
    const std::vector<double> dense = Scenario::densify_points(pts, capability, scen_.start, scen_.end);
    for (size_t iy = 0; iy < dense.size(); ++iy) ms[iy * (size_t)n_ + (size_t)m] = dense[iy];
  }
}

void EnsembleCore::upload_member_series() {
  const size_t np = (size_t)npad_, ns = (size_t)scen_.ns();
  std::vector<double> flat;
  for (const MemberSeriesDef &d : kMemberSeries) {
    const std::vector<double> &ms = member_series_[d.k];
    if (ms.empty()) continue;
    if (!d_mseries_[d.k]) check(hipMalloc(&d_mseries_[d.k], sizeof(double) * ns * np), "hipMalloc member series");
    flat.assign(ns * np, 0.0);
    for (size_t iy = (size_t)d.lag; iy < ns; ++iy)
      for (size_t l = 0; l < np; ++l)
        flat[iy * np + l] = ms[(iy - (size_t)d.lag) * n_ + (size_t)std::min(member_of_lane_[l], n_ - 1)];
    check(hipMemcpyAsync(d_mseries_[d.k], flat.data(), sizeof(double) * flat.size(),
                         hipMemcpyHostToDevice, stream_), "upload member series");
    check(hipStreamSynchronize(stream_), "sync member series");
  }
  upload_args();
  mseries_dirty_ = false;
}

// the kernels' argument block: buffers + scenario scalars; the constraint mask also carries the
// constraints that only some members hold
void EnsembleCore::upload_args() {
  HxArgs a;
  a.buf = buffers();
  a.kc = kc_;
  for (int k = 0; k < HXM_N; ++k)
    if (!member_series_[k].empty()) a.kc.con_mask |= constraint_bit(k);
  member_con_mask_ = a.kc.con_mask & ~kc_.con_mask;
  check(hipMemcpyAsync(d_args_, &a, sizeof a, hipMemcpyHostToDevice, stream_), "upload args");
  check(hipStreamSynchronize(stream_), "sync args");
}

// N2O and halocarbon recurrences per member on the device (hx_gas_kernel): parameter rows in
// lane order, the member-independent series the recurrences read, and the two result series
// the extended run kernel picks up instead of the per-year table's columns.
void EnsembleCore::run_gas_kernel() {
  gas_dirty_ = false;
  auto fr = [](void *p) { if (p) (void)hipFree(p); };
  if (gas_member_.empty()) {  // back to the host's shared series
    if (d_mseries_[HXM_N2O]) {
      sync();
      fr(d_mseries_[HXM_N2O]); fr(d_mseries_[HXM_RF_OTHER]);
      d_mseries_[HXM_N2O] = d_mseries_[HXM_RF_OTHER] = nullptr;
      upload_args();
    }
    return;
  }
  const Scenario &s = scen_;
  const size_t np = (size_t)npad_, ns = (size_t)s.ns();
  const size_t nh = s.halocarbons.size();
  std::vector<size_t> horder(nh);  // forcings are summed in std::map key order ("RF_<gas>")
  for (size_t h = 0; h < nh; ++h) horder[h] = h;
  std::sort(horder.begin(), horder.end(), [&](size_t a, size_t b) {
    return s.halocarbons[a].name < s.halocarbons[b].name;
  });
  auto member_value = [&](const std::string &cap, double dflt, size_t lane) {
    auto it = gas_member_.find(cap);
    if (it == gas_member_.end()) return dflt;
    return it->second[(size_t)std::min(member_of_lane_[lane], n_ - 1)];
  };
  std::vector<double> par((3 + 3 * nh) * np), ser((4 + 2 * nh) * ns), h0(nh);
  for (size_t l = 0; l < np; ++l) {
    par[l] = member_value("N0", s.scalar("N2O", "N0"), l);
    par[np + l] = member_value("TN2O0", s.scalar("N2O", "TN2O0"), l);
    par[2 * np + l] = member_value("UC_N2O", s.scalar("N2O", "UC_N2O"), l);
    for (size_t k = 0; k < nh; ++k) {
      const Halocarbon &H = s.halocarbons[horder[k]];
      par[(3 + 3 * k) * np + l] = member_value("tau_" + H.name, H.tau, l);
      par[(4 + 3 * k) * np + l] = component_disabled(H.name + "_halocarbon")
                                      ? 0.0 : member_value("rho_" + H.name, H.rho, l);
      par[(5 + 3 * k) * np + l] = member_value("delta_" + H.name, H.delta, l);
    }
  }
  const auto &n2o_em = s.series("N2O", "N2O_emissions");
  const auto &n2o_nat = s.series("N2O", "N2O_natural_emissions");
  const std::vector<double> nanv(ns, std::nan(""));
  auto con = [&](const std::string &sec, const std::string &key) -> const std::vector<double> & {
    return s.has_series(sec, key) ? s.series(sec, key) : nanv;
  };
  const auto &n2o_con = con("N2O", "N2O_constrain");
  const auto albedo = s.has_series("simpleNbox", "RF_albedo") ? s.series("simpleNbox", "RF_albedo")
                                                                : std::vector<double>(ns, -0.2);
  const auto misc = s.has_series("forcing", "RF_misc") ? s.series("forcing", "RF_misc")
                                                        : std::vector<double>(ns, 0.0);
  for (size_t iy = 0; iy < ns; ++iy) {
    ser[iy] = n2o_em[iy] + n2o_nat[iy];
    ser[ns + iy] = n2o_con[iy];
    ser[2 * ns + iy] = albedo[iy];
    ser[3 * ns + iy] = misc[iy];
  }
  for (size_t k = 0; k < nh; ++k) {
    const Halocarbon &H = s.halocarbons[horder[k]];
    const auto &hc = con(H.name + "_halocarbon", H.name + "_constrain");
    h0[k] = H.H0;
    for (size_t iy = 0; iy < ns; ++iy) {
      const double emissMol = H.emissions[iy] / H.molarMass * 1.0;  // halocarbon_component.cpp:194-199
      ser[(4 + 2 * k) * ns + iy] = emissMol / (0.1 * 1.8);
      ser[(5 + 2 * k) * ns + iy] = hc[iy];
    }
  }
  sync();
  fr(d_gas_par_); fr(d_gas_ser_);
  d_gas_par_ = d_gas_ser_ = nullptr;
  double *d_h0 = nullptr;
  check(hipMalloc(&d_gas_par_, sizeof(double) * par.size()), "hipMalloc gas parameters");
  check(hipMalloc(&d_gas_ser_, sizeof(double) * (ser.size() + nh)), "hipMalloc gas series");
  d_h0 = d_gas_ser_ + ser.size();
  if (!d_mseries_[HXM_N2O]) {
    check(hipMalloc(&d_mseries_[HXM_N2O], sizeof(double) * ns * np), "hipMalloc N2O series");
    check(hipMalloc(&d_mseries_[HXM_RF_OTHER], sizeof(double) * ns * np), "hipMalloc RF series");
  }
  check(hipMemcpyAsync(d_gas_par_, par.data(), sizeof(double) * par.size(), hipMemcpyHostToDevice,
                       stream_), "upload gas parameters");
  check(hipMemcpyAsync(d_gas_ser_, ser.data(), sizeof(double) * ser.size(), hipMemcpyHostToDevice,
                       stream_), "upload gas series");
  check(hipMemcpyAsync(d_h0, h0.data(), sizeof(double) * nh, hipMemcpyHostToDevice, stream_),
        "upload gas preindustrial values");
  check(hx_launch_gas(d_gas_par_, d_gas_ser_, d_h0, (int)nh, (int)ns, npad_,
                      const_cast<double *>(d_mseries_[HXM_N2O]),
                      const_cast<double *>(d_mseries_[HXM_RF_OTHER]), stream_), "gas kernel");
  check(hipStreamSynchronize(stream_), "gas kernel sync");
  upload_args();
}

// HECTOR_AMD_TIMING=1: wall time of prepare()'s stages on stderr (tools/prof/e2e_times.py)
namespace {
struct StageClock {
  bool on;
  std::chrono::steady_clock::time_point t0;
  StageClock() : on(std::getenv("HECTOR_AMD_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void lap(const char *what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[hector_amd timing] %-28s %8.3f ms\n", what,
                 std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
}  // namespace

void EnsembleCore::lane_of_member(int *out) {
  prepare();
  std::memcpy(out, lane_of_member_.data(), sizeof(int) * (size_t)n_);
}

void EnsembleCore::set_member_sorting(bool on) {
  if (on == sort_members_) return;
  sort_members_ = on;
  params_dirty_ = true;
  need_spinup_ = true;
  last_iy_ = 0;
}

// Lanes of a wavefront execute the solver in lock-step, so a wavefront costs the
// maximum over its members of steps / stashes per year.  Members with similar
// perturbed parameters follow similar schedules: tile the members by the first two
// varying parameters (quantile bins of the first, sorted by the second inside a
// bin).  On the ECS/Q10 ensemble this brings the wave-max stash count from 2.69 to
// 2.06 per year (member mean 2.00) and steps from 4.63 to 3.62 (mean 3.52).
// A stable sort of `order` by a double key (ascending, or descending with ties kept in order),
// as std::stable_sort with `key[a] < key[b]` gives it, in linear time: least-significant-digit
// radix passes over the order-preserving integer image of the keys (11 bits a pass; a pass whose
// digit is the same for every key is skipped).  Sorting 65 536 members with std::stable_sort and
// an indirect comparison took 8 ms per parameter upload -- as long as the year loop itself.
namespace {
void radix_stable_sort(std::vector<int> &order, size_t from, size_t to, const std::vector<double> &key,
                       bool descending) {
  const size_t n = to - from;
  if (n < 2) return;
  struct Item { uint64_t k; int idx; };
  std::vector<Item> a(n), b(n);
  for (size_t i = 0; i < n; ++i) {
    const int idx = order[from + i];
    const double v = key[(size_t)idx] + 0.0;  // (-0.0 compares equal to 0.0: one image for both)
    uint64_t u;
    std::memcpy(&u, &v, sizeof u);
    u ^= (u >> 63) ? ~uint64_t(0) : (uint64_t(1) << 63);  // ascending doubles -> ascending integers
    a[i] = Item{descending ? ~u : u, idx};
  }
  constexpr int BITS = 11, NPASS = 6, RAD = 1 << BITS;
  std::vector<uint32_t> hist((size_t)NPASS * RAD, 0);
  for (size_t i = 0; i < n; ++i)
    for (int p = 0; p < NPASS; ++p) ++hist[(size_t)p * RAD + ((a[i].k >> (p * BITS)) & (RAD - 1))];
  for (int p = 0; p < NPASS; ++p) {
    uint32_t *h = hist.data() + (size_t)p * RAD;
    if (h[(a[0].k >> (p * BITS)) & (RAD - 1)] == n) continue;  // every key has this digit
    uint32_t sum = 0;
    for (int d = 0; d < RAD; ++d) { const uint32_t c = h[d]; h[d] = sum; sum += c; }
    for (size_t i = 0; i < n; ++i) b[h[(a[i].k >> (p * BITS)) & (RAD - 1)]++] = a[i];
    a.swap(b);
  }
  for (size_t i = 0; i < n; ++i) order[from + i] = a[i].idx;
}
}  // namespace

// ---- a fitted cost model on the parameter key ------------------------------------------------
// What a member's solver costs (dopri5 steps, stashes) is a smooth function of its perturbed
// parameters.  After a complete run the core fits  cost ~ quadratic in the standardised varying
// parameter rows  to the measured per-member costs (least squares on a sample of members) and
// files it in a process-wide registry under the scenario's per-year table, the biome count and
// the list of varying rows.  A core with the same key -- the next, larger ensemble of a study, the
// next iteration of a calibration loop after a setvar -- orders its lanes by the PREDICTED cost
// from its first run on, costliest wavefronts first like the measured order, instead of waiting
// for a complete run of its own.  (A pilot in time cannot do this: the members' costs of the
// first 160 years of SSP2-4.5 have a rank correlation of 0.2 with those of the rest; a quadratic
// in S and Q10 has 0.89 with the total, and orders 131 072 members as well as their measured
// costs do -- scratch analysis on the oracle's schedules)
namespace {
struct CostModel {
  std::vector<int> rows;            // varying parameter rows the model knows
  std::vector<double> mean, sd;     // their standardisation at fit time
  bool cross = true;                // with the products x_k x_l (k < l)
  std::vector<double> beta;         // 1, x_k, x_k^2, [x_k x_l]
};
std::mutex g_cost_mu;
std::map<uint64_t, CostModel> g_cost_models;

size_t cost_terms(size_t k, bool cross) { return 1 + 2 * k + (cross ? k * (k - 1) / 2 : 0); }
void cost_features(const CostModel &m, const double *x, double *f) {
  const size_t k = m.rows.size();
  size_t o = 0;
  f[o++] = 1.0;
  for (size_t a = 0; a < k; ++a) f[o++] = x[a];
  for (size_t a = 0; a < k; ++a) f[o++] = x[a] * x[a];
  if (m.cross)
    for (size_t a = 0; a < k; ++a)
      for (size_t b = a + 1; b < k; ++b) f[o++] = x[a] * x[b];
}
}  // namespace

uint64_t EnsembleCore::cost_model_key(const std::vector<int> &varying) const {
  uint64_t h = 1469598103934665603ull;   // FNV-1a
  auto mix = [&](const void *p, size_t n) {
    const unsigned char *c = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
  };
  mix(shared_.data(), shared_.size() * sizeof(double));
  mix(&B_, sizeof B_);
  for (int r : varying) mix(&r, sizeof r);
  //  ... and what else shapes a member's schedule: the values of the rows every member
  // shares, and which constraints the scenario / the members carry -- a model fitted to another
  // workload of the same scenario is not applied to this one
  const int cm = kc_.con_mask | member_con_mask_;
  mix(&cm, sizeof cm);
  for (int r = 0; r < HX_NPARAM(B_); ++r)
    if (row_uniform_[r] && !params_[(size_t)r].empty()) mix(&params_[(size_t)r][0], sizeof(double));
  return h;
}

// ---- models that outlive the process -----------------------------------------------------------
// A genuine one-shot run -- fresh process, one core -- has no earlier core to learn from.  The
// registry is therefore seeded from a FILE of fitted models, one text record per model:
//   model <key hex> <k> <cross> | rows ... | mean ... | sd ... | beta ...
// read once per process, before the first lookup: $HECTOR_AMD_COST_MODELS if set, else
// <directory of this library>/../data/cost_models.txt -- the models shipped with the scenarios
// (tools/make_cost_models.py fits them on the GPU for the shipped scenarios' perturbed-parameter
// ensembles and writes the file through hx_cost_models_export).  A record whose key matches no
// core is never used; a model only orders lanes, results do not depend on it.
namespace {
std::once_flag g_cost_file_once;
bool parse_cost_models(std::istream &in, std::map<uint64_t, CostModel> &out) {
  std::string line;
  bool any = false;
  while (std::getline(in, line)) {
    std::istringstream ls(line);
    std::string tag, keyhex;
    size_t k = 0; int cross = 0;
    if (!(ls >> tag) || tag != "model") continue;
    if (!(ls >> keyhex >> k >> cross) || k == 0 || k > 24) continue;
    CostModel m;
    m.cross = cross != 0;
    const size_t nt = cost_terms(k, m.cross);
    auto section = [&](const char *name, size_t n, auto &vec) {
      std::string bar, nm;
      if (!(ls >> bar >> nm) || bar != "|" || nm != name) return false;
      vec.resize(n);
      for (size_t i = 0; i < n; ++i) if (!(ls >> vec[i])) return false;
      return true;
    };
    if (!section("rows", k, m.rows) || !section("mean", k, m.mean) || !section("sd", k, m.sd) ||
        !section("beta", nt, m.beta)) continue;
    bool ok = true;
    for (double v : m.beta) ok = ok && std::isfinite(v);
    for (double v : m.sd) ok = ok && std::isfinite(v) && v > 0;
    if (!ok) continue;
    out[std::strtoull(keyhex.c_str(), nullptr, 16)] = std::move(m);
    any = true;
  }
  return any;
}
std::string default_cost_models_path() {
  if (const char *e = std::getenv("HECTOR_AMD_COST_MODELS")) return e;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&default_cost_models_path), &info) && info.dli_fname) {
    std::string lib = info.dli_fname;
    const size_t sl = lib.rfind('/');
    return (sl == std::string::npos ? std::string(".") : lib.substr(0, sl)) + "/../data/cost_models.txt";
  }
  return "";
}
void seed_cost_models_once() {
  std::call_once(g_cost_file_once, [] {
    const std::string path = default_cost_models_path();
    if (path.empty()) return;
    std::ifstream in(path);
    if (!in) return;
    std::map<uint64_t, CostModel> got;
    if (!parse_cost_models(in, got)) return;
    std::lock_guard<std::mutex> lk(g_cost_mu);
    for (auto &kv : got) g_cost_models.emplace(kv.first, std::move(kv.second));   // (a fitted one stays)
  });
}
}  // namespace

int hx_cost_models_load_file(const char *path) {
  std::ifstream in(path);
  if (!in) return -1;
  std::map<uint64_t, CostModel> got;
  if (!parse_cost_models(in, got)) return 0;
  std::lock_guard<std::mutex> lk(g_cost_mu);
  int n = 0;
  for (auto &kv : got) { g_cost_models[kv.first] = std::move(kv.second); ++n; }
  return n;
}

int hx_cost_models_export_file(const char *path) {
  std::ofstream out(path, std::ios::trunc);
  if (!out) return -1;
  out << "# hector_amd lane-cost models: cost ~ quadratic in the standardised varying parameter rows\n"
         "# (EnsembleCore::fit_cost_model; written by hx_cost_models_export, read at first use)\n";
  out.precision(17);
  std::lock_guard<std::mutex> lk(g_cost_mu);
  int n = 0;
  for (const auto &kv : g_cost_models) {
    const CostModel &m = kv.second;
    char key[32];
    std::snprintf(key, sizeof key, "%016llx", (unsigned long long)kv.first);
    out << "model " << key << ' ' << m.rows.size() << ' ' << (m.cross ? 1 : 0) << " | rows";
    for (int r : m.rows) out << ' ' << r;
    out << " | mean";
    for (double v : m.mean) out << ' ' << v;
    out << " | sd";
    for (double v : m.sd) out << ' ' << v;
    out << " | beta";
    for (double v : m.beta) out << ' ' << v;
    out << '\n';
    ++n;
  }
  return out ? n : -1;
}

// member_cost: [n_] measured cost per member (member order)
void EnsembleCore::fit_cost_model(const std::vector<double> &member_cost) {
  std::vector<int> varying;
  for (int r = 0; r < HX_NPARAM(B_); ++r) if (!row_uniform_[r]) varying.push_back(r);
  if (varying.empty() || varying.size() > 24 || n_ < 256) return;
  CostModel m;
  m.rows = varying;
  const size_t k = varying.size();
  m.cross = k <= 8;
  const size_t nt = cost_terms(k, m.cross);
  const size_t stride = std::max<size_t>(1, (size_t)n_ / 16384);
  std::vector<size_t> sample;
  for (size_t i = 0; i < (size_t)n_; i += stride) sample.push_back(i);
  if (sample.size() < 4 * nt) return;
  m.mean.assign(k, 0.0); m.sd.assign(k, 1.0);
  for (size_t a = 0; a < k; ++a) {
    const std::vector<double> &r = params_[(size_t)varying[a]];
    double mu = 0, var = 0;
    for (size_t i : sample) mu += r[i];
    mu /= (double)sample.size();
    for (size_t i : sample) var += (r[i] - mu) * (r[i] - mu);
    m.mean[a] = mu;
    m.sd[a] = var > 0 ? std::sqrt(var / (double)sample.size()) : 1.0;
  }
  // normal equations, Cholesky with a small ridge
  std::vector<double> A(nt * nt, 0.0), b(nt, 0.0), f(nt), x(k);
  for (size_t i : sample) {
    for (size_t a = 0; a < k; ++a) x[a] = (params_[(size_t)varying[a]][i] - m.mean[a]) / m.sd[a];
    cost_features(m, x.data(), f.data());
    for (size_t p = 0; p < nt; ++p) {
      b[p] += f[p] * member_cost[i];
      for (size_t q = 0; q <= p; ++q) A[p * nt + q] += f[p] * f[q];
    }
  }
  for (size_t p = 0; p < nt; ++p) A[p * nt + p] += 1e-9 * (double)sample.size();
  for (size_t p = 0; p < nt; ++p) {     // A = L L^T in the lower triangle
    for (size_t q = 0; q <= p; ++q) {
      double v = A[p * nt + q];
      for (size_t t = 0; t < q; ++t) v -= A[p * nt + t] * A[q * nt + t];
      if (p == q) { if (!(v > 0)) return; A[p * nt + p] = std::sqrt(v); }
      else A[p * nt + q] = v / A[q * nt + q];
    }
  }
  for (size_t p = 0; p < nt; ++p) {     // L y = b
    double v = b[p];
    for (size_t t = 0; t < p; ++t) v -= A[p * nt + t] * b[t];
    b[p] = v / A[p * nt + p];
  }
  for (size_t p = nt; p-- > 0;) {       // L^T beta = y
    double v = b[p];
    for (size_t t = p + 1; t < nt; ++t) v -= A[t * nt + p] * b[t];
    b[p] = v / A[p * nt + p];
  }
  m.beta = b;
  for (double v : m.beta) if (!std::isfinite(v)) return;
  std::lock_guard<std::mutex> lk(g_cost_mu);
  g_cost_models[cost_model_key(varying)] = std::move(m);
}

// -> predicted cost per member (member order), or false if the registry holds no model for this
// core's scenario / biome count / varying rows
bool EnsembleCore::predict_cost(const std::vector<int> &varying, std::vector<double> &out) const {
  if (!cost_model_ || varying.empty()) return false;
  seed_cost_models_once();
  CostModel m;
  {
    std::lock_guard<std::mutex> lk(g_cost_mu);
    auto it = g_cost_models.find(cost_model_key(varying));
    if (it == g_cost_models.end()) return false;
    m = it->second;
  }
  const size_t k = m.rows.size(), nt = cost_terms(k, m.cross);
  if (m.beta.size() != nt) return false;
  out.resize((size_t)n_);
  std::vector<double> f(nt), x(k);
  for (int i = 0; i < n_; ++i) {
    for (size_t a = 0; a < k; ++a) x[a] = (params_[(size_t)m.rows[a]][(size_t)i] - m.mean[a]) / m.sd[a];
    cost_features(m, x.data(), f.data());
    double v = 0;
    for (size_t p = 0; p < nt; ++p) v += m.beta[p] * f[p];
    out[(size_t)i] = v;
  }
  return true;
}

// Will run() take the two-wavefront flavour (hx_run_kernel<HX_B1W2>)?  One biome, no carbon
// tracking, at least hx_set_two_wave_from() members (default: more wavefronts than SIMDs).
bool EnsembleCore::two_wave_expected() const {
  const int w2_from = two_wave_from_ < 0 ? simds_ * HX_WAVE + 1 : two_wave_from_;
  // (members with their own ocean diffusivity: by default the one-wavefront kernel in two rounds --
  // since round 5 it requests a member's kernel-table entries ahead of their use, which the
  // two-wavefront flavour has no registers for: 131 072 such members 13.5 against 14.3 ms;
  // hx_set_two_wave_from with an explicit size still takes the flavour)
  if (two_wave_from_ < 0 && !row_uniform_[HXP_DIFF]) return false;
  return B_ == 1 && trk_iy() < 0 && !d_track_out_f_ && w2_from > 0 && n_ >= w2_from;
}

void EnsembleCore::assign_lanes() {
  member_of_lane_.resize((size_t)npad_);
  lane_of_member_.resize((size_t)n_);
  std::vector<int> order((size_t)n_);
  for (int i = 0; i < n_; ++i) order[(size_t)i] = i;
  std::vector<int> varying;
  for (int r = 0; r < HX_NPARAM(B_); ++r)
    if (!row_uniform_[r]) varying.push_back(r);
  if (sort_members_ && n_ > HX_WAVE && !varying.empty()) {
    radix_stable_sort(order, 0, (size_t)n_, params_[varying[0]], false);
    if (varying.size() > 1) {
      // second key: the second perturbed parameter, or -- with more than two -- the sum of
      // the standardised remaining ones (per-biome Q10s and warming factors act together)
      std::vector<double> q((size_t)n_, 0.0);
      for (size_t k = 1; k < varying.size(); ++k) {
        const std::vector<double> &r = params_[varying[k]];
        double mean = 0, var = 0;
        for (int i = 0; i < n_; ++i) mean += r[(size_t)i];
        mean /= n_;
        for (int i = 0; i < n_; ++i) var += (r[(size_t)i] - mean) * (r[(size_t)i] - mean);
        const double sd = std::sqrt(var / n_);
        if (sd > 0) for (int i = 0; i < n_; ++i) q[(size_t)i] += (r[(size_t)i] - mean) / sd;
      }
      int nbins = (int)std::lround(std::sqrt((double)n_ / HX_WAVE));
      nbins = std::max(1, nbins);
      const int per = (n_ + nbins - 1) / nbins;
      for (int b0 = 0; b0 < n_; b0 += per)
        radix_stable_sort(order, (size_t)b0, (size_t)std::min(n_, b0 + per), q, false);
    }
  }
  // the cost the lanes are ordered by: measured (a complete run of this core), or predicted by a
  // model fitted to an earlier core's measurements (see fit_cost_model), where the order matters:
  // more wavefronts than SIMDs
  const std::vector<double> *cost = nullptr;
  std::vector<double> predicted;
  lane_order_source_ = 0;
  if (sort_members_ && n_ > HX_WAVE && (int)lane_cost_.size() == n_) { cost = &lane_cost_; lane_order_source_ = 1; }
  else if (sort_members_ && calibrate_lanes_ && n_ / HX_WAVE > simds_ && predict_cost(varying, predicted)) {
    cost = &predicted; lane_order_source_ = 2;
  }
  if (cost) {
    // costliest first (ties keep the parameter order): wavefronts of members that
    // really take the same number of steps and stashes, the expensive ones dispatched first
    radix_stable_sort(order, 0, (size_t)n_, *cost, true);
    // Two resident wavefronts per SIMD (the two-wavefront flavour): the first `simds_` wavefronts
    // get a SIMD each, the next ones join them in dispatch order -- wavefront simds_ + k next to
    // wavefront k.  In descending order the costliest would share its SIMD with a median one and
    // a median one with the cheapest; with the second batch ASCENDING the costliest pairs with the
    // cheapest and every SIMD carries about twice the mean.
    // (Only then: the kernels that keep one wavefront per SIMD -- several biomes, carbon tracking,
    // hx_set_two_wave_from(0) -- run their second batch after the first, and costliest-first is
    // the order that ends the launch soonest there.)
    const int W = n_ / HX_WAVE;   // full wavefronts
    if (pair_costly_with_cheap_ && W > simds_ && two_wave_expected()) {
      const int hi = std::min(W, 2 * simds_);
      for (int a = simds_, b = hi - 1; a < b; ++a, --b)
        std::swap_ranges(order.begin() + (size_t)a * HX_WAVE, order.begin() + (size_t)(a + 1) * HX_WAVE,
                         order.begin() + (size_t)b * HX_WAVE);
    }
  }
  else if (sort_members_ && key_order_mode_ > 0 && n_ / HX_WAVE > simds_ && two_wave_expected()) {
    // (experiment switch HECTOR_AMD_KEY_ORDER, no measured cost yet: 1 = the second batch of
    // wavefronts reversed, so that the ends of the parameter order share SIMDs; 2 = the whole
    // order reversed as well -- the last parameter bins dispatched first)
    const int W = n_ / HX_WAVE, hi = std::min(W, 2 * simds_);
    auto swap_waves = [&](int a, int b) {
      std::swap_ranges(order.begin() + (size_t)a * HX_WAVE, order.begin() + (size_t)(a + 1) * HX_WAVE,
                       order.begin() + (size_t)b * HX_WAVE);
    };
    if (key_order_mode_ == 2) for (int a = 0, b = W - 1; a < b; ++a, --b) swap_waves(a, b);
    for (int a = simds_, b = hi - 1; a < b; ++a, --b) swap_waves(a, b);
  }
  for (int l = 0; l < npad_; ++l) member_of_lane_[(size_t)l] = order[(size_t)std::min(l, n_ - 1)];
  for (int l = 0; l < n_; ++l) lane_of_member_[(size_t)order[(size_t)l]] = l;
}

// reset(startDate) after a complete run: adopt the measured lane order (see the header)
void EnsembleCore::maybe_calibrate_lanes() {
  if (!calibrate_lanes_ || !sort_members_ || n_ <= HX_WAVE || !lane_cost_.empty() || !d_cost_) return;
  if (cost_from_iy_ != 0 || last_iy_ != scen_.ns() - 1 || last_run_pair_) return;
  bool adopt = true;
#ifndef HX_HOST_EMULATION
  {  // With no more wavefronts than SIMDs every wavefront has a SIMD to itself from start to end
    // and the launch lasts as long as its costliest one under any order: nothing to gain for THIS
    // core -- but its measurements still make the cost model that a larger core of the same study
    // orders its first run by (once per core: cost_fitted_).
    if (npad_ / HX_WAVE <= simds_ && !std::getenv("HECTOR_AMD_CALIBRATE_ALWAYS")) adopt = false;
  }
#endif
  if (!adopt && (cost_fitted_ || !cost_model_ || n_ < 4096)) return;
  sync();
  std::vector<double> tmp((size_t)npad_);
  check(hipMemcpy(tmp.data(), d_cost_, sizeof(double) * (size_t)npad_, hipMemcpyDeviceToHost), "lane cost");
  std::vector<double> member_cost((size_t)n_);
  for (int i = 0; i < n_; ++i) member_cost[(size_t)i] = tmp[(size_t)lane_of_member_[(size_t)i]];
  if (cost_model_) { fit_cost_model(member_cost); cost_fitted_ = true; }
  if (!adopt) return;
  lane_cost_ = std::move(member_cost);
  // nothing to gain if the order stays (e.g. every member alike)
  const std::vector<int> before = lane_of_member_;
  assign_lanes();
  if (lane_of_member_ == before) return;
  params_dirty_ = true;   // parameters, state and series move to their new lanes:
  need_spinup_ = true;    // upload and spin up again (prepare())
}

void EnsembleCore::upload_params() {
  const size_t np = (size_t)npad_;
  StageClock clk;
  assign_lanes();
  clk.lap("  assign_lanes");
  // Only what moved goes to the device: a row that was set since the last upload, and -- when
  // the lane order changed -- the rows that differ between members (a uniform row reads the same
  // in every order).  The calibration loop (new values of two or three parameters for every
  // member, reset, run) sends those rows instead of the whole table.
  const int nrows = HX_NPARAM(B_);
  const bool all = rows_all_dirty_ || (int)row_dirty_.size() != nrows;
  const bool order_changed = all || uploaded_order_ != member_of_lane_;
  std::vector<int> send;
  for (int r = 0; r < nrows; ++r)
    if (all || row_dirty_[(size_t)r] || (order_changed && !row_uniform_[(size_t)r])) send.push_back(r);
  std::vector<double> flat(np * send.size());
  for (size_t k = 0; k < send.size(); ++k) {
    const std::vector<double> &row = params_[(size_t)send[k]];
    double *dst = flat.data() + k * np;
    if (row_uniform_[(size_t)send[k]]) std::fill(dst, dst + np, row[0]);
    else for (size_t l = 0; l < np; ++l) dst[l] = row[(size_t)member_of_lane_[l]];
  }
  clk.lap("  flatten rows");
  if (order_changed)
    check(hipMemcpyAsync(d_lane_of_member_, lane_of_member_.data(), sizeof(int) * (size_t)n_,
                         hipMemcpyHostToDevice, stream_), "upload lane map");
  for (size_t k = 0; k < send.size();) {  // runs of adjacent rows travel together
    size_t e = k + 1;
    while (e < send.size() && send[e] == send[e - 1] + 1) ++e;
    check(hipMemcpyAsync(d_params_ + (size_t)send[k] * np, flat.data() + k * np,
                         sizeof(double) * np * (e - k), hipMemcpyHostToDevice, stream_), "upload params");
    k = e;
  }
  {
    std::vector<double> u((size_t)nrows);
    for (int r = 0; r < nrows; ++r) u[(size_t)r] = params_[(size_t)r][0];
    check(hipMemcpyAsync(d_uparams_, u.data(), sizeof(double) * u.size(), hipMemcpyHostToDevice,
                         stream_), "upload uniform params");
    check(hipStreamSynchronize(stream_), "sync uniform params");
  }
  check(hipStreamSynchronize(stream_), "sync params");
  clk.lap("  H2D params (waited)");
  row_dirty_.assign((size_t)nrows, 0);
  rows_all_dirty_ = false;
  order_changed_ = order_changed;
  uploaded_order_ = member_of_lane_;
  // DOECLIM convolution kernel: one shared table when every member has the same
  // diffusivity (wave-uniform scalar loads in the run kernel), else Ker[ns][npad]
  ker_per_member_ = !row_uniform_[HXP_DIFF];
  check(hipMemsetAsync(d_ker_, 0, sizeof(double) * ((size_t)scen_.ns() + 96) * (ker_per_member_ ? np : 1),
                       stream_), "zero ker");
  check(hx_launch_doeclim_kernel(d_params_ + (size_t)HXP_DIFF * np, d_ker_, scen_.ns(),
                                 ker_per_member_ ? npad_ : 1, ker_per_member_ ? npad_ : 1,
                                 stream_), "doeclim kernel table");
  check(hx_launch_derive(d_params_, d_derived_, d_ker_, ker_per_member_ ? 1 : 0, scen_.ns(), B_,
                         npad_, stream_), "derive kernel");
  // lane 0's derived constants behind the uniform parameter values (HxBuffers::uderived): rows
  // that are the same for every member are read through scalar loads
  check(hipMemcpy2DAsync(d_uparams_ + HX_NPARAM(B_), sizeof(double), d_derived_, sizeof(double) * np,
                         sizeof(double), (size_t)HX_NDERIVED(B_), hipMemcpyDeviceToDevice, stream_),
        "uniform derived constants");
  {
    bool lo_any = false;
    for (double v : params_[HXP_LO_RATIO]) if (v != 0.0) lo_any = true;
    kc_.con_mask = (kc_.con_mask & ~HXC_LO) | (lo_any ? HXC_LO : 0);
    kc_.trk_iy = trk_iy();
    upload_args();
  }
  params_dirty_ = false;
}

// SimpleNbox::prepareToRun's checks of the biome parameters (simpleNbox-runtime.cpp:94-107) and
// ForcingComponent::prepareToRun's of the forcing efficacies (forcing_component.cpp:291-293): the
// reference throws there, before anything runs -- so does this, naming the biome and the first
// member at fault (a configuration error, unlike the model errors a run reports as status flags).
void EnsembleCore::check_parameters() const {
  auto fail = [&](const char *what, int b, size_t i) {
    throw std::runtime_error(std::string(what) + " (biome '" + biome_names_[(size_t)b] + "', member " +
                             std::to_string(i) + ")");
  };
  for (int b = 0; b < B_; ++b) {
    const int r = HXP_NGLOBAL + b * HXPB_N;
    for (size_t i = 0; i < (size_t)n_; ++i) {
      const double beta = params_[r + HXPB_BETA][i], q10 = params_[r + HXPB_Q10][i],
                   fv = params_[r + HXPB_F_NPPV][i], fd = params_[r + HXPB_F_NPPD][i],
                   fl = params_[r + HXPB_F_LITTERD][i];
      if (!(beta >= 0.0)) fail("beta < 0", b, i);
      if (!(q10 > 0.0)) fail("q10_rh <= 0.0", b, i);
      if (!(fv >= 0.0)) fail("f_nppv <0", b, i);
      if (!(fd >= 0.0)) fail("f_nppd <0", b, i);
      if (!(fv + fd <= 1.0)) fail("f_nppv + f_nppd >1", b, i);
      if (!(fl >= 0.0 && fl <= 1.0)) fail("f_litterd <0 or >1", b, i);
    }
  }
  if (!(kc_.delta_ch4 >= -1 && kc_.delta_ch4 <= 1)) throw std::runtime_error("bad delta CH4 value");
  if (!(kc_.delta_n2o >= -1 && kc_.delta_n2o <= 1)) throw std::runtime_error("bad delta N2O value");
  if (!(kc_.delta_co2 >= -1 && kc_.delta_co2 <= 1)) throw std::runtime_error("bad delta CO2 value");
}

void EnsembleCore::prepare() {
  check(hipSetDevice(device_), "hipSetDevice");
  StageClock clk;
  if (layout_dirty_) { alloc_device(); clk.lap("alloc_device"); }
  if (shared_dirty_ || params_dirty_ || need_spinup_) prewarm_begin();
  if (shared_dirty_) {  // dated inputs changed: rebuild the per-year table
    build_shared();
    check(hipMemcpyAsync(d_shared_, shared_.data(), sizeof(double) * shared_.size(),
                         hipMemcpyHostToDevice, stream_), "upload shared");
    check(hipStreamSynchronize(stream_), "sync shared");
    shared_dirty_ = false;
    params_dirty_ = true;  // HxConst (constraint mask, preindustrial values) is re-uploaded
  }
  if (params_dirty_) {
    check_parameters();
    clk.lap("check_parameters");
    upload_params();
    clk.lap("upload_params");
    if (order_changed_)
      for (int k = 0; k < HXM_N; ++k) if (!member_series_[k].empty()) mseries_dirty_ = true;  // lanes have moved
    gas_dirty_ = true;
  }
  if (gas_dirty_ || (!gas_member_.empty() && !d_mseries_[HXM_N2O])) run_gas_kernel();
  if (mseries_dirty_) upload_member_series();
  clk.lap("gas kernel / member series");
  if (!need_spinup_) return;
  // Spinup is independent of every parameter that is not in the spinup set
  // (SURVEY 3f): if those rows are uniform, spin up one prototype wavefront and
  // broadcast its state.
  bool uniform = true;
  for (const ParamDef &d : kParams) {
    if (!d.spinup) continue;
    if (d.per_biome) {
      for (int b = 0; b < B_; ++b)
        if (!row_uniform_[HXP_NGLOBAL + b * HXPB_N + d.row]) uniform = false;
    } else if (!row_uniform_[d.row]) uniform = false;
  }
  const size_t np = (size_t)npad_;
  // A shared spinup whose inputs have not changed since it ran -- the parameters set in between
  // are ones the spinup does not see (kParams: spinup = false; the calibration loop's S, Q10,
  // beta, diffusivity ...) -- is not run again: every lane's post-spinup state, step count and
  // year-0 outputs are the prototype's, whatever the lane order, and still on the device.  Only
  // the members' own derive-time flags are applied anew.  (last_spinup_ms() then reports 0.)
  const bool reuse = spin_valid_ && uniform && spin_uniform_ && !std::getenv("HECTOR_AMD_ALWAYS_SPINUP");
  spin_uniform_ = uniform;
  if (reuse) {
    check(hipMemcpyAsync(d_state_, d_state_ + np * HX_NSTATE(B_), sizeof(double) * np * HX_NSTATE(B_),
                         hipMemcpyDeviceToDevice, stream_), "restore state");
    check(hipMemcpyAsync(d_status_, d_status_ + 2 * np, sizeof(unsigned) * np, hipMemcpyDeviceToDevice,
                         stream_), "restore status");
  } else {
    // (an all-member spinup wants the SIMDs, and is itself full-chip work; experiments:
    // HECTOR_AMD_PREWARM_STOP=spinup -- the loop ends ahead of a shared spinup too)
    if (!uniform || std::getenv("HECTOR_AMD_PREWARM_STOP")) prewarm_end();
    check(hipEventRecord(ev0_, stream_), "event");
    check(hx_launch_spinup(B_, d_args_, uniform ? 1 : npad_, d_spin_steps_, stream_), "spinup");
    check(hx_launch_alk(d_args_, uniform ? 1 : npad_, stream_), "alkalinity tuning");
    if (uniform) {
      check(hx_launch_broadcast(d_state_, HX_NSTATE(B_), npad_, stream_), "broadcast state");
      check(hx_launch_broadcast_u32(d_status_, npad_, stream_), "broadcast status");
      check(hx_launch_broadcast_u32(reinterpret_cast<unsigned *>(d_spin_steps_), npad_, stream_),
            "broadcast steps");
      for (int v = 0; v < HXO_NVAR; ++v)
        if (d_out_[v]) check(hx_launch_broadcast(d_out_[v], 1, npad_, stream_), "broadcast out");
    }
    // the spinup's own status, before the members' derive-time flags: what a later reuse starts from
    check(hipMemcpyAsync(d_status_ + 2 * np, d_status_, sizeof(unsigned) * np, hipMemcpyDeviceToDevice,
                         stream_), "keep spinup status");
  }
  check(hx_launch_or_flags(d_status_, d_derived_ + (size_t)HXD_FLAG * npad_, npad_, stream_),
        "derive flags");
  // snapshot of the post-spinup state for reset(startDate)
  if (!reuse)
    check(hipMemcpyAsync(d_state_ + np * HX_NSTATE(B_), d_state_, sizeof(double) * np * HX_NSTATE(B_),
                         hipMemcpyDeviceToDevice, stream_), "snapshot state");
  check(hipMemcpyAsync(d_status_ + np, d_status_, sizeof(unsigned) * np, hipMemcpyDeviceToDevice,
                       stream_), "snapshot status");
  if (reuse) {
    spin_ms_ = 0.0;
    clk.lap("post-spinup state reused");
  } else {
    check(hipEventRecord(ev1_, stream_), "event");
    check(hipEventSynchronize(ev1_), "spinup sync");
    float ms = 0;
    check(hipEventElapsedTime(&ms, ev0_, ev1_), "elapsed");
    spin_ms_ = ms;
    clk.lap("spinup + snapshot (waited)");
  }
  spin_valid_ = uniform;
  need_spinup_ = false;
  last_iy_ = 0;
  hist_valid_to_ = 0;
  dirty_from_iy_ = -1;
}

void EnsembleCore::reset(double date) {
  if (date < scen_.start) {  // core.cpp:511-549: rerun spinup
    need_spinup_ = true; spin_valid_ = false;
    last_iy_ = 0;
    dirty_from_iy_ = -1;
    return;
  }
  const int iy = (int)date - scen_.start;
  if (iy == 0 && !(need_spinup_ || layout_dirty_ || params_dirty_)) maybe_calibrate_lanes();
  if (need_spinup_ || layout_dirty_ || params_dirty_) {
    if (iy != 0) throw std::runtime_error("reset: the core has pending changes from date 0");
    prepare();
    return;
  }
  if (iy == last_iy_) return;
  if (iy > last_iy_) throw std::runtime_error("reset: date is after the current date");
  const size_t np = (size_t)npad_, rows = (size_t)HX_NSTATE(B_);
  sync();
  if (iy == 0) {  // the post-spinup snapshot
    check(hipMemcpyAsync(d_state_, d_state_ + np * rows, sizeof(double) * np * rows,
                         hipMemcpyDeviceToDevice, stream_), "restore state");
    check(hipMemcpyAsync(d_status_, d_status_ + np, sizeof(unsigned) * np, hipMemcpyDeviceToDevice,
                         stream_), "restore status");
  } else {
    if (!d_hist_ || iy > hist_valid_to_)
      throw std::runtime_error("reset: no state history for that date (enable_history() before "
                               "running; only reset(0) and reset(startDate) work without it)");
    check(hipMemcpyAsync(d_state_, d_hist_ + (size_t)iy * rows * np, sizeof(double) * np * rows,
                         hipMemcpyDeviceToDevice, stream_), "restore state from history");
    // a member that failed after `date` is healthy again at `date` (the reference's reset()+run()
    // recovers once the inputs are fixed)
    check(hipMemcpyAsync(d_status_, d_hist_status_ + (size_t)iy * np, sizeof(unsigned) * np,
                         hipMemcpyDeviceToDevice, stream_), "restore status from history");
  }
  last_iy_ = iy;
  cost_from_iy_ = -1;  // (the lane-cost sums restart with the next launch from startDate)
  if (dirty_from_iy_ >= iy) dirty_from_iy_ = -1;
}

// ---- clocks up while the host prepares (hx_prewarm_kernel, hx_kernels.hip) -----------------------
// Begun when prepare() finds an upload or a spinup to do (the GPU has been idle for a while and
// will be for milliseconds more), ended by run() right ahead of the run kernel; bounded by
// prewarm_ms_ on the device's own clock.  hx_set_prewarm / HECTOR_AMD_PREWARM_MS = 0: off.
void EnsembleCore::prewarm_begin() {
#ifndef HX_HOST_EMULATION
  if (prewarm_ms_ <= 0 || prewarm_on_) return;
  {  // only after an idle gap: a calibration loop's next iteration finds the clocks where it left them
    const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (last_gpu_activity_s_ >= 0 && now - last_gpu_activity_s_ < 0.005) return;
  }
  if (!aux_stream_) check(hipStreamCreateWithFlags(&aux_stream_, hipStreamNonBlocking), "prewarm stream");
  if (!d_prewarm_) check(hipMalloc((void **)&d_prewarm_, 64 + sizeof(double) * 1024), "prewarm flag");
  check(hipMemsetAsync(d_prewarm_, 0, 64, aux_stream_), "prewarm flag");
  // three small wavefronts a SIMD (16 registers each: every other launch still finds room; the
  // spinup and alkalinity kernels raise their own priority in the instruction arbiter).  One a SIMD
  // brings a 131 072-member first run to 1.06-1.09 of the steady state, three to 1.04
  // (profiles/r06_one_shot_prewarm.txt)
  int waves = 3 * simds_;
  if (const char *e = std::getenv("HECTOR_AMD_PREWARM_WAVES")) waves = std::max(1, std::min(8 * simds_, std::atoi(e)));
  // (experiments: HECTOR_AMD_PREWARM_MEM=1 -- the loop also sweeps the SST output array)
  const bool sweep = std::getenv("HECTOR_AMD_PREWARM_MEM") && d_out_[HXO_SST];
  check(hx_launch_prewarm(reinterpret_cast<const int *>(d_prewarm_), (long long)prewarm_ms_ * 100000ll,
                          reinterpret_cast<double *>(d_prewarm_ + 64), waves,
                          sweep ? d_out_[HXO_SST] : nullptr,
                          sweep ? (unsigned long long)scen_.ns() * (unsigned long long)npad_ : 0ull, aux_stream_),
        "prewarm kernel");
  prewarm_on_ = true;
#endif
}
// (on the core's stream: the flag rises when everything queued before it -- upload, spinup -- is done)
void EnsembleCore::prewarm_end() {
#ifndef HX_HOST_EMULATION
  if (!prewarm_on_) return;
  check(hipMemsetAsync(d_prewarm_, 1, 4, stream_), "prewarm stop");
  prewarm_on_ = false;
  last_run_prewarmed_ = true;
#endif
}

void EnsembleCore::run(double runtodate) {
  prepare();
  if (runtodate < 0.0) runtodate = scen_.end;
  if (runtodate > scen_.end)
    throw std::runtime_error("Requested run-to date is after the configured end date.");
  const int target = (int)runtodate - scen_.start;
  if (dirty_from_iy_ >= 0 && dirty_from_iy_ < last_iy_) {
    // rcpp_hector.cpp:160-166: a core that is not clean is reset before it runs
    const int back = (d_hist_ && dirty_from_iy_ <= hist_valid_to_) ? dirty_from_iy_ : 0;
    dirty_from_iy_ = -1;
    reset((double)(scen_.start + back));
  }
  dirty_from_iy_ = -1;
  if (target < last_iy_ + 1) return;  // core.cpp:455-460: models not run
  if (last_iy_ == 0) {  // the lane-cost sums of a pass over the scenario start here
    check(hipMemsetAsync(d_cost_, 0, sizeof(double) * (size_t)npad_, stream_), "zero lane cost");
    cost_from_iy_ = 0;
  }
  last_run_prewarmed_ = false;
  prewarm_end();
  check(hipEventRecord(ev0_, stream_), "event");
  // one launch for the whole span: wavefronts are independent (each does its own DOECLIM
  // history pass every HX_DBLK years), so there is no global barrier to wait at
  const bool hf = d_out_[HXO_HEATFLUX] || d_out_[HXO_FLUX_MIXED] || d_out_[HXO_FLUX_INTERIOR];
  const int con_mask = kc_.con_mask | member_con_mask_;
  bool ext = con_mask != 0;  // extended kernel: constraints or the extra diagnostics
  for (int v = HXO_NPP; v < HXO_NVAR; ++v) if (d_out_[v]) ext = true;
  for (int k = 0; k < HXM_N; ++k) if (d_mseries_[k]) ext = true;
  // (the extended kernel without the NBP machinery -- five solver variables, one interval set --
  // unless the scenario or a member holds an NBP constraint; HECTOR_AMD_EXTENDED_NBP=1: always
  // the one with it, the tests hold the two against each other)
  const bool force_nbp = getenv("HECTOR_AMD_EXTENDED_NBP") != nullptr;
  int con = ext ? (((con_mask & HXC_NBP) || force_nbp) ? 1 : -1) : 0;
  // (and without the constraint / warming-ratio / per-member-series code altogether when only a
  // diagnostic output made the run "extended": the plain kernel + diagnostics, CON = -2 --
  // 65 536 members with NPP recorded 6.75 -> 6.3 ms; HECTOR_AMD_EXTENDED_CONS=1: never, the tests
  // hold the two against each other)
  if (con == -1 && con_mask == 0 && !getenv("HECTOR_AMD_EXTENDED_CONS")) {
    bool any_ms = false;
    for (int k = 0; k < HXM_N; ++k) if (d_mseries_[k]) any_ms = true;
    // (built for one to four biomes and the two-wavefront flavour: hx_kernels.hip, launch_run_b)
    if (!any_ms && B_ >= 1 && B_ <= 4) con = -2;
  }
  if (d_track_out_f_) {
    if (con_mask & (HXC_CO2 | HXC_NBP))
      throw std::runtime_error("carbon tracking together with a CO2 or NBP constraint is not "
                               "supported (the constraint residual is an untracked source)");
    con = 2;
  }
  // small ensembles (too few wavefronts to fill the SIMDs): two wavefronts per 64 members
  // (it serves scenario-wide CO2 / tas / RF_tot / CH4 constraints -- concentration-driven runs -- with
  // shared diffusivity; an NBP constraint, constraint or emission series per member and a
  // land-ocean warming ratio take the extended run kernel, which is also taken for diagnostics
  // this one does not record itself)
  const int pair_cons = con_mask & (HXC_CO2 | HXC_TAS | HXC_FTOT | HXC_CH4);
  bool plain = (con_mask & ~(HXC_CO2 | HXC_TAS | HXC_FTOT | HXC_CH4)) == 0 && !d_track_out_f_ &&
               !(pair_cons && ker_per_member_);
  for (int k = 0; k < HXM_N; ++k) if (d_mseries_[k]) plain = false;
  bool pair = hx_pair_available() && B_ <= 4 && plain && n_ <= pair_max_members_;
  // (two to four biomes -- the land side owns the biome loops --: with shared diffusivity and without
  // the heat-flux sum)
  if (B_ > 1 && (ker_per_member_ || d_out_[HXO_HEATFLUX])) pair = false;
  for (int v = 0; v < HXO_NVAR && pair; ++v)
    if (d_out_[v]) {  // what hx_pair_kernel records
      static const int ok[] = {HXO_SST, HXO_TLAND, HXO_CO2, HXO_TGAV, HXO_NSTASH, HXO_RF_TOT, HXO_RF_CO2,
                               HXO_ATMOS_C, HXO_OCEAN_C, HXO_OCEAN_UPTAKE, HXO_HL_PH, HXO_LL_PH, HXO_CH4,
                               HXO_O3, HXO_NBP, HXO_VEG_C, HXO_DET_C, HXO_SOIL_C, HXO_PERMAFROST_C,
                               HXO_THAWED_C, HXO_EARTH_C, HXO_HEATFLUX, HXO_NPP, HXO_RH, HXO_RH_DET,
                               HXO_RH_SOIL, HXO_RH_CH4, HXO_F_FROZEN, HXO_GMST};
      if (std::find(std::begin(ok), std::end(ok), v) == std::end(ok)) pair = false;
      // (the NPP / RH diagnostics of a split core are sums of weighted per-biome parts: run kernels)
      if (B_ > 1 && (v == HXO_NPP || v == HXO_RH || v == HXO_RH_DET || v == HXO_RH_SOIL)) pair = false;
    }
  last_run_pair_ = pair;
  // more wavefronts than SIMDs: the one-biome kernel built for two resident wavefronts per SIMD
  const bool w2 = !pair && con <= 1 && two_wave_expected();
  last_run_w2_ = w2;
  last_run_con_ = con;
  if (pair)
    check(hx_launch_run_pair(d_args_, npad_, d_out_[HXO_HEATFLUX] != nullptr, ker_per_member_, last_iy_, target,
                             stream_, pair_cons != 0, B_), "run kernel (pair)");
  else
  check(hx_launch_run(B_, d_args_, npad_, hf || con == 2, ker_per_member_, con, last_iy_, target,
                      stream_, w2, simds_ / 4),
        "run kernel");
  check(hipEventRecord(ev1_, stream_), "event");
  run_timed_ = true;
  slr_valid_to_ = -1;
  if (d_hist_) hist_valid_to_ = target;  // slabs last_iy_+1..target were just (re)written
  last_iy_ = target;
}

void EnsembleCore::sync() {
  check(hipStreamSynchronize(stream_), "stream sync");
  last_gpu_activity_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (run_timed_) {
    float ms = 0;
    check(hipEventElapsedTime(&ms, ev0_, ev1_), "elapsed");
    run_ms_ = ms;
    run_timed_ = false;
  }
}

// Start and end of every wavefront of the last year-loop launch (HxBuffers::wave_clk), relative to
// the earliest start, in ticks of the constant 100 MHz clock.  -> wavefronts written (0 before the
// first run); the small-ensemble kernel has two per 64 members.
int EnsembleCore::wave_clock(long long *ticks, int cap) {
  if (!d_wave_clk_ || last_iy_ == 0) return 0;
  sync();
  const int nw = (npad_ / HX_WAVE) * (last_run_pair_ ? 2 : 1);
  std::vector<long long> t((size_t)nw * 2);
  check(hipMemcpy(t.data(), d_wave_clk_, sizeof(long long) * t.size(), hipMemcpyDeviceToHost), "wave clock");
  long long t0 = t[0];
  for (int w = 0; w < nw; ++w) t0 = std::min(t0, t[(size_t)2 * w]);
  const int n = std::min(nw, cap);
  for (int w = 0; w < n; ++w) { ticks[2 * w] = t[(size_t)2 * w] - t0; ticks[2 * w + 1] = t[(size_t)2 * w + 1] - t0; }
  return n;
}

// Variables answered on the host: scenario INPUT series, the member-independent gas cycles
// (N2O, halocarbons: run while the per-year table is built) and the forcings that are a
// member's scalar times a shared series.  Returns false if `capability` is not one of them;
// with out_host == nullptr only answers the question.
bool EnsembleCore::fetch_host(const std::string &capability_in, int year0, int year1,
                              double *out_host) {
  // the R accessors RF_CF4() ... return "Fadj<gas>", which the forcing component maps to
  // "RF_<gas>" (forcing_component.cpp:56-75, 577-583)
  const std::string capability = capability_in.compare(0, 4, "Fadj") == 0
                                     ? "RF_" + capability_in.substr(4) : capability_in;
  const int ns = scen_.ns();
  std::vector<double> ser;          // member-independent series [ns]
  const std::vector<double> *scale = nullptr;  // per-member factor (parameter row)
  bool relative = false, computed = false;     // relative to the forcing base year / needs a run
  auto input = [&](const std::string &sec, const std::string &key, double dflt) {
    return scen_.has_series(sec, key) ? scen_.series(sec, key) : std::vector<double>((size_t)ns, dflt);
  };
  for (const MemberSeriesDef &d : kMemberSeries)
    if (capability == d.name && !member_series_[d.k].empty()) {
      if (!out_host) return true;
      if (year0 < scen_.start || year1 > scen_.end || year1 < year0)
        throw std::runtime_error("fetchvars: dates must lie between startDate and endDate");
      std::copy(member_series_[d.k].begin() + (size_t)(year0 - scen_.start) * n_,
                member_series_[d.k].begin() + (size_t)(year1 - scen_.start + 1) * n_, out_host);
      return true;
    }
  for (const DatedDef &d : kDated)
    if (capability == d.name && scen_.has_series(d.sections[0], capability))
      ser = scen_.series(d.sections[0], capability);
  const std::string csuf = "_concentration", esuf = "_emissions";
  auto ends = [&](const std::string &suf) {
    return capability.size() > suf.size() &&
           capability.compare(capability.size() - suf.size(), suf.size(), suf) == 0;
  };
  if (capability == "N2O_concentration" && !gas_member_.empty() &&
      (gas_member_.count("N0") || gas_member_.count("TN2O0") || gas_member_.count("UC_N2O")))
    return false;  // differs between members: answered from the device series (fetchvars)
  if (ser.empty() && capability == "N2O_concentration") {
    ser.resize((size_t)ns);
    for (int iy = 0; iy < ns; ++iy) ser[(size_t)iy] = shared_[(size_t)iy * HXSH_STRIDE + HXSH_N2O];
    computed = true;
  }
  if (ser.empty() && (ends(csuf) || capability.compare(0, 3, "RF_") == 0)) {
    // a gas whose lifetime / efficiency differs between members: its recurrence per member, here
    // on the host for the one gas asked for (halocarbon_component.cpp:181-229)
    for (size_t h = 0; h < scen_.halocarbons.size(); ++h) {
      const Halocarbon &H = scen_.halocarbons[h];
      const bool conc = capability == H.name + csuf, rf = capability == "RF_" + H.name;
      if (!(conc || rf)) continue;
      if (!(gas_member_.count("tau_" + H.name) || (rf && (gas_member_.count("rho_" + H.name) ||
                                                           gas_member_.count("delta_" + H.name)))))
        break;
      if (!out_host) return true;
      if (shared_dirty_ || need_spinup_ || gas_dirty_)
        throw std::runtime_error("fetchvars: run the core after changing inputs");
      if (year0 < scen_.start || year1 > last_date() || year1 < year0)
        throw std::runtime_error("fetchvars: dates must lie between startDate and the current date");
      auto val = [&](const std::string &cap, double dflt, int i) {
        auto it = gas_member_.find(cap);
        return it == gas_member_.end() ? dflt : it->second[(size_t)i];
      };
      const std::string ckey = H.name + "_halocarbon." + H.name + "_constrain";
      const std::vector<double> *hc = scen_.has_series(H.name + "_halocarbon", H.name + "_constrain")
                                          ? &scen_.series(H.name + "_halocarbon", H.name + "_constrain") : nullptr;
      const int base = kc_.baseyear_idx;
      std::vector<double> v((size_t)ns);
      for (int i = 0; i < n_; ++i) {
        const double tau = val("tau_" + H.name, H.tau, i), rho = val("rho_" + H.name, H.rho, i),
                     delta = val("delta_" + H.name, H.delta, i);
        const double expfac = std::exp(-(1 / tau));
        double c = H.H0;
        for (int iy = 0; iy < ns; ++iy) {
          if (iy >= 1) {
            const double dconc = (H.emissions[(size_t)iy] / H.molarMass * 1.0) / (0.1 * 1.8);
            c = c * expfac + dconc * tau * (1.0 - expfac);
            if (hc && !std::isnan((*hc)[(size_t)iy])) c = (*hc)[(size_t)iy];
          }
          const double rf_un = rho * c;
          v[(size_t)iy] = conc ? c : (iy >= 1 ? rf_un + delta * rf_un : 0.0);
        }
        for (int y = year0; y <= year1; ++y) {
          const int iy = y - scen_.start;
          double x = v[(size_t)iy];
          if (rf) x = (iy >= base) ? x - v[(size_t)base] : 0.0;
          out_host[(size_t)(y - year0) * n_ + i] = x;
        }
      }
      return true;
    }
  }
  if (ser.empty() && (ends(csuf) || ends(esuf) || capability.compare(0, 3, "RF_") == 0)) {
    for (size_t h = 0; h < scen_.halocarbons.size(); ++h) {
      const Halocarbon &H = scen_.halocarbons[h];
      if (capability == H.name + csuf) { ser = halo_conc_[h]; computed = true; }
      else if (capability == H.name + esuf) ser = H.emissions;
      else if (capability == "RF_" + H.name) {  // halocarbon_component.cpp:205-229
        ser.resize((size_t)ns);
        for (int iy = 0; iy < ns; ++iy) {
          const double rf_un = H.rho * halo_conc_[h][(size_t)iy];
          ser[(size_t)iy] = rf_un + H.delta * rf_un;
        }
        ser[0] = 0.0;  // the component has not run at startDate
        relative = computed = true;
      }
    }
  }
  if (ser.empty() && capability.compare(0, 3, "RF_") == 0) {
    // forcing_component.cpp:430-487; alpha = aero_scalar, volscl = vol_scalar
    const auto bc = input("bc", "BC_emissions", 0), oc = input("oc", "OC_emissions", 0),
               so2 = input("so2", "SO2_emissions", 0), nh3 = input("nh3", "NH3_emissions", 0);
    const double aci_beta = 2.279759, s_BCOC = 111.05064063;
    const double s_SO2 = (260.34644166 * 1000) * (32.065 / 64.066);
    auto scaled = [&](const std::vector<double> &e, const char *rho_key) {
      const double rho = scen_.scalar("forcing", rho_key);
      ser.resize((size_t)ns);
      for (int iy = 0; iy < ns; ++iy) ser[(size_t)iy] = rho * e[(size_t)iy];
      scale = &params_[HXP_AERO];
    };
    if (capability == "RF_BC") scaled(bc, "rho_bc");
    else if (capability == "RF_OC") scaled(oc, "rho_oc");
    else if (capability == "RF_SO2") scaled(so2, "rho_so2");
    else if (capability == "RF_NH3") scaled(nh3, "rho_nh3");
    else if (capability == "RF_aci") {
      ser.resize((size_t)ns);
      for (int iy = 0; iy < ns; ++iy)
        ser[(size_t)iy] = -1 * aci_beta * std::log(1 + (so2[(size_t)iy] / s_SO2) +
                                                   ((bc[(size_t)iy] + oc[(size_t)iy]) / s_BCOC));
      scale = &params_[HXP_AERO];
    } else if (capability == "RF_vol") {
      ser = input("so2", "SV", 0);
      scale = &params_[HXP_VOL];
    } else if (capability == "RF_albedo") ser = input("simpleNbox", "RF_albedo", -0.2);
    else if (capability == "RF_misc") ser = input("forcing", "RF_misc", 0);
    if (!ser.empty()) relative = computed = true;
  }
  if (ser.empty()) return false;
  if (!out_host) return true;
  const int last = computed ? last_date() : scen_.end;
  if (computed && (shared_dirty_ || need_spinup_))
    throw std::runtime_error("fetchvars: run the core after changing inputs");
  if (year0 < scen_.start || year1 > last || year1 < year0)
    throw std::runtime_error("fetchvars: dates must lie between startDate and the current date");
  const int base = kc_.baseyear_idx;
  for (int y = year0; y <= year1; ++y) {
    const int iy = y - scen_.start;
    double *row = out_host + (size_t)(y - year0) * n_;
    for (int i = 0; i < n_; ++i) {
      const double f = scale ? (*scale)[(size_t)i] : 1.0;
      double v = f * ser[(size_t)iy];
      if (relative) v = (iy >= base) ? v - f * ser[(size_t)base] : 0.0;  // :507-527
      row[i] = v;
    }
  }
  return true;
}

namespace {
// component and unit of every output variable, as the reference's output stream prints them
// (src/csv_outputstream_visitor.cpp:126-365; unit names src/unitval.cpp:30-165)
struct VarInfo { const char *variable, *component, *units; };
const VarInfo kVarInfo[] = {
    {"NBP", "simpleNbox", "Pg C/yr"}, {"NPP", "simpleNbox", "Pg C/yr"}, {"RH", "simpleNbox", "Pg C/yr"},
    {"rh_det", "simpleNbox", "Pg C/yr"}, {"rh_soil", "simpleNbox", "Pg C/yr"},
    {"rh_ch4", "simpleNbox", "Pg C/yr"}, {"CO2_concentration", "simpleNbox", "ppmv CO2"},
    {"atmos_co2", "simpleNbox", "Pg C"}, {"atmos_c_residual", "simpleNbox", "Pg C"},
    {"veg_c", "simpleNbox", "Pg C"}, {"detritus_c", "simpleNbox", "Pg C"},
    {"soil_c", "simpleNbox", "Pg C"}, {"permafrost_c", "simpleNbox", "Pg C"},
    {"thawedp_c", "simpleNbox", "Pg C"}, {"f_frozen", "simpleNbox", "(unitless)"},
    {"earth_c", "simpleNbox", "Pg C"}, {"detritus_tempfert", "simpleNbox", "(unitless)"},
    {"soil_tempfert", "simpleNbox", "(unitless)"},
    {"global_tas", "temperature", "degC"}, {"gmst", "temperature", "degC"},
    {"heatflux_mixed", "temperature", "W/m2"}, {"heatflux_interior", "temperature", "W/m2"},
    {"heatflux", "temperature", "W/m2"}, {"land_tas", "temperature", "degC"},
    {"sst", "temperature", "degC"}, {"ocean_tas", "temperature", "degC"},
    {"HL_ocean_uptake", "ocean", "Pg C/yr"}, {"LL_ocean_uptake", "ocean", "Pg C/yr"},
    {"DO_ocean_c", "ocean", "Pg C"}, {"HL_ocean_c", "ocean", "Pg C"}, {"IO_ocean_c", "ocean", "Pg C"},
    {"LL_ocean_c", "ocean", "Pg C"}, {"ML_ocean_c", "ocean", "Pg C"},
    {"HL_DIC", "ocean", "umol/kg"}, {"LL_DIC", "ocean", "umol/kg"}, {"DIC", "ocean", "umol/kg"},
    {"HL_downwelling", "ocean", "Pg C/yr"}, {"ocean_uptake", "ocean", "Pg C/yr"},
    {"HL_OmegaAr", "ocean", "(unitless)"}, {"LL_OmegaAr", "ocean", "(unitless)"},
    {"HL_OmegaCa", "ocean", "(unitless)"}, {"LL_OmegaCa", "ocean", "(unitless)"},
    {"HL_PCO2", "ocean", "uatm"}, {"LL_PCO2", "ocean", "uatm"}, {"PCO2", "ocean", "uatm"},
    {"HL_pH", "ocean", "pH"}, {"LL_pH", "ocean", "pH"}, {"pH", "ocean", "pH"},
    {"HL_sst", "ocean", "degC"}, {"LL_sst", "ocean", "degC"}, {"ocean_c", "ocean", "Pg C"},
    {"HL_CO3", "ocean", "umol/kg"}, {"LL_CO3", "ocean", "umol/kg"}, {"CO3", "ocean", "umol/kg"},
    {"HL_Revelle", "ocean", "(unitless)"}, {"LL_Revelle", "ocean", "(unitless)"},
    {"ocean_timesteps", "ocean", "(unitless)"}, {"timesteps", "ocean", "(unitless)"},
    {"solver_steps", "carbon-cycle-solver", "(unitless)"},
    {"slr", "slr", "cm"}, {"slr_no_ice", "slr", "cm"}, {"sl_rc", "slr", "cm/yr"},
    {"sl_rc_no_ice", "slr", "cm/yr"},
    {"O3_concentration", "ozone", "DU O3"}, {"TAU_OH", "OH", "Years"},
    {"CH4_concentration", "CH4", "ppbv CH4"}, {"N2O_concentration", "N2O", "ppbv N2O"},
};
}  // namespace

// getunits(var) of the R package (R/units.R) and the component that owns the variable
void EnsembleCore::var_info(const std::string &capability_in, std::string *component,
                            std::string *units) const {
  std::string cap = capability_in;
  const size_t dot = cap.find('.');
  if (dot != std::string::npos) cap = cap.substr(dot + 1);  // "<biome>.<variable>"
  if (cap.compare(0, 4, "Fadj") == 0) cap = "RF_" + cap.substr(4);
  auto set = [&](const std::string &c, const std::string &u) {
    if (component) *component = c;
    if (units) *units = u;
  };
  for (const VarInfo &v : kVarInfo) if (cap == v.variable) return set(v.component, v.units);
  if (cap.compare(0, 3, "RF_") == 0 && cap.find("_constrain") == std::string::npos)
    return set("forcing", "W/m2");
  for (const ParamDef &d : kParams)
    if (cap == d.name) {
      const bool temp = d.row == HXP_S || d.row == HXP_DIFF || d.row == HXP_QCO2 || d.row == HXP_LO_RATIO;
      const bool forc = d.row == HXP_AERO || d.row == HXP_VOL;
      const bool ocean = !d.per_biome && d.row >= HXP_TT && d.row <= HXP_PRE_ID;
      return set(d.per_biome ? "simpleNbox" : temp ? "temperature" : forc ? "forcing" : ocean ? "ocean" : "simpleNbox",
                 d.units);
    }
  {
    std::string u;
    const std::string sec = shared_param_section(scen_, cap, &u);
    if (!sec.empty()) return set(sec, u);
  }
  for (const DatedDef &d : kDated) if (cap == d.name) return set(d.sections[0], d.units);
  for (const Halocarbon &h : scen_.halocarbons) {
    if (cap == h.name + "_emissions") return set(h.name + "_halocarbon", "Gg");
    if (cap == h.name + "_concentration" || cap == h.name + "_constrain")
      return set(h.name + "_halocarbon", "pptv");
  }
  throw std::runtime_error("Caller is requesting unknown variable: " + capability_in);
}

void EnsembleCore::set_tracking_date(int year) {
  if (year <= 0) year = 0;
  const int before = trk_iy();
  tracking_year_ = year;
  if (trk_iy() == before) return;
  layout_dirty_ = true;
  need_spinup_ = true; spin_valid_ = false;
  last_iy_ = 0;
}

std::vector<std::string> EnsembleCore::tracking_pools() const {
  // pool names as the fluxpools carry them (simpleNbox.cpp:45-79, ocean_component.cpp:246-258)
  std::vector<std::string> n = {"atmos_co2", "earth_c"};  // D_ATMOSPHERIC_CO2, simpleNbox.cpp:65
  for (int b = 0; b < B_; ++b) {
    const std::string pre = (B_ == 1 && biome_names_[0] == "global") ? "" : biome_names_[(size_t)b] + ".";
    for (const char *k : {"veg_c", "detritus_c", "soil_c", "permafrost_c", "thawedp_c"}) n.push_back(pre + k);
  }
  for (const char *k : {"HL", "LL", "intermediate", "deep"}) n.push_back(k);
  return n;
}

void EnsembleCore::tracking_data(int member, int year0, int year1, double *values,
                                 double *fractions, unsigned long long *source_masks) {
  if (!d_track_out_f_) throw std::runtime_error("carbon tracking is off (set trackingDate first)");
  if (member < 0 || member >= n_) throw std::runtime_error("tracking_data: bad member index");
  if (year0 < tracking_year_ || year1 > last_date() || year1 < year0)
    throw std::runtime_error("tracking_data: dates must lie between trackingDate and the current date");
  sync();
  const size_t TP = (size_t)(2 + 5 * B_ + 4);
  const size_t vr = (size_t)hx_track_value_rows(B_), W = vr / TP - 1;
  const size_t k0 = (size_t)(year0 - tracking_year_), ny = (size_t)(year1 - year0 + 1);
  const int lane = lane_of_member_[(size_t)member];
  // this member's lane of its block's tiles ([block][slot][row][64]; per year: TP values, then the
  // TP mask words as bit patterns -- W = 1, or 2 for more than 64 pools, per pool, low words first)
  const size_t blk = (size_t)lane / 64, l = (size_t)lane % 64;
  const size_t slots = (size_t)(scen_.ns() - trk_iy() + 1);
  std::vector<double> col(ny * vr);
  // (the block's slots are contiguous: rows of consecutive years follow each other)
  check(hipMemcpy2D(fractions, sizeof(double),
                    d_track_out_f_ + (blk * slots + k0 + 1) * TP * TP * 64 + l, 64 * sizeof(double),
                    sizeof(double), ny * TP * TP, hipMemcpyDeviceToHost), "tracking fractions");
  check(hipMemcpy2D(col.data(), sizeof(double),
                    d_track_out_v_ + (blk * slots + k0 + 1) * vr * 64 + l, 64 * sizeof(double),
                    sizeof(double), ny * vr, hipMemcpyDeviceToHost), "tracking values");
  for (size_t y = 0; y < ny; ++y) {
    std::memcpy(values + y * TP, col.data() + y * vr, sizeof(double) * TP);
    if (source_masks)  // which sources a pool's map holds (a held source can have fraction 0)
      for (size_t p = 0; p < TP; ++p)
        for (size_t w = 0; w < W; ++w)
          std::memcpy(source_masks + (y * TP + p) * W + w, col.data() + y * vr + TP * (1 + w) + p,
                      sizeof(double));
  }
}

std::string EnsembleCore::run_name() const { return scen_.text("core", "run_name", ""); }

bool EnsembleCore::host_output(const std::string &capability) {
  return fetch_host(capability, 0, 0, nullptr);
}

// Diagnostics derived on the device from recorded outputs; the result is a lane-ordered
// [ny][npad] block in d_diag_.
void EnsembleCore::compute_derived(const std::string &capability, int iy0, int ny) {
  const DerivedDef *d = derived_of(capability);
  auto need = [&](const char *name) -> const double * {
    const int v = out_index(name);
    if (!d_out_[v])
      throw std::runtime_error("variable " + capability + " needs " + name +
                               ": enable it (or " + capability + ") with set_outputs()");
    return d_out_[v];
  };
  const size_t np = (size_t)npad_, ns = (size_t)scen_.ns();
  if (d->kind >= DK_SLR) {
    if (!d_slr_) check(hipMalloc(&d_slr_, sizeof(double) * 4 * ns * np), "hipMalloc slr");
    if (slr_valid_to_ != last_iy_) {
      check(hipMemsetAsync(d_slr_, 0, sizeof(double) * 4 * ns * np, stream_), "zero slr");
      check(hx_launch_slr(need("global_tas"), npad_, scen_.start, last_iy_, d_slr_, ns * np, stream_),
            "slr kernel");
      slr_valid_to_ = last_iy_;
    }
    check(hipMemcpyAsync(d_diag_, d_slr_ + (size_t)(d->kind - DK_SLR) * ns * np + (size_t)iy0 * np,
                         sizeof(double) * (size_t)ny * np, hipMemcpyDeviceToDevice, stream_), "slr rows");
    return;
  }
  HxDiagArgs a{};
  a.npad = npad_; a.iy0 = iy0; a.ny = ny; a.base_idx = kc_.baseyear_idx;
  a.shared = d_shared_;
  a.n2o_members = d_mseries_[HXM_N2O];
  a.lo_ratio = d_params_ + (size_t)HXP_LO_RATIO * np;
  a.sqrtN0 = kc_.sqrtN0; a.sqrtM0 = kc_.sqrtM0; a.M0f = kc_.M0f;
  a.delta_n2o = kc_.delta_n2o; a.delta_ch4 = kc_.delta_ch4;
  const bool hl = d->box == 0;
  // oceanbox.cpp:97-99 (deltaT), ocean_component.cpp:233-238 (box volumes)
  a.deltaT = hl ? -16.4 : 2.9;
  const double area = 3.6e14;
  a.inv_vol = 1.0 / (area * (hl ? 0.15 : 1 - 0.15) * 100.0);
  switch (d->kind) {
    case HXG_TEMP: a.sst = need("sst"); break;
    case HXG_DIC: a.sst = need("sst"); a.carbon = need(hl ? "HL_ocean_c" : "LL_ocean_c"); break;
    case HXG_REVELLE: a.carbon = need(hl ? "HL_ocean_c" : "LL_ocean_c");  // fall through
    case HXG_CO3: case HXG_OMEGA_AR: case HXG_OMEGA_CA:
      a.sst = need("sst"); a.ph = need(hl ? "HL_pH" : "LL_pH");
      a.pco2 = need(hl ? "HL_PCO2" : "LL_PCO2");
      break;
    case HXG_OCEAN_TAS: a.sst = d_out_[HXO_SST]; a.tgav = need("global_tas"); break;
    case HXG_RF_N2O: a.co2 = need("CO2_concentration"); a.ch4 = need("CH4_concentration"); break;
    case HXG_RF_CH4: case HXG_RF_H2O: a.ch4 = need("CH4_concentration"); break;
    case HXG_RF_O3: a.o3 = need("O3_concentration"); break;
    default: throw std::runtime_error("unknown diagnostic");
  }
  check(hx_launch_diag(d->kind, a, d_diag_, stream_), "diag kernel");
}

// A disabled component registers no capabilities: asking for its variables fails like
// Core::sendMessage does for an unknown capability (core.cpp:716-778).
void EnsembleCore::check_component_enabled(const std::string &capability_in) const {
  const std::string cap = capability_in.compare(0, 4, "Fadj") == 0 ? "RF_" + capability_in.substr(4)
                                                                   : capability_in;
  std::string sec;
  const bool so2_off = component_disabled("so2");
  if (cap == "RF_BC" || cap == "RF_OC" || cap == "RF_SO2" || cap == "RF_NH3" || cap == "RF_aci") {
    for (const char *s : {"bc", "oc", "so2", "nh3"}) if (component_disabled(s)) sec = s;
  } else if (cap == "RF_vol" && so2_off) sec = "so2";
  else if ((cap == "O3_concentration" || cap == "RF_O3_trop") && component_disabled("ozone")) sec = "ozone";
  else if ((cap == "CH4_concentration" || cap == "RF_CH4" || cap == "RF_H2O_strat") && component_disabled("CH4")) sec = "CH4";
  else if ((cap == "N2O_concentration" || cap == "RF_N2O") && component_disabled("N2O")) sec = "N2O";
  else if ((cap == "RF_CO2" || cap == "RF_CH4" || cap == "RF_H2O_strat" || cap == "RF_N2O") &&
           (component_disabled("N2O") || component_disabled("CH4")))
    sec = component_disabled("N2O") ? "N2O" : "CH4";   // (not computed: forcing_component.cpp:315-317)
  else if (cap == "TAU_OH" && component_disabled("OH")) sec = "OH";
  else if ((cap == "slr" || cap == "sl_rc" || cap == "slr_no_ice" || cap == "sl_rc_no_ice") &&
           component_disabled("slr")) sec = "slr";
  else
    for (const Halocarbon &h : scen_.halocarbons)
      if ((cap == "RF_" + h.name || cap == h.name + "_concentration") &&
          component_disabled(h.name + "_halocarbon")) sec = h.name + "_halocarbon";
  if (!sec.empty())
    throw std::runtime_error("Caller is requesting unknown variable: " + capability_in +
                             " (component [" + sec + "] is disabled)");
}

// row_pitch: doubles between the rows (years) of out_host; 0 or n_members = contiguous.  A shard
// of a Fleet writes its block of members straight into the whole ensemble's [year][member] array.
void EnsembleCore::fetchvars(const std::string &capability, int year0, int year1,
                             double *out_host, size_t row_pitch) {
  check_component_enabled(capability);
  if (row_pitch == (size_t)n_) row_pitch = 0;
  if (row_pitch) {
    static const char *const whole[] = {"pH", "PCO2", "DIC", "CO3", "ML_ocean_c"};
    bool on_host = fetch_host(capability, year0, year1, nullptr);
    for (const char *w : whole) if (capability == w) on_host = true;
    if (on_host) {  // answered or combined on the host: through a contiguous block
      const size_t ny = (size_t)(year1 - year0 + 1);
      std::vector<double> part(ny * (size_t)n_);
      fetchvars(capability, year0, year1, part.data(), 0);
      for (size_t y = 0; y < ny; ++y)
        std::memcpy(out_host + y * row_pitch, part.data() + y * (size_t)n_, sizeof(double) * (size_t)n_);
      return;
    }
  }
  if (fetch_host(capability, year0, year1, out_host)) return;
  {  // whole-surface values: area-weighted low/high latitude (ocean_component.cpp:466-503)
    static const char *const combos[][3] = {{"pH", "LL_pH", "HL_pH"}, {"PCO2", "LL_PCO2", "HL_PCO2"},
                                            {"DIC", "LL_DIC", "HL_DIC"}, {"CO3", "LL_CO3", "HL_CO3"},
                                            {"ML_ocean_c", "LL_ocean_c", "HL_ocean_c"}};
    for (auto &c : combos)
      if (capability == c[0]) {
        const size_t cnt = (size_t)(year1 - year0 + 1) * (size_t)n_;
        std::vector<double> hl(cnt);
        fetchvars(c[1], year0, year1, out_host);
        fetchvars(c[2], year0, year1, hl.data());
        const bool sum = capability == "ML_ocean_c";
        const double part_high = 0.15, part_low = 1 - 0.15;
        for (size_t i = 0; i < cnt; ++i)
          out_host[i] = sum ? out_host[i] + hl[i] : part_low * out_host[i] + part_high * hl[i];
        return;
      }
  }
  const bool n2o_members = capability == "N2O_concentration" && d_mseries_[HXM_N2O];
  const DerivedDef *dd = derived_of(capability);
  const int v = (dd || n2o_members) ? -1 : out_index(capability);
  if (!dd && !n2o_members && !d_out_[v])
    throw std::runtime_error("variable " + capability + " was not enabled with set_outputs()");
  if (year0 < scen_.start || year1 > last_date() || year1 < year0)
    throw std::runtime_error("fetchvars: dates must lie between startDate and the current date");
  sync();
  const int iy0 = year0 - scen_.start, ny = year1 - year0 + 1;
  const size_t need = (size_t)ny * (size_t)n_;
  if (need > gather_cap_) {
    if (d_gather_) (void)hipFree(d_gather_);
    d_gather_ = nullptr;
    check(hipMalloc(&d_gather_, sizeof(double) * need), "hipMalloc gather");
    gather_cap_ = need;
  }
  const double *src = nullptr;
  if (dd) {
    const size_t dneed = (size_t)ny * (size_t)npad_;
    if (dneed > diag_cap_) {
      if (d_diag_) (void)hipFree(d_diag_);
      d_diag_ = nullptr;
      check(hipMalloc(&d_diag_, sizeof(double) * dneed), "hipMalloc diag");
      diag_cap_ = dneed;
    }
    compute_derived(capability, iy0, ny);
    src = d_diag_;
  } else if (n2o_members) {
    src = d_mseries_[HXM_N2O] + (size_t)iy0 * npad_;
  } else {
    src = d_out_[v] + (size_t)iy0 * npad_;
  }
  check(hx_launch_gather(src, d_lane_of_member_, d_gather_, n_, npad_, ny, stream_), "gather");
  if (row_pitch)
    check(hipMemcpy2DAsync(out_host, sizeof(double) * row_pitch, d_gather_, sizeof(double) * (size_t)n_,
                           sizeof(double) * (size_t)n_, (size_t)ny, hipMemcpyDeviceToHost, stream_), "fetch");
  else
    check(hipMemcpyAsync(out_host, d_gather_, sizeof(double) * need, hipMemcpyDeviceToHost, stream_),
          "fetch");
  check(hipStreamSynchronize(stream_), "fetch sync");
}

const double *EnsembleCore::device_var(const std::string &capability, int *npad) const {
  const int v = out_index(capability);
  if (!d_out_[v])
    throw std::runtime_error("variable " + capability + " was not enabled with set_outputs()");
  if (npad) *npad = npad_;
  return d_out_[v];
}

void EnsembleCore::stats_device(const std::string &capability, int year0, int year1,
                                double *d_stats) {
  stats_async(capability, year0, year1, d_stats);
  check(hipStreamSynchronize(stream_), "stats sync");
}

void EnsembleCore::stats_async(const std::string &capability, int year0, int year1,
                               double *d_stats) {
  const int v = out_index(capability);
  if (!d_out_[v])
    throw std::runtime_error("variable " + capability + " was not enabled with set_outputs()");
  if (year0 < scen_.start || year1 > last_date() || year1 < year0)
    throw std::runtime_error("stats: dates must lie between startDate and the current date");
  check(hx_launch_stats(d_out_[v], n_, npad_, year0 - scen_.start, year1 - year0 + 1, d_stats,
                        stream_), "stats kernel");
}

void EnsembleCore::status(unsigned *out_host) {
  prepare();
  sync();
  std::vector<unsigned> tmp((size_t)npad_);
  check(hipMemcpy(tmp.data(), d_status_, sizeof(unsigned) * (size_t)npad_, hipMemcpyDeviceToHost),
        "status");
  for (int i = 0; i < n_; ++i) out_host[i] = tmp[(size_t)lane_of_member_[(size_t)i]];
}

void EnsembleCore::state_row(int row, double *out_host) {
  prepare();
  sync();
  if (row < 0 || row >= HX_NSTATE(B_)) throw std::runtime_error("state_row: bad row");
  std::vector<double> tmp((size_t)npad_);
  check(hipMemcpy(tmp.data(), d_state_ + (size_t)row * npad_, sizeof(double) * (size_t)npad_,
                  hipMemcpyDeviceToHost), "state row");
  for (int i = 0; i < n_; ++i) out_host[i] = tmp[(size_t)lane_of_member_[(size_t)i]];
}

int EnsembleCore::spinup_steps(int member) {
  prepare();
  sync();
  int v = 0;
  if (member < 0 || member >= n_) throw std::runtime_error("spinup_steps: bad member index");
  check(hipMemcpy(&v, d_spin_steps_ + lane_of_member_[(size_t)member], sizeof(int),
                  hipMemcpyDeviceToHost), "steps");
  return v;
}

}  // namespace hx
