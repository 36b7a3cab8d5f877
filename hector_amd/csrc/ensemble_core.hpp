// ensemble_core.hpp -- host runtime for an N-member Hector ensemble on one GPU.
//
// Mirrors the reference's Core for the year-loop path (inst/include/core.hpp:37-114,
// src/core.cpp:302-549): init from an INI/scenario, setData through capability
// strings (incl. "<biome>.<var>"), prepareToRun (incl. spinup), run(runToDate)
// callable repeatedly with increasing dates, reset(date), getData -- except that
// every parameter and every result carries a member axis.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "hx_layout.h"
#include "hx_scenario.hpp"

namespace hx {

// The process-wide registry of lane-cost models (ensemble_core.cpp, fit_cost_model) to / from a
// text file: -> number of models written / read, -1 if the file cannot be opened.
int hx_cost_models_export_file(const char *path);
int hx_cost_models_load_file(const char *path);

class EnsembleCore {
 public:
  EnsembleCore(const std::string &scenario_path, int n_members, int device);
  ~EnsembleCore();
  EnsembleCore(const EnsembleCore &) = delete;
  EnsembleCore &operator=(const EnsembleCore &) = delete;

  int n_members() const { return n_; }
  int n_biomes() const { return B_; }
  int start_date() const { return scen_.start; }
  int end_date() const { return scen_.end; }
  int last_date() const { return scen_.start + last_iy_; }
  const std::vector<std::string> &biomes() const { return biome_names_; }

  // SETDATA for a model parameter: nvalues == 1 (all members) or n_members.
  // capability = reference capability string (component_data.hpp), optionally
  // "<biome>.<capability>".  units: "" / nullptr skips the check, otherwise it
  // must equal the reference's unit string for that variable (unitval.cpp).
  // Marks the core dirty from date 0 (spinup reruns if a spinup-relevant
  // parameter changed), like R/messages.R:107-140.
  void setvar(const std::string &capability, const double *values, int nvalues,
              const char *units);
  // current value of a parameter for every member (GETDATA without date)
  void getvar(const std::string &capability, double *out) const;

  // split_biome (R/biome.R:61-130): replace the single biome by n new ones,
  // pools and npp_flux0 partitioned by the given fractions (nullptr = equal),
  // other parameters copied; forces a re-spinup.
  void split_biome(const std::vector<std::string> &names, const double *fveg,
                   const double *fdet, const double *fsoil, const double *fpf,
                   const double *fnpp);
  void split_biome_of(const std::string &old_biome, const std::vector<std::string> &names,
                      const double *fveg, const double *fdet, const double *fsoil,
                      const double *fpf, const double *fnpp);
  // create_biome_impl / delete_biome_impl / rename_biome (src/rcpp_hector.cpp; SimpleNbox::
  // createBiome, deleteBiome, renameBiome): at most HX_BDYN biomes (1..HX_MAXB run unrolled
  // kernels, more the looped ones)
  void create_biome(const std::string &biome);
  void delete_biome(const std::string &biome);
  void rename_biome(const std::string &oldname, const std::string &newname);
  int biome_index(const std::string &biome) const;

  // which output variables are recorded per year (capability strings);
  // sst and land_tas are always recorded (the model needs their history).
  void set_outputs(const std::vector<std::string> &capabilities);
  // lane assignment: sort members by their perturbed parameters so that the lanes of a
  // wavefront follow similar solver schedules (default on; results do not depend on it)
  void set_member_sorting(bool on);
  // Lane order by MEASURED cost: the run kernel adds up what every member's solver did (dopri5
  // steps, stashes); after a run that covered startDate..endDate, the next reset(startDate)
  // reorders the lanes by it, costliest wavefronts first (they are dispatched first, so an
  // ensemble of more wavefronts than SIMDs ends on cheap ones), and spins up again.  Only for
  // ensembles of more wavefronts than the GPU has SIMDs (otherwise the launch lasts as long as
  // its costliest wavefront under any order).  Results do not depend on the order.  Default on; a parameter change falls back to the parameter key
  // until the next complete run.
  void set_lane_calibration(bool on) { calibrate_lanes_ = on; }
  bool lanes_calibrated() const { return !lane_cost_.empty(); }
  // what the lanes of the last upload are ordered by: 0 the parameter key, 1 this core's measured
  // cost, 2 the cost a model predicts that was fitted to an earlier core's measurements (same
  // scenario table, biome count and varying rows, this process; ensemble_core.cpp: fit_cost_model)
  int lane_order_source() const { return lane_order_source_; }
  void set_cost_model(bool on) { cost_model_ = on; }
  // the chip's clocks kept up while prepare() uploads and spins up after an idle gap, so that the
  // run kernel behind it does not start at ramping clocks (ensemble_core.cpp: prewarm_begin);
  // ms: the most the busy loop may last on the device's own clock, 0 = off
  void set_prewarm(int ms) { prewarm_ms_ = ms < 0 ? 0 : (ms > 200 ? 200 : ms); }
  int prewarm_ms() const { return prewarm_ms_; }
  bool last_run_prewarmed() const { return last_run_prewarmed_; }
  // keep every year's component state in HBM (272 B per member-year for one biome) so that
  // reset(date) can go back to any computed year, like the reference's tseries records
  void enable_history(bool on);
  // keep what the reference's output stream sees after every spinup step (its "spinup = 1" rows,
  // csv_outputstream_visitor.cpp:86-95): the carbon-cycle variables, max_spinup x 21 x 8 B per member
  void enable_spinup_record(bool on);
  static const std::vector<std::string> &spinup_record_vars();
  // values[step * nvars + v] for steps 1..spinup_steps(member); returns the step count
  int spinup_record(int member, double *values, int max_steps);
  // SETDATA with dates for a scenario input series (emissions, SV, RF_albedo...): the same
  // new values for every member.  Like R/messages.R:107-140 the core becomes dirty from
  // min(year) - 1; the next run() resets there (needs the history) or to startDate.
  void setvar_dated(const std::string &capability, const int *years, const double *values,
                    int n, const char *units);
  void lane_of_member(int *out);
  // SETDATA with dates and a different value for every member (ffi_emissions, luc_emissions,
  // daccs_uptake, luc_uptake, CH4_emissions): values[i * n_members + member] for years[i] --
  // the reference's pattern of re-running a period with new emissions per run
  // (vignettes/ex_hector_apply.Rmd), for all members at once.
  void setvar_dated_members(const std::string &capability, const int *years, const double *values,
                            int nyears, const char *units);
  static const char *const *output_capabilities(int *count);

  void reset(double date);      // Core::reset: date < startDate => redo spinup
  void run(double runtodate);   // Core::run; < 0 => endDate.  Asynchronous.
  void sync();                  // wait for the stream
  // GETDATA with dates: out[(year - year0) * n + member]
  void fetchvars(const std::string &capability, int year0, int year1, double *out_host,
                 size_t row_pitch = 0);
  bool host_output(const std::string &capability);
  // Carbon tracking (Core::trackingDate, get_tracking_data): origins of every pool's carbon from
  // `year` on.  year <= 0 or beyond endDate switches it off.
  void set_tracking_date(int year);
  int tracking_date() const { return tracking_year_; }
  std::vector<std::string> tracking_pools() const;
  // values[ny][TP], fractions[ny][TP][TP], source_masks[ny][TP] (bit s: source s is in the pool's
  // map; optional) of one member for year0..year1 (>= the tracking date)
  void tracking_data(int member, int year0, int year1, double *values, double *fractions,
                     unsigned long long *source_masks = nullptr);
  std::string run_name() const;
  void var_info(const std::string &capability, std::string *component, std::string *units) const;
  const std::vector<std::string> &halocarbon_names() const { return halo_names_; }
  // device pointer to the [ns][npad] array of an output variable
  const double *device_var(const std::string &capability, int *npad) const;
  // per-year ensemble statistics {count,sum,sumsq,min,max} into a DEVICE buffer
  // of (year1-year0+1)*5 doubles (caller-owned, e.g. a torch tensor for RCCL)
  void stats_device(const std::string &capability, int year0, int year1, double *d_stats);
  // the same, queued on the core's stream without waiting (the fleet's collective follows it)
  void stats_async(const std::string &capability, int year0, int year1, double *d_stats);
  int device() const { return device_; }
  void status(unsigned *out_host);
  void state_row(int row, double *out_host);
  int spinup_steps(int member);

  double last_run_kernel_ms() const { return run_ms_; }
  // "output=0" in a component's section: the output stream leaves its rows out (core.cpp:257-262)
  bool component_output_enabled(const std::string &section) const { return scen_.scalar(section, "output", 1.0) > 0; }
  void set_pair_kernel_limit(int max_members) { pair_max_members_ = max_members < 0 ? 0 : max_members; }
  // Ensembles of at least min_members take the one-biome kernel built for two resident
  // wavefronts per SIMD (hx_run_kernel<HX_B1W2>): < 0 the default -- more wavefronts than the
  // device has SIMDs --, 0 never
  void set_two_wave_from(int min_members) { two_wave_from_ = min_members; }
  int wave_clock(long long *ticks, int cap);   // [wavefront][start, end] of the last launch, 100 MHz ticks
  const char *last_run_kernel() const { return last_run_pair_ ? "pair" : last_run_w2_ ? "run2" : "run"; }
  // which instantiation family the last run() asked for (the CON template argument of
  // hx_run_kernel): 0 plain, -2 plain + diagnostics, -1 extended, 1 extended with the NBP
  // machinery, 2 carbon tracking
  int last_run_variant() const { return last_run_con_; }
  double last_spinup_ms() const { return spin_ms_; }
  hipStream_t stream() const { return stream_; }

 private:
  struct ParamRef { int row; bool per_biome; const char *units; bool affects_spinup; };
  int resolve_param(const std::string &capability, const ParamRef **ref) const;
  int out_index(const std::string &capability) const;
  void build_shared();
  void alloc_device();
  void free_device();
  void upload_params();
  void prepare();  // upload + spinup when dirty
  HxBuffers buffers() const;
  void check(hipError_t e, const char *what) const;

  Scenario scen_;
  int n_, npad_, B_, device_;
  std::vector<std::string> biome_names_;
  std::vector<std::vector<double>> params_;  // [row][npad]
  std::vector<bool> row_uniform_;
  std::vector<int> member_of_lane_, lane_of_member_;  // lane <-> member (size npad / n)
  bool sort_members_ = true, calibrate_lanes_ = true;
  std::vector<double> lane_cost_;  // [n_] measured cost per member (empty: parameter key)
  double *d_cost_ = nullptr;
  long long *d_wave_clk_ = nullptr;   // HxBuffers::wave_clk
  double *d_bscratch_ = nullptr;  // [nbiome][npad] f_new_thaw of the seven- and eight-biome kernels
  int cost_from_iy_ = -1;         // d_cost_ covers the years cost_from_iy_+1..last_iy_ (-1: nothing)
  void maybe_calibrate_lanes();
  uint64_t cost_model_key(const std::vector<int> &varying) const;
  void fit_cost_model(const std::vector<double> &member_cost);
  bool predict_cost(const std::vector<int> &varying, std::vector<double> &out) const;
  int lane_order_source_ = 0;
  bool cost_model_ = true, cost_fitted_ = false;
  void assign_lanes();
  bool params_dirty_ = true, need_spinup_ = true, layout_dirty_ = true, ker_per_member_ = false;
  // What the device holds of the parameter table: the rows set since the last upload and the lane
  // order of that upload (upload_params() sends only what moved).
  std::vector<char> row_dirty_;
  bool rows_all_dirty_ = true, order_changed_ = true;
  std::vector<int> uploaded_order_;
  // The post-spinup snapshot on the device belongs to the current spinup inputs (a shared spinup
  // is then not repeated when only parameters it does not see were set: prepare()).
  bool spin_valid_ = false;
  int last_iy_ = 0;
  HxConst kc_{};
  std::vector<double> member_series_[HXM_N];  // host [ns][n_], member order; empty = shared
  // tas_constrain / RF_tot_constrain interpolate between the dates they were given at
  // (temperature_component.cpp:112, forcing_component.cpp:112): per-member POINTS, year -> [n_]
  // (NaN = none for that member), from which the dense member series is rebuilt
  std::map<int, std::vector<double>> member_points_[HXM_N];
  void densify_member_constraint(int k, const std::string &capability);
  double *d_mseries_[HXM_N] = {};
  bool mseries_dirty_ = false;
  void upload_member_series();
  void upload_args();
  bool component_disabled(const std::string &section) const;  // "enabled=0" in the INI section
  void check_component_enabled(const std::string &capability) const;
  // N2O / halocarbon parameters that differ between members: capability -> [n_] (member order)
  std::map<std::string, std::vector<double>> gas_member_;
  double *d_gas_par_ = nullptr, *d_gas_ser_ = nullptr;
  void run_gas_kernel();  // fills the per-member N2O and halocarbon-forcing series
  bool gas_dirty_ = false;
  int member_con_mask_ = 0;  // HXC_* bits that only per-member constraint series contribute
  void init_from_scenario();
  std::vector<std::string> halo_names_;
  std::vector<std::vector<double>> halo_conc_;  // [gas][ns] halocarbon concentrations, pptv
  std::vector<double> shared_, ker_;
  bool out_enabled_[HXO_NVAR];
  // device
  double *d_params_ = nullptr, *d_state_ = nullptr, *d_shared_ = nullptr, *d_ker_ = nullptr;
  double *d_out_[HXO_NVAR];
  unsigned *d_status_ = nullptr;
  int *d_spin_steps_ = nullptr;
  HxArgs *d_args_ = nullptr;
  double *d_uparams_ = nullptr;
  int tracking_year_ = 0;  // 0 = off
  double *d_track_out_f_ = nullptr, *d_track_out_v_ = nullptr;
  // SimpleNbox::run starts tracking when runToDate == trackingDate (simpleNbox-runtime.cpp:215-220):
  // a date at or before startDate, or past endDate, never engages
  int trk_iy() const {
    return (tracking_year_ > scen_.start && tracking_year_ <= scen_.end) ? tracking_year_ - scen_.start : -1;
  }
  double *d_derived_ = nullptr, *d_dpart_ = nullptr, *d_gather_ = nullptr, *d_hist_ = nullptr;
  unsigned *d_hist_status_ = nullptr;  // [ns][npad] status bits of every year (with d_hist_)
  void remap_biome_outputs(const std::vector<int> &old_of_new);
  bool history_ = false, shared_dirty_ = false;
  bool spin_record_ = false, spin_uniform_ = false;
  double *d_spin_rec_ = nullptr;
  int hist_valid_to_ = 0;   // history slabs 1..hist_valid_to_ are valid
  int dirty_from_iy_ = -1;  // pending auto-reset target (R wrapper's reset_date)
  int *d_lane_of_member_ = nullptr;
  size_t gather_cap_ = 0, diag_cap_ = 0;
  double *d_diag_ = nullptr, *d_slr_ = nullptr;
  int slr_valid_to_ = -1;
  void check_parameters() const;
  bool fetch_host(const std::string &capability, int year0, int year1, double *out_host);
  void compute_derived(const std::string &capability, int iy0, int ny);
  hipStream_t stream_ = nullptr;
  hipStream_t aux_stream_ = nullptr;          // the prewarm loop's (non-blocking)
  unsigned char *d_prewarm_ = nullptr;        // its stop flag (+ a sink)
  int prewarm_ms_ = 50;
  bool prewarm_on_ = false, last_run_prewarmed_ = false;
  double last_gpu_activity_s_ = -1.0;         // host clock of the last launch / wait of this core
  void prewarm_begin();
  void prewarm_end();
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  bool run_timed_ = false;
  int pair_max_members_ = 32768;  // ensembles up to this size use the two-wavefront kernel (0: never)
  bool last_run_pair_ = false;
  int two_wave_from_ = -1;        // see set_two_wave_from()
  bool pair_costly_with_cheap_ = true;   // lane order by measured cost: see assign_lanes()
  int key_order_mode_ = 0;               // HECTOR_AMD_KEY_ORDER (experiments, see assign_lanes)
  bool last_run_w2_ = false;
  int last_run_con_ = 0;
  bool two_wave_expected() const;   // run() will take hx_run_kernel<HX_B1W2> (see assign_lanes)
  int simds_ = 1024;              // SIMDs of this core's device (4 per compute unit)
  mutable double run_ms_ = 0, spin_ms_ = 0;
};

}  // namespace hx
