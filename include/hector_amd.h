/*
 * hector_amd.h -- C ABI of libhector_amd.so: an MI355X-native ensemble
 * integrator for Hector's coupled carbon-cycle / climate year loop.
 *
 * Each entry point names the reference interface it replaces (file:line in
 * JGCRI/hector v3.5.0).  The reference drives ONE member through
 *   Core::mkcore/getcore/delcore   inst/include/core.hpp:105-109, src/core.cpp:813-857
 *   Core::init + INIToCoreReader   src/rcpp_hector.cpp:31-86 (newcore_impl)
 *   Core::sendMessage(SETDATA/GETDATA, capability, message_data)
 *                                  src/core.cpp:716-778, src/rcpp_hector.cpp:282-356
 *   Core::run / Core::reset / Core::shutDown
 *                                  src/core.cpp:448-549, src/rcpp_hector.cpp:88-181
 * This library keeps those verbs and the capability strings of
 * inst/include/component_data.hpp, and adds a member axis: every parameter is a
 * vector over members and every result is [year][member].
 *
 * Conventions: plain C types only; every function returns 0 on success and a
 * non-zero code on failure, with the message available from hx_last_error()
 * (the reference throws h_exception; nothing is thrown across this ABI).
 * Per-member MODEL errors (mass balance, >8 solver retries, negative pool...)
 * do not fail a call: they set bits in the member's status word (hx_status).
 * A core is bound to one GPU (hx_newcore) or to a list of GPUs (hx_newcore_devices: contiguous
 * member blocks, one per GPU, no exchange during integration); there is no CPU execution
 * path -- hx_newcore fails if no HIP device is present.  Not thread-safe per handle (like Core).
 */
#ifndef HECTOR_AMD_H
#define HECTOR_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hx_core hx_core; /* opaque; replaces the int index of Core::mkcore */

/* "hip" for the product library.  (A test-only host-emulation build reports
 * "host-emulation"; the Python loader refuses it outside tests.) */
const char *hx_backend(void);
/* "built with HIP x.y.z (<compiler>), gfx950; runtime <hipRuntimeGetVersion>, driver
 * <hipDriverGetVersion>": the toolchain pairing behind a run's figures.  hx_backend()'s first call
 * warns on stderr when the runtime's major version is not the build's. */
const char *hx_build_info(void);
const char *hx_last_error(void);

/* newcore(inifile, ...)  R/hector.R:81-87, src/rcpp_hector.cpp:31-86.
 * `scenario` is a Hector INI file (csv: tables resolved like the reference) or
 * a dense scenario pack (.hxs).  Creates an n_members ensemble on GPU `device`,
 * all members at the INI's parameter values; one biome "global", or the biomes the INI
 * defines with "<biome>.<variable>" keys (at most 32; carbon tracking: 24). */
int hx_newcore(const char *scenario, int n_members, int device, hx_core **out);

/* The same ensemble over SEVERAL GPUs of the node.  The reference keeps many
 * independent cores in one process through its registry (Core::mkcore / getcore / delcore,
 * inst/include/core.hpp:105-109, src/core.cpp:813-857) and the host loops over them; here the
 * handle is that registry: shard s is an n_members / n_devices block of consecutive members on
 * devices[s] (the remainder goes to the first shards), every call of this header is routed to the
 * shards -- per-member arguments sliced, results returned in member order -- and hx_run queues
 * the kernels of every GPU before it returns.  Nothing is exchanged while the model runs; the one
 * collective is hx_ensemble_stats.  Each device may appear once (RCCL needs one rank per GPU;
 * HECTOR_AMD_FLEET_REHEARSAL=1 admits duplicates on a smaller box and exchanges the statistics
 * with device copies instead). */
int hx_newcore_devices(const char *scenario, int n_members, const int *devices, int n_devices,
                       hx_core **out);
/* devices[n_shards], offsets[n_shards + 1] (offsets[s] = first member of shard s); NULL = skip */
int hx_shards(hx_core *core, int *n_shards, int *devices, int *offsets);

/* Per-year ensemble statistics {count, sum, sum of squares, min, max} of nvars recorded outputs
 * over EVERY member on every GPU (and every process that joined the communicator):
 * [nvars][year1 - year0 + 1][5] doubles into host memory out_host and / or device memory d_out
 * (on the first shard's GPU); either may be NULL.  Every GPU reduces its own members
 * (wavefront shuffles), ONE ncclAllGather over RCCL / xGMI hands every rank all blocks (44 KB per
 * rank and variable), and every rank folds them in rank order -- the result is bit-identical on
 * all ranks.  The north star's "RCCL gather of Tgav / CO2 stats"; the reference has no
 * counterpart (its hosts aggregate fetchvars() data frames in R).  A one-GPU core that joined
 * no communicator does no collective.  Returns when the result is in place. */
int hx_ensemble_stats(hx_core *core, int nvars, const char *const *capabilities, int year0,
                      int year1, double *out_host, double *d_out);
/* One process per GPU (MPI / torchrun style hosts): ONE process calls hx_comm_unique_id and
 * hands the 128 bytes to the others by its own means; every process then joins with its
 * process rank.  The communicator has n_procs * n_shards ranks, this core's shards are ranks
 * proc_rank * n_shards + s, and hx_ensemble_stats reduces over all of them.  A core made by
 * hx_newcore_devices that never calls this gets a communicator of its own shards at the first
 * hx_ensemble_stats.  RCCL is loaded when first needed (librccl.so.1 -- the copy already in the
 * process if there is one; HECTOR_AMD_RCCL overrides the path). */
int hx_comm_unique_id(char *id128);
int hx_comm_init_rank(hx_core *core, int n_procs, int proc_rank, const char *id128);
/* world: ranks of the communicator (0 = none yet); backend: "rccl <version> via <library>" */
int hx_comm_info(hx_core *core, int *world, int *first_rank, const char **backend);

/* shutdown(core)  src/rcpp_hector.cpp:88-101 (Core::shutDown + delcore) */
int hx_shutdown(hx_core *core);

/* setvar(core, NA, var, values, unit)  R/messages.R:107-140 ->
 * sendmessage(SETDATA)  src/rcpp_hector.cpp:282-356 -> Core::setData.
 * capability: component_data.hpp string, optionally "<biome>.<capability>".
 * nvalues is 1 (every member) or n_members.  units: NULL/"" = unchecked, else it
 * must match the reference's unit string (e.g. "degC" for S).  Like the R
 * wrapper this invalidates results from date 0 (the next run respins if needed).
 * Parameters of the member-independent components (delta_co2, rho_bc, M0, Tsoil, Tstrat ...)
 * take one value for the whole core -- except those of the N2O and halocarbon components
 * (N0, TN2O0, UC_N2O: src/n2o_component.cpp:95-130; tau_<gas> = the INI key "tau" of
 * [<gas>_halocarbon], rho_<gas>, delta_<gas>: src/halocarbon_component.cpp:118-150), which the
 * reference perturbs per run: with one value per member their recurrences run per member on
 * the device ahead of the year loop (16 B per member-year of HBM). */
int hx_setvar(hx_core *core, const char *capability, const double *values, int nvalues,
              const char *units);
/* setvar(core, dates, var, values, unit) for a scenario INPUT series (emissions, SV,
 * RF_albedo, RF_misc ...) or a constraint (CO2_constrain, NBP_constrain, tas_constrain,
 * RF_tot_constrain, CH4_constrain, N2O_constrain, <gas>_constrain; NaN removes a date);
 * R/messages.R:107-140 with dates: the same new values for every member.  Marks the core dirty from min(year)-1: the next hx_run first resets there (if the
 * state history is enabled, else to startDate), like run() does for a core that is not clean
 * (src/rcpp_hector.cpp:160-166). */
int hx_setvar_dated(hx_core *core, const char *capability, const int *years, const double *values,
                    int n, const char *units);
/* The same with a different value for every member -- the reference's "re-run a period with new
 * emissions per run" pattern (vignettes/ex_hector_apply.Rmd; one reset/setvar/run per run
 * there), for all members in one run: values[i * n_members + member] is the value of
 * years[i].  ffi_emissions, luc_emissions, daccs_uptake, luc_uptake, CH4_emissions, and the
 * constraints CO2_constrain, NBP_constrain, tas_constrain, RF_tot_constrain, CH4_constrain
 * (NaN = no constraint for that member and year; a CH4 constraint at startDate is core-wide)
 * -- 8 B per member-year of HBM each, once used. */
int hx_setvar_dated_members(hx_core *core, const char *capability, const int *years,
                            const double *values, int nyears, const char *units);
/* Keep every year's component state in HBM so hx_reset can return to any computed date --
 * what the reference's per-component tseries records provide (src/simpleNbox.cpp:708-840,
 * src/ocean_component.cpp:767-846).  272 B per member-year (one biome); default off. */
int hx_enable_history(hx_core *core, int on);

/* The spinup as the reference's output stream sees it: CSVOutputStreamVisitor is visited after
 * every spinup step with spinup = 1 (src/core.cpp:402-408, src/csv_outputstream_visitor.cpp:86-95,
 * the year column holds the step number).  hx_enable_spinup_record(core, 1) before the first run
 * keeps the carbon-cycle variables of those rows -- the ones that move during the spinup: NBP, NPP,
 * RH, rh_det, rh_soil, atmos_co2, atmos_c_residual, the land pools, earth_c, the ocean boxes'
 * carbon and the air-sea / HL->DO fluxes -- at max_spinup x 21 x 8 B of HBM per member.
 * hx_spinup_record: *names / *nvars = the capability names (values == NULL: only that);
 * values[(step-1) * nvars + v] for steps 1 .. *steps of `member`.  hector-amd writes them as the
 * spinup = 1 rows of outputstream_<run_name>.csv. */
int hx_enable_spinup_record(hx_core *core, int on);
int hx_spinup_record(hx_core *core, int member, const char *const **names, int *nvars, double *values,
                     int max_steps, int *steps);
/* fetchvars(core, NA, var) for parameters: GETDATA without a date. out[n_members] */
int hx_getvar(hx_core *core, const char *capability, double *out);

/* split_biome(core, "global", names, fveg_c, ...)  R/biome.R:61-130.
 * names: n_biomes C strings.  Fraction arrays may be NULL (equal split / same as fveg). */
int hx_split_biome(hx_core *core, int n_biomes, const char *const *names, const double *fveg,
                   const double *fdet, const double *fsoil, const double *fpf,
                   const double *fnpp);

/* split_biome(core, old_biome, new_biomes, ...) for a core that already has several biomes:
 * the new biomes are appended to the biome list, old_biome is deleted (R/biome.R:61-130). */
int hx_split_biome_of(hx_core *core, const char *old_biome, int n_biomes,
                      const char *const *names, const double *fveg, const double *fdet,
                      const double *fsoil, const double *fpf, const double *fnpp);

/* create_biome_impl / delete_biome_impl / rename_biome  (src/rcpp_hector.cpp:359-400;
 * SimpleNbox::createBiome / deleteBiome / renameBiome, src/simpleNbox.cpp:864-1060).  A created
 * biome has empty pools and npp_flux0 = 0 and the other parameters of the most recent biome
 * (set them with hx_setvar("<biome>.veg_c", ...), like R's create_biome does); at most 32 biomes
 * (1-8 run fully unrolled kernels, 9-32 kernels that loop over the biomes).
 * All three invalidate the run (spinup again), like any parameter change. */
int hx_create_biome(hx_core *core, const char *biome);
int hx_delete_biome(hx_core *core, const char *biome);
int hx_rename_biome(hx_core *core, const char *oldname, const char *newname);

/* get_biome_list(core)  R/biome.R:8-16: "global", the names given to hx_split_biome, or the
 * biomes an INI file defines with "<biome>.<variable>" keys in [simpleNbox]
 * (src/simpleNbox.cpp:190-330).  Per-biome parameters are "<biome>.beta" ..., per-biome pools
 * "<biome>.veg_c", ".detritus_c", ".soil_c", ".permafrost_c", ".thawedp_c" are outputs. */
int hx_biomes(hx_core *core, const char *const **names, int *count);

/* Select which per-year outputs are recorded (capability strings such as
 * "CO2_concentration", "global_tas", "RF_tot"...).  sst and land_tas are always
 * recorded.  The reference records everything always (tseries in every
 * component); here it is opt-in because each variable costs 8 B/member-year. */
int hx_set_outputs(hx_core *core, int nvars, const char *const *capabilities);
int hx_output_capabilities(const char *const **names, int *count);
/* Names of the scenario's halocarbon components ("CF4", "HFC23" ...; one
 * HalocarbonComponent each in the reference, src/core.cpp:120-175): "<name>_concentration",
 * "<name>_emissions", "RF_<name>" and "<name>_constrain" are fetchvars / setvar capabilities. */
int hx_halocarbons(hx_core *core, const char *const **names, int *count);

/* Internal lane assignment: by default members are mapped to GPU lanes sorted by their
 * perturbed parameters (wavefronts then follow similar solver schedules); every result is
 * returned in the caller's member order either way and does not depend on this switch.
 * hx_device_var exposes the raw lane-ordered arrays; hx_lane_of_member maps them. */
int hx_set_member_sorting(hx_core *core, int on);
/* The order is refined by MEASURED cost: the run kernel adds up every member's dopri5 steps and
 * stashes, and the first hx_reset(startDate) after a run that covered startDate..endDate reorders
 * the lanes by it -- wavefronts of members that really walk the same schedule, the costliest
 * dispatched first, so that an ensemble of more wavefronts than SIMDs does not end on its most
 * expensive ones; where two wavefronts share a SIMD (hx_set_two_wave_from) the second batch in
 * ascending cost, so that the costliest shares with the cheapest -- and spins up again (once; a parameter change falls back to the parameter
 * key until the next complete run).  Only ensembles of more wavefronts than the GPU has SIMDs
 * (65 536 members on an MI355X) are reordered: a smaller one lasts as long as its costliest
 * wavefront under any order.  Default on; results do not depend on it.
 * Cost to know about: that one hx_reset(startDate) uploads the reordered rows and, unless the
 * spinup is shared by all members, integrates the spinup again before it returns (it BLOCKS for
 * hx_last_spinup_ms, which then reports that spinup; 0 when the shared spinup was reused), and a
 * hx_setvar afterwards falls back to the parameter key -- a calibration loop that changes
 * parameters every iteration (setvar / reset / run) gains nothing from it and should switch it
 * off with hx_set_lane_calibration(core, 0).
 * hx_lanes_calibrated: 1 once the measured order is in use. */
int hx_set_lane_calibration(hx_core *core, int on);
int hx_lanes_calibrated(hx_core *core, int *yes);
/* A one-shot run has no measured costs of its own.  After a complete run every core fits
 * cost ~ quadratic in its varying parameter rows (standardised) to what it measured and files the
 * model under its scenario table, biome count and varying rows, process-wide; a LATER core with
 * the same key (the larger ensemble of the same study, the next iteration of a calibration loop
 * after hx_setvar) orders its lanes by the predicted cost from its first run on -- where the
 * order matters, more wavefronts than SIMDs.  hx_lane_order_source: what the lanes of the last
 * upload are ordered by -- 0 the parameter key, 1 this core's measured cost, 2 the model's
 * prediction.  hx_set_cost_model(core, 0) / HECTOR_AMD_COST_MODEL=0: neither fit nor use one.
 * Results do not depend on the order.  No counterpart in the reference. */
int hx_lane_order_source(hx_core *core, int *source);
int hx_set_cost_model(hx_core *core, int on);
/* The registry as a file, so that a fresh process -- a genuine one-shot run -- starts with models:
 * hx_cost_models_export writes every model the process holds (text, one record per model),
 * hx_cost_models_load adds a file's models to the registry (count: how many).  Without either
 * call the library reads $HECTOR_AMD_COST_MODELS, else <its directory>/../data/cost_models.txt
 * -- the models shipped with the scenarios (tools/make_cost_models.py) -- once, before the first
 * lookup.  A model is keyed on the scenario's per-year table, the biome count, the varying rows,
 * the values of the uniform rows and the constraint mask; it only orders lanes.  No counterpart
 * in the reference. */
int hx_cost_models_export(const char *path, int *count);
int hx_cost_models_load(const char *path, int *count);
int hx_lane_of_member(hx_core *core, int *out /* n_members */);

/* reset(core, date)  src/rcpp_hector.cpp:103-151 -> Core::reset src/core.cpp:511-549.
 * date < startDate (e.g. 0): rerun the spinup on the next run; date == startDate:
 * back to the post-spinup state; any other computed date if hx_enable_history is on. */
int hx_reset(hx_core *core, double date);

/* run(core, runtodate)  src/rcpp_hector.cpp:153-181 -> Core::run src/core.cpp:448-509.
 * runtodate < 0: run to endDate.  Callable repeatedly with increasing dates.
 * hx_run returns after the GPU work is queued; hx_sync waits for it.
 * Prepares the core first (parameter upload, spinup) if anything changed. */
int hx_run(hx_core *core, double runtodate);
int hx_sync(hx_core *core);

/* fetchvars(core, dates, var)  R/messages.R:46-88: GETDATA per (variable, year).
 * out[(year - year0) * n_members + member], host memory. */
int hx_fetchvars(hx_core *core, const char *capability, int year0, int year1, double *out);
/* Same data without leaving the GPU: device pointer to the variable's
 * [n_years_total][npad] array (row = year - startDate, npad >= n_members), columns in
 * LANE order (see hx_lane_of_member). */
int hx_device_var(hx_core *core, const char *capability, const double **d_ptr, int *npad);
/* ... of one shard of a multi-GPU core (memory of that shard's GPU, npad of that shard) */
int hx_device_var_shard(hx_core *core, int shard, const char *capability, const double **d_ptr,
                        int *npad);
/* Per-year ensemble statistics {count, sum, sum of squares, min, max} of one
 * variable into a caller-owned DEVICE buffer of (year1-year0+1)*5 doubles --
 * the sufficient statistics that a multi-GPU job all-reduces over RCCL.  The kernel runs on the
 * core's stream (hx_stream) and has finished when the call returns; work the caller queued on
 * d_stats on another stream (e.g. a fill) must have completed before the call.  On a core that
 * spans several GPUs or joined a communicator this is hx_ensemble_stats for one variable (d_stats
 * on the first shard's GPU). */
int hx_stats_device(hx_core *core, const char *capability, int year0, int year1,
                    double *d_stats);

/* per-member model-error bitmask (HX_ERR_* below), host array of n_members */
int hx_status(hx_core *core, unsigned *out);
int hx_spinup_steps(hx_core *core, int member, int *steps);

/* Diagnostic: one row of the per-member state table (hx_layout.h HxStateRow /
 * HxBiomeState numbering) at the current date, host array of n_members.  The
 * reference exposes the same quantities as undated GETDATA (e.g. ocean box
 * carbon, max timestep: src/ocean_component.cpp:422-512). */
int hx_state_row(hx_core *core, int row, double *out);

/* Carbon tracking: get_tracking_data(core)  R/hector.R -> Core::getTrackingData
 * (src/core.cpp:199-209, src/csv_tracking_visitor.cpp): where the carbon of every pool
 * originated, from Core::trackingDate on.  Switch it on with
 * hx_setvar(core, "trackingDate", &year, 1, NULL) before running.  Pools (hx_tracking_pools):
 * atmos_c, earth_c, [<biome>.]veg_c/detritus_c/soil_c/permafrost_c/thawedp_c, and the ocean
 * boxes HL, LL, intermediate, deep.  hx_tracking_data: for one member and year0..year1,
 * values[(y-year0)*TP + pool] (Pg C), fractions[((y-year0)*TP + pool)*TP + source] and, if
 * source_masks is not NULL, source_masks[((y-year0)*TP + pool)*W + s/64] with bit s%64 set where
 * source s is in the pool's map (the rows CSVFluxPoolVisitor::print_pool writes); W = (TP+63)/64
 * words per pool: 1 up to 11 biomes (TP = 6 + 5 biomes <= 64), 2 from 12 biomes on.  A [core]
 * trackingDate in the INI switches tracking on as well; hector-amd then writes
 * tracking_<run_name>.csv.  Any biome count (1-16).  Costs TP*TP*8 B per member-year of HBM
 * (968 B for one biome, 59 KB for sixteen; a record that does not fit is refused with the
 * numbers); not combinable with a CO2 or NBP constraint. */
int hx_tracking_pools(hx_core *core, const char *const **names, int *count);
int hx_tracking_data(hx_core *core, int member, int year0, int year1, double *values,
                     double *fractions, unsigned long long *source_masks);

/* getunits(var)  R/units.R (unit strings of src/unitval.cpp:30-165) and the component that owns
 * the variable (IModelComponent::getComponentName, as printed by the output stream): for
 * parameters, inputs, constraints and outputs.  The strings stay valid until the next call on
 * this thread. */
int hx_var_info(hx_core *core, const char *capability, const char **component, const char **units);
/* Core::getRun_name (src/core.cpp:215-220): the INI's [core] run_name, "" if none */
int hx_run_name(hx_core *core, const char **name);
/* Unit vectors for function-level parity tests (no core needed).
 * hx_unit_csys: oceancsys::ocean_csys_run (src/ocean_csys.cpp:166-366) for n independent
 *   (T degC, box carbon Pg C, alkalinity mol/kg) triples in a box of `volume` m3;
 *   out[4*i..] = {pCO2 uatm, pH, Tr, status bits}.
 * hx_unit_doeclim_kernel: TemperatureComponent::prepareToRun's Ker[ns]
 *   (src/temperature_component.cpp:303-371) for one diffusivity (cm2/s). */
int hx_unit_csys(int device, int n, const double *Tc, const double *carbon, const double *alk,
                 double volume, double *out);
int hx_unit_doeclim_kernel(int device, double diff, int ns, double *out);

/* core metadata: startDate, endDate, current date, members, biomes */
int hx_dates(hx_core *core, int *start, int *end, int *current);
int hx_sizes(hx_core *core, int *n_members, int *n_biomes);

/* HIP-event time of the last run / spinup launch on the core's stream, ms */
int hx_last_run_ms(hx_core *core, double *ms);
int hx_last_spinup_ms(hx_core *core, double *ms);
/* the core's hipStream_t, as void* */
int hx_stream(hx_core *core, void **stream);
int hx_stream_shard(hx_core *core, int shard, void **stream);

/* Small ensembles -- too few 64-member wavefronts to occupy the GPU's 1 024 SIMDs, BASELINE
 * configs[1] -- are run by a kernel that gives every 64 members TWO wavefronts (ocean / climate and
 * land, hx_dev_pair.h): same model, same decisions, ~20 % shorter launch.  It serves
 * ensembles of one to four biomes without an NBP constraint, a land-ocean warming ratio, per-member series
 * (scenario-wide CO2 / tas / RF_tot / CH4 constraints -- concentration-driven runs -- are served,
 * with shared diffusivity) or diagnostics beyond CO2,
 * tas, RF_tot, RF_CO2, SST, land tas, timesteps, the carbon pools (atmos_co2, ocean_c, veg_c,
 * detritus_c, soil_c, permafrost_c, thawedp_c, earth_c), NBP, NPP, RH and its parts, f_frozen,
 * ocean_uptake, heatflux, gmst, HL_pH, LL_pH and the
 * CH4 / O3 concentrations (any scalar parameter may differ between members, diffusivity included);
 * everything else takes the one-wavefront kernels.
 * hx_set_pair_kernel_limit: ensembles of up to max_members use it (default 32 768 = one workgroup
 * per two SIMDs; 0 = never; the environment variable HECTOR_AMD_PAIR_MAX_MEMBERS sets the default of
 * new cores).  hx_last_run_kernel: "run" or "pair", whichever the last hx_run took. */
int hx_set_pair_kernel_limit(hx_core *core, int max_members);
/* Large ensembles -- more wavefronts of 64 members than the GPU has SIMDs (65 536 members on an
 * MI355X), e.g. BASELINE configs[3]'s 131 072 members per GPU -- are run by a flavour of the
 * one-biome kernel compiled for TWO resident wavefronts per SIMD (at most 256 registers and 20 KB
 * of LDS a wavefront; hx_dev_member.h, HX_B1W2): the second wavefront issues into the slots a
 * dependent fp64 chain of the first leaves empty.  Same model code, same decisions; it serves what
 * the one-biome kernels serve -- shared or per-member diffusivity, heat flux, and the extended kernel's
 * constraints, land-ocean warming ratio and diagnostics -- except carbon tracking.
 * hx_set_two_wave_from: ensembles of at least min_members members use it (< 0: the default, one
 * more wavefront than the device has SIMDs; 0: never; the environment variable
 * HECTOR_AMD_TWO_WAVE_FROM sets the default of new cores).  hx_last_run_kernel then says "run2".
 * With the default (< 0) an ensemble whose members differ in ocean heat diffusivity stays on the
 * one-wavefront kernel, which is the faster one for it (131 072 such members 13.5 against 14.3 ms). */
int hx_set_two_wave_from(hx_core *core, int min_members);
/* A run kernel launched after an idle gap -- the first run of a fresh core, behind 10-15 ms of
 * upload and spinup -- executes at ramping clocks (+6 % at 65 536 members, +11-15 % at 131 072:
 * profiles/r06_prewarm_curve.txt).  While hx_run's preparation uploads and spins up after such a
 * gap, a busy loop of three small wavefronts a SIMD keeps the chip's clocks up; hx_run stops it right
 * ahead of the run kernel, and it stops by itself after `ms` milliseconds of the device's own
 * clock (default 50; 0: off; HECTOR_AMD_PREWARM_MS sets the default of new cores).
 * hx_last_run_prewarmed: whether the last hx_run's kernel was launched behind it.  Results do
 * not depend on it.  No counterpart in the reference. */
int hx_set_prewarm(hx_core *core, int ms);
int hx_last_run_prewarmed(hx_core *core, int *yes);

/* Core::outputEnabled (src/core.cpp:257-262, 688-695): 0 if the scenario's section of that component
 * says output=0 -- the output stream visitor then leaves the component's rows out
 * (src/csv_outputstream_visitor.cpp, every visit()); hector-amd's stream does the same. */
int hx_component_output(hx_core *core, const char *component, int *enabled);
int hx_last_run_kernel(hx_core *core, const char **name);
/* Which family of run-kernel instantiations the last hx_run asked for: 0 the plain kernel;
 * -2 the plain kernel plus the diagnostics the reference's output stream writes (NPP, RH and its
 * parts, the ocean boxes' carbon / pCO2 / uptake, gmst ...: csv_outputstream_visitor.cpp:126-365)
 * when no constraint, land-ocean warming ratio or per-member series exists anywhere; -1 the
 * extended kernel (those too: temperature_component.cpp:510-525,586-625, forcing_component.cpp:
 * 498-505, ch4_component.cpp:156-157, simpleNbox-runtime.cpp:567-603); 1 the extended kernel with
 * the NBP constraint's machinery (simpleNbox-runtime.cpp:343-383); 2 carbon tracking. */
int hx_last_run_variant(hx_core *core, int *variant);
/* When every wavefront of the last hx_run's year-loop launch started and ended: ticks[2 w] and
 * ticks[2 w + 1] for wavefront w of shard `shard` (64 members in lane order; the small-ensemble
 * kernel has two wavefronts per 64 members), in ticks of the device's constant 100 MHz clock
 * relative to the earliest start; *n_waves = wavefronts written (at most cap; 0 before a run).
 * The launch lasts as long as its last wavefront: this is the launch's tail, wavefront by
 * wavefront -- what bench.py reports as roofline.wave_time and what decides whether another lane
 * order or more resident wavefronts can shorten a launch.  No counterpart in the reference
 * (src/core.cpp:483-504 runs one member). */
int hx_wave_clock(hx_core *core, int shard, long long *ticks, int cap, int *n_waves);

#define HX_ERR_MASS 1u     /* mass not conserved        simpleNbox-runtime.cpp:553-563 */
#define HX_ERR_RETRIES 2u  /* solver retries exhausted  carbon-cycle-solver.cpp:242-294 */
#define HX_ERR_NEGPOOL 4u  /* negative pool             fluxpool.hpp:100-102 */
#define HX_ERR_SPINUP 8u   /* did not spin up           core.cpp:394-420 */
#define HX_ERR_SINGULAR 16u /* DOECLIM matrix singular  temperature_component.cpp:84-86 */
#define HX_ERR_ROOT 32u    /* carbonate root not found  ocean_csys.cpp:134-156 */
#define HX_ERR_STEPFAIL 64u /* >500 rejected ODE steps  (odeint failed_step_checker), or more than
                             * 20 000 accepted ones in a year (step size collapsed) */
/* A member with a flag is not integrated any further -- the reference throws at that point. */

#ifdef __cplusplus
}
#endif
#endif
