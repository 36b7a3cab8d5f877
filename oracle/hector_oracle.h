/*
 * hector_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU restatement (plain C99) of the reference's coupled
 * carbon-cycle / climate year loop for ONE ensemble member:
 *   Core::run year loop            src/core.cpp:448-509
 *   CarbonCycleSolver::run         src/carbon-cycle-solver.cpp:222-303
 *   SimpleNbox (land + atmosphere) src/simpleNbox-runtime.cpp, src/simpleNbox.cpp
 *   OceanComponent / oceanbox / oceancsys
 *   TemperatureComponent (DOECLIM) src/temperature_component.cpp
 *   ForcingComponent + CH4/OH/O3/N2O/halocarbon components
 * plus the numerics of the Boost routines the reference calls (odeint dopri5
 * controlled stepper, newton_raphson_iterate, brent_find_minima, lognormal cdf)
 * restated from their published algorithms (Boost is not in /root/reference;
 * unpinned, CI uses distro libboost-dev, comments mention 1.81).
 *
 * PINNING: validated against the reference's own golden trajectory
 * tests/testthat/compdata/hector_comp.csv (fixture tests/golden/
 * hector_comp_ssp245.txt) -- see tests/test_oracle_golden.py.  Perturbed-
 * parameter and multi-biome members have no reference vectors ("parity
 * unpinned" for those inputs beyond the reference's property tests).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library.  The product (hector_amd/csrc) never links or calls it.
 */
#ifndef HECTOR_ORACLE_H
#define HECTOR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define HXO_MAXB 32     /* max biomes */
#define HXO_NHALO 26

/* output variable ids: out[var * ns + (year - start)] */
enum {
  HXO_CO2 = 0,      /* CO2_concentration ppmv */
  HXO_TGAV,         /* global_tas */
  HXO_RF_TOT,
  HXO_RF_CO2,
  HXO_HEATFLUX,
  HXO_OCEAN_C,
  HXO_HL_PH,
  HXO_ATMOS_C,
  HXO_SST,
  HXO_PERMAFROST_C,
  HXO_TLAND,        /* land_tas */
  HXO_CH4,
  HXO_N2O,
  HXO_O3,
  HXO_VEG_C,
  HXO_DET_C,
  HXO_SOIL_C,
  HXO_THAWED_C,
  HXO_EARTH_C,
  HXO_NSTASH,       /* ocean "timesteps" (stashes) in the year */
  HXO_MAXTS,        /* ocean max_timestep at year end */
  HXO_SOLVER_DT,    /* solver dt at year end */
  HXO_NSTEPS,       /* accepted dopri5 steps in the year */
  HXO_NRHS,         /* RHS evaluations in the year */
  HXO_LL_PH,
  HXO_PCO2_HL,
  HXO_PCO2_LL,
  HXO_OCEAN_UPTAKE, /* annualflux_sum */
  HXO_NBP,
  HXO_RF_CH4,
  HXO_RF_N2O,
  /* diagnostics (csv_outputstream_visitor.cpp writes all of these every year) */
  HXO_NPP, HXO_RH, HXO_RH_DET, HXO_RH_SOIL, HXO_RH_CH4, HXO_F_FROZEN, HXO_CA_RESIDUAL,
  HXO_GMST, HXO_FLUX_MIXED, HXO_FLUX_INTERIOR, HXO_OCEAN_TAS,
  HXO_HL_UPTAKE, HXO_LL_UPTAKE, HXO_C_HL, HXO_C_LL, HXO_C_IO, HXO_C_DO,
  HXO_DIC_HL, HXO_DIC_LL, HXO_HL_DO,
  HXO_OMEGAAR_HL, HXO_OMEGAAR_LL, HXO_OMEGACA_HL, HXO_OMEGACA_LL,
  HXO_TEMP_HL, HXO_TEMP_LL, HXO_CO3_HL, HXO_CO3_LL, HXO_REVELLE_HL, HXO_REVELLE_LL,
  HXO_TAU_OH,
  HXO_RF_H2O, HXO_RF_O3, HXO_RF_BC, HXO_RF_OC, HXO_RF_SO2, HXO_RF_NH3, HXO_RF_ACI,
  HXO_RF_VOL, HXO_RF_ALBEDO, HXO_RF_MISC, HXO_RF_HALO, /* sum of the halocarbon forcings */
  HXO_SLR, HXO_SL_RC, HXO_SLR_NO_ICE, HXO_SL_RC_NO_ICE,
  /* per biome (first 4 biomes): 11 variables each, index HXO_BIOME0 + 11 * biome + k, k in the
   * order veg_c, detritus_c, soil_c, permafrost_c, thawedp_c, NPP, RH, rh_ch4, f_frozen,
   * detritus_tempfert, soil_tempfert */
  HXO_BIOME0,
  HXO_NVAR = HXO_BIOME0 + 44
};

typedef struct hxo_scenario hxo_scenario;

typedef struct {
  /* temperature */
  double S, diff, qco2;
  /* forcing scalars */
  double aero_scalar, vol_scalar;
  /* simpleNbox global */
  double C0;
  int nbiome;
  /* per biome */
  double beta[HXO_MAXB], q10_rh[HXO_MAXB], warmingfactor[HXO_MAXB];
  double npp_flux0[HXO_MAXB], veg_c[HXO_MAXB], detritus_c[HXO_MAXB],
      soil_c[HXO_MAXB], permafrost_c[HXO_MAXB];
  double f_nppv[HXO_MAXB], f_nppd[HXO_MAXB], f_litterd[HXO_MAXB];
  double rh_ch4_frac[HXO_MAXB], pf_mu[HXO_MAXB], pf_sigma[HXO_MAXB],
      fpf_static[HXO_MAXB];
  /* ocean */
  double tt, tu, twi, tid, preind_surface_c, preind_interdeep_c;
  /* temperature: land-ocean warming ratio override, 0 = off */
  double lo_warming_ratio;
} hxo_params;

/* Load a scenario pack (.hxs, see tools/import_scenario.py). NULL on error. */
hxo_scenario *hxo_scenario_load(const char *path);
void hxo_scenario_free(hxo_scenario *);
int hxo_scenario_start(const hxo_scenario *);
int hxo_scenario_end(const hxo_scenario *);
int hxo_scenario_max_spinup(const hxo_scenario *);

/* Fill *p with the scenario's own (INI) values, one "global" biome. */
void hxo_params_default(const hxo_scenario *, hxo_params *p);

/* Split biome 0 of *p into n equal biomes (R/biome.R:61-130 semantics:
 * pools and npp_flux0 times 1/n, other parameters copied). */
void hxo_params_split_equal(hxo_params *p, int n);

/* Spin up + run one member from startDate to run_to (<= endDate).
 * out: HXO_NVAR * ns doubles (ns = end-start+1), zero-filled first.
 * Returns 0 on success, else a bitmask of model errors (mass balance, >8
 * retries, negative pool, spinup failure ...). spinup_steps (may be NULL)
 * receives the number of spinup steps taken. */
int hxo_run_member(const hxo_scenario *, const hxo_params *, int run_to,
                   double *out, int *spinup_steps);

/* Test-suite probe: multiply every pool by 1 + rel * xi (xi in [-1, 1), deterministic) at the end
 * of every model year, and every flux the alkalinity tuner evaluates, in the runs this thread
 * makes from now on; 0 switches it off.  Used to measure how strongly a member amplifies
 * rounding-level differences (tests/test_random_sweep.py). */
void hxo_set_rounding_noise(double rel);

/* The same with carbon tracking from `tracking_date` on (Core::trackingDate; fluxpool source
 * maps, inst/include/fluxpool.hpp): TP = 2 + 5 * nbiome + 4 pools in the order atmos_c, earth_c,
 * per biome {veg_c, detritus_c, soil_c, permafrost_c, thawedp_c}, ocean {HL, LL, intermediate,
 * deep}; trk_v[ns][TP] pool values, trk_f[ns][TP][TP] fraction of pool p that originated in
 * pool s (rows of years before tracking_date stay untouched).  PARITY UNPINNED: the reference
 * ships no tracking vectors; its own tests check structure and that fractions sum to one. */
int hxo_run_member_tracking(const hxo_scenario *, const hxo_params *, int run_to,
                            int tracking_date, double *out, int *spinup_steps, double *trk_f,
                            double *trk_v);

/* The spinup alone (Core::run_spinup, core.cpp:394-420) with the state the output-stream visitor
 * sees after every step (its spinup = 1 rows, csv_outputstream_visitor.cpp:86-95):
 * spin_out[HXO_NVAR][max_spinup], row step-1.  *steps (may be NULL) = number of steps. */
int hxo_run_member_spinup(const hxo_scenario *s, const hxo_params *p, double *spin_out, int *steps);

/* Run members [0,n) whose parameters differ from *base only in S and q10_rh[0]
 * (the BASELINE config 2-4 ensemble); writes co2[n*ns], tgav[n*ns] (may be
 * NULL).  Used as bench.py's cpu_baseline leg.  Returns OR of error masks. */
int hxo_run_ensemble_ecs_q10(const hxo_scenario *, const hxo_params *base,
                             int n, const double *S, const double *q10,
                             int run_to, double *co2, double *tgav);

/* Run members [0,n), each with its own parameter set.  Members whose spinup-relevant
 * parameters (C0, ocean transports and preindustrial carbon, npp_flux0, initial pools, the
 * NPP / litter fractions) equal those of member 0 start from member 0's spun-up state -- the
 * sharing hxo_run_ensemble_ecs_q10 uses (bit-identical to a spinup of their own); the others
 * spin up themselves.  co2 / tgav [n*ns], timesteps [n*ns] (stashes per year), errs [n] may
 * each be NULL.  Returns the OR of the error masks. */
int hxo_run_ensemble(const hxo_scenario *, const hxo_params *params, int n, int run_to,
                     double *co2, double *tgav, unsigned char *timesteps, int *errs);

/* unit vectors for tests */
/* carbonate chemistry: T (degC), carbon (PgC), alk (mol/kg), box volume (m3)
 * -> out[0]=PCO2o, out[1]=pH, out[2]=Tr, out[3]=K0, out[4]=h, out[5]=CO3 */
void hxo_csys(double Tc, double carbon, double alk, double volume, double *out);
/* DOECLIM kernel Ker[ns] for a diffusivity */
void hxo_doeclim_kernel(double diff, int ns, double *ker);

#define HXO_ERR_MASS 1
#define HXO_ERR_RETRIES 2
#define HXO_ERR_NEGPOOL 4
#define HXO_ERR_SPINUP 8
#define HXO_ERR_SINGULAR 16
#define HXO_ERR_ROOT 32
#define HXO_ERR_STEPFAIL 64

#ifdef __cplusplus
}
#endif
#endif
