/*
 * hector_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see hector_oracle.h).
 *
 * One-member scalar restatement of the reference's year loop.  Citations are
 * reference file:line (relative to /root/reference).  Operation order follows
 * the reference expression by expression; compile with -ffp-contract=off.
 */
#include "hector_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* scenario pack                                                       */
/* ------------------------------------------------------------------ */
typedef struct {
  char name[24];
  double tau, rho, delta, H0, molarMass;
  double *em;  /* [ns] */
  double *con; /* [ns] concentration constraint, NaN = none that year; NULL = none */
  int disabled; /* "enabled=0" in the gas's section: Core removes the component (core.cpp:251-256) */
} hxo_halo;

struct hxo_scenario {
  int start, end, ns;
  /* scalars */
  double tt, tu, twi, tid, preind_surface_c, preind_interdeep_c;
  int spinup_chem, do_spinup, max_spinup;
  /* components switched off with "enabled=0" (core.cpp:251-256): the forcing component then
   * finds no such capability and leaves the forcing out (forcing_component.cpp:392-484) */
  int off_bc, off_oc, off_so2, off_nh3, off_ozone;
  /* [N2O] or [CH4] (with [OH] and [ozone], which ask for the CH4 concentration at run time:
   * oh_component.cpp:150, o3_component.cpp:134) disabled: the forcing component finds
   * D_CH4_CONC / D_N2O_CONC missing and skips the major greenhouse gases altogether -- CO2, N2O,
   * CH4 and stratospheric H2O (forcing_component.cpp:315-389) */
  int off_n2o, off_ch4;
  double C0, npp_flux0, veg_c, detritus_c, soil_c, permafrost_c;
  double f_nppv, f_nppd, f_litterd, beta, q10_rh;
  double eps_abs, eps_rel, dt, eps_spinup;
  double M0, Tsoil, Tstrat, UC_CH4;
  double TOH0, CNOX, CCO, CNMVOC, CCH4;
  double PO3;
  double N0, UC_N2O, TN2O0;
  double baseyear, aero_scalar, vol_scalar, delta_co2, delta_ch4, delta_n2o;
  double rho_bc, rho_oc, rho_so2, rho_nh3;
  double S, diff, qco2;
  /* series [ns] */
  double *ffi, *daccs, *luc_e, *luc_u, *albedo, *so2, *sv, *ch4n, *ch4_em;
  double *nox_oh, *co_oh, *nmvoc_oh, *nox_o3, *co_o3, *nmvoc_o3;
  double *n2o_nat, *n2o_em, *rf_misc, *bc, *oc, *nh3;
  /* constraints [ns]: NaN where the reference's tseries has no value for that date
   * (the pack writer applies each series' exists / interpolation rule); NULL = unset */
  double *co2_con, *nbp_con, *tas_con, *ftot_con, *ch4_con, *n2o_con;
  double lo_warming_ratio;
  int nhalo;
  hxo_halo halo[HXO_NHALO];
};

static hxo_halo *find_halo(hxo_scenario *s, const char *section, int create) {
  /* section is "<gas>_halocarbon" */
  const char *suf = strstr(section, "_halocarbon");
  if (!suf) return NULL;
  char gas[24];
  size_t n = (size_t)(suf - section);
  if (n >= sizeof gas) return NULL;
  memcpy(gas, section, n);
  gas[n] = 0;
  for (int i = 0; i < s->nhalo; i++)
    if (!strcmp(s->halo[i].name, gas)) return &s->halo[i];
  if (!create || s->nhalo >= HXO_NHALO) return NULL;
  hxo_halo *h = &s->halo[s->nhalo++];
  memset(h, 0, sizeof *h);
  strcpy(h->name, gas);
  h->H0 = 0.0; /* halocarbon_component.cpp:61 default: no preindustrial */
  return h;
}

static void set_scalar(hxo_scenario *s, const char *sec, const char *key,
                       const char *val) {
  double v = strtod(val, NULL);
#define SC(SEC, KEY, FIELD) \
  if (!strcmp(sec, SEC) && !strcmp(key, KEY)) { s->FIELD = v; return; }
  if (!strcmp(sec, "core") && !strcmp(key, "startDate")) { s->start = (int)v; return; }
  if (!strcmp(sec, "core") && !strcmp(key, "endDate")) { s->end = (int)v; return; }
  if (!strcmp(sec, "core") && !strcmp(key, "do_spinup")) { s->do_spinup = (int)v; return; }
  if (!strcmp(sec, "core") && !strcmp(key, "max_spinup")) { s->max_spinup = (int)v; return; }
  if (!strcmp(sec, "ocean") && !strcmp(key, "spinup_chem")) { s->spinup_chem = (int)v; return; }
  if (!strcmp(key, "enabled")) {
    const int off = v <= 0;
    if (!strcmp(sec, "bc")) s->off_bc = off;
    else if (!strcmp(sec, "oc")) s->off_oc = off;
    else if (!strcmp(sec, "so2")) s->off_so2 = off;
    else if (!strcmp(sec, "nh3")) s->off_nh3 = off;
    else if (!strcmp(sec, "ozone")) s->off_ozone = off;
    else if (!strcmp(sec, "N2O")) s->off_n2o = off;
    else if (!strcmp(sec, "CH4")) s->off_ch4 = off;
    else if (!strcmp(sec, "OH")) { /* (runs only together with CH4 off; nothing else reads it) */ }
    else { hxo_halo *hh = find_halo(s, sec, 1); if (hh) hh->disabled = off; }
    return;
  }
  SC("ocean", "tt", tt) SC("ocean", "tu", tu) SC("ocean", "twi", twi)
  SC("ocean", "tid", tid) SC("ocean", "preind_surface_c", preind_surface_c)
  SC("ocean", "preind_interdeep_c", preind_interdeep_c)
  SC("simpleNbox", "C0", C0) SC("simpleNbox", "npp_flux0", npp_flux0)
  SC("simpleNbox", "veg_c", veg_c) SC("simpleNbox", "detritus_c", detritus_c)
  SC("simpleNbox", "soil_c", soil_c) SC("simpleNbox", "permafrost_c", permafrost_c)
  SC("simpleNbox", "f_nppv", f_nppv) SC("simpleNbox", "f_nppd", f_nppd)
  SC("simpleNbox", "f_litterd", f_litterd) SC("simpleNbox", "beta", beta)
  SC("simpleNbox", "q10_rh", q10_rh)
  SC("carbon-cycle-solver", "eps_abs", eps_abs)
  SC("carbon-cycle-solver", "eps_rel", eps_rel)
  SC("carbon-cycle-solver", "dt", dt)
  SC("carbon-cycle-solver", "eps_spinup", eps_spinup)
  SC("CH4", "M0", M0) SC("CH4", "Tsoil", Tsoil) SC("CH4", "Tstrat", Tstrat)
  SC("CH4", "UC_CH4", UC_CH4)
  SC("OH", "TOH0", TOH0) SC("OH", "CNOX", CNOX) SC("OH", "CCO", CCO)
  SC("OH", "CNMVOC", CNMVOC) SC("OH", "CCH4", CCH4)
  SC("ozone", "PO3", PO3)
  SC("N2O", "N0", N0) SC("N2O", "UC_N2O", UC_N2O) SC("N2O", "TN2O0", TN2O0)
  SC("forcing", "baseyear", baseyear) SC("forcing", "aero_scalar", aero_scalar)
  SC("forcing", "vol_scalar", vol_scalar) SC("forcing", "delta_co2", delta_co2)
  SC("forcing", "delta_ch4", delta_ch4) SC("forcing", "delta_n2o", delta_n2o)
  SC("forcing", "rho_bc", rho_bc) SC("forcing", "rho_oc", rho_oc)
  SC("forcing", "rho_so2", rho_so2) SC("forcing", "rho_nh3", rho_nh3)
  SC("temperature", "S", S) SC("temperature", "diff", diff)
  SC("temperature", "qco2", qco2)
  SC("temperature", "lo_warming_ratio", lo_warming_ratio)
#undef SC
  hxo_halo *h = find_halo(s, sec, 1);
  if (h) {
    if (!strcmp(key, "tau")) h->tau = v;
    else if (!strncmp(key, "rho_", 4)) h->rho = v;
    else if (!strncmp(key, "delta_", 6)) h->delta = v;
    else if (!strcmp(key, "H0")) h->H0 = v;
    else if (!strcmp(key, "molarMass")) h->molarMass = v;
  }
}

static void set_series(hxo_scenario *s, const char *sec, const char *key,
                       double *vals) {
#define SE(SEC, KEY, FIELD) \
  if (!strcmp(sec, SEC) && !strcmp(key, KEY)) { s->FIELD = vals; return; }
  SE("simpleNbox", "ffi_emissions", ffi) SE("simpleNbox", "daccs_uptake", daccs)
  SE("simpleNbox", "luc_emissions", luc_e) SE("simpleNbox", "luc_uptake", luc_u)
  SE("simpleNbox", "RF_albedo", albedo)
  SE("so2", "SO2_emissions", so2) SE("so2", "SV", sv)
  SE("CH4", "CH4N", ch4n) SE("CH4", "CH4_emissions", ch4_em)
  SE("OH", "NOX_emissions", nox_oh) SE("OH", "CO_emissions", co_oh)
  SE("OH", "NMVOC_emissions", nmvoc_oh)
  SE("ozone", "NOX_emissions", nox_o3) SE("ozone", "CO_emissions", co_o3)
  SE("ozone", "NMVOC_emissions", nmvoc_o3)
  SE("N2O", "N2O_natural_emissions", n2o_nat) SE("N2O", "N2O_emissions", n2o_em)
  SE("forcing", "RF_misc", rf_misc)
  SE("bc", "BC_emissions", bc) SE("oc", "OC_emissions", oc)
  SE("nh3", "NH3_emissions", nh3)
  SE("simpleNbox", "CO2_constrain", co2_con) SE("simpleNbox", "NBP_constrain", nbp_con)
  SE("temperature", "tas_constrain", tas_con) SE("forcing", "RF_tot_constrain", ftot_con)
  SE("CH4", "CH4_constrain", ch4_con) SE("N2O", "N2O_constrain", n2o_con)
#undef SE
  hxo_halo *h = find_halo(s, sec, 1);
  if (h && strstr(key, "_emissions")) { h->em = vals; return; }
  if (h && strstr(key, "_constrain")) { h->con = vals; return; }
  free(vals);
}

hxo_scenario *hxo_scenario_load(const char *path) {
  FILE *f = fopen(path, "r");
  if (!f) return NULL;
  hxo_scenario *s = (hxo_scenario *)calloc(1, sizeof *s);
  s->do_spinup = 1;
  s->max_spinup = 2000;
  s->eps_abs = s->eps_rel = 1e-6;
  s->dt = 0.3; /* carbon-cycle-solver.cpp:35 */
  s->aero_scalar = s->vol_scalar = 1.0;
  s->preind_surface_c = 900.0;    /* ocean_component.cpp:90-91 defaults */
  s->preind_interdeep_c = 37100.0;
  size_t cap = 1 << 20;
  char *line = (char *)malloc(cap);
  int ok = 0;
  while (fgets(line, (int)cap, f)) {
    char kind[16], sec[64], key[64];
    int off = 0;
    if (sscanf(line, "%15s", kind) != 1) continue;
    if (!strcmp(kind, "HXS")) { ok = 1; continue; }
    if (!strcmp(kind, "scalar")) {
      char val[256];
      if (sscanf(line, "%*s %63s %63s %255s", sec, key, val) == 3)
        set_scalar(s, sec, key, val);
    } else if (!strcmp(kind, "series")) {
      int y0, n;
      if (sscanf(line, "%*s %63s %63s %d %d%n", sec, key, &y0, &n, &off) != 4)
        continue;
      double *v = (double *)malloc(sizeof(double) * (size_t)n);
      char *p = line + off;
      for (int i = 0; i < n; i++) v[i] = strtod(p, &p);
      if (s->start == 0) s->start = y0;
      set_series(s, sec, key, v);
    }
  }
  free(line);
  fclose(f);
  s->ns = s->end - s->start + 1;
  if (!ok || s->ns <= 1) { hxo_scenario_free(s); return NULL; }
  if (s->baseyear == 0.0) s->baseyear = s->start + 1;  /* forcing_component.cpp:278-283 */
  return s;
}

void hxo_scenario_free(hxo_scenario *s) {
  if (!s) return;
  double **ser[] = {&s->ffi, &s->daccs, &s->luc_e, &s->luc_u, &s->albedo,
                    &s->so2, &s->sv, &s->ch4n, &s->ch4_em, &s->nox_oh,
                    &s->co_oh, &s->nmvoc_oh, &s->nox_o3, &s->co_o3,
                    &s->nmvoc_o3, &s->n2o_nat, &s->n2o_em, &s->rf_misc,
                    &s->bc, &s->oc, &s->nh3};
  for (size_t i = 0; i < sizeof ser / sizeof ser[0]; i++) free(*ser[i]);
  for (int i = 0; i < s->nhalo; i++) { free(s->halo[i].em); free(s->halo[i].con); }
  free(s->co2_con); free(s->nbp_con); free(s->tas_con); free(s->ftot_con);
  free(s->ch4_con); free(s->n2o_con);
  free(s);
}
int hxo_scenario_start(const hxo_scenario *s) { return s->start; }
int hxo_scenario_end(const hxo_scenario *s) { return s->end; }
int hxo_scenario_max_spinup(const hxo_scenario *s) { return s->max_spinup; }

void hxo_params_default(const hxo_scenario *s, hxo_params *p) {
  memset(p, 0, sizeof *p);
  p->S = s->S; p->diff = s->diff; p->qco2 = s->qco2;
  p->aero_scalar = s->aero_scalar; p->vol_scalar = s->vol_scalar;
  p->C0 = s->C0;
  p->nbiome = 1;
  p->beta[0] = s->beta; p->q10_rh[0] = s->q10_rh;
  p->warmingfactor[0] = 1.0;          /* simpleNbox.cpp:93 */
  p->npp_flux0[0] = s->npp_flux0; p->veg_c[0] = s->veg_c;
  p->detritus_c[0] = s->detritus_c; p->soil_c[0] = s->soil_c;
  p->permafrost_c[0] = s->permafrost_c;
  p->f_nppv[0] = s->f_nppv; p->f_nppd[0] = s->f_nppd;
  p->f_litterd[0] = s->f_litterd;
  p->rh_ch4_frac[0] = 0.023;           /* simpleNbox.cpp:100-103 */
  p->pf_sigma[0] = 0.986; p->pf_mu[0] = 1.67; p->fpf_static[0] = 0.74;
  p->tt = s->tt; p->tu = s->tu; p->twi = s->twi; p->tid = s->tid;
  p->preind_surface_c = s->preind_surface_c;
  p->preind_interdeep_c = s->preind_interdeep_c;
  p->lo_warming_ratio = s->lo_warming_ratio;
}

void hxo_params_split_equal(hxo_params *p, int n) {
  /* R/biome.R:61-130: pools and npp_flux0 * (1/n); other parameters copied */
  double f = 1.0 / (double)n;
  double v = p->veg_c[0], d = p->detritus_c[0], so = p->soil_c[0],
         pf = p->permafrost_c[0], np = p->npp_flux0[0];
  for (int b = n - 1; b >= 0; b--) {
    p->beta[b] = p->beta[0]; p->q10_rh[b] = p->q10_rh[0];
    p->warmingfactor[b] = p->warmingfactor[0];
    p->f_nppv[b] = p->f_nppv[0]; p->f_nppd[b] = p->f_nppd[0];
    p->f_litterd[b] = p->f_litterd[0];
    p->rh_ch4_frac[b] = p->rh_ch4_frac[0]; p->pf_mu[b] = p->pf_mu[0];
    p->pf_sigma[b] = p->pf_sigma[0]; p->fpf_static[b] = p->fpf_static[0];
    p->veg_c[b] = v * f; p->detritus_c[b] = d * f; p->soil_c[b] = so * f;
    p->permafrost_c[b] = pf * f; p->npp_flux0[b] = np * f;
  }
  p->nbiome = n;
}

/* ------------------------------------------------------------------ */
/* Boost numerics restated                                             */
/* ------------------------------------------------------------------ */

/* boost::math::tools::evaluate_polynomial, runtime count: plain Horner from
 * the top coefficient (called through polynomial<double>::evaluate,
 * src/ocean_csys.cpp:107-117). */
static double horner(const double *a, int count, double z) {
  double sum = a[count - 1];
  for (int i = count - 2; i >= 0; --i) {
    sum *= z;
    sum += a[i];
  }
  return sum;
}

static double sgn(double x) { return (x > 0) - (x < 0); }

/* boost::math::tools::newton_raphson_iterate(f, guess, min, max, digits),
 * Boost >= 1.71 logic (roots.hpp); f = quintic and its derivative.
 * Call site src/ocean_csys.cpp:152-153 with digits = int(53*0.6) = 31. */
static double newton_quintic(const double *a, const double *da, double guess,
                             double min, double max, int digits, int *err) {
  double f0 = 0, f1, last_f0 = 0;
  double result = guess;
  double factor = ldexp(1.0, 1 - digits);
  double delta = DBL_MAX, delta1 = DBL_MAX, delta2 = DBL_MAX;
  double max_range_f = 0, min_range_f = 0;
  long count = 100000;
  do {
    last_f0 = f0;
    delta2 = delta1;
    delta1 = delta;
    f0 = horner(a, 6, result);
    f1 = horner(da, 5, result);
    --count;
    if (0 == f0) break;
    if (f1 == 0) {
      /* detail::handle_zero_derivative: bisect toward the root */
      if (last_f0 == 0) {
        guess = (result == min) ? max : min;
        last_f0 = horner(a, 6, guess);
        delta = guess - result;
      }
      if (sgn(last_f0) * sgn(f0) < 0)
        delta = (delta < 0) ? (result - min) / 2 : (result - max) / 2;
      else
        delta = (delta < 0) ? (result - max) / 2 : (result - min) / 2;
    } else {
      delta = f0 / f1;
    }
    if (fabs(delta * 2) > fabs(delta2)) {
      double shift = (delta > 0) ? (result - min) / 2 : (result - max) / 2;
      if ((result != 0) && (fabs(shift) > fabs(result)))
        delta = sgn(delta) * fabs(result) * (double)1.1f;
      else
        delta = shift;
      delta1 = 3 * delta;
      delta2 = 3 * delta;
    }
    guess = result;
    result -= delta;
    if (result <= min) {
      delta = 0.5 * (guess - min);
      result = guess - delta;
      if ((result == min) || (result == max)) break;
    } else if (result >= max) {
      delta = 0.5 * (guess - max);
      result = guess - delta;
      if ((result == min) || (result == max)) break;
    }
    if (delta > 0) { max = guess; max_range_f = f0; }
    else { min = guess; min_range_f = f0; }
    if (max_range_f * min_range_f > 0) { *err |= HXO_ERR_ROOT; return guess; }
  } while (count && (fabs(result * factor) < fabs(delta)));
  (void)last_f0;
  return result;
}

/* ------------------------------------------------------------------ */
/* ocean carbonate chemistry   src/ocean_csys.cpp                      */
/* ------------------------------------------------------------------ */
typedef struct {
  double S, volumeofbox, As, U, alk;
  /* outputs */
  double K0, Kh, Kw, Tr, PCO2o, pH, TCO2o, HCO3, CO3, OmegaCa, OmegaAr, h;
} csys_t;

/* oceancsys::convertToDIC  src/ocean_csys.cpp:403-408 (umol/kg) */
static double csys_dic_umol(const csys_t *c, double carbon) {
  double dic = ((carbon * 1e15) * (1.0 / 12.01) * (1.0 / 1027.0) *
                (1.0 / c->volumeofbox));
  return dic * 1e6;
}

/* find_largest_root  src/ocean_csys.cpp:134-156 */
static double find_largest_root(double *a, int *err) {
  const int degree = 5;
  double da[5];
  for (int i = 1; i < 6; ++i) da[i - 1] = a[i] * (double)i;
  double max = pow(fabs(a[0] / (2.0 * a[degree])), 1.0 / degree);
  for (int i = 1; i < degree; ++i) {
    double m = pow(fabs(a[i] / a[degree]), 1.0 / (double)(degree - i));
    max = (max < m) ? m : max; /* std::max(max, m) */
  }
  max *= 2.0;
  int get_digits = (int)(53 * 0.6);
  return newton_quintic(a, da, max - 0.001, 0.0, max, get_digits, err);
}

/* oceancsys::ocean_csys_run  src/ocean_csys.cpp:166-366 */
static void csys_run(csys_t *c, double Tc, double carbon, int *err) {
  double tmp, tmp1, tmp2, tmp3;
  const double S = c->S, alk = c->alk;
  const double dic = csys_dic_umol(c, carbon) / 1e6;
  const double Tk = Tc + 273.15;

  tmp1 = -58.0931 + 90.5069 * (100 / Tk) + 22.2940 * log(Tk / 100);
  tmp2 = S * (0.027766 - 0.025888 * (Tk / 100) +
              0.0050578 * ((Tk / 100) * (Tk / 100)));
  const double lnK0 = tmp1 + tmp2;
  c->K0 = exp(lnK0);

  const double Sc =
      2073.1 - (125.62 * Tc) + (3.6276 * Tc * Tc) - (0.043219 * Tc * Tc * Tc);

  tmp1 = -13847.26 / Tk + 148.96502 - 23.6521 * log(Tk);
  tmp2 = +(118.67 / Tk - 5.977 + 1.0495 * log(Tk)) * sqrt(S) - 0.01615 * S;
  const double lnKw = tmp1 + tmp2;
  c->Kw = exp(lnKw);

  tmp = 9345.17 / Tk - 60.2409 + 23.3585 * log(Tk / 100);
  const double nKhwe74 =
      tmp + S * (0.023517 - 0.00023656 * Tk + 0.0047036e-4 * Tk * Tk);
  c->Kh = exp(nKhwe74);

  const double pK1mehr = 3633.86 / Tk - 61.2172 + 9.6777 * log(Tk) -
                         0.011555 * S + 0.0001152 * S * S;
  const double K1_val = pow(10, -pK1mehr);

  const double pK2mehr = 471.78 / Tk + 25.9290 - 3.16967 * log(Tk) -
                         0.01781 * S + 0.0001122 * S * S;
  const double K2_val = pow(10.0, -pK2mehr);

  tmp1 = (-8966.90 - 2890.53 * sqrt(S) - 77.942 * S +
          1.728 * pow(S, (3.0 / 2.0)) - 0.0996 * S * S) /
         Tk;
  tmp2 = +148.0248 + 137.1942 * sqrt(S) + 1.62142 * S;
  tmp3 = +(-24.4344 - 25.085 * sqrt(S) - 0.2474 * S) * log(Tk) +
         0.053105 * sqrt(S) * Tk;
  const double lnKb = tmp1 + tmp2 + tmp3;
  const double Kb_val = exp(lnKb);

  tmp1 = -171.9065 - 0.077993 * Tk + 2839.319 / Tk + 71.595 * log10(Tk);
  tmp2 = +(-0.77712 + 0.0028426 * Tk + 178.34 / Tk) * sqrt(S);
  tmp3 = -0.07711 * S + 0.0041249 * pow(S, 1.5);
  const double Kspc = pow(10.0, tmp1 + tmp2 + tmp3);

  tmp1 = -171.945 - 0.077993 * Tk + 2903.293 / Tk + 71.595 * log10(Tk);
  tmp2 = +(-0.068393 + 0.0017276 * Tk + 88.135 / Tk) * sqrt(S);
  tmp3 = -0.10018 * S + 0.0059415 * pow(S, 1.5);
  const double Kspa = pow(10.0, tmp1 + tmp2 + tmp3);

  const double bor = 1 * (416.0 * (S / 35.0)) * 1.e-6;
  const double Kw_val = c->Kw;

  double a[6];
  const double p5 = -1.0;
  const double p4 = -alk - Kb_val - K1_val;
  const double p3 = dic * K1_val - alk * (Kb_val + K1_val) + Kb_val * bor +
                    Kw_val - Kb_val * K1_val - K1_val * K2_val;
  tmp = dic * (Kb_val * K1_val + 2.0 * K1_val * K2_val) -
        alk * (Kb_val * K1_val + K1_val * K2_val) + Kb_val * bor * K1_val;
  const double p2 =
      tmp + (Kw_val * Kb_val + Kw_val * K1_val - Kb_val * K1_val * K2_val);
  tmp = 2.0 * dic * Kb_val * K1_val * K2_val - alk * Kb_val * K1_val * K2_val +
        Kb_val * bor * K1_val * K2_val;
  const double p1 = tmp + (Kw_val * Kb_val * K1_val + Kw_val * K1_val * K2_val);
  const double p0 = Kw_val * Kb_val * K1_val * K2_val;
  a[0] = p0; a[1] = p1; a[2] = p2; a[3] = p3; a[4] = p4; a[5] = p5;

  const double h = find_largest_root(a, err);
  c->h = h;

  const double co2st = dic / (1.0 + K1_val / h + K1_val * K2_val / h / h);
  const double hco3 = dic / (1.0 + h / K1_val + K2_val / h);
  const double co3 = dic / (1.0 + h / K2_val + h * h / K1_val / K2_val);
  const double million = 1e6;
  c->TCO2o = co2st * million;
  c->HCO3 = hco3 * million;
  c->CO3 = co3 * million;
  c->PCO2o = co2st * million / c->Kh;
  c->pH = -log10(h);
  c->Tr = (0.585 * c->K0 * pow(Sc, -0.5) * c->U * c->U);
  const double calcium = 0.02128 / 40.087 * (S / 1.80655);
  c->OmegaCa = ((co3 * calcium) / Kspc);
  c->OmegaAr = ((co3 * calcium) / Kspa);
}

/* calc_annual_surface_flux  src/ocean_csys.cpp:375-396 */
static double csys_annual_flux(const csys_t *c, double CO2_conc,
                               double cpoolscale) {
  double monthly = ((CO2_conc - c->PCO2o * cpoolscale) * c->Tr);
  return (monthly * c->As * 12.0) / 1e15;
}

void hxo_csys(double Tc, double carbon, double alk, double volume,
              double *out) {
  csys_t c;
  memset(&c, 0, sizeof c);
  c.S = 34.5; c.U = 6.7; c.alk = alk; c.volumeofbox = volume; c.As = 1.0;
  int err = 0;
  csys_run(&c, Tc, carbon, &err);
  out[0] = c.PCO2o; out[1] = c.pH; out[2] = c.Tr; out[3] = c.K0;
  out[4] = c.h; out[5] = c.CO3;
}

/* ------------------------------------------------------------------ */
/* member state                                                        */
/* ------------------------------------------------------------------ */
enum { HL = 0, LL = 1, IO = 2, DO = 3 };
#define SNBOX_ATMOS 0
#define SNBOX_VEG 1
#define SNBOX_DET 2
#define SNBOX_SOIL 3
#define SNBOX_PERMAFROST 4
#define SNBOX_THAWEDP 5
#define SNBOX_OCEAN 6
#define SNBOX_EARTH 7
#define PGC_TO_PPMVCO2 (1.0 / 2.13)      /* carbon-cycle-model.hpp:29 */
#define PPMVCO2_TO_PGC (1.0 / PGC_TO_PPMVCO2)
#define CARBON_CYCLE_RETRY 1234
#define MAX_RETRIES 8                    /* carbon-cycle-solver.hpp:23 */
#define MB_EPSILON 0.001                 /* simpleNbox.hpp:37 */
#define Q10_TEMPN 200                    /* simpleNbox-runtime.cpp:1042 */

/* ---- carbon tracking: a pool or flux with the origin of its carbon (fluxpool.hpp) ---- */
#define TP_MAX (2 + 5 * HXO_MAXB + 4) /* atmos, earth, 5 pools per biome, 4 ocean boxes */
typedef struct {
  double val;
  double f[TP_MAX]; /* fraction that originated in pool s (0 if s is not in the map) */
  unsigned __int128 mask; /* which sources are in the map (ctmap keys); 86 pools for 16 biomes */
} tv_t;

typedef struct {
  const hxo_scenario *sc;
  const hxo_params *pa;
  int B;
  int err;
  int core_in_spinup;
  /* ---- ocean ---- */
  double carbon[4], additions[4], subtractions[4];
  double atmflux[2], ao_flux[2], oa_flux[2], preind_flux[2], Tbox[2], deltaT[2];
  int active_chem;
  csys_t chem[2];
  /* connections: from, to, k in the reference's compute order */
  double k_LL_HL, k_LL_IO, k_HL_DO, k_IO_LL, k_IO_HL, k_IO_DO, k_DO_IO;
  double max_timestep, lastflux_annualized;
  int reduced_timestep_timeout, timesteps, ocean_in_spinup;
  double annualflux_sum, annualflux_sumHL, annualflux_sumLL;
  double ocean_ODEstartdate, SST, ocean_CO2_conc;
  /* ---- simpleNbox ---- */
  double atmos_c, earth_c, cumulative_pf_ch4, masstot;
  double veg_c[HXO_MAXB], detritus_c[HXO_MAXB], soil_c[HXO_MAXB],
      permafrost_c[HXO_MAXB], thawed_c[HXO_MAXB];
  double co2fert[HXO_MAXB], tempfertd[HXO_MAXB], tempferts[HXO_MAXB],
      f_frozen[HXO_MAXB], f_new_thaw[HXO_MAXB];
  double tempferts_recorded[HXO_MAXB]; /* tempferts_tv at last record_state */
  double RH_ch4_sum;                   /* sum_b RH_ch4[b] as last recorded */
  double cum_luc_va, end_of_spinup_vegc, npp_luc_adjust, nbp;
  double cur_luc_e, cur_luc_u, cur_ffi, cur_daccs;
  int snbox_in_spinup, has_been_run_before;
  double ODEstartdate;
  double *Tland_record; /* [ns], index Y-start holds Tland_record[Y] */
  int Tland_first;      /* first index set, -1 if none */
  /* ---- solver ---- */
  double t, dt;
  long nsteps_year, nrhs_year;
  /* ---- gases ---- */
  double ch4_prev, n2o_prev, tau_oh, o3, ch4, n2o;
  double M0_eff, N0_eff; /* preindustrial values after prepareToRun's constraint override */
  double halo_conc[HXO_NHALO], halo_rf[HXO_NHALO];
  /* ---- forcing ---- */
  int have_base;
  double base_tot, base_co2, base_ch4, base_n2o;
  double rf_tot, rf_co2, rf_ch4, rf_n2o;
  /* ---- DOECLIM ---- */
  int ns;
  double *Ker, *forcing, *temp, *temp_landair, *temp_sst, *heatflux_mixed,
      *heatflux_interior;
  double A[4], IB[4], taucfl, taukls, taucfs, tauksl, taudif, powtoheat;
  double tas_land, sst_now; /* undated D_LAND_TAS, D_SST */
  double Ca_residual;
  /* carbon tracking shadow (values mirror the fluxpools' running values) */
  int trk_on, trk_iy, TP;
  tv_t trk[TP_MAX];      /* the pools */
  tv_t trk_addn[4];      /* oceanbox::CarbonAdditions */
  double trk_subn[4];    /* oceanbox::CarbonSubtractions (value only) */
  tv_t trk_atm_copy;     /* OceanComponent::atmosphere_cpool: the atmosphere as of SimpleNbox::run */
  tv_t trk_ao[2], trk_oa[2];
  double *trk_out_f, *trk_out_v; /* [ns][TP][TP], [ns][TP] or NULL */
  double hl_do;                                  /* annual_box_fluxes[HL -> DO] */
  double final_npp, final_rh, final_rh_det, final_rh_soil;
  double final_npp_b[HXO_MAXB], final_rh_b[HXO_MAXB];
  double temp_surface_now, flux_mixed_now, flux_interior_now;
  double rf_item_v[16];                          /* RF_* relative to the base year, see year_forcing */
  double rf_base_v[16];
  double slr[4]; int slr_have; double refperiod_tgav;
} member_t;

/* tseries::exists for a constraint held as a NaN-filled dense series */
static int con_has(const double *c, int iy) { return c && !isnan(c[iy]); }

/* DOECLIM hard-coded parameters  inst/include/temperature_component.hpp:77-98 */
static const double d_dt = 1, d_ak = 0.31, d_bk = 1.59, d_csw = 0.13,
                    d_earth_area = 5100656E8,
                    d_secs_per_Year = 60.0 * 60.0 * 24.0 * 365.2422,
                    d_rlam = 1.43, d_zbot = 4000.0, d_bsi = 1.3, d_cal = 0.52,
                    d_cas = 7.80, d_flnd = 0.29, d_fso = 0.95;

/* kernel  src/temperature_component.cpp:303-371 */
static void doeclim_kernel(double diff, int ns, double *Ker) {
  const double dt = d_dt;
  double kcon = d_secs_per_Year / 10000;
  double keff = kcon * diff;
  double taubot = pow(d_zbot, 2) / keff;
  double *KT0 = (double *)calloc((size_t)ns * 7, sizeof(double));
  double *KTA1 = KT0 + ns, *KTB1 = KTA1 + ns, *KTA2 = KTB1 + ns,
         *KTB2 = KTA2 + ns, *KTA3 = KTB2 + ns, *KTB3 = KTA3 + ns;
  KT0[ns - 1] = 4.0 - 2.0 * pow(2.0, 0.5);
  KTA1[ns - 1] =
      -8.0 * exp(-taubot / dt) + 4.0 * pow(2.0, 0.5) * exp(-0.5 * taubot / dt);
  KTB1[ns - 1] = 4.0 * pow((M_PI * taubot / dt), 0.5) *
                 (1.0 + erf(pow(0.5 * taubot / dt, 0.5)) -
                  2.0 * erf(pow(taubot / dt, 0.5)));
  KTA2[ns - 1] = 8.0 * exp(-4.0 * taubot / dt) -
                 4.0 * pow(2.0, 0.5) * exp(-2.0 * taubot / dt);
  KTB2[ns - 1] = -8.0 * pow((M_PI * taubot / dt), 0.5) *
                 (1.0 + erf(pow((2.0 * taubot / dt), 0.5)) -
                  2.0 * erf(2.0 * pow((taubot / dt), 0.5)));
  KTA3[ns - 1] = -8.0 * exp(-9.0 * taubot / dt) +
                 4.0 * pow(2.0, 0.5) * exp(-4.5 * taubot / dt);
  KTB3[ns - 1] = 12.0 * pow((M_PI * taubot / dt), 0.5) *
                 (1.0 + erf(pow((4.5 * taubot / dt), 0.5)) -
                  2.0 * erf(3.0 * pow((taubot / dt), 0.5)));
  for (int i = 0; i < (ns - 1); i++) {
    KT0[i] = 4.0 * pow((double)(ns - i), 0.5) -
             2.0 * pow((double)(ns + 1 - i), 0.5) -
             2.0 * pow((double)(ns - 1 - i), 0.5);
    KTA1[i] =
        -8.0 * pow((double)(ns - i), 0.5) * exp(-taubot / dt / (double)(ns - i)) +
        4.0 * pow((double)(ns + 1 - i), 0.5) *
            exp(-taubot / dt / (double)(ns + 1 - i)) +
        4.0 * pow((double)(ns - 1 - i), 0.5) *
            exp(-taubot / dt / (double)(ns - 1 - i));
    KTB1[i] = 4.0 * pow((M_PI * taubot / dt), 0.5) *
              (erf(pow((taubot / dt / (double)(ns - 1 - i)), 0.5)) +
               erf(pow((taubot / dt / (double)(ns + 1 - i)), 0.5)) -
               2.0 * erf(pow((taubot / dt / (double)(ns - i)), 0.5)));
    KTA2[i] = 8.0 * pow((double)(ns - i), 0.5) *
                  exp(-4.0 * taubot / dt / (double)(ns - i)) -
              4.0 * pow((double)(ns + 1 - i), 0.5) *
                  exp(-4.0 * taubot / dt / (double)(ns + 1 - i)) -
              4.0 * pow((double)(ns - 1 - i), 0.5) *
                  exp(-4.0 * taubot / dt / (double)(ns - 1 - i));
    KTB2[i] = -8.0 * pow((M_PI * taubot / dt), 0.5) *
              (erf(2.0 * pow((taubot / dt / (double)(ns - 1 - i)), 0.5)) +
               erf(2.0 * pow((taubot / dt / (double)(ns + 1 - i)), 0.5)) -
               2.0 * erf(2.0 * pow((taubot / dt / (double)(ns - i)), 0.5)));
    KTA3[i] = -8.0 * pow((double)(ns - i), 0.5) *
                  exp(-9.0 * taubot / dt / (double)(ns - i)) +
              4.0 * pow((double)(ns + 1 - i), 0.5) *
                  exp(-9.0 * taubot / dt / (double)(ns + 1 - i)) +
              4.0 * pow((double)(ns - 1 - i), 0.5) *
                  exp(-9.0 * taubot / dt / (double)(ns - 1 - i));
    KTB3[i] = 12.0 * pow((M_PI * taubot / dt), 0.5) *
              (erf(3.0 * pow((taubot / dt / (double)(ns - 1 - i)), 0.5)) +
               erf(3.0 * pow((taubot / dt / (double)(ns + 1 - i)), 0.5)) -
               2.0 * erf(3.0 * pow((taubot / dt / (double)(ns - i)), 0.5)));
  }
  for (int i = 0; i < ns; i++)
    Ker[i] = KT0[i] + KTA1[i] + KTB1[i] + KTA2[i] + KTB2[i] + KTA3[i] + KTB3[i];
  free(KT0);
}

void hxo_doeclim_kernel(double diff, int ns, double *ker) {
  doeclim_kernel(diff, ns, ker);
}

/* TemperatureComponent::prepareToRun  src/temperature_component.cpp:196-413 */
static void doeclim_prepare(member_t *m) {
  const double dt = d_dt, flnd = d_flnd, bsi = d_bsi, rlam = d_rlam,
               ak = d_ak, bk = d_bk, cal = d_cal, cas = d_cas, fso = d_fso;
  const double S = m->pa->S, qco2 = m->pa->qco2, diff = m->pa->diff;
  const int ns = m->ns;
  double B[4], C[4];
  for (int i = 0; i < 4; i++) { B[i] = 0.0; C[i] = 0.0; }
  double kcon = d_secs_per_Year / 10000;
  double ocean_area = (1.0 - flnd) * d_earth_area;
  double cnum = rlam * flnd + bsi * (1.0 - flnd);
  double cden = rlam * flnd - ak * (rlam - bsi);
  double cfl = flnd * cnum / cden * qco2 / S - bk * (rlam - bsi) / cden;
  double cfs = (rlam * flnd - ak / (1.0 - flnd) * (rlam - bsi)) * cnum / cden *
                   qco2 / S +
               rlam * flnd / (1.0 - flnd) * bk * (rlam - bsi) / cden;
  double kls = bk * rlam * flnd / cden - ak * flnd * cnum / cden * qco2 / S;
  double keff = kcon * diff;
  m->powtoheat = ocean_area * d_secs_per_Year / pow(10.0, 22);
  double taucfs = cas / cfs;
  double taucfl = cal / cfl;
  double taudif = pow(cas, 2) / pow(d_csw, 2) * M_PI / keff;
  double tauksl = (1.0 - flnd) * cas / kls;
  double taukls = flnd * cal / kls;
  m->taucfs = taucfs; m->taucfl = taucfl; m->taudif = taudif;
  m->tauksl = tauksl; m->taukls = taukls;

  doeclim_kernel(diff, ns, m->Ker);

  C[0] = 1.0 / pow(taucfl, 2.0) + 1.0 / pow(taukls, 2.0) +
         2.0 / taucfl / taukls + bsi / taukls / tauksl;
  C[1] = -1 * bsi / pow(taukls, 2.0) - bsi / taucfl / taukls -
         bsi / taucfs / taukls - pow(bsi, 2.0) / taukls / tauksl;
  C[2] = -1 * bsi / pow(tauksl, 2.0) - 1.0 / taucfs / tauksl -
         1.0 / taucfl / tauksl - 1.0 / taukls / tauksl;
  C[3] = 1.0 / pow(taucfs, 2.0) + pow(bsi, 2.0) / pow(tauksl, 2.0) +
         2.0 * bsi / taucfs / tauksl + bsi / taukls / tauksl;
  for (int i = 0; i < 4; i++) C[i] = C[i] * (pow(dt, 2.0) / 12.0);

  B[0] = 1.0 + dt / (2.0 * taucfl) + dt / (2.0 * taukls);
  B[1] = -dt / (2.0 * taukls) * bsi;
  B[2] = -dt / (2.0 * tauksl);
  B[3] = 1.0 + dt / (2.0 * taucfs) + dt / (2.0 * tauksl) * bsi +
         2.0 * fso * pow((dt / taudif), 0.5);
  m->A[0] = 1.0 - dt / (2.0 * taucfl) - dt / (2.0 * taukls);
  m->A[1] = dt / (2.0 * taukls) * bsi;
  m->A[2] = dt / (2.0 * tauksl);
  m->A[3] = 1.0 - dt / (2.0 * taucfs) - dt / (2.0 * tauksl) * bsi +
            m->Ker[ns - 1] * fso * pow((dt / taudif), 0.5);
  for (int i = 0; i < 4; i++) {
    B[i] = B[i] + C[i];
    m->A[i] = m->A[i] + C[i];
  }
  /* invert_1d_2x2_matrix  src/temperature_component.cpp:81-94 */
  double temp_d = (B[0] * B[3] - B[1] * B[2]);
  if (temp_d == 0) m->err |= HXO_ERR_SINGULAR;
  double temp = 1 / temp_d;
  m->IB[0] = temp * B[3];
  m->IB[1] = temp * -1 * B[1];
  m->IB[2] = temp * -1 * B[2];
  m->IB[3] = temp * B[0];
}

/* TemperatureComponent::run  src/temperature_component.cpp:417-557 */
static void doeclim_run(member_t *m, int tstep, double rf_total) {
  const double dt = d_dt, flnd = d_flnd, bsi = d_bsi, cal = d_cal,
               cas = d_cas, fso = d_fso;
  const int ns = m->ns;
  double *QL = m->forcing, *QO = m->forcing;
  m->forcing[tstep] = rf_total;
  double DQ1 = 0.0, DQ2 = 0.0, QC1 = 0.0, QC2 = 0.0, DelQL = 0.0, DelQO = 0.0,
         DPAST1 = 0.0, DPAST2 = 0.0, DTEAUX1 = 0.0, DTEAUX2 = 0.0;
  m->temp[tstep] = 0.0;
  m->temp_landair[tstep] = 0.0;
  m->temp_sst[tstep] = 0.0;
  m->heatflux_mixed[tstep] = 0.0;
  m->heatflux_interior[tstep] = 0.0;
  if (tstep > 0) {
    DelQL = QL[tstep] - QL[tstep - 1];
    DelQO = QO[tstep] - QO[tstep - 1];
    QC1 = (DelQL / cal * (1.0 / m->taucfl + 1.0 / m->taukls) -
           bsi * DelQO / cas / m->taukls);
    QC2 = (DelQO / cas * (1.0 / m->taucfs + bsi / m->tauksl) -
           DelQL / cal / m->tauksl);
    QC1 = QC1 * pow(dt, 2.0) / 12.0;
    QC2 = QC2 * pow(dt, 2.0) / 12.0;
    DQ1 = 0.5 * dt / cal * (QL[tstep] + QL[tstep - 1]);
    DQ2 = 0.5 * dt / cas * (QO[tstep] + QO[tstep - 1]);
    DQ1 = DQ1 + QC1;
    DQ2 = DQ2 + QC2;
    for (int i = 0; i <= tstep; i++)
      DPAST2 = DPAST2 + m->temp_sst[i] * m->Ker[ns - tstep + i - 1];
    DPAST2 = DPAST2 * fso * pow((dt / m->taudif), 0.5);
    DTEAUX1 = m->A[0] * m->temp_landair[tstep - 1] + m->A[1] * m->temp_sst[tstep - 1];
    DTEAUX2 = m->A[2] * m->temp_landair[tstep - 1] + m->A[3] * m->temp_sst[tstep - 1];
    m->temp_landair[tstep] = m->IB[0] * (DQ1 + DPAST1 + DTEAUX1) +
                             m->IB[1] * (DQ2 + DPAST2 + DTEAUX2);
    m->temp_sst[tstep] = m->IB[2] * (DQ1 + DPAST1 + DTEAUX1) +
                         m->IB[3] * (DQ2 + DPAST2 + DTEAUX2);
  } else {
    m->temp_landair[0] = 0.0;
    m->temp_sst[0] = 0.0;
  }
  m->temp[tstep] = flnd * m->temp_landair[tstep] +
                   (1.0 - flnd) * bsi * m->temp_sst[tstep];
  /* user-supplied temperature  temperature_component.cpp:510-525 */
  if (con_has(m->sc->tas_con, tstep)) {
    m->temp[tstep] = m->sc->tas_con[tstep];
    m->temp_landair[tstep] =
        (m->temp[tstep] - (1.0 - flnd) * bsi * m->temp_sst[tstep]) / flnd;
    m->temp_sst[tstep] =
        (m->temp[tstep] - flnd * m->temp_landair[tstep]) / ((1.0 - flnd) * bsi);
  }
  if (tstep > 0) {
    m->heatflux_mixed[tstep] = cas * (m->temp_sst[tstep] - m->temp_sst[tstep - 1]);
    for (int i = 0; i < tstep; i++)
      m->heatflux_interior[tstep] =
          m->heatflux_interior[tstep] + m->temp_sst[i] * m->Ker[ns - tstep + i];
    m->heatflux_interior[tstep] =
        cas * fso / pow((m->taudif * dt), 0.5) *
        (2.0 * m->temp_sst[tstep] - m->heatflux_interior[tstep]);
  } else {
    m->heatflux_mixed[0] = 0.0;
    m->heatflux_interior[0] = 0.0;
  }
  /* setoutputs  src/temperature_component.cpp:706-746 */
  m->temp_surface_now = flnd * m->temp_landair[tstep] + (1.0 - flnd) * m->temp_sst[tstep];
  m->flux_mixed_now = m->heatflux_mixed[tstep];
  m->flux_interior_now = m->heatflux_interior[tstep];
  m->tas_land = m->temp_landair[tstep];
  m->sst_now = m->temp_sst[tstep];
  /* land-ocean warming ratio override :722-739; what D_LAND_TAS / D_SST return :586-625 */
  const double lo = m->pa->lo_warming_ratio;
  if (lo != 0) {
    double temp_oceanair_constrain = m->temp[tstep] / ((lo * flnd) + (1 - flnd));
    double temp_landair_constrain = temp_oceanair_constrain * lo;
    double temp_sst_constrain = temp_oceanair_constrain / bsi;
    m->tas_land = temp_landair_constrain;
    m->sst_now = temp_sst_constrain;
  }
}

/* ------------------------------------------------------------------ */
/* carbon tracking arithmetic   inst/include/fluxpool.hpp:166-298     */
/* ------------------------------------------------------------------ */
enum { TP_ATM = 0, TP_EARTH = 1 };
#define TP_LAND(b, k) (2 + 5 * (b) + (k)) /* k: 0 veg 1 det 2 soil 3 permafrost 4 thawed */
#define TP_OCEAN(m, box) (2 + 5 * (m)->B + (box))

static tv_t tv_self(int self, double val) { /* fluxpool::set: ctmap[name] = 1 */
  tv_t r;
  memset(&r, 0, sizeof r);
  r.val = val; r.f[self] = 1.0; r.mask = (unsigned __int128)1 << self;
  return r;
}
/* flux_from_fluxpool / flux_from_unitval: the pool's origins, another value */
static tv_t tv_from(const tv_t *pool, double val) { tv_t r = *pool; r.val = val; return r; }
/* operator+ : origins mixed by value; a zero total shares equally among the sources */
static tv_t tv_add(tv_t a, tv_t b, int TP) {
  tv_t r;
  memset(&r, 0, sizeof r);
  r.val = a.val + b.val;
  r.mask = a.mask | b.mask;
  int nsrc = 0;
  for (int s = 0; s < TP; s++) if ((r.mask >> s) & 1) nsrc++;
  for (int s = 0; s < TP; s++) {
    if (!((r.mask >> s) & 1)) continue;
    const double pool_s = a.val * a.f[s] + b.val * b.f[s];
    r.f[s] = (r.val != 0.0) ? pool_s / r.val : 1.0 / nsrc;
  }
  return r;
}
static tv_t tv_sub(tv_t a, tv_t b) { a.val = a.val - b.val; return a; } /* operator- */
static tv_t tv_mul(tv_t a, double k) { a.val = a.val * k; return a; }   /* operator* */

/* ------------------------------------------------------------------ */
/* ocean component                                                     */
/* ------------------------------------------------------------------ */
static double ocean_totalcpool(const member_t *m) {
  /* src/ocean_component.cpp:325-328 */
  return m->carbon[DO] + m->carbon[IO] + m->carbon[LL] + m->carbon[HL];
}

/* OceanComponent::prepareToRun  src/ocean_component.cpp:202-319 */
static void ocean_prepare(member_t *m) {
  const hxo_params *p = m->pa;
  const double part_high = 0.15;
  const double part_low = 1 - part_high;
  const double spy = 60 * 60 * 24 * 365.25;
  const double thick_LL = 100, thick_HL = 100;
  const double thick_inter = 1000 - thick_LL;
  const double thick_deep = 3777 - thick_inter - thick_LL;
  const double ocean_area = 3.6e14;
  const double LL_volume = ocean_area * part_low * thick_LL;
  const double HL_volume = ocean_area * part_high * thick_HL;
  const double I_volume = ocean_area * thick_inter;
  const double D_volume = ocean_area * thick_deep;
  const double LL_vol_frac = LL_volume / (LL_volume + HL_volume);
  const double HL_vol_frac = 1 - LL_vol_frac;
  const double I_vol_frac = I_volume / (I_volume + D_volume);
  const double D_vol_frac = 1 - I_vol_frac;
  m->carbon[LL] = LL_vol_frac * p->preind_surface_c;
  m->carbon[HL] = HL_vol_frac * p->preind_surface_c;
  m->carbon[IO] = I_vol_frac * p->preind_interdeep_c;
  m->carbon[DO] = D_vol_frac * p->preind_interdeep_c;
  for (int i = 0; i < 4; i++) m->additions[i] = m->subtractions[i] = 0.0;
  m->preind_flux[HL] = 1.000;
  m->preind_flux[LL] = -1.000;
  m->active_chem = m->sc->spinup_chem;
  double LL_HL = (p->tt * spy) / LL_volume;
  double HL_DO = ((p->tt + p->tu) * spy) / HL_volume;
  double DO_IO = ((p->tt + p->tu) * spy) / D_volume;
  double IO_HL = (p->tu * spy) / I_volume;
  double IO_LL = (p->tt * spy) / I_volume;
  double IO_LLex = (p->twi * spy) / I_volume;
  double LL_IOex = (p->twi * spy) / LL_volume;
  double DO_IOex = (p->tid * spy) / D_volume;
  double IO_DOex = (p->tid * spy) / I_volume;
  m->k_LL_HL = LL_HL;
  m->k_LL_IO = LL_IOex;
  m->k_HL_DO = HL_DO;
  m->k_IO_LL = IO_LL + IO_LLex;
  m->k_IO_HL = IO_HL;
  m->k_IO_DO = IO_DOex;
  m->k_DO_IO = DO_IO + DO_IOex;
  memset(m->chem, 0, sizeof m->chem);
  m->deltaT[HL] = -16.4;
  m->chem[HL].S = 34.5; m->chem[HL].volumeofbox = HL_volume;
  m->chem[HL].As = ocean_area * part_high; m->chem[HL].U = 6.7;
  m->deltaT[LL] = 2.9;
  m->chem[LL].S = 34.5; m->chem[LL].volumeofbox = LL_volume;
  m->chem[LL].As = ocean_area * part_low; m->chem[LL].U = 6.7;
  m->annualflux_sum = m->annualflux_sumHL = m->annualflux_sumLL = 0.0;
  m->SST = 0.0;
  m->lastflux_annualized = 0.0;
  m->max_timestep = 1.0;           /* ocean_component.cpp:76-77 */
  m->reduced_timestep_timeout = 0;
  m->Tbox[HL] = m->Tbox[LL] = -999;
  m->atmflux[0] = m->atmflux[1] = 0;
  m->ao_flux[0] = m->ao_flux[1] = m->oa_flux[0] = m->oa_flux[1] = 0;
}

static void box_separate(member_t *m, int b) { /* oceanbox.cpp:262-271 */
  if (m->atmflux[b] > 0) { m->ao_flux[b] = m->atmflux[b]; m->oa_flux[b] = 0.0; }
  else { m->ao_flux[b] = 0.0; m->oa_flux[b] = -m->atmflux[b]; }
}

static void box_transfer(member_t *m, int from, int to, double k, double yf) {
  double closs = m->carbon[from] * k * yf; /* oceanbox.cpp:246 */
  m->additions[to] = m->additions[to] + closs;
  m->subtractions[from] = m->subtractions[from] + closs;
  if (from == HL && to == DO) m->hl_do = m->hl_do + closs; /* annual_box_fluxes :254-255 */
}

/* oceanbox::compute_fluxes  src/oceanbox.cpp:203-260 */
static void box_compute_fluxes(member_t *m, int b, double CO2_conc, double yf,
                               int do_circ) {
  if (b == HL || b == LL) {
    if (m->active_chem) {
      csys_run(&m->chem[b], m->Tbox[b], m->carbon[b], &m->err);
      m->atmflux[b] = csys_annual_flux(&m->chem[b], CO2_conc, 1.0);
    } else {
      m->atmflux[b] = m->preind_flux[b];
    }
    m->atmflux[b] = m->atmflux[b] * yf;
    box_separate(m, b);
  }
  if (do_circ) {
    switch (b) { /* connection order: src/ocean_component.cpp:277-283 */
    case HL: box_transfer(m, HL, DO, m->k_HL_DO, yf); break;
    case LL: box_transfer(m, LL, HL, m->k_LL_HL, yf);
             box_transfer(m, LL, IO, m->k_LL_IO, yf); break;
    case IO: box_transfer(m, IO, LL, m->k_IO_LL, yf);
             box_transfer(m, IO, HL, m->k_IO_HL, yf);
             box_transfer(m, IO, DO, m->k_IO_DO, yf); break;
    case DO: box_transfer(m, DO, IO, m->k_DO_IO, yf); break;
    }
  }
}

static void box_update_state(member_t *m, int b) { /* oceanbox.cpp:297-303 */
  double ao = (b < 2) ? m->ao_flux[b] : 0.0, oa = (b < 2) ? m->oa_flux[b] : 0.0;
  m->carbon[b] = m->carbon[b] + m->additions[b] + ao - oa - m->subtractions[b];
  m->additions[b] = 0.0;
  m->subtractions[b] = 0.0;
}

/* oceanbox::fmin  src/oceanbox.cpp:335-350 */
static __thread double g_rounding_noise; /* conditioning probe of the test suite, see below */
static __thread unsigned long long noise_state;
static double noise_xi(void) { /* deterministic, in [-1, 1) */
  noise_state = noise_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)((noise_state >> 33) & 0xFFFFF) / 524288.0 - 1.0;
}
static double box_fmin(member_t *m, int b, double alk, double f_target) {
  m->chem[b].alk = alk;
  csys_run(&m->chem[b], m->Tbox[b], m->carbon[b], &m->err);
  double flux = csys_annual_flux(&m->chem[b], m->ocean_CO2_conc, 1.0);
  if (g_rounding_noise != 0.0) flux *= 1.0 + g_rounding_noise * noise_xi();
  return fabs(flux - f_target);
}

/* oceanbox::chem_equilibrate  src/oceanbox.cpp:382-445, with
 * boost::math::tools::brent_find_minima(f, min, max, bits) restated
 * (minima.hpp).  NB: the alkalinity that survives is the LAST point handed to
 * f, not the returned minimiser (fmin sets alk as a side effect). */
static void box_chem_equilibrate(member_t *m, int b) {
  double alk_min = 2100e-6, alk_max = 2750e-6;
  double f_target = m->preind_flux[b];
  for (double alk1 = alk_min; alk1 <= alk_max; alk1 += (alk_max - alk_min) / 20)
    (void)box_fmin(m, b, alk1, f_target);
  int bits = (int)(53 * 0.6);
  /* brent_find_minima */
  bits = (53 / 2 < bits) ? 53 / 2 : bits;
  double tolerance = ldexp(1.0, 1 - bits);
  double min = alk_min, max = alk_max;
  double x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
  static const double golden = 0.3819660f;
  x = w = v = max;
  fw = fv = fx = box_fmin(m, b, x, f_target);
  delta2 = delta = 0;
  long count = 1000000;
  do {
    mid = (min + max) / 2;
    fract1 = tolerance * fabs(x) + tolerance / 4;
    fract2 = 2 * fract1;
    if (fabs(x - mid) <= (fract2 - (max - min) / 2)) break;
    if (fabs(delta2) > fract1) {
      double r = (x - w) * (fx - fv);
      double q = (x - v) * (fx - fw);
      double p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = fabs(q);
      double td = delta2;
      delta2 = delta;
      if ((fabs(p) >= fabs(q * td / 2)) || (p <= q * (min - x)) ||
          (p >= q * (max - x))) {
        delta2 = (x >= mid) ? min - x : max - x;
        delta = golden * delta2;
      } else {
        delta = p / q;
        u = x + delta;
        if (((u - min) < fract2) || ((max - u) < fract2))
          delta = (mid - x) < 0 ? -fabs(fract1) : fabs(fract1);
      }
    } else {
      delta2 = (x >= mid) ? min - x : max - x;
      delta = golden * delta2;
    }
    u = (fabs(delta) >= fract1)
            ? (x + delta)
            : (delta > 0 ? (x + fabs(fract1)) : (x - fabs(fract1)));
    fu = box_fmin(m, b, u, f_target);
    if (fu <= fx) {
      if (u >= x) min = x; else max = x;
      v = w; w = x; x = u;
      fv = fw; fw = fx; fx = fu;
    } else {
      if (u < x) min = u; else max = u;
      if ((fu <= fw) || (w == x)) {
        v = w; w = u; fv = fw; fw = fu;
      } else if ((fu <= fv) || (v == x) || (v == w)) {
        v = u; fv = fu;
      }
    }
  } while (--count);
}

/* OceanComponent::run  src/ocean_component.cpp:356-407 */
static void ocean_run(member_t *m) {
  /* CO2_conc = D_CO2_CONC at runToDate: not yet recorded -> flat extrapolation
   * of atmos_c_ts = the current atmos_c (tseries.hpp:319-334,
   * h_interpolator.cpp:115-118) */
  m->ocean_CO2_conc = m->atmos_c * PGC_TO_PPMVCO2;
  m->SST = m->sst_now;
  m->ocean_in_spinup = m->core_in_spinup;
  m->annualflux_sum = m->annualflux_sumHL = m->annualflux_sumLL = 0.0;
  m->timesteps = 0;
  /* new_year: Tbox = SST + MEAN_TOS_TEMP + deltaT  oceanbox.cpp:97-99,309-323 */
  m->Tbox[HL] = m->SST + 18 + m->deltaT[HL];
  m->Tbox[LL] = m->SST + 18 + m->deltaT[LL];
  m->atmflux[HL] = 0.0;
  m->atmflux[LL] = 0.0;
  m->hl_do = 0.0; /* new_year  oceanbox.cpp:309-313 */
  if (!m->sc->spinup_chem && !m->ocean_in_spinup && !m->active_chem) {
    m->active_chem = 1;
    box_chem_equilibrate(m, HL);
    box_chem_equilibrate(m, LL);
  }
  box_compute_fluxes(m, HL, m->ocean_CO2_conc, 1.0, 0);
  box_compute_fluxes(m, LL, m->ocean_CO2_conc, 1.0, 0);
}

/* OceanComponent::calcderivs  src/ocean_component.cpp:603-626 */
static int ocean_calcderivs(const member_t *m, double t, const double c[],
                            double dcdt[]) {
  const double yearfraction = (t - m->ocean_ODEstartdate);
  const double cpooldiff = c[SNBOX_OCEAN] - ocean_totalcpool(m);
  const double surfacepools = m->carbon[LL] + m->carbon[HL];
  const double cpoolscale = (surfacepools + cpooldiff) / surfacepools;
  double CO2_conc = c[SNBOX_ATMOS] * PGC_TO_PPMVCO2;
  double flux;
  if (m->ocean_in_spinup && !m->sc->spinup_chem)
    flux = m->preind_flux[HL] + m->preind_flux[LL];
  else
    flux = csys_annual_flux(&m->chem[HL], CO2_conc, cpoolscale) +
           csys_annual_flux(&m->chem[LL], CO2_conc, cpoolscale);
  dcdt[SNBOX_OCEAN] = flux;
  if (yearfraction > m->max_timestep) return CARBON_CYCLE_RETRY;
  return 0;
}

/* The origins side of OceanComponent::stashCValues: oceanbox::compute_fluxes step 4
 * (oceanbox.cpp:240-257), separate_surface_fluxes (:262-271) and update_state (:297-303),
 * boxes in the order HL, LL, intermediate, deep.  Called with the pre-update box carbon. */
static void trk_ocean_stash(member_t *m, double yf) {
  const int TP = m->TP;
  static const int from_[7] = {HL, LL, LL, IO, IO, IO, DO};
  static const int to_[7] = {DO, HL, IO, LL, HL, DO, IO};
  const double k_[7] = {m->k_HL_DO, m->k_LL_HL, m->k_LL_IO, m->k_IO_LL,
                        m->k_IO_HL, m->k_IO_DO, m->k_DO_IO};
  for (int b = 0; b < 4; b++) m->trk[TP_OCEAN(m, b)].val = m->carbon[b];
  for (int i = 0; i < 7; i++) {
    tv_t closs = tv_mul(tv_mul(m->trk[TP_OCEAN(m, from_[i])], k_[i]), yf); /* carbon * k * yf */
    m->trk_addn[to_[i]] = tv_add(m->trk_addn[to_[i]], closs, TP);        /* add_carbon */
    m->trk_subn[from_[i]] = m->trk_subn[from_[i]] + closs.val;
  }
  for (int b = 0; b < 2; b++) { /* surface boxes: final separate_surface_fluxes */
    if (m->atmflux[b] > 0) {
      m->trk_ao[b] = tv_from(&m->trk_atm_copy, m->atmflux[b]);
      m->trk_oa[b] = tv_from(&m->trk[TP_OCEAN(m, b)], 0.0);
    } else {
      m->trk_ao[b] = tv_from(&m->trk_atm_copy, 0.0);
      m->trk_oa[b] = tv_from(&m->trk[TP_OCEAN(m, b)], -m->atmflux[b]);
    }
  }
  for (int b = 0; b < 4; b++) { /* carbon + CarbonAdditions + ao_flux - oa_flux - CarbonSubtractions */
    tv_t c = tv_add(m->trk[TP_OCEAN(m, b)], m->trk_addn[b], TP);
    if (b < 2) {
      c = tv_add(c, m->trk_ao[b], TP);
      c = tv_sub(c, m->trk_oa[b]);
    } else { /* ao/oa of the deeper boxes: zero-valued fluxpools named after the box */
      c = tv_add(c, tv_from(&m->trk_atm_copy, 0.0), TP);
    }
    c.val = c.val - m->trk_subn[b];
    m->trk[TP_OCEAN(m, b)] = c;
    m->trk_addn[b] = tv_self(TP_OCEAN(m, b), 0.0);
    m->trk_subn[b] = 0.0;
  }
}

/* OceanComponent::stashCValues  src/ocean_component.cpp:653-763 */
static void ocean_stash(member_t *m, double t, const double c[]) {
  const double yearfraction = (t - m->ocean_ODEstartdate);
  m->timesteps++;
  const int in_partial_year = (t != (int)(t));
  double CO2_conc = c[SNBOX_ATMOS] * PGC_TO_PPMVCO2;
  box_compute_fluxes(m, HL, CO2_conc, yearfraction, 1);
  box_compute_fluxes(m, LL, CO2_conc, yearfraction, 1);
  box_compute_fluxes(m, IO, CO2_conc, yearfraction, 1);
  box_compute_fluxes(m, DO, CO2_conc, yearfraction, 1);
  double currentflux = m->atmflux[HL] + m->atmflux[LL];
  double solver_flux = c[SNBOX_OCEAN] - ocean_totalcpool(m);
  double adjustment = 0.0;
  if (currentflux) adjustment = (solver_flux - currentflux) / 2.0;
  m->atmflux[HL] = m->atmflux[HL] + adjustment;
  m->atmflux[LL] = m->atmflux[LL] + adjustment;
  box_separate(m, HL);
  box_separate(m, LL);
  double cflux_annualdiff = solver_flux / yearfraction - m->lastflux_annualized;
  if (cflux_annualdiff > 0.1) {
    double r = m->max_timestep * 0.5;
    m->max_timestep = (0.3 < r) ? r : 0.3; /* max(MIN, ts*FACTOR) */
    m->reduced_timestep_timeout = 20;
  } else if (!in_partial_year && m->reduced_timestep_timeout) {
    int r = m->reduced_timestep_timeout - 1;
    m->reduced_timestep_timeout = (0 < r) ? r : 0;
    if (!m->reduced_timestep_timeout) {
      double q = m->max_timestep / 0.5;
      m->max_timestep = (q < 1.0) ? q : 1.0; /* min(MAX, ts/FACTOR) */
      if (m->max_timestep < 1.0) m->reduced_timestep_timeout = 20;
    }
  }
  double lastflux = m->atmflux[LL] + m->atmflux[HL];
  m->annualflux_sumHL = m->annualflux_sumHL + m->atmflux[HL];
  m->annualflux_sumLL = m->annualflux_sumLL + m->atmflux[LL];
  m->annualflux_sum = m->annualflux_sum + lastflux;
  m->lastflux_annualized = lastflux / yearfraction;
  if (m->trk_on) trk_ocean_stash(m, yearfraction);
  box_update_state(m, HL);
  box_update_state(m, LL);
  box_update_state(m, IO);
  box_update_state(m, DO);
  if (m->trk_on)
    for (int b = 0; b < 4; b++) m->trk[TP_OCEAN(m, b)].val = m->carbon[b];
  m->ocean_ODEstartdate = t;
}

/* ------------------------------------------------------------------ */
/* simpleNbox                                                          */
/* ------------------------------------------------------------------ */
static double snb_npp(const member_t *m, int b) { /* runtime.cpp:622-635 */
  double npp = m->pa->npp_flux0[b];
  npp = npp * m->co2fert[b];
  npp = npp * m->npp_luc_adjust;
  return npp;
}
static double snb_rh_fda(const member_t *m, int b) { /* :653-665 */
  return (m->detritus_c[b] * 0.25) * m->tempfertd[b];
}
static double snb_rh_fsa(const member_t *m, int b) { /* :671-683 */
  return (m->soil_c[b] * 0.02) * m->tempferts[b];
}
static double snb_rh_ftpa_co2(const member_t *m, int b) { /* :689-701 */
  double tpfc = m->thawed_c[b] * (1 - m->pa->fpf_static[b]);
  double tpflux = tpfc * 0.02;
  return tpflux * m->tempferts[b] * (1.0 - m->pa->rh_ch4_frac[b]);
}
static double snb_rh_ftpa_ch4(const member_t *m, int b) { /* :707-711 */
  return snb_rh_ftpa_co2(m, b) / (1.0 - m->pa->rh_ch4_frac[b]) *
         m->pa->rh_ch4_frac[b];
}
static double snb_rh(const member_t *m, int b) { /* :717-721 */
  return snb_rh_fda(m, b) + snb_rh_fsa(m, b) + snb_rh_ftpa_co2(m, b);
}
/* compute_pf_thaw_refreeze  runtime.cpp:744-772 */
static void snb_pf(const member_t *m, int b, double rh_co2, double rh_ch4,
                   double *thaw, double *refreeze_tp, double *refreeze_soil) {
  double biome_c_thaw = m->permafrost_c[b] * m->f_new_thaw[b];
  double pf_refreeze_tp = 0.0, pf_refreeze_soil = 0.0;
  if (biome_c_thaw < 0) {
    const double pf_refreeze = -biome_c_thaw;
    biome_c_thaw = 0.0;
    const double thawed_remaining = m->thawed_c[b] - rh_co2 - rh_ch4;
    pf_refreeze_tp = (thawed_remaining < pf_refreeze) ? thawed_remaining
                                                      : pf_refreeze;
  }
  *thaw = biome_c_thaw; *refreeze_tp = pf_refreeze_tp;
  *refreeze_soil = pf_refreeze_soil;
}

static double sum_b(const double *x, int B) {
  double s = 0.0;
  for (int b = 0; b < B; b++) s = s + x[b];
  return s;
}

/* SimpleNbox::getCValues  runtime.cpp:247-258 */
static void snb_getCValues(member_t *m, double t, double c[]) {
  c[SNBOX_ATMOS] = m->atmos_c;
  c[SNBOX_VEG] = sum_b(m->veg_c, m->B);
  c[SNBOX_DET] = sum_b(m->detritus_c, m->B);
  c[SNBOX_SOIL] = sum_b(m->soil_c, m->B);
  c[SNBOX_PERMAFROST] = sum_b(m->permafrost_c, m->B);
  c[SNBOX_THAWEDP] = sum_b(m->thawed_c, m->B);
  c[SNBOX_OCEAN] = ocean_totalcpool(m);
  m->ocean_ODEstartdate = t;
  c[SNBOX_EARTH] = m->earth_c;
  m->ODEstartdate = t;
}

/* SimpleNbox::calcderivs  runtime.cpp:781-934 */
static int snb_calcderivs(member_t *m, double t, const double c[],
                          double dcdt[]) {
  m->nrhs_year++;
  const int omodel_err = ocean_calcderivs(m, t, c, dcdt);
  const double ao_exchange = dcdt[SNBOX_OCEAN];
  double ocean_uptake = 0.0, ocean_release = 0.0;
  if (ao_exchange >= 0.0) ocean_uptake = ao_exchange;
  else ocean_release = -ao_exchange;
  double npp_current = 0, npp_fav = 0, npp_fad = 0, npp_fas = 0;
  double rh_fda_current = 0, rh_fsa_current = 0, rh_ftpa_co2_current = 0,
         rh_ftpa_ch4_current = 0;
  const hxo_params *p = m->pa;
  for (int b = 0; b < m->B; b++) {
    double npp_biome = snb_npp(m, b);
    npp_current = npp_current + npp_biome;
    npp_fav = npp_fav + npp_biome * p->f_nppv[b];
    npp_fad = npp_fad + npp_biome * p->f_nppd[b];
    npp_fas = npp_fas + npp_biome * (1 - p->f_nppv[b] - p->f_nppd[b]);
    rh_fda_current = rh_fda_current + snb_rh_fda(m, b);
    rh_fsa_current = rh_fsa_current + snb_rh_fsa(m, b);
    rh_ftpa_co2_current = rh_ftpa_co2_current + snb_rh_ftpa_co2(m, b);
    rh_ftpa_ch4_current = rh_ftpa_ch4_current + snb_rh_ftpa_ch4(m, b);
  }
  double rh_current = rh_fda_current + rh_fsa_current + rh_ftpa_co2_current;
  double litter_flux = 0, litter_fvd = 0, litter_fvs = 0;
  for (int b = 0; b < m->B; b++) {
    double v = m->veg_c[b] * 0.035;
    litter_flux = litter_flux + v;
    litter_fvd = litter_fvd + v * p->f_litterd[b];
    litter_fvs = litter_fvs + v * (1 - p->f_litterd[b]);
  }
  double detsoil_flux = 0;
  for (int b = 0; b < m->B; b++)
    detsoil_flux = detsoil_flux + m->detritus_c[b] * 0.6;
  const double total = c[SNBOX_VEG] + c[SNBOX_DET] + c[SNBOX_SOIL];
  double luc_fva = m->cur_luc_e * c[SNBOX_VEG] / total;
  double luc_fda = m->cur_luc_e * c[SNBOX_DET] / total;
  double luc_fsa = m->cur_luc_e * c[SNBOX_SOIL] / total;
  double luc_fav = m->cur_luc_u;
  double ch4ox_current = 0.0;
  double pf_thaw_c = 0, pf_refreeze_tp = 0, pf_refreeze_soil = 0;
  if (!m->snbox_in_spinup) {
    for (int b = 0; b < m->B; b++) {
      double x, y, z;
      snb_pf(m, b, snb_rh_ftpa_co2(m, b), snb_rh_ftpa_ch4(m, b), &x, &y, &z);
      pf_thaw_c = pf_thaw_c + x;
      pf_refreeze_tp = pf_refreeze_tp + y;
      pf_refreeze_soil = pf_refreeze_soil + z;
    }
  }
  /* NBP constraint: NPP and RH adjusted equally  runtime.cpp:871-898 */
  {
    const int it = (int)round(t) - m->sc->start;
    if (!m->snbox_in_spinup && it >= 0 && it < m->ns && con_has(m->sc->nbp_con, it)) {
      const double nbp = npp_current - rh_current - m->cur_luc_e + m->cur_luc_u;
      const double diff = m->sc->nbp_con[it] - nbp;
      const double npp_current_old = npp_current;
      npp_current = npp_current + diff / 2.0;
      const double npp_ratio = npp_current / npp_current_old;
      npp_fav = npp_fav * npp_ratio;
      npp_fad = npp_fad * npp_ratio;
      npp_fas = npp_fas * npp_ratio;
      const double rh_current_old = rh_current;
      rh_current = rh_current - diff / 2.0;
      const double rh_ratio = rh_current / rh_current_old;
      rh_fda_current = rh_fda_current * rh_ratio;
      rh_fsa_current = rh_fsa_current * rh_ratio;
      rh_ftpa_co2_current = rh_ftpa_co2_current * rh_ratio;
    }
  }
  dcdt[SNBOX_ATMOS] = m->cur_ffi - m->cur_daccs + m->cur_luc_e - m->cur_luc_u +
                      ch4ox_current - ocean_uptake + ocean_release -
                      npp_current + rh_current;
  dcdt[SNBOX_VEG] = npp_fav - litter_flux - luc_fva + luc_fav;
  dcdt[SNBOX_DET] =
      npp_fad + litter_fvd - detsoil_flux - rh_fda_current - luc_fda;
  dcdt[SNBOX_SOIL] = npp_fas + litter_fvs + detsoil_flux - rh_fsa_current -
                     pf_refreeze_soil - luc_fsa;
  dcdt[SNBOX_PERMAFROST] = -pf_thaw_c + pf_refreeze_soil + pf_refreeze_tp;
  dcdt[SNBOX_THAWEDP] =
      pf_thaw_c - pf_refreeze_tp - rh_ftpa_ch4_current - rh_ftpa_co2_current;
  dcdt[SNBOX_OCEAN] = ocean_uptake - ocean_release;
  dcdt[SNBOX_EARTH] = -m->cur_ffi + m->cur_daccs;
  return omodel_err;
}

/* Tland_record.get(i): exact if recorded, flat extrapolation before the first
 * record (tseries.hpp:310-334, h_interpolator.cpp:103-122) */
static double tland_record_get(const member_t *m, int year) {
  int idx = year - m->sc->start;
  if (m->Tland_first < 0) return 0.0;
  if (idx < m->Tland_first) idx = m->Tland_first;
  return m->Tland_record[idx];
}

/* SimpleNbox::slowparameval  runtime.cpp:945-1072 */
static void snb_slowparameval(member_t *m, double t) {
  const hxo_scenario *s = m->sc;
  const hxo_params *p = m->pa;
  m->ocean_in_spinup = m->core_in_spinup; /* ocean slowparameval :630-633 */
  if (m->snbox_in_spinup) {
    m->cur_luc_e = m->cur_luc_u = m->cur_ffi = m->cur_daccs = 0.0;
  } else {
    int iy = (int)t - s->start;
    m->cur_luc_e = s->luc_e[iy];
    m->cur_luc_u = s->luc_u[iy];
    m->cur_ffi = s->ffi[iy];
    m->cur_daccs = s->daccs[iy];
  }
  m->npp_luc_adjust =
      (m->end_of_spinup_vegc - m->cum_luc_va) / m->end_of_spinup_vegc;
  for (int b = 0; b < m->B; b++) {
    if (m->snbox_in_spinup) m->co2fert[b] = 1.0;
    else /* calc_co2fert :614-616 */
      m->co2fert[b] =
          1 + p->beta[b] * log((m->atmos_c * PGC_TO_PPMVCO2) / p->C0);
  }
  const double Tland = m->tas_land;
  const int have_last = (t > s->start);
  for (int b = 0; b < m->B; b++) {
    if (m->snbox_in_spinup) {
      m->tempfertd[b] = 1.0; m->tempferts[b] = 1.0;
      m->f_frozen[b] = 1.0; m->f_new_thaw[b] = 0.0;
    } else {
      double wf = p->warmingfactor[b];
      const double Tland_biome = Tland * wf;
      m->tempfertd[b] = pow(p->q10_rh[b], (Tland_biome / 10.0));
      m->f_new_thaw[b] = 0.0;
      if (m->permafrost_c[b]) {
        double f_frozen_current = 1.0;
        if (Tland_biome > 0) {
          /* boost lognormal cdf = erfc(-(ln x - mu)/(sigma*sqrt2))/2 */
          double diff = (log(Tland_biome) - p->pf_mu[b]) /
                        (p->pf_sigma[b] * 1.4142135623730950488016887242096981);
          f_frozen_current = 1 - erfc(-diff) / 2;
        }
        m->f_new_thaw[b] = m->f_frozen[b] - f_frozen_current;
        m->f_frozen[b] = f_frozen_current;
      }
      double Tland_rm = 0.0;
      if (t > s->start + 0) {
        for (int i = (int)(t - 0 - Q10_TEMPN); i < t - 0; i++)
          Tland_rm += tland_record_get(m, i) * wf;
        Tland_rm /= Q10_TEMPN;
      }
      m->tempferts[b] = pow(p->q10_rh[b], (Tland_rm / 10.0));
      double tempferts_last = have_last ? m->tempferts_recorded[b] : 0.0;
      if (m->tempferts[b] < tempferts_last) m->tempferts[b] = tempferts_last;
    }
  }
}

/* The origins side of SimpleNbox::stashCValues (simpleNbox-runtime.cpp:289-540): the same
 * sequence of fluxpool operations, pool by pool and biome by biome; called with the pools as
 * they are when the stash begins (the ocean's part has already run, like omodel->stashCValues). */
static void trk_land_stash(member_t *m, const double c[], double yf, double npp_total,
                           double npp_rh_total, double rh_adj, double newveg, double newdet,
                           double newsoil, double newpermafrost, double newthawedpf,
                           double newatmos) {
  const hxo_params *p = m->pa;
  const int TP = m->TP;
  tv_t *T = m->trk;
  T[TP_ATM].val = m->atmos_c; T[TP_EARTH].val = m->earth_c;
  for (int b = 0; b < m->B; b++) {
    T[TP_LAND(b, 0)].val = m->veg_c[b]; T[TP_LAND(b, 1)].val = m->detritus_c[b];
    T[TP_LAND(b, 2)].val = m->soil_c[b]; T[TP_LAND(b, 3)].val = m->permafrost_c[b];
    T[TP_LAND(b, 4)].val = m->thawed_c[b];
  }
  const tv_t ffi_flux = tv_from(&T[TP_EARTH], m->cur_ffi);  /* :295-296 */
  const tv_t ccs_flux = tv_from(&T[TP_ATM], m->cur_daccs);
  const tv_t oa_flux = tv_add(m->trk_oa[LL], m->trk_oa[HL], TP);  /* get_oaflux: LL + HL */
  const tv_t ao_flux = tv_add(m->trk_ao[LL], m->trk_ao[HL], TP);
  const double permafrost_total = sum_b(m->permafrost_c, m->B);
  const double total = c[SNBOX_VEG] + c[SNBOX_DET] + c[SNBOX_SOIL];
  for (int b = 0; b < m->B; b++) {
    tv_t *veg = &T[TP_LAND(b, 0)], *det = &T[TP_LAND(b, 1)], *soil = &T[TP_LAND(b, 2)],
         *pf = &T[TP_LAND(b, 3)], *tp = &T[TP_LAND(b, 4)], *atm = &T[TP_ATM];
    const double wt = (snb_npp(m, b) + snb_rh(m, b)) / npp_rh_total;
    const double wt_pf = permafrost_total > 0 ? m->permafrost_c[b] / permafrost_total : 0;
    const double veg_frac = veg->val / total, det_frac = det->val / total,
                 soil_frac = soil->val / total;
    /* every flux of this biome is drawn before any pool moves (:414-452) */
    const tv_t luc_fva = tv_mul(tv_from(veg, m->cur_luc_e * veg_frac), yf);
    const tv_t luc_fda = tv_mul(tv_from(det, m->cur_luc_e * det_frac), yf);
    const tv_t luc_fsa = tv_mul(tv_from(soil, m->cur_luc_e * soil_frac), yf);
    const tv_t luc_fav = tv_mul(tv_from(atm, m->cur_luc_u), yf);
    const double npp_biome = npp_total * wt;
    const tv_t npp_fav = tv_mul(tv_from(atm, npp_biome * p->f_nppv[b]), yf);
    const tv_t npp_fad = tv_mul(tv_from(atm, npp_biome * p->f_nppd[b]), yf);
    const tv_t npp_fas = tv_mul(tv_from(atm, npp_biome * (1 - p->f_nppv[b] - p->f_nppd[b])), yf);
    const double rh_fda_adj = snb_rh_fda(m, b) * rh_adj, rh_fsa_adj = snb_rh_fsa(m, b) * rh_adj,
                 rh_co2_adj = snb_rh_ftpa_co2(m, b) * rh_adj, rh_ch4_adj = snb_rh_ftpa_ch4(m, b) * rh_adj;
    const tv_t rh_fda_flux = tv_mul(tv_from(det, rh_fda_adj), yf);
    const tv_t rh_fsa_flux = tv_mul(tv_from(soil, rh_fsa_adj), yf);
    const tv_t rh_fpa_co2 = tv_mul(tv_from(tp, rh_co2_adj), yf);
    const tv_t rh_fpa_ch4 = tv_mul(tv_from(tp, rh_ch4_adj), yf);
    /* luc fluxes :455-460 (the detritus line of the reference has no effect: an expression
     * statement without assignment) */
    *atm = tv_add(tv_add(tv_sub(tv_add(*atm, luc_fva, TP), luc_fav), luc_fda, TP), luc_fsa, TP);
    *veg = tv_sub(tv_add(*veg, luc_fav, TP), luc_fva);
    *soil = tv_sub(*soil, luc_fsa);
    /* npp fluxes :463-467 */
    *veg = tv_add(*veg, npp_fav, TP);
    *det = tv_add(*det, npp_fad, TP);
    *soil = tv_add(*soil, npp_fas, TP);
    *atm = tv_sub(tv_sub(tv_sub(*atm, npp_fav), npp_fad), npp_fas);
    /* rh fluxes :470-476 */
    *atm = tv_add(tv_add(tv_add(*atm, rh_fda_flux, TP), rh_fsa_flux, TP), rh_fpa_co2, TP);
    *det = tv_sub(*det, rh_fda_flux);
    *soil = tv_sub(*soil, rh_fsa_flux);
    *tp = tv_sub(tv_sub(*tp, rh_fpa_co2), rh_fpa_ch4);
    if (!m->snbox_in_spinup) { /* permafrost thaw and refreeze :484-503 */
      /* compute_pf_thaw_refreeze :744-772 on the pools as they are at this point (the thawed
       * pool has already lost this stash's respiration) */
      double x = pf->val * m->f_new_thaw[b], y = 0.0, z = 0.0;
      if (x < 0) {
        const double pf_refreeze = -x;
        x = 0.0;
        const double thawed_remaining = tp->val - rh_co2_adj - rh_ch4_adj;
        y = (thawed_remaining < pf_refreeze) ? thawed_remaining : pf_refreeze;
      }
      const tv_t pf_thaw = tv_mul(tv_from(pf, x), yf);
      const tv_t pf_refreeze_tp = tv_mul(tv_from(tp, y), yf);
      const tv_t pf_refreeze_soil = tv_mul(tv_from(soil, z), yf);
      *pf = tv_add(tv_add(tv_sub(*pf, pf_thaw), pf_refreeze_tp, TP), pf_refreeze_soil, TP);
      *tp = tv_sub(tv_add(*tp, pf_thaw, TP), pf_refreeze_tp);
      *soil = tv_sub(*soil, pf_refreeze_soil);
    }
    /* litter :506-511, detritus -> soil :514-521 */
    const tv_t litter = tv_mul(*veg, 0.035 * yf);
    const tv_t litter_fvd = tv_mul(litter, p->f_litterd[b]);
    const tv_t litter_fvs = tv_mul(litter, 1 - p->f_litterd[b]);
    *det = tv_add(*det, litter_fvd, TP);
    *soil = tv_add(*soil, litter_fvs, TP);
    *veg = tv_sub(*veg, litter);
    const tv_t detsoil = tv_mul(*det, 0.6 * yf);
    *soil = tv_add(*soil, detsoil, TP);
    *det = tv_sub(*det, detsoil);
    /* adjust_pool_to_val(solver value, false): the value only :524-530 */
    veg->val = newveg * wt; det->val = newdet * wt; soil->val = newsoil * wt;
    pf->val = newpermafrost * wt_pf; tp->val = newthawedpf * wt_pf;
  }
  /* :534-541 */
  T[TP_EARTH] = tv_add(tv_sub(T[TP_EARTH], ffi_flux), ccs_flux, TP);
  T[TP_ATM] = tv_sub(tv_add(T[TP_ATM], ffi_flux, TP), ccs_flux);
  T[TP_ATM] = tv_sub(tv_add(T[TP_ATM], oa_flux, TP), ao_flux);
  T[TP_EARTH].val = c[SNBOX_EARTH];
  T[TP_ATM].val = newatmos;
}

/* SimpleNbox::stashCValues  runtime.cpp:270-609 (no tracking, no constraints) */
static void snb_stash(member_t *m, double t, const double c[]) {
  const hxo_params *p = m->pa;
  const double yf = (t - m->ODEstartdate);
  if (!(yf >= 0 && yf <= 1)) m->err |= HXO_ERR_STEPFAIL;
  ocean_stash(m, t, c);
  double npp_total = 0, rh_total = 0;
  for (int b = 0; b < m->B; b++) npp_total = npp_total + snb_npp(m, b);
  for (int b = 0; b < m->B; b++) rh_total = rh_total + snb_rh(m, b);
  const double permafrost_total = sum_b(m->permafrost_c, m->B);
  double alf = npp_total - rh_total - m->cur_luc_e + m->cur_luc_u;
  double npp_rh_total = npp_total + rh_total;
  double newveg = c[SNBOX_VEG], newdet = c[SNBOX_DET], newsoil = c[SNBOX_SOIL],
         newpermafrost = c[SNBOX_PERMAFROST];
  double solver_tpf = c[SNBOX_THAWEDP];
  if (fabs(solver_tpf) < 1e-10) solver_tpf = 0.0;
  if (c[0] < 0 || newveg < 0 || newdet < 0 || newsoil < 0 ||
      newpermafrost < 0 || solver_tpf < 0)
    m->err |= HXO_ERR_NEGPOOL;
  /* NBP constraint  runtime.cpp:343-383 */
  double rh_nbp_constraint_adjust = 1.0;
  {
    const int it = (int)round(t) - m->sc->start;
    if (!m->core_in_spinup && it >= 0 && it < m->ns && con_has(m->sc->nbp_con, it)) {
      const double diff = m->sc->nbp_con[it] - alf;
      npp_total = npp_total + diff / 2.0;
      rh_nbp_constraint_adjust = (rh_total - diff / 2.0) / rh_total;
      rh_total = rh_total - diff / 2.0;
      const double pool_diff = diff * yf;
      const double total_land =
          c[SNBOX_DET] + c[SNBOX_VEG] + c[SNBOX_SOIL] + c[SNBOX_THAWEDP];
      newdet = newdet + pool_diff * c[SNBOX_DET] / total_land;
      newveg = newveg + pool_diff * c[SNBOX_VEG] / total_land;
      newsoil = newsoil + pool_diff * c[SNBOX_SOIL] / total_land;
      solver_tpf = solver_tpf + pool_diff * c[SNBOX_THAWEDP] / total_land;
      m->carbon[DO] = (-pool_diff) + m->carbon[DO]; /* M_DUMP_TO_DEEP_OCEAN */
      alf = npp_total - rh_total - m->cur_luc_e + m->cur_luc_u;
    }
  }
  m->nbp = alf;
  const double total = c[SNBOX_VEG] + c[SNBOX_DET] + c[SNBOX_SOIL];
  m->cum_luc_va = m->cum_luc_va +
                  ((m->cur_luc_e - m->cur_luc_u) * c[SNBOX_VEG] / total);
  if (m->trk_on)
    trk_land_stash(m, c, yf, npp_total, npp_rh_total, rh_nbp_constraint_adjust, newveg, newdet,
                   newsoil, newpermafrost, solver_tpf, c[SNBOX_ATMOS]);
  m->final_npp = m->final_rh = m->final_rh_det = m->final_rh_soil = 0.0;
  for (int b = 0; b < m->B; b++) {
    const double wt = (snb_npp(m, b) + snb_rh(m, b)) / npp_rh_total;
    /* final_npp / final_rh* :420-440, summed over biomes as D_NPP, D_RH ... report them */
    m->final_npp = m->final_npp + npp_total * wt;
    m->final_npp_b[b] = npp_total * wt;
    {
      const double a = snb_rh_fda(m, b) * rh_nbp_constraint_adjust;
      const double bb = snb_rh_fsa(m, b) * rh_nbp_constraint_adjust;
      const double cc = snb_rh_ftpa_co2(m, b) * rh_nbp_constraint_adjust;
      const double dd = snb_rh_ftpa_ch4(m, b) * rh_nbp_constraint_adjust;
      m->final_rh = m->final_rh + (a + bb + cc + dd);
      m->final_rh_b[b] = a + bb + cc + dd;
      m->final_rh_det = m->final_rh_det + a;
      m->final_rh_soil = m->final_rh_soil + bb;
    }
    const double wt_pf =
        permafrost_total > 0 ? m->permafrost_c[b] / permafrost_total : 0;
    double rh_ftpa_ch4_adj = snb_rh_ftpa_ch4(m, b) * rh_nbp_constraint_adjust;
    double rh_fpa_ch4_flux = rh_ftpa_ch4_adj * yf;
    m->cumulative_pf_ch4 += rh_fpa_ch4_flux;
    m->veg_c[b] = newveg * wt;
    m->detritus_c[b] = newdet * wt;
    m->soil_c[b] = newsoil * wt;
    m->permafrost_c[b] = newpermafrost * wt_pf;
    m->thawed_c[b] = solver_tpf * wt_pf;
  }
  m->earth_c = c[SNBOX_EARTH];
  m->atmos_c = c[SNBOX_ATMOS];
  double sum = 0.0;
  for (int i = 0; i < 8; i++) sum += c[i];
  sum += m->cumulative_pf_ch4;
  const double diff = fabs(sum - m->masstot);
  /* !(<=) rather than (>): a NaN state raises the flag as well (the reference would not notice) */
  if (m->masstot > 0.0 && !(diff <= MB_EPSILON)) m->err |= HXO_ERR_MASS;
  m->masstot = sum;
  const int it = (int)t - m->sc->start;
  if (m->core_in_spinup ||
      (t == floor(t) && it >= 0 && it < m->ns && con_has(m->sc->co2_con, it))) {
    double match = (m->core_in_spinup ? p->C0 : m->sc->co2_con[it]) / PGC_TO_PPMVCO2;
    double residual = m->atmos_c - match;
    /* deepOceanCarbonDump  ocean_component.cpp:146-154 */
    m->carbon[DO] = residual + m->carbon[DO];
    m->atmos_c = m->atmos_c - residual;
    m->Ca_residual = residual;
  } else {
    m->Ca_residual = 0.0;
  }
  m->ODEstartdate = t;
}

/* SimpleNbox::record_state  simpleNbox.cpp:789-840 (what later code reads) */
static void snb_record_state(member_t *m) {
  double s = 0.0;
  for (int b = 0; b < m->B; b++) {
    double v = m->snbox_in_spinup ? 0.0 : snb_rh_ftpa_ch4(m, b);
    s = s + v;
    m->tempferts_recorded[b] = m->tempferts[b];
  }
  m->RH_ch4_sum = s;
}

/* ------------------------------------------------------------------ */
/* odeint: controlled_runge_kutta<runge_kutta_dopri5>, integrate_adaptive */
/* ------------------------------------------------------------------ */
#define NC 8
typedef struct {
  double dxdt[NC];
  int first_call;
} stepper_t;

/* runge_kutta_dopri5::do_step_impl (FSAL, with error)  */
static int dopri5_step(member_t *m, const double *in, const double *dxdt_in,
                       double t, double *out, double *dxdt_out, double dt,
                       double *xerr) {
  const double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
  const double b21 = 1.0 / 5;
  const double b31 = 3.0 / 40, b32 = 9.0 / 40;
  const double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
  const double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187,
               b53 = 64448.0 / 6561, b54 = -212.0 / 729;
  const double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247,
               b64 = 49.0 / 176, b65 = -5103.0 / 18656;
  const double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192,
               c5 = -2187.0 / 6784, c6 = 11.0 / 84;
  const double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695,
               dc4 = c4 - 393.0 / 640, dc5 = c5 - (-92097.0 / 339200),
               dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
  double xt[NC], k2[NC], k3[NC], k4[NC], k5[NC], k6[NC];
  int st;
  for (int i = 0; i < NC; i++) xt[i] = 1.0 * in[i] + dt * b21 * dxdt_in[i];
  if ((st = snb_calcderivs(m, t + dt * a2, xt, k2))) return st;
  for (int i = 0; i < NC; i++)
    xt[i] = 1.0 * in[i] + dt * b31 * dxdt_in[i] + dt * b32 * k2[i];
  if ((st = snb_calcderivs(m, t + dt * a3, xt, k3))) return st;
  for (int i = 0; i < NC; i++)
    xt[i] = 1.0 * in[i] + dt * b41 * dxdt_in[i] + dt * b42 * k2[i] +
            dt * b43 * k3[i];
  if ((st = snb_calcderivs(m, t + dt * a4, xt, k4))) return st;
  for (int i = 0; i < NC; i++)
    xt[i] = 1.0 * in[i] + dt * b51 * dxdt_in[i] + dt * b52 * k2[i] +
            dt * b53 * k3[i] + dt * b54 * k4[i];
  if ((st = snb_calcderivs(m, t + dt * a5, xt, k5))) return st;
  for (int i = 0; i < NC; i++)
    xt[i] = 1.0 * in[i] + dt * b61 * dxdt_in[i] + dt * b62 * k2[i] +
            dt * b63 * k3[i] + dt * b64 * k4[i] + dt * b65 * k5[i];
  if ((st = snb_calcderivs(m, t + dt, xt, k6))) return st;
  for (int i = 0; i < NC; i++)
    out[i] = 1.0 * in[i] + dt * c1 * dxdt_in[i] + dt * c3 * k3[i] +
             dt * c4 * k4[i] + dt * c5 * k5[i] + dt * c6 * k6[i];
  if ((st = snb_calcderivs(m, t + dt, out, dxdt_out))) return st;
  for (int i = 0; i < NC; i++)
    xerr[i] = dt * dc1 * dxdt_in[i] + dt * dc3 * k3[i] + dt * dc4 * k4[i] +
              dt * dc5 * k5[i] + dt * dc6 * k6[i] + dt * dc7 * dxdt_out[i];
  return 0;
}

/* controlled_runge_kutta<..., fsal>::try_step: 1 = success, 0 = fail,
 * CARBON_CYCLE_RETRY if the RHS threw (carbon-cycle-solver.cpp:175-186) */
static int try_step(member_t *m, stepper_t *st, double *x, double *t,
                    double *dt) {
  const double eps_abs = m->sc->eps_abs, eps_rel = m->sc->eps_rel;
  int rc;
  if (st->first_call) {
    if ((rc = snb_calcderivs(m, *t, x, st->dxdt))) return rc;
    st->first_call = 0;
  }
  double xnew[NC], dxdtnew[NC], xerr[NC];
  if ((rc = dopri5_step(m, x, st->dxdt, *t, xnew, dxdtnew, *dt, xerr)))
    return rc;
  /* default_error_checker::error: max_i |err|/(eps_abs+eps_rel*(|x|+dt|dxdt|)) */
  double max_rel_err = 0.0;
  for (int i = 0; i < NC; i++) {
    double e = fabs(xerr[i]) /
               (eps_abs + eps_rel * (1.0 * fabs(x[i]) +
                                     (1.0 * *dt) * fabs(st->dxdt[i])));
    if (e > max_rel_err) max_rel_err = e;
  }
  if (max_rel_err > 1.0) {
    /* default_step_adjuster::decrease_step, error_order = 4 */
    double f = 9.0 / 10.0 * pow(max_rel_err, -1.0 / (4 - 1));
    *dt *= (f > 1.0 / 5.0) ? f : 1.0 / 5.0;
    return 0;
  }
  *t += *dt;
  /* increase_step, stepper_order = 5 */
  if (max_rel_err < 0.5) {
    double error = max_rel_err;
    double lo = pow(5.0, -5.0);
    error = (lo > error) ? lo : error;
    *dt *= 9.0 / 10.0 * pow(error, -1.0 / 5);
  }
  memcpy(x, xnew, sizeof xnew);
  memcpy(st->dxdt, dxdtnew, sizeof dxdtnew);
  m->nsteps_year++;
  return 1;
}

/* integrate_adaptive(controlled stepper): returns 0 or CARBON_CYCLE_RETRY.
 * The observer writes the solver's t (carbon-cycle-solver.cpp:196-200). */
static int integrate_adaptive(member_t *m, double *x, double start_time,
                              double end_time, double dt) {
  stepper_t st;
  st.first_call = 1;
  while ((end_time - start_time) > DBL_EPSILON) { /* less_with_sign, dt>0 */
    m->t = start_time; /* observer */
    if (((start_time + dt) - end_time) > DBL_EPSILON) dt = end_time - start_time;
    int res, fails = 0;
    do {
      res = try_step(m, &st, x, &start_time, &dt);
      if (res == CARBON_CYCLE_RETRY) return res;
      if (++fails > 500) { m->err |= HXO_ERR_STEPFAIL; return 0; }
      /* odeint has no limit on accepted steps; like the kernels (HX_MAX_STEPS_PER_YEAR) the
       * oracle gives up on a member whose step size has collapsed */
      if (m->nsteps_year > 20000) { m->err |= HXO_ERR_STEPFAIL; return 0; }
    } while (res == 0);
  }
  m->t = start_time;
  return 0;
}

/* CarbonCycleSolver::run  src/carbon-cycle-solver.cpp:222-303 */
static void solver_run(member_t *m, const double tnew) {
  double c[NC];
  snb_getCValues(m, m->t, c);
  snb_slowparameval(m, m->t);
  int retry = 0;
  while (m->t < tnew && retry < MAX_RETRIES) {
    double t_start = m->t;
    double t_target = tnew;
    while (m->t < t_target && retry < MAX_RETRIES) {
      int stat = integrate_adaptive(m, c, t_start, t_target, m->dt);
      if (m->err) { snb_record_state(m); return; } /* an exception in the reference: run over */
      if (stat == CARBON_CYCLE_RETRY) {
        ++retry;
        t_target = t_start + (t_target - t_start) / 2.0;
        m->t = t_start;
        m->dt = t_target - m->t;
        snb_getCValues(m, m->t, c);
      }
    }
    if (retry < MAX_RETRIES) {
      retry = 0;
      snb_stash(m, m->t, c);
      if (m->err) { snb_record_state(m); return; }
    }
  }
  if (m->t != tnew) m->err |= HXO_ERR_RETRIES;
  snb_record_state(m);
}

/* ------------------------------------------------------------------ */
/* gases + forcing                                                     */
/* ------------------------------------------------------------------ */
typedef struct { const char *name; double v; } rf_item;
static int rf_cmp(const void *a, const void *b) {
  return strcmp(((const rf_item *)a)->name, ((const rf_item *)b)->name);
}

static void year_gases(member_t *m, int iy) {
  const hxo_scenario *s = m->sc;
  /* OHComponent::run  src/oh_component.cpp:137-178 */
  const double previous_ch4 = m->ch4_prev;
  double toh = 0.0;
  if (previous_ch4 != s->M0) {
    const double a = s->CCH4 * ((1.0 * log(previous_ch4)) - log(s->M0));
    const double b = s->CNOX * ((1.0 * s->nox_oh[iy]) - s->nox_oh[0]);
    const double cc = s->CCO * ((1.0 * +s->co_oh[iy]) - s->co_oh[0]);
    const double d = s->CNMVOC * ((1.0 * +s->nmvoc_oh[iy]) - s->nmvoc_oh[0]);
    toh = a + b + cc + d;
  }
  m->tau_oh = s->TOH0 * exp(-toh);
  /* CH4Component::run  src/ch4_component.cpp:152-199 */
  {
    const double current_ch4em = s->ch4_em[iy];
    const double current_toh = m->tau_oh;
    const double rh_ch4 = m->RH_ch4_sum * (1000.0 * 16.04 / 12.01);
    const double ch4n = s->ch4n[iy];
    const double emisTocon = (current_ch4em + rh_ch4 + ch4n) / s->UC_CH4;
    const double soil_sink = previous_ch4 / s->Tsoil;
    const double strat_sink = previous_ch4 / s->Tstrat;
    const double oh_sink = previous_ch4 / current_toh;
    const double dCH4 = emisTocon - soil_sink - strat_sink - oh_sink;
    m->ch4 = previous_ch4 + dCH4;
    if (con_has(s->ch4_con, iy)) m->ch4 = s->ch4_con[iy]; /* :156-157 */
  }
  /* OzoneComponent::run  src/o3_component.cpp:126-146 */
  m->o3 = (5 * log(m->ch4)) + (0.125 * s->nox_o3[iy]) +
          (0.0011 * s->co_o3[iy]) + (0.0033 * s->nmvoc_o3[iy]);
  /* N2OComponent::run  src/n2o_component.cpp:152-191 */
  {
    double previous_n2o = m->n2o_prev;
    double tau = s->TN2O0 * (pow(previous_n2o / m->N0_eff, -0.05));
    const double current_n2oem = s->n2o_em[iy] + s->n2o_nat[iy];
    const double dN2O = current_n2oem / s->UC_N2O - previous_n2o / tau;
    m->n2o = previous_n2o + dN2O;
    if (con_has(s->n2o_con, iy)) m->n2o = s->n2o_con[iy]; /* n2o_component.cpp:157-158 */
  }
  /* HalocarbonComponent::run  src/halocarbon_component.cpp:181-229 */
  for (int h = 0; h < s->nhalo; h++) {
    const hxo_halo *H = &s->halo[h];
    double Ha = m->halo_conc[h];
    const double timestep = 1.0;
    const double alpha = 1 / H->tau;
    double emissMol = H->em[iy] / H->molarMass * timestep;
    double concDeltaEmiss = emissMol / (0.1 * 1.8);
    double expfac = exp(-alpha);
    Ha = Ha * expfac + concDeltaEmiss * H->tau * (1.0 - expfac);
    if (con_has(H->con, iy)) Ha = H->con[iy]; /* halocarbon_component.cpp:189-191 */
    m->halo_conc[h] = Ha;
    double rf_unadjusted = H->rho * Ha;
    m->halo_rf[h] = rf_unadjusted + H->delta * rf_unadjusted;
  }
  m->ch4_prev = m->ch4;
  m->n2o_prev = m->n2o;
}

/* ForcingComponent::run  src/forcing_component.cpp:300-532 */
static void year_forcing(member_t *m, int year, double CO2_conc) {
  const hxo_scenario *s = m->sc;
  const hxo_params *p = m->pa;
  int iy = year - s->start;
  if (year < s->baseyear) {
    m->rf_tot = m->rf_co2 = 0.0; m->rf_ch4 = m->rf_n2o = 0;
    memset(m->rf_item_v, 0, sizeof m->rf_item_v);
    return;
  }
  const double a1 = -2.4785e-7, b1 = 7.5906e-4, c1 = -2.1492e-3, d1 = 5.2488;
  const double a2 = -3.4197e-4, b2 = 2.5455e-4, c2 = -2.4357e-4, d2 = 0.12173;
  const double a3 = -8.9603e-5, b3 = -1.2462e-4, d3 = 0.045194;
  const double aci_beta = 2.279759, s_BCOC = 111.05064063;
  const double s_SO2 = (260.34644166 * 1000) * (32.065 / 64.066);
  static char names[HXO_NHALO][32];
  rf_item f[48];
  int nf = 0;
  double C0 = p->C0, M0 = m->M0_eff, N0 = m->N0_eff, Ma = m->ch4, Na = m->n2o;
  double C_alpha_max = C0 - (b1 / (2 * a1));
  double n2o_alpha = c1 * sqrt(Na);
  double alpha_prime;
  if (CO2_conc > C_alpha_max) alpha_prime = d1 - (pow(b1, 2) / (4 * a1));
  else if (C0 < CO2_conc && CO2_conc < C_alpha_max)
    alpha_prime = d1 + a1 * pow((CO2_conc - C0), 2) + b1 * (CO2_conc - C0);
  else alpha_prime = d1;
  double sarf_co2 = (alpha_prime + n2o_alpha) * log(CO2_conc / C0);
  double fco2 = (sarf_co2 * s->delta_co2) + sarf_co2;
  const int major = !(s->off_ch4 || s->off_n2o);  /* :315-317 checkCapability of all three */
  if (!major) fco2 = 0.0;
  if (major) { f[nf].name = "RF_CO2"; f[nf++].v = fco2; }
  double sarf_n2o = (a2 * sqrt(CO2_conc) + b2 * sqrt(Na) + c2 * sqrt(Ma) + d2) *
                    (sqrt(Na) - sqrt(N0));
  double fn2o = (s->delta_n2o * sarf_n2o) + sarf_n2o;
  if (!major) fn2o = 0.0;
  if (major) { f[nf].name = "RF_N2O"; f[nf++].v = fn2o; }
  double sarf_ch4 = (a3 * sqrt(Ma) + b3 * sqrt(Na) + d3) * (sqrt(Ma) - sqrt(M0));
  double fch4 = (s->delta_ch4 * sarf_ch4) + sarf_ch4;
  if (!major) fch4 = 0.0;
  if (major) { f[nf].name = "RF_CH4"; f[nf++].v = fch4; }
  const double Ma_base = 1831, stratH2O_base = 0.0485;
  if (major) {
  f[nf].name = "RF_H2O_strat";
  f[nf++].v = stratH2O_base * ((Ma - M0) / (Ma_base - M0));
  }
  if (!s->off_ozone) { f[nf].name = "RF_O3_trop"; f[nf++].v = 0.042 * m->o3; }  /* :392 */
  for (int h = 0; h < s->nhalo; h++) {
    if (s->halo[h].disabled) continue;  /* :413-419 checkCapability */
    snprintf(names[h], sizeof names[h], "RF_%s", s->halo[h].name);
    f[nf].name = names[h]; f[nf++].v = m->halo_rf[h];
  }
  double E_BC = s->bc[iy], E_OC = s->oc[iy], E_NH3 = s->nh3[iy],
         E_SO2 = s->so2[iy];
  double alpha = p->aero_scalar;
  if (!(s->off_bc || s->off_oc || s->off_so2 || s->off_nh3)) {  /* :422-425: all four or none */
  f[nf].name = "RF_BC"; f[nf++].v = alpha * s->rho_bc * E_BC;
  f[nf].name = "RF_OC"; f[nf++].v = alpha * s->rho_oc * E_OC;
  f[nf].name = "RF_SO2"; f[nf++].v = alpha * s->rho_so2 * E_SO2;
  f[nf].name = "RF_NH3"; f[nf++].v = alpha * s->rho_nh3 * E_NH3;
  f[nf].name = "RF_aci";
  f[nf++].v = alpha * (-1 * aci_beta *
                       log(1 + (E_SO2 / s_SO2) + ((E_BC + E_OC) / s_BCOC)));
  }
  f[nf].name = "RF_albedo"; f[nf++].v = s->albedo[iy];
  if (!s->off_so2) { f[nf].name = "RF_vol"; f[nf++].v = p->vol_scalar * s->sv[iy]; }  /* :478 */
  f[nf].name = "RF_misc"; f[nf++].v = s->rf_misc ? s->rf_misc[iy] : 0.0;
  /* Ftot = sum over std::map<string,unitval> in key order (:489-492) */
  qsort(f, (size_t)nf, sizeof f[0], rf_cmp);
  double Ftot = 0.0;
  for (int i = 0; i < nf; i++) Ftot = Ftot + f[i].v;
  if (con_has(s->ftot_con, iy)) Ftot = s->ftot_con[iy]; /* :498-505 */
  if (year == s->baseyear) {
    m->have_base = 1; m->base_tot = Ftot; m->base_co2 = fco2;
    m->base_ch4 = fch4; m->base_n2o = fn2o;
  }
  {
    static const char *const want[] = {"RF_H2O_strat", "RF_O3_trop", "RF_BC", "RF_OC", "RF_SO2",
                                       "RF_NH3", "RF_aci", "RF_vol", "RF_albedo", "RF_misc"};
    double v[16] = {0};
    for (int i = 0; i < nf; i++) {
      int hit = -1;
      for (int k = 0; k < 10; k++) if (!strcmp(f[i].name, want[k])) hit = k;
      if (hit >= 0) v[hit] = f[i].v;
    }
    for (int h = 0; h < s->nhalo; h++) if (!s->halo[h].disabled) v[10] += m->halo_rf[h];
    if (year == s->baseyear) memcpy(m->rf_base_v, v, sizeof v);
    for (int k = 0; k < 11; k++) m->rf_item_v[k] = v[k] - m->rf_base_v[k];
  }
  m->rf_tot = Ftot - m->base_tot;
  m->rf_co2 = fco2 - m->base_co2;
  m->rf_ch4 = fch4 - m->base_ch4;
  m->rf_n2o = fn2o - m->base_n2o;
}

/* ------------------------------------------------------------------ */
/* driver                                                              */
/* ------------------------------------------------------------------ */
static void record_outputs_stride(member_t *m, int iy, double *out, int ns) {
#define O(V) out[(V) * ns + iy]
  O(HXO_CO2) = m->atmos_c * PGC_TO_PPMVCO2;
  O(HXO_ATMOS_C) = m->atmos_c;
  O(HXO_OCEAN_C) = m->carbon[DO] + m->carbon[IO] + m->carbon[LL] + m->carbon[HL];
  O(HXO_HL_PH) = m->chem[HL].pH;
  O(HXO_LL_PH) = m->chem[LL].pH;
  O(HXO_PCO2_HL) = m->chem[HL].PCO2o;
  O(HXO_PCO2_LL) = m->chem[LL].PCO2o;
  O(HXO_PERMAFROST_C) = sum_b(m->permafrost_c, m->B);
  O(HXO_VEG_C) = sum_b(m->veg_c, m->B);
  O(HXO_DET_C) = sum_b(m->detritus_c, m->B);
  O(HXO_SOIL_C) = sum_b(m->soil_c, m->B);
  O(HXO_THAWED_C) = sum_b(m->thawed_c, m->B);
  O(HXO_EARTH_C) = m->earth_c;
  O(HXO_NSTASH) = m->timesteps;
  O(HXO_MAXTS) = m->max_timestep;
  O(HXO_SOLVER_DT) = m->dt;
  O(HXO_NSTEPS) = (double)m->nsteps_year;
  O(HXO_NRHS) = (double)m->nrhs_year;
  O(HXO_OCEAN_UPTAKE) = m->annualflux_sum;
  O(HXO_NBP) = m->nbp;
  O(HXO_NPP) = m->final_npp; O(HXO_RH) = m->final_rh;
  O(HXO_RH_DET) = m->final_rh_det; O(HXO_RH_SOIL) = m->final_rh_soil;
  O(HXO_RH_CH4) = m->RH_ch4_sum;
  {
    /* f_frozen_weighted_mean  simpleNbox.cpp:492-514 */
    const double ptot = sum_b(m->permafrost_c, m->B);
    double ff = 0.0;
    if (ptot > 0.0) for (int b = 0; b < m->B; b++) ff += (m->permafrost_c[b] / ptot) * m->f_frozen[b];
    else ff = 1.0;
    O(HXO_F_FROZEN) = ff;
  }
  O(HXO_CA_RESIDUAL) = m->Ca_residual;
  O(HXO_HL_UPTAKE) = m->annualflux_sumHL; O(HXO_LL_UPTAKE) = m->annualflux_sumLL;
  O(HXO_C_HL) = m->carbon[HL]; O(HXO_C_LL) = m->carbon[LL];
  O(HXO_C_IO) = m->carbon[IO]; O(HXO_C_DO) = m->carbon[DO];
  O(HXO_DIC_HL) = csys_dic_umol(&m->chem[HL], m->carbon[HL]);
  O(HXO_DIC_LL) = csys_dic_umol(&m->chem[LL], m->carbon[LL]);
  O(HXO_HL_DO) = m->hl_do;
  O(HXO_OMEGAAR_HL) = m->chem[HL].OmegaAr; O(HXO_OMEGAAR_LL) = m->chem[LL].OmegaAr;
  O(HXO_OMEGACA_HL) = m->chem[HL].OmegaCa; O(HXO_OMEGACA_LL) = m->chem[LL].OmegaCa;
  O(HXO_TEMP_HL) = m->Tbox[HL]; O(HXO_TEMP_LL) = m->Tbox[LL];
  O(HXO_CO3_HL) = m->chem[HL].CO3; O(HXO_CO3_LL) = m->chem[LL].CO3;
  if (m->active_chem) { /* calc_revelle  oceanbox.cpp:278-292 */
    O(HXO_REVELLE_HL) = csys_dic_umol(&m->chem[HL], m->carbon[HL]) / m->chem[HL].CO3;
    O(HXO_REVELLE_LL) = csys_dic_umol(&m->chem[LL], m->carbon[LL]) / m->chem[LL].CO3;
  }
  O(HXO_TAU_OH) = m->tau_oh;
  for (int b = 0; b < m->B && b < 4; b++) {
    const int k0 = HXO_BIOME0 + 11 * b;
    O(k0 + 0) = m->veg_c[b]; O(k0 + 1) = m->detritus_c[b]; O(k0 + 2) = m->soil_c[b];
    O(k0 + 3) = m->permafrost_c[b]; O(k0 + 4) = m->thawed_c[b];
    O(k0 + 5) = m->final_npp_b[b]; O(k0 + 6) = m->final_rh_b[b];
    O(k0 + 7) = m->snbox_in_spinup ? 0.0 : snb_rh_ftpa_ch4(m, b);
    O(k0 + 8) = m->f_frozen[b]; O(k0 + 9) = m->tempfertd[b]; O(k0 + 10) = m->tempferts[b];
  }
#undef O
}

static void record_outputs(member_t *m, int iy, double *out) { record_outputs_stride(m, iy, out, m->ns); }

/* prepareToRun of every component (core.cpp:372-376) */
static void member_prepare(member_t *m, const hxo_scenario *s, const hxo_params *p,
                           double *buf) {
  const int ns = s->ns;
  memset(m, 0, sizeof *m);
  m->sc = s; m->pa = p; m->B = p->nbiome; m->ns = ns;
  memset(buf, 0, sizeof(double) * (size_t)ns * 8);
  m->Ker = buf; m->forcing = buf + ns; m->temp = buf + 2 * ns;
  m->temp_landair = buf + 3 * ns; m->temp_sst = buf + 4 * ns;
  m->heatflux_mixed = buf + 5 * ns; m->heatflux_interior = buf + 6 * ns;
  m->Tland_record = buf + 7 * ns;
  m->Tland_first = -1;
  m->trk_iy = -1;
  ocean_prepare(m);
  /* SimpleNbox: simpleNbox.cpp:45-79, runtime.cpp:66-190 */
  m->earth_c = 5500;
  m->cum_luc_va = 0.0; m->npp_luc_adjust = 1.0; m->masstot = 0.0;
  for (int b = 0; b < m->B; b++) {
    m->veg_c[b] = p->veg_c[b]; m->detritus_c[b] = p->detritus_c[b];
    m->soil_c[b] = p->soil_c[b]; m->permafrost_c[b] = p->permafrost_c[b];
    m->thawed_c[b] = 0.0;
    m->co2fert[b] = 1.0; m->tempfertd[b] = 1.0; m->tempferts[b] = 1.0;
    m->f_frozen[b] = 1.0; m->f_new_thaw[b] = 0.0;
  }
  m->end_of_spinup_vegc = sum_b(m->veg_c, m->B);
  m->cumulative_pf_ch4 = 0.0;
  m->has_been_run_before = 0;
  m->atmos_c = p->C0 * PPMVCO2_TO_PGC;
  /* solver: carbon-cycle-solver.cpp:118-133 */
  m->t = s->start; m->dt = s->dt;
  /* gases: prepareToRun of CH4/OH/N2O/halocarbons */
  /* ch4_component.cpp:137-147, n2o_component.cpp:137-145: a constraint at startDate replaces
   * the preindustrial value (OH keeps the INI M0: its prepareToRun runs first) */
  m->M0_eff = con_has(s->ch4_con, 0) ? s->ch4_con[0] : s->M0;
  m->N0_eff = con_has(s->n2o_con, 0) ? s->n2o_con[0] : s->N0;
  m->ch4_prev = m->M0_eff; m->n2o_prev = m->N0_eff; m->tau_oh = s->TOH0;
  m->ch4 = m->M0_eff; m->n2o = m->N0_eff; m->o3 = s->PO3;
  for (int h = 0; h < s->nhalo; h++) m->halo_conc[h] = s->halo[h].H0;
  doeclim_prepare(m);
  m->tas_land = 0.0; m->sst_now = 0.0;
}

/* Core::run_spinup core.cpp:394-420; returns the number of steps.  spin_out (may be NULL):
 * [variable][max_spinup] -- the state after every step as the output-stream visitor would see it
 * (CSVOutputStreamVisitor is visited after every spinup step with spinup = 1, core.cpp:402-408,
 * csv_outputstream_visitor.cpp:86-95), row step-1. */
static int member_spinup_rec(member_t *m, double *spin_out) {
  const hxo_scenario *s = m->sc;
  int step = 0, spunup = 0;
  if (s->do_spinup) {
    m->core_in_spinup = 1;
    int first = 1;
    double c_old[NC], c_new[NC];
    while (!spunup && ++step < s->max_spinup) {
      ocean_run(m);            /* ocean.run_spinup = run(step) */
      m->snbox_in_spinup = 1;  /* simpleNbox.run_spinup */
      /* solver.run_spinup  carbon-cycle-solver.cpp:313-370 */
      if (first) {
        first = 0;
        m->t = step - 1;
        snb_getCValues(m, m->t, c_old);
        snb_record_state(m);
      }
      snb_getCValues(m, m->t, c_old);
      solver_run(m, (double)step);
      snb_getCValues(m, (double)step, c_new);
      double max_dcdt = 0.0;
      for (int i = 0; i < NC; i++) {
        double d = fabs(c_new[i] - c_old[i]);
        if (d > max_dcdt) max_dcdt = d;
      }
      spunup = (max_dcdt < s->eps_spinup);
      if (spunup) m->t = s->start;
      snb_record_state(m);
      if (spin_out) record_outputs_stride(m, step - 1, spin_out, s->max_spinup);
    }
    if (!spunup) m->err |= HXO_ERR_SPINUP;
    m->core_in_spinup = 0;
  }
  return step;
}

static int member_spinup(member_t *m) { return member_spinup_rec(m, NULL); }

/* slrComponent::run + compute_slr  src/slr_component.cpp:116-232 (Vermeer & Rahmstorf 2009).
 * tgav is recorded from startDate+1; nothing is computed before the end of the reference
 * period (1980); dT/dt is the interpolator's derivative of the series AS KNOWN WHEN the date
 * is computed (h_interpolator.cpp:132-167): slope of the only adjacent segment at either end
 * of the series, the mean of the two adjacent slopes inside it. */
static void slr_compute(member_t *m, int date, int lastdate, double *out) {
  const hxo_scenario *s = m->sc;
  const int ns = m->ns, first = s->start + 1;
  const double *tg = m->temp;
#define TG(Y) tg[(Y) - s->start]
  const double T = TG(date) - m->refperiod_tgav;
  double dTdt = 0.0;
  if (lastdate - first + 1 > 2) {
    if (date == first) dTdt = (TG(first + 1) - TG(first)) / 1.0;
    else if (date == lastdate) dTdt = (TG(date) - TG(date - 1)) / 1.0;
    else {
      double slopePrev = (TG(date) - TG(date - 1)) / 1.0;
      double slopeNext = (TG(date + 1) - TG(date)) / 1.0;
      dTdt = (slopePrev + slopeNext) / 2.0;
    }
  }
#undef TG
  const int iy = date - s->start;
  const double a = 0.56, b = -4.9, T0 = -0.41;
  const double dHdt = a * (T - T0) + b * dTdt;
  out[HXO_SL_RC * ns + iy] = dHdt;
  double to_date = (date - 1 >= first) ? out[HXO_SLR * ns + iy - 1] : 0.0;
  out[HXO_SLR * ns + iy] = to_date + dHdt;
  const double a_ni = 0.08, b_ni = 2.5, T0_ni = -0.375;
  const double dHdt_ni = a_ni * (T - T0_ni) + b_ni * dTdt;
  out[HXO_SL_RC_NO_ICE * ns + iy] = dHdt_ni;
  double to_date_ni = (date - 1 >= first) ? out[HXO_SLR_NO_ICE * ns + iy - 1] : 0.0;
  out[HXO_SLR_NO_ICE * ns + iy] = to_date_ni + dHdt_ni;
}

static void slr_run(member_t *m, int year, double *out) {
  const hxo_scenario *s = m->sc;
  const int lo = 1951, hi = 1980;
  if (s->start + 1 > lo) return; /* the reference needs tgav over the reference period */
  if (year == hi) {
    double sum = 0.0;
    for (int i = lo; i <= hi; i++) sum += m->temp[i - s->start];
    m->refperiod_tgav = sum / (hi - lo + 1);
    for (int i = s->start + 1; i <= hi; i++) slr_compute(m, i, hi, out);
  }
  if (year > hi) slr_compute(m, year, year, out);
}

/* Core::run core.cpp:483-504, component order SURVEY 3c */
/* Conditioning probe of the TEST SUITE (never set by anything else): with rel != 0 every pool is
 * multiplied by 1 + rel * xi (xi pseudo-random in [-1, 1)) at the end of every model year, and so
 * is every air-sea flux the alkalinity tuner evaluates (box_fmin) -- the size of rounding
 * differences between two faithful implementations (another libm, FMA contraction).  How far the
 * trajectory moves tells how strongly this member amplifies such differences: the ocean boxes
 * under one explicit-Euler stash a year, and Brent's branch decisions on a V-shaped objective
 * whose two flanks give equal values.  Per thread. */

static void member_main(member_t *m, int run_to, double *out) {
  const hxo_scenario *s = m->sc;
  const int ns = s->ns;
  m->nsteps_year = m->nrhs_year = 0;
  m->timesteps = 0;
  record_outputs(m, 0, out);
  out[HXO_HL_PH * ns] = m->chem[HL].pH;
  out[HXO_CH4 * ns] = m->M0_eff; out[HXO_N2O * ns] = m->N0_eff; out[HXO_O3 * ns] = s->PO3;
  for (int year = s->start + 1; year <= run_to && year <= s->end; year++) {
    int iy = year - s->start;
    m->nsteps_year = m->nrhs_year = 0;
    year_gases(m, iy);
    ocean_run(m);
    /* SimpleNbox::run  runtime.cpp:203-228 */
    m->snbox_in_spinup = m->core_in_spinup;
    if (!m->has_been_run_before) {
      m->end_of_spinup_vegc = sum_b(m->veg_c, m->B);
      m->has_been_run_before = 1;
    }
    m->Tland_record[iy] = m->tas_land;
    if (m->Tland_first < 0) m->Tland_first = iy;
    if (iy == m->trk_iy && !m->trk_on) { /* start_tracking: every pool is 100 % itself */
      m->trk_on = 1;
      m->TP = 2 + 5 * m->B + 4;
      m->trk[TP_ATM] = tv_self(TP_ATM, m->atmos_c);
      m->trk[TP_EARTH] = tv_self(TP_EARTH, m->earth_c);
      for (int b = 0; b < m->B; b++) {
        m->trk[TP_LAND(b, 0)] = tv_self(TP_LAND(b, 0), m->veg_c[b]);
        m->trk[TP_LAND(b, 1)] = tv_self(TP_LAND(b, 1), m->detritus_c[b]);
        m->trk[TP_LAND(b, 2)] = tv_self(TP_LAND(b, 2), m->soil_c[b]);
        m->trk[TP_LAND(b, 3)] = tv_self(TP_LAND(b, 3), m->permafrost_c[b]);
        m->trk[TP_LAND(b, 4)] = tv_self(TP_LAND(b, 4), m->thawed_c[b]);
      }
      for (int b = 0; b < 4; b++) {
        m->trk[TP_OCEAN(m, b)] = tv_self(TP_OCEAN(m, b), m->carbon[b]);
        m->trk_addn[b] = tv_self(TP_OCEAN(m, b), 0.0);
        m->trk_subn[b] = 0.0;
      }
    }
    if (m->trk_on) { /* set_atmosphere_sources(atmos_c)  runtime.cpp:225-227 */
      m->trk_atm_copy = m->trk[TP_ATM];
      m->trk_atm_copy.val = m->atmos_c;
    }
    solver_run(m, (double)year);
    double CO2_conc = m->atmos_c * PGC_TO_PPMVCO2;
    year_forcing(m, year, CO2_conc);
    doeclim_run(m, iy, m->rf_tot);
    record_outputs(m, iy, out);
    out[HXO_TGAV * ns + iy] = m->temp[iy];
    out[HXO_SST * ns + iy] = m->sst_now;     /* lo_sst / lo_temp_landair when the */
    out[HXO_TLAND * ns + iy] = m->tas_land;  /* warming ratio is set, :586-625     */
    out[HXO_HEATFLUX * ns + iy] =
        m->heatflux_mixed[iy] + d_fso * m->heatflux_interior[iy];
    out[HXO_RF_TOT * ns + iy] = m->rf_tot;
    out[HXO_RF_CO2 * ns + iy] = m->rf_co2;
    out[HXO_RF_CH4 * ns + iy] = m->rf_ch4;
    out[HXO_RF_N2O * ns + iy] = m->rf_n2o;
    out[HXO_CH4 * ns + iy] = m->ch4;
    out[HXO_N2O * ns + iy] = m->n2o;
    out[HXO_O3 * ns + iy] = m->o3;
    out[HXO_GMST * ns + iy] = m->temp_surface_now;
    out[HXO_FLUX_MIXED * ns + iy] = m->flux_mixed_now;
    out[HXO_FLUX_INTERIOR * ns + iy] = m->flux_interior_now;
    out[HXO_OCEAN_TAS * ns + iy] =
        (m->pa->lo_warming_ratio != 0) ? m->sst_now * d_bsi : d_bsi * m->temp_sst[iy];
    for (int k = 0; k < 11; k++) out[(HXO_RF_H2O + k) * ns + iy] = m->rf_item_v[k];
    slr_run(m, year, out);
    if (m->err) break; /* the reference aborts the run at the first error (h_exception) */
    if (g_rounding_noise != 0.0) { /* conditioning probe, see hxo_set_rounding_noise */
      double *v[2 + 4 + 3 * HXO_MAXB];
      int nv = 0;
      v[nv++] = &m->atmos_c;
      for (int b = 0; b < 4; b++) v[nv++] = &m->carbon[b];
      for (int b = 0; b < m->B; b++) {
        v[nv++] = &m->veg_c[b]; v[nv++] = &m->detritus_c[b]; v[nv++] = &m->soil_c[b];
      }
      for (int k = 0; k < nv; k++) *v[k] *= 1.0 + g_rounding_noise * noise_xi();
    }
    if (m->trk_on && m->trk_out_f) { /* CSVFluxPoolVisitor: pools and their source fractions */
      const int TP = m->TP;
      for (int pl = 0; pl < TP; pl++) {
        m->trk_out_v[(size_t)iy * TP + pl] = m->trk[pl].val;
        for (int sc = 0; sc < TP; sc++)
          m->trk_out_f[((size_t)iy * TP + pl) * TP + sc] = m->trk[pl].f[sc];
      }
    }
  }
}

void hxo_set_rounding_noise(double rel) {
  g_rounding_noise = rel;
  noise_state = 0x9E3779B97F4A7C15ULL;
}

int hxo_run_member(const hxo_scenario *s, const hxo_params *p, int run_to,
                   double *out, int *spinup_steps) {
  return hxo_run_member_tracking(s, p, run_to, -1, out, spinup_steps, NULL, NULL);
}

int hxo_run_member_tracking(const hxo_scenario *s, const hxo_params *p, int run_to,
                            int tracking_date, double *out, int *spinup_steps, double *trk_f,
                            double *trk_v) {
  member_t M, *m = &M;
  const int ns = s->ns;
  memset(out, 0, sizeof(double) * (size_t)HXO_NVAR * (size_t)ns);
  double *buf = (double *)malloc(sizeof(double) * (size_t)ns * 8);
  member_prepare(m, s, p, buf);
  m->trk_iy = tracking_date - s->start; /* Core::trackingDate, 9999 = never */
  if (tracking_date < 0) m->trk_iy = -1;
  m->trk_out_f = trk_f; m->trk_out_v = trk_v;
  int step = member_spinup(m);
  if (spinup_steps) *spinup_steps = step;
  member_main(m, run_to, out);
  free(buf);
  return m->err;
}

/* The spinup alone, with what the output stream sees after every step (spinup = 1 rows):
 * spin_out [HXO_NVAR][s->max_spinup], row step-1; returns the error mask, *steps the step count. */
int hxo_run_member_spinup(const hxo_scenario *s, const hxo_params *p, double *spin_out, int *steps) {
  member_t M, *m = &M;
  memset(spin_out, 0, sizeof(double) * (size_t)HXO_NVAR * (size_t)s->max_spinup);
  double *buf = (double *)malloc(sizeof(double) * (size_t)s->ns * 8);
  member_prepare(m, s, p, buf);
  const int step = member_spinup_rec(m, spin_out);
  if (steps) *steps = step;
  free(buf);
  return m->err;
}

/* The ensemble of BASELINE configs 2-4: members differ only in S and q10_rh,
 * which do not enter the spinup, so the spinup is done once and its
 * end state is the starting point of every member -- the same sharing the GPU
 * path uses; tests check it equals per-member hxo_run_member bit for bit. */
int hxo_run_ensemble_ecs_q10(const hxo_scenario *s, const hxo_params *base,
                             int n, const double *S, const double *q10,
                             int run_to, double *co2, double *tgav) {
  int err = 0;
  const int ns = s->ns;
  double *out = (double *)malloc(sizeof(double) * (size_t)HXO_NVAR * (size_t)ns);
  double *buf0 = (double *)malloc(sizeof(double) * (size_t)ns * 8);
  double *buf = (double *)malloc(sizeof(double) * (size_t)ns * 8);
  member_t M0;
  member_prepare(&M0, s, base, buf0);
  (void)member_spinup(&M0);
  for (int i = 0; i < n; i++) {
    hxo_params p = *base;
    p.S = S[i];
    for (int b = 0; b < p.nbiome; b++) p.q10_rh[b] = q10[i];
    member_t M = M0;
    memcpy(buf, buf0, sizeof(double) * (size_t)ns * 8);
    M.pa = &p;
    M.Ker = buf; M.forcing = buf + ns; M.temp = buf + 2 * ns;
    M.temp_landair = buf + 3 * ns; M.temp_sst = buf + 4 * ns;
    M.heatflux_mixed = buf + 5 * ns; M.heatflux_interior = buf + 6 * ns;
    M.Tland_record = buf + 7 * ns;
    doeclim_prepare(&M); /* A, IB, time scales depend on S */
    memset(out, 0, sizeof(double) * (size_t)HXO_NVAR * (size_t)ns);
    member_main(&M, run_to, out);
    err |= M.err;
    if (co2) memcpy(co2 + (size_t)i * ns, out + HXO_CO2 * ns, sizeof(double) * ns);
    if (tgav) memcpy(tgav + (size_t)i * ns, out + HXO_TGAV * ns, sizeof(double) * ns);
  }
  free(out); free(buf); free(buf0);
  return err;
}

/* see hector_oracle.h */
static int same_spinup_params(const hxo_params *a, const hxo_params *b) {
  if (a->nbiome != b->nbiome || a->C0 != b->C0 || a->tt != b->tt || a->tu != b->tu ||
      a->twi != b->twi || a->tid != b->tid || a->preind_surface_c != b->preind_surface_c ||
      a->preind_interdeep_c != b->preind_interdeep_c)
    return 0;
  for (int k = 0; k < a->nbiome; k++)
    if (a->npp_flux0[k] != b->npp_flux0[k] || a->veg_c[k] != b->veg_c[k] ||
        a->detritus_c[k] != b->detritus_c[k] || a->soil_c[k] != b->soil_c[k] ||
        a->permafrost_c[k] != b->permafrost_c[k] || a->f_nppv[k] != b->f_nppv[k] ||
        a->f_nppd[k] != b->f_nppd[k] || a->f_litterd[k] != b->f_litterd[k])
      return 0;
  return 1;
}

int hxo_run_ensemble(const hxo_scenario *s, const hxo_params *params, int n, int run_to,
                     double *co2, double *tgav, unsigned char *timesteps, int *errs) {
  int err = 0;
  const int ns = s->ns;
  double *out = (double *)malloc(sizeof(double) * (size_t)HXO_NVAR * (size_t)ns);
  double *buf0 = (double *)malloc(sizeof(double) * (size_t)ns * 8);
  double *buf = (double *)malloc(sizeof(double) * (size_t)ns * 8);
  member_t M0;
  member_prepare(&M0, s, &params[0], buf0);
  (void)member_spinup(&M0);
  for (int i = 0; i < n; i++) {
    member_t M;
    if (same_spinup_params(&params[0], &params[i])) {
      M = M0;
      memcpy(buf, buf0, sizeof(double) * (size_t)ns * 8);
      M.pa = &params[i];
      M.Ker = buf; M.forcing = buf + ns; M.temp = buf + 2 * ns;
      M.temp_landair = buf + 3 * ns; M.temp_sst = buf + 4 * ns;
      M.heatflux_mixed = buf + 5 * ns; M.heatflux_interior = buf + 6 * ns;
      M.Tland_record = buf + 7 * ns;
      doeclim_prepare(&M); /* A, IB, time scales, Ker depend on S, diff, qco2 */
    } else {
      member_prepare(&M, s, &params[i], buf);
      (void)member_spinup(&M);
    }
    memset(out, 0, sizeof(double) * (size_t)HXO_NVAR * (size_t)ns);
    member_main(&M, run_to, out);
    err |= M.err;
    if (errs) errs[i] = M.err;
    if (co2) memcpy(co2 + (size_t)i * ns, out + HXO_CO2 * ns, sizeof(double) * ns);
    if (tgav) memcpy(tgav + (size_t)i * ns, out + HXO_TGAV * ns, sizeof(double) * ns);
    if (timesteps)
      for (int y = 0; y < ns; y++)
        timesteps[(size_t)i * ns + y] = (unsigned char)out[(size_t)HXO_NSTASH * ns + y];
  }
  free(out); free(buf); free(buf0);
  return err;
}
