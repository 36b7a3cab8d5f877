// hx_main.cpp -- command-line front end: `hector-amd <ini>` runs a scenario and writes
// output/outputstream_<run_name>.csv in the reference's long format, like the reference's
// wrapper (src/main.cpp:36-128 + CSVOutputStreamVisitor, src/csv_outputstream_visitor.cpp):
//
//   # Output from hector-amd ... on <date>
//   year,run_name,spinup,component,variable,value,units
//
// One row per (year, variable).  Differences from the reference, by design of the ensemble
// path: the spinup = 1 rows (one set per spinup step, the step number in the year column) hold the
// carbon-cycle variables of simpleNbox and the ocean only -- the rest does not run in the spinup
// (--no-spinup-rows leaves them out; off beyond 4 096 members); with --members N > 1 the run_name column is "<run_name>.<member>"; values are
// printed with 6 significant digits like the reference (forcings with 4,
// csv_outputstream_visitor.cpp:129) unless --precision is given.
// Talks to libhector_amd.so through the C ABI only.
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <exception>
#include <string>
#include <vector>

#include "../../include/hector_amd.h"

namespace {

// the variables of one year in the order the visitor prints them (csv_outputstream_visitor.cpp:
// 126-365); component and unit strings come from the library (hx_var_info)
const char *const kStreamVars[] = {
    "NBP", "NPP", "RH", "rh_det", "rh_soil", "rh_ch4", "CO2_concentration", "atmos_co2",
    "atmos_c_residual", "veg_c", "detritus_c", "soil_c", "permafrost_c", "thawedp_c", "f_frozen",
    "earth_c", "global_tas", "gmst", "heatflux_mixed", "heatflux_interior", "heatflux",
    "land_tas", "sst", "HL_ocean_uptake", "LL_ocean_uptake", "DO_ocean_c", "HL_ocean_c",
    "IO_ocean_c", "LL_ocean_c", "HL_DIC", "LL_DIC", "HL_downwelling", "ocean_uptake",
    "HL_OmegaAr", "LL_OmegaAr", "HL_OmegaCa", "LL_OmegaCa", "HL_PCO2", "LL_PCO2", "HL_pH",
    "LL_pH", "HL_sst", "LL_sst", "ocean_c", "HL_CO3", "LL_CO3", "HL_Revelle", "LL_Revelle",
    "O3_concentration", "TAU_OH", "CH4_concentration", "N2O_concentration",
};
const char *const kForcings[] = {"RF_BC", "RF_CH4", "RF_CO2", "RF_H2O_strat", "RF_N2O", "RF_NH3",
                                 "RF_O3_trop", "RF_OC", "RF_SO2", "RF_aci", "RF_albedo",
                                 "RF_misc", "RF_tot", "RF_vol"};
const char *const kSlr[] = {"sl_rc", "slr", "sl_rc_no_ice", "slr_no_ice"};
// with more than one biome the visitor adds these per biome (:169-198)
const char *const kBiomeVars[] = {"NPP", "RH", "rh_ch4", "veg_c", "detritus_c", "soil_c", "permafrost_c",
                                  "thawedp_c", "f_frozen", "detritus_tempfert", "soil_tempfert"};

void die(const std::string &msg, int code) {
  std::fprintf(stderr, "* Program exception:\n%s\n", msg.c_str());
  std::exit(code);
}
void ck(int rc) { if (rc) die(hx_last_error(), 1); }

std::vector<double> parse_values(const std::string &txt) {
  std::vector<double> v;
  size_t p = 0;
  while (p <= txt.size()) {
    size_t q = txt.find(',', p);
    if (q == std::string::npos) q = txt.size();
    v.push_back(std::strtod(txt.substr(p, q - p).c_str(), nullptr));
    p = q + 1;
  }
  return v;
}

}  // namespace

static int run_main(int argc, char **argv) {
  std::string scenario, outdir = "output/";
  int members = 1, device = 0, precision = 0, runto = -1;
  bool spinup_rows = true;  // the stream's spinup = 1 rows (one set per spinup step)
  std::vector<int> devices;  // --devices 0,1,...: the members sharded over several GPUs
  std::vector<std::pair<std::string, std::string>> params;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) die("missing value after " + a, 1);
      return argv[++i];
    };
    if (a == "--members") members = std::atoi(next().c_str());
    else if (a == "--device") device = std::atoi(next().c_str());
    else if (a == "--devices") for (double d : parse_values(next())) devices.push_back((int)d);
    else if (a == "--output-dir") { outdir = next(); if (outdir.back() != '/') outdir += '/'; }
    else if (a == "--precision") precision = std::atoi(next().c_str());
    else if (a == "--run-to") runto = std::atoi(next().c_str());
    else if (a == "--no-spinup-rows") spinup_rows = false;
    else if (a == "--set") {  // --set S=2.5,3.0,4.1   one value, or one per member
      const std::string kv = next();
      const size_t eq = kv.find('=');
      if (eq == std::string::npos) die("--set needs capability=value[,value...]", 1);
      params.emplace_back(kv.substr(0, eq), kv.substr(eq + 1));
    } else if (a == "-h" || a == "--help") {
      std::printf("Usage: hector-amd <config file name> [--members N] [--set cap=v[,v...]]...\n"
                  "       [--run-to year] [--output-dir dir] [--precision digits] [--device i]\n"
                  "       [--no-spinup-rows]\n"
                  "       [--devices i,j,...]   (the members in contiguous blocks over several GPUs)\n");
      return 0;
    } else if (scenario.empty()) scenario = a;
    else die("unexpected argument " + a, 1);
  }
  // src/main.cpp:47-60
  if (scenario.empty()) die("Usage: <program> <config file name>", 1);
  if (FILE *f = std::fopen(scenario.c_str(), "r")) std::fclose(f);
  else die("Couldn't find input file " + scenario, 1);

  hx_core *core = nullptr;
  if (devices.empty()) ck(hx_newcore(scenario.c_str(), members, device, &core));
  else ck(hx_newcore_devices(scenario.c_str(), members, devices.data(), (int)devices.size(), &core));
  for (auto &kv : params) {
    const std::vector<double> v = parse_values(kv.second);
    if ((int)v.size() != 1 && (int)v.size() != members)
      die("--set " + kv.first + ": need 1 or " + std::to_string(members) + " values", 1);
    ck(hx_setvar(core, kv.first.c_str(), v.data(), (int)v.size(), nullptr));
  }
  int start = 0, end = 0, cur = 0;
  ck(hx_dates(core, &start, &end, &cur));
  if (runto < 0 || runto > end) runto = end;

  const char *const *halo = nullptr;
  int nhalo = 0;
  ck(hx_halocarbons(core, &halo, &nhalo));
  std::vector<std::string> halo_names(halo, halo + nhalo);

  const char *const *bio = nullptr;
  int nbio = 0;
  ck(hx_biomes(core, &bio, &nbio));
  std::vector<std::string> biome_vars;  // "<biome>.<variable>" rows, only with several biomes
  if (nbio > 1)
    for (int b = 0; b < nbio; ++b)
      for (const char *v : kBiomeVars) biome_vars.push_back(std::string(bio[b]) + "." + v);

  // everything the stream prints
  std::vector<std::string> wanted;
  for (auto &bv : biome_vars) wanted.push_back(bv);
  for (const char *v : kStreamVars) wanted.push_back(v);
  for (const char *f : kForcings) wanted.push_back(f);
  for (const char *s : kSlr) wanted.push_back(s);
  {
    std::vector<const char *> ptr;
    for (auto &w : wanted) ptr.push_back(w.c_str());
    ck(hx_set_outputs(core, (int)ptr.size(), ptr.data()));
  }
  // the spinup as the stream sees it (csv_outputstream_visitor.cpp:86-95): 336 KB of HBM per
  // member -- for the ensembles a stream file makes sense for
  if (members > 4096) spinup_rows = false;
  if (spinup_rows) ck(hx_enable_spinup_record(core, 1));
  ck(hx_run(core, (double)runto));
  ck(hx_sync(core));

  std::vector<unsigned> status((size_t)members);
  ck(hx_status(core, status.data()));
  for (int i = 0; i < members; ++i)
    if (status[(size_t)i])
      std::fprintf(stderr, "member %d: model error flags 0x%x (see HX_ERR_* in hector_amd.h)\n", i,
                   status[(size_t)i]);

  // fetch: variable -> [year][member]
  const int y0 = start + 1, ny = runto - y0 + 1;
  std::map<std::string, std::vector<double>> data;
  auto fetch = [&](const std::string &name) {
    std::vector<double> &d = data[name];
    d.resize((size_t)ny * members);
    ck(hx_fetchvars(core, name.c_str(), y0, runto, d.data()));
  };
  for (auto &w : wanted) fetch(w);
  for (auto &h : halo_names) { fetch(h + "_concentration"); fetch("RF_" + h); }

  const char *rn_c = nullptr;
  ck(hx_run_name(core, &rn_c));
  const std::string rn = rn_c ? rn_c : "";
  mkdir(outdir.c_str(), 0777);  // ensure_dir_exists(OUTPUT_DIRECTORY)
  const std::string path = outdir + (rn.empty() ? "outputstream.csv" : "outputstream_" + rn + ".csv");
  FILE *out = std::fopen(path.c_str(), "w");
  if (!out) die("cannot write " + path, 1);
  {
    std::time_t t = std::time(nullptr);
    std::string when = std::ctime(&t);
    while (!when.empty() && when.back() == '\n') when.pop_back();
    std::fprintf(out, "# Output from hector-amd (%s) version 3.5.0-compatible on %s\n", hx_backend(),
                 when.c_str());
    std::fprintf(out, "year,run_name,spinup,component,variable,value,units\n");
  }
  int base = 1750;
  {
    // forcing rows start at the base year (csv_outputstream_visitor.cpp:131-132)
    double b = 0;
    if (hx_getvar(core, "baseyear", &b) == 0 && b > 0) base = (int)b;
  }
  const int slr_from = 1990;  // max(refperiod_high, normalize_year), :300-318
  std::map<std::string, bool> comp_out;  // output=0 in a component's section silences its rows
  auto row = [&](int year, const std::string &run, const char *comp, const std::string &var,
                 double v, const char *units, int prec) {
    auto it = comp_out.find(comp);
    if (it == comp_out.end()) {
      int en = 1;
      ck(hx_component_output(core, comp, &en));
      it = comp_out.emplace(comp, en != 0).first;
    }
    if (!it->second) return;
    std::fprintf(out, "%d,%s,0,%s,%s,%.*g,%s\n", year, run.c_str(), comp, var.c_str(), prec, v, units);
  };
  const int p_def = precision > 0 ? precision : 6, p_rf = precision > 0 ? precision : 4;
  struct Info { std::string component, units; };
  std::map<std::string, Info> info;
  for (const char *v : kStreamVars) {
    const char *c = nullptr, *u = nullptr;
    ck(hx_var_info(core, v, &c, &u));
    info[v] = Info{c, u};
  }
  // spinup = 1 rows: after every spinup step the visitor prints the model as it stands, the step
  // number in the year column.  Written here: the carbon-cycle variables of simpleNbox and the
  // ocean (the ones that move; hx_spinup_record).  Temperature, gases and forcing do not run in
  // the spinup and the ocean chemistry is off (spinup_chem = 0): what the reference prints for
  // them there is initial or undefined state and is left out.
  std::vector<double> sp;
  const char *const *sp_names = nullptr;
  int sp_nv = 0, max_spin = 0;
  if (spinup_rows) {
    ck(hx_spinup_record(core, 0, &sp_names, &sp_nv, nullptr, 0, nullptr));
    for (int mbr = 0; mbr < members; ++mbr) {
      int st = 0;
      ck(hx_spinup_steps(core, mbr, &st));
      if (st > max_spin) max_spin = st;
    }
    sp.resize((size_t)(max_spin > 0 ? max_spin : 1) * sp_nv);
  }
  auto sp_row = [&](int step, const std::string &run, const char *comp, const std::string &var,
                    double v, const char *units) {
    auto it = comp_out.find(comp);
    if (it == comp_out.end()) {
      int en = 1;
      ck(hx_component_output(core, comp, &en));
      it = comp_out.emplace(comp, en != 0).first;
    }
    if (it->second)
      std::fprintf(out, "%d,%s,1,%s,%s,%.*g,%s\n", step, run.c_str(), comp, var.c_str(), p_def, v, units);
  };
  for (int mbr = 0; mbr < members; ++mbr) {
    const std::string run = members > 1 ? rn + "." + std::to_string(mbr) : rn;
    if (spinup_rows && max_spin > 0) {
      int steps = 0;
      ck(hx_spinup_record(core, mbr, nullptr, nullptr, sp.data(), max_spin, &steps));
      std::map<std::string, int> col;
      for (int v = 0; v < sp_nv; ++v) col[sp_names[v]] = v;
      static const char *const land[] = {"NBP", "NPP", "RH", "rh_det", "rh_soil", "rh_ch4", "CO2_concentration",
                                         "atmos_co2", "atmos_c_residual", "veg_c", "detritus_c", "soil_c",
                                         "permafrost_c", "thawedp_c", "f_frozen", "earth_c"};
      static const char *const sea[] = {"HL_ocean_uptake", "LL_ocean_uptake", "DO_ocean_c", "HL_ocean_c",
                                        "IO_ocean_c", "LL_ocean_c", "HL_downwelling", "ocean_uptake", "ocean_c"};
      for (int st = 1; st <= steps; ++st) {
        const double *r = sp.data() + (size_t)(st - 1) * sp_nv;
        auto val = [&](const std::string &v) {
          if (v == "rh_ch4") return r[col["RH"]];                 // (the stream's alias, :151)
          if (v == "CO2_concentration") return r[col["atmos_co2"]] * (1.0 / 2.13);
          if (v == "f_frozen") return 1.0;                        // (no thaw in the spinup)
          if (v == "ocean_c")
            return r[col["DO_ocean_c"]] + r[col["IO_ocean_c"]] + r[col["LL_ocean_c"]] + r[col["HL_ocean_c"]];
          return r[col[v]];
        };
        for (const char *v : land) sp_row(st, run, "simpleNbox", v, val(v), info[v].units.c_str());
        for (const char *v : sea) sp_row(st, run, "ocean", v, val(v), info[v].units.c_str());
      }
    }
    for (int y = y0; y <= runto; ++y) {
      const size_t o = (size_t)(y - y0) * members + mbr;
      if (y >= base) {
        std::map<std::string, double> rf;  // the reference's map order
        for (const char *f : kForcings) rf[f] = data[f][o];
        for (auto &h : halo_names) rf["RF_" + h] = data["RF_" + h][o];
        for (auto &kv : rf) row(y, run, "forcing", kv.first, kv.second, "W/m2", p_rf);
      }
      std::string last_comp;
      for (const char *vname : kStreamVars) {
        const Info &vi = info[vname];
        if (vi.component == "temperature" && last_comp != "temperature")
          for (auto &h : halo_names)  // halocarbon components come before temperature
            row(y, run, (h + "_halocarbon").c_str(), h + "_concentration",
                data[h + "_concentration"][o], "pptv", p_def);
        if (vi.component == "ozone" && last_comp != "ozone" && ny > 0) {
          // slrComponent: nothing until 1990, then the back years in one go (:300-318)
          if (y == slr_from)
            for (int yy = y0; yy < y; ++yy) {
              row(yy, run, "slr", "slr", data["slr"][(size_t)(yy - y0) * members + mbr], "cm", p_def);
              row(yy, run, "slr", "slr_no_ice",
                  data["slr_no_ice"][(size_t)(yy - y0) * members + mbr], "cm", p_def);
            }
          if (y >= slr_from) {
            row(y, run, "slr", "sl_rc", data["sl_rc"][o], "cm/yr", p_def);
            row(y, run, "slr", "slr", data["slr"][o], "cm", p_def);
            row(y, run, "slr", "sl_rc_no_ice", data["sl_rc_no_ice"][o], "cm/yr", p_def);
            row(y, run, "slr", "slr_no_ice", data["slr_no_ice"][o], "cm", p_def);
          }
        }
        // the reference's stream prints final_rh under the name rh_ch4 (csv_outputstream_visitor.cpp:
        // 151); the file is kept identical, fetchvars("rh_ch4") returns the CH4 respiration itself
        const bool rh_alias = !std::strcmp(vname, "rh_ch4");
        row(y, run, vi.component.c_str(), vname, data[rh_alias ? "RH" : vname][o], vi.units.c_str(), p_def);
        if (!std::strcmp(vname, "earth_c"))  // biome rows close the simpleNbox block
          for (auto &bv : biome_vars) {
            const char *u = nullptr;
            ck(hx_var_info(core, bv.c_str(), nullptr, &u));
            row(y, run, "simpleNbox", bv, data[bv][o], u, p_def);
          }
        last_comp = vi.component;
      }
    }
  }
  std::fclose(out);
  std::printf("wrote %s (%d member%s, %d-%d)\n", path.c_str(), members, members > 1 ? "s" : "", y0,
              runto);
  // tracking_<run_name>.csv (src/main.cpp:91-105, CSVFluxPoolVisitor): when the INI's
  // [core] trackingDate falls inside the run
  double tdate = 9999.0;
  {
    std::vector<double> td((size_t)members);
    ck(hx_getvar(core, "trackingDate", td.data()));
    tdate = td[0];
  }
  if (tdate > y0 && tdate <= runto) {
    const char *const *pools = nullptr;
    int tp = 0;
    ck(hx_tracking_pools(core, &pools, &tp));
    const int t0 = (int)tdate, nyt = runto - t0 + 1;
    std::vector<double> tv((size_t)nyt * tp), tfr((size_t)nyt * tp * tp);
    const int tw = (tp + 63) / 64;  // mask words per pool
    std::vector<unsigned long long> tm((size_t)nyt * tp * tw);
    for (int mbr = 0; mbr < members; ++mbr) {  // one file per member ("<run_name>.<member>")
      const std::string run = members > 1 ? rn + "." + std::to_string(mbr) : rn;
      const std::string tpath = outdir + (run.empty() ? "tracking.csv" : "tracking_" + run + ".csv");
      FILE *tf = std::fopen(tpath.c_str(), "w");
      if (!tf) die("cannot write " + tpath, 1);
      std::fprintf(tf, "year,component,pool_name,pool_value,pool_units,source_name,source_fraction\n");
      ck(hx_tracking_data(core, mbr, t0, runto, tv.data(), tfr.data(), tm.data()));
      for (int y = 0; y < nyt; ++y)
        for (int p = 0; p < tp; ++p)
          for (int s2 = 0; s2 < tp; ++s2)
            if (tm[((size_t)y * tp + p) * tw + s2 / 64] >> (s2 % 64) & 1ull)
              std::fprintf(tf, "%d,%s,%s,%.*g,Pg C,%s,%.*g\n", t0 + y,
                           p >= tp - 4 ? "ocean" : "simpleNbox", pools[p], p_def,
                           tv[(size_t)y * tp + p], pools[s2], p_def,
                           tfr[((size_t)y * tp + p) * tp + s2]);
      std::fclose(tf);
      std::printf("wrote %s\n", tpath.c_str());
    }
  }
  ck(hx_shutdown(core));
  return 0;
}

// src/main.cpp:116-128: model errors (here: everything the C ABI reports) exit with 1, any other
// C++ exception with 2, anything else with 3
int main(int argc, char **argv) {
  try {
    return run_main(argc, argv);
  } catch (std::exception &e) {
    std::fprintf(stderr, "* Standard exception: %s\n", e.what());
    return 2;
  } catch (...) {
    std::fprintf(stderr, "* Other exception!\n");
    return 3;
  }
}
