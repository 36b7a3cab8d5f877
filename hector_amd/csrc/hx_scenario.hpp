// hx_scenario.hpp -- scenario inputs of one Hector configuration (what
// INIToCoreReader + CSVTableReader push into Core::setData, reference
// src/ini_to_core_reader.cpp:100-180, src/csv_table_reader.cpp:115-198),
// held as plain scalars and dense per-year series on startDate..endDate.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace hx {

struct Halocarbon {
  std::string name;  // e.g. "CF4" (section "<name>_halocarbon")
  double tau = 0, rho = 0, delta = 0, H0 = 0, molarMass = 0;
  std::vector<double> emissions;
};

class Scenario {
 public:
  // Load a dense scenario pack (.hxs, written by tools/import_scenario.py) or a
  // Hector INI file (with its csv: tables).  Throws std::runtime_error.
  static Scenario load(const std::string &path);

  int start = 0, end = 0;
  int ns() const { return end - start + 1; }

  double scalar(const std::string &section, const std::string &key) const;
  double scalar(const std::string &section, const std::string &key, double dflt) const;
  bool has_scalar(const std::string &section, const std::string &key) const;
  void validate_keys() const;  // every section a component, every key one it reads
  std::vector<std::string> scalar_keys(const std::string &section) const;
  void set_scalar(const std::string &section, const std::string &key, double v);
  std::string text(const std::string &section, const std::string &key,
                   const std::string &dflt) const;  // a non-numeric INI value (run_name)
  const std::vector<double> &series(const std::string &section,
                                    const std::string &key) const;
  bool has_series(const std::string &section, const std::string &key) const;
  void set_series_value(const std::string &section, const std::string &key, int year, double v);
  // Constraint series ("*_constrain") are dense like the others but hold NaN where the
  // reference's tseries would not return a value: CO2/NBP/CH4/N2O/halocarbon constraints exist
  // only at the dates given (tseries::exists); tas_constrain interpolates between its first
  // and last date (temperature_component.cpp:510-511); RF_tot_constrain also applies, flat,
  // before its first date (forcing_component.cpp:498).
  static bool is_constraint(const std::string &key);
  void set_constraint_point(const std::string &section, const std::string &key, int year,
                            double v);
  // the dates a constraint was given at (empty if none) and the dense [ns] series those
  // points mean under the rules above -- also used per member (setvar_dated_members)
  const std::map<int, double> &constraint_points(const std::string &section,
                                                 const std::string &key) const;
  static std::vector<double> densify_points(const std::map<int, double> &pts,
                                            const std::string &key, int start, int end);

  std::vector<Halocarbon> halocarbons;
  std::string source;

 private:
  static Scenario load_pack(const std::string &path);
  static Scenario load_ini(const std::string &path);
  void finish();
  void densify_constraint(const std::string &name);
  std::map<std::string, std::map<int, double>> con_points_;  // "section.key" -> given dates
  std::map<std::string, std::string> scalars_;           // "section.key" -> text
  std::map<std::string, std::vector<double>> series_;    // "section.key" -> [ns]
};

}  // namespace hx
