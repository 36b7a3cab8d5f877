// hx_scenario.cpp -- scenario front end: dense packs and Hector INI files.
//
// INI semantics follow the reference reader (file:line in /root/reference):
//   key=value, key[year]=value, key=csv:<file>      src/ini_to_core_reader.cpp:100-180
//   ';' starts a comment when at line start or after whitespace   src/ini.c:35-45,88-110
//   csv tables: header row names the column, first column is the date, rows
//   starting with ';' and the UNITS row are skipped, blank cells ignored;
//   a relative csv path is resolved against the INI's directory
//                                                   src/csv_table_reader.cpp:115-198
// A series read at a date: exact hit -> that value; else linear interpolation
// between neighbours; flat beyond either end; a one-point series is a constant
//                             inst/include/tseries.hpp:302-334, src/h_interpolator.cpp:103-122
#include "hx_scenario.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace hx {
namespace {

std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n");
  if (a == std::string::npos) return "";
  size_t b = s.find_last_not_of(" \t\r\n");
  return s.substr(a, b - a + 1);
}

double to_double(const std::string &s, const std::string &what) {
  char *end = nullptr;
  const std::string t = trim(s);
  double v = std::strtod(t.c_str(), &end);
  if (t.empty() || end == t.c_str() || *end != 0)
    throw std::runtime_error("Could not convert '" + s + "' to a number (" + what + ")");
  return v;
}

std::vector<std::string> split_commas(const std::string &line) {
  std::vector<std::string> out;
  size_t p = 0;
  while (true) {
    size_t q = line.find(',', p);
    if (q == std::string::npos) { out.push_back(line.substr(p)); break; }
    out.push_back(line.substr(p, q - p));
    p = q + 1;
  }
  return out;
}

std::string dirname_of(const std::string &p) {
  size_t s = p.find_last_of('/');
  return s == std::string::npos ? "." : p.substr(0, s);
}

bool file_exists(const std::string &p) { return std::ifstream(p).good(); }

using Points = std::map<double, double>;

double tseries_get(const Points &pts, double t) {
  if (pts.size() == 1) return pts.begin()->second;
  auto it = pts.find(t);
  if (it != pts.end()) return it->second;
  if (t < pts.begin()->first) return pts.begin()->second;
  if (t > pts.rbegin()->first) return pts.rbegin()->second;
  auto hi = pts.upper_bound(t);
  auto lo = std::prev(hi);
  return lo->second + (t - lo->first) * (hi->second - lo->second) / (hi->first - lo->first);
}

Points read_csv_column(const std::string &path, const std::string &column) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Could not open csv table " + path);
  std::string line;
  std::vector<std::string> header;
  size_t ci = 0;
  Points pts;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    const std::string t = trim(line);
    if (t.empty() || t[0] == ';') continue;
    auto row = split_commas(line);
    if (header.empty()) {
      header = row;
      for (size_t c = 1; c < header.size() && ci == 0; ++c)
        if (trim(header[c]) == column) ci = c;
      if (ci == 0)
        throw std::runtime_error("Could not find a column for " + column + " in " + path);
      continue;
    }
    const std::string idx = trim(row[0]);
    if (idx == "UNITS" || idx.empty()) continue;
    if (ci >= row.size()) continue;
    const std::string cell = trim(row[ci]);
    if (cell.empty()) continue;
    pts[to_double(idx, "csv index")] = to_double(cell, column);
  }
  return pts;
}

}  // namespace

double Scenario::scalar(const std::string &section, const std::string &key) const {
  auto it = scalars_.find(section + "." + key);
  if (it == scalars_.end())
    throw std::runtime_error("scenario has no scalar " + section + "." + key);
  return to_double(it->second, section + "." + key);
}
double Scenario::scalar(const std::string &section, const std::string &key, double dflt) const {
  auto it = scalars_.find(section + "." + key);
  return it == scalars_.end() ? dflt : to_double(it->second, section + "." + key);
}
std::string Scenario::text(const std::string &section, const std::string &key,
                           const std::string &dflt) const {
  auto it = scalars_.find(section + "." + key);
  return it == scalars_.end() ? dflt : it->second;
}
std::vector<std::string> Scenario::scalar_keys(const std::string &section) const {
  std::vector<std::string> out;
  const std::string pre = section + ".";
  for (auto &kv : scalars_)
    if (kv.first.compare(0, pre.size(), pre) == 0) out.push_back(kv.first.substr(pre.size()));
  return out;
}
void Scenario::set_scalar(const std::string &section, const std::string &key, double v) {
  char buf[40];
  std::snprintf(buf, sizeof buf, "%.17g", v);
  scalars_[section + "." + key] = buf;
  for (auto &h : halocarbons)
    if (section == h.name + "_halocarbon") {
      if (key == "tau") h.tau = v;
      else if (key == "rho_" + h.name) h.rho = v;
      else if (key == "delta_" + h.name) h.delta = v;
      else if (key == "H0") h.H0 = v;
      else if (key == "molarMass") h.molarMass = v;
    }
}
bool Scenario::has_scalar(const std::string &section, const std::string &key) const {
  return scalars_.count(section + "." + key) != 0;
}
const std::vector<double> &Scenario::series(const std::string &section,
                                            const std::string &key) const {
  auto it = series_.find(section + "." + key);
  if (it == series_.end())
    throw std::runtime_error("scenario has no series " + section + "." + key);
  return it->second;
}
bool Scenario::has_series(const std::string &section, const std::string &key) const {
  return series_.count(section + "." + key) != 0;
}

bool Scenario::is_constraint(const std::string &key) {
  const std::string suf = "_constrain";
  return key.size() > suf.size() && key.compare(key.size() - suf.size(), suf.size(), suf) == 0;
}

std::vector<double> Scenario::densify_points(const std::map<int, double> &pts,
                                             const std::string &key, int start, int end) {
  std::vector<double> v((size_t)(end - start + 1), std::nan(""));
  const bool interp = (key == "tas_constrain" || key == "RF_tot_constrain");
  if (!pts.empty()) {
    for (int y = start; y <= end; ++y) {
      auto hit = pts.find(y);
      if (hit != pts.end()) { v[(size_t)(y - start)] = hit->second; continue; }
      if (!interp) continue;
      if (y > pts.rbegin()->first) continue;
      if (y < pts.begin()->first) {
        if (key == "RF_tot_constrain") v[(size_t)(y - start)] = pts.begin()->second;
        continue;
      }
      auto hi = pts.upper_bound(y);
      auto lo = std::prev(hi);
      v[(size_t)(y - start)] = lo->second + ((double)y - lo->first) * (hi->second - lo->second) /
                                               (double)(hi->first - lo->first);
    }
  }
  return v;
}

void Scenario::densify_constraint(const std::string &name) {
  series_[name] = densify_points(con_points_[name], name.substr(name.find('.') + 1), start, end);
}

const std::map<int, double> &Scenario::constraint_points(const std::string &section,
                                                         const std::string &key) const {
  static const std::map<int, double> none;
  auto it = con_points_.find(section + "." + key);
  return it == con_points_.end() ? none : it->second;
}

void Scenario::set_constraint_point(const std::string &section, const std::string &key, int year,
                                    double v) {
  if (year < start || year > end) throw std::runtime_error("date outside startDate..endDate");
  const std::string name = section + "." + key;
  if (std::isnan(v)) con_points_[name].erase(year);
  else con_points_[name][year] = v;
  densify_constraint(name);
}

void Scenario::set_series_value(const std::string &section, const std::string &key, int year,
                                double v) {
  if (is_constraint(key)) { set_constraint_point(section, key, year, v); return; }
  auto it = series_.find(section + "." + key);
  if (it == series_.end()) {
    series_[section + "." + key] = std::vector<double>((size_t)ns(), 0.0);
    it = series_.find(section + "." + key);
  }
  if (year < start || year > end) throw std::runtime_error("date outside startDate..endDate");
  it->second[(size_t)(year - start)] = v;
  for (auto &h : halocarbons)
    if (section == h.name + "_halocarbon" && key == h.name + "_emissions")
      h.emissions[(size_t)(year - start)] = v;
}

namespace {
struct IniKey { const char *component, *key; };
const IniKey kIniKeys[] = {
#include "hx_ini_keys.inc"
};
}  // namespace

// Every key of every section has to be one its component reads: the reference's components throw
// "Unknown variable name while parsing <component>: <key>" from setData() for anything else
// (e.g. src/temperature_component.cpp, src/core.cpp:242-243), and a section that names no
// component fails in Core::getComponentByName.  "enabled" / "output" are the core's, for every
// component (core.cpp:251-262); simpleNbox keys may carry a biome ("<biome>.<key>").
void Scenario::validate_keys() const {
  const std::string suf = "_halocarbon";
  auto check = [&](const std::string &name) {
    const size_t dot = name.find('.');
    if (dot == std::string::npos) return;
    const std::string sec = name.substr(0, dot);
    std::string key = name.substr(dot + 1);
    std::string comp = sec, gas;
    if (sec.size() > suf.size() && !sec.compare(sec.size() - suf.size(), suf.size(), suf)) {
      comp = "*_halocarbon";
      gas = sec.substr(0, sec.size() - suf.size());
    }
    if (comp == "simpleNbox") {
      const size_t d2 = key.find('.');
      if (d2 != std::string::npos) key = key.substr(d2 + 1);
    }
    bool known_component = false;
    for (const IniKey &k : kIniKeys) {
      if (comp != k.component) continue;
      known_component = true;
      std::string want = k.key;
      const size_t g = want.find("<gas>");
      if (g != std::string::npos) want.replace(g, 5, gas);
      if (want == key) return;
    }
    if (!known_component)
      throw std::runtime_error("Component not found: " + sec + " (section of " + source + ")");
    if (sec != "core" && (key == "enabled" || key == "output")) return;
    throw std::runtime_error("Unknown variable name while parsing " + sec + ": " + key);
  };
  for (auto &kv : scalars_) check(kv.first);
  for (auto &kv : series_) check(kv.first);
  for (auto &kv : con_points_) check(kv.first);
}

void Scenario::finish() {
  validate_keys();
  start = (int)scalar("core", "startDate");
  end = (int)scalar("core", "endDate");
  if (end <= start) throw std::runtime_error("scenario: endDate <= startDate");
  // halocarbon sections "<gas>_halocarbon"
  std::map<std::string, Halocarbon> hc;
  std::vector<std::string> order;
  const std::string suf = "_halocarbon";
  auto gas_of = [&](const std::string &k, std::string &gas, std::string &key) {
    size_t dot = k.find('.');
    if (dot == std::string::npos) return false;
    const std::string sec = k.substr(0, dot);
    if (sec.size() <= suf.size() || sec.compare(sec.size() - suf.size(), suf.size(), suf))
      return false;
    gas = sec.substr(0, sec.size() - suf.size());
    key = k.substr(dot + 1);
    return true;
  };
  std::string gas, key;
  for (auto &kv : scalars_) {
    if (!gas_of(kv.first, gas, key)) continue;
    if (!hc.count(gas)) { hc[gas].name = gas; order.push_back(gas); }
    Halocarbon &h = hc[gas];
    const double v = to_double(kv.second, kv.first);
    if (key == "tau") h.tau = v;
    else if (key == "rho_" + gas) h.rho = v;
    else if (key == "delta_" + gas) h.delta = v;
    else if (key == "H0") h.H0 = v;
    else if (key == "molarMass") h.molarMass = v;
  }
  for (auto &kv : series_) {
    if (!gas_of(kv.first, gas, key)) continue;
    if (key != gas + "_emissions") continue;
    if (!hc.count(gas)) { hc[gas].name = gas; order.push_back(gas); }
    hc[gas].emissions = kv.second;
  }
  // dense constraint series (packs): the given dates are the non-NaN entries
  for (auto &kv : series_) {
    const std::string key = kv.first.substr(kv.first.find('.') + 1);
    if (!is_constraint(key) || con_points_.count(kv.first)) continue;
    std::map<int, double> &pts = con_points_[kv.first];
    for (size_t i = 0; i < kv.second.size(); ++i)
      if (!std::isnan(kv.second[i])) pts[start + (int)i] = kv.second[i];
  }
  halocarbons.clear();
  for (auto &g : order) {
    Halocarbon &h = hc[g];
    if (h.emissions.empty()) h.emissions.assign(ns(), 0.0);
    if (h.tau == 0 || h.molarMass <= 0)
      throw std::runtime_error("halocarbon " + g + ": tau/molarMass missing");
    halocarbons.push_back(h);
  }
}

Scenario Scenario::load_pack(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Input file " + path + " does not exist.");
  Scenario s;
  s.source = path;
  std::string line;
  bool magic = false;
  std::map<std::string, int> first_year;
  while (std::getline(f, line)) {
    std::istringstream is(line);
    std::string kind, sec, key;
    if (!(is >> kind)) continue;
    if (kind == "HXS") { magic = true; continue; }
    if (kind == "scalar") {
      std::string val;
      if (is >> sec >> key >> val) s.scalars_[sec + "." + key] = val;
    } else if (kind == "series") {
      int y0, n;
      if (!(is >> sec >> key >> y0 >> n)) continue;
      if (n <= 0) throw std::runtime_error("series " + sec + "." + key + ": bad length");
      first_year[sec + "." + key] = y0;
      std::vector<double> v((size_t)n);
      std::string tok;
      for (int i = 0; i < n; ++i) {
        if (!(is >> tok)) throw std::runtime_error("short series " + sec + "." + key);
        v[(size_t)i] = std::strtod(tok.c_str(), nullptr);
      }
      s.series_[sec + "." + key] = std::move(v);
    }
  }
  if (!magic) throw std::runtime_error(path + " is not a scenario pack");
  // a pack's series are dense over startDate..endDate: everything downstream indexes them by
  // year - startDate without further checks
  {
    const int y0 = (int)s.scalar("core", "startDate"), y1 = (int)s.scalar("core", "endDate");
    for (auto &kv : s.series_)
      if (first_year[kv.first] != y0 || (int)kv.second.size() != y1 - y0 + 1)
        throw std::runtime_error("series " + kv.first + " of " + path + " does not cover " +
                                 std::to_string(y0) + ".." + std::to_string(y1));
  }
  s.finish();
  return s;
}

Scenario Scenario::load_ini(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Input file " + path + " does not exist.");
  Scenario s;
  s.source = path;
  const std::string dir = dirname_of(path);
  std::map<std::string, Points> pts;
  std::string line, section;
  while (std::getline(f, line)) {
    std::string t = trim(line);
    if (t.empty() || t[0] == ';' || t[0] == '#') continue;
    if (t[0] == '[') {
      size_t e = t.find(']');
      if (e == std::string::npos) throw std::runtime_error("bad section line: " + line);
      section = trim(t.substr(1, e - 1));
      continue;
    }
    bool prev_ws = false;
    for (size_t i = 0; i < t.size(); ++i) {
      if (t[i] == ';' && prev_ws) { t = trim(t.substr(0, i)); break; }
      prev_ws = (t[i] == ' ' || t[i] == '\t');
    }
    size_t eq = t.find_first_of("=:");
    if (eq == std::string::npos) continue;
    std::string name = trim(t.substr(0, eq)), value = trim(t.substr(eq + 1));
    size_t lb = name.find('[');
    if (lb != std::string::npos) {
      size_t rb = name.find(']', lb);
      if (rb == std::string::npos) throw std::runtime_error("bad index in " + name);
      const double year = to_double(name.substr(lb + 1, rb - lb - 1), name);
      pts[section + "." + name.substr(0, lb)][year] = to_double(value, name);
    } else if (value.compare(0, 4, "csv:") == 0) {
      std::string p = value.substr(4);
      if (!file_exists(p)) p = dir + "/" + p;
      for (auto &kv : read_csv_column(p, name)) pts[section + "." + name][kv.first] = kv.second;
    } else {
      s.scalars_[section + "." + name] = value;
    }
  }
  const int y0 = (int)s.scalar("core", "startDate"), y1 = (int)s.scalar("core", "endDate");
  s.start = y0; s.end = y1;
  for (auto &kv : pts) {
    const std::string key = kv.first.substr(kv.first.find('.') + 1);
    if (is_constraint(key)) {
      for (auto &pt : kv.second)
        if (pt.first >= y0 && pt.first <= y1 && pt.first == std::floor(pt.first))
          s.con_points_[kv.first][(int)pt.first] = pt.second;
      s.densify_constraint(kv.first);
      continue;
    }
    std::vector<double> v;
    for (int y = y0; y <= y1; ++y) v.push_back(tseries_get(kv.second, (double)y));
    s.series_[kv.first] = std::move(v);
  }
  s.finish();
  return s;
}

Scenario Scenario::load(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Input file " + path + " does not exist.");
  std::string first;
  std::getline(f, first);
  f.close();
  if (first.compare(0, 3, "HXS") == 0) return load_pack(path);
  return load_ini(path);
}

}  // namespace hx
