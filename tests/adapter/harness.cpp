// TEST INFRASTRUCTURE: drives include/hector_amd_component.hpp the way the reference's Core
// drives an IModelComponent (init, setData, prepareToRun, one run() per model year, GETDATA
// messages, reset), through a minimal Core-shaped set of types -- the reference's own headers
// need Boost, which this image lacks.  Prints "year variable value units" lines.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>

#include "hector_amd_component.hpp"

namespace host {  // shaped like inst/include/{imodel_component,message_data,unitval,core}.hpp
struct unitval {
  double v = 0; std::string u;
  unitval() {}
  unitval(double val, const std::string &units) : v(val), u(units) {}
  double value() const { return v; }
  std::string unitsName() const { return u; }
};
struct Core {
  static double undefinedIndex() { return -1.0; }
  std::map<std::string, std::string> caps;
  void registerCapability(const std::string &name, const std::string &comp) { caps[name] = comp; }
};
struct message_data {
  // (the reference's constructors, inst/include/message_data.hpp:31-46, signature for signature:
  // tests/test_adapter_interface_pin.py holds them against that header)
  message_data() : date(Core::undefinedIndex()), isVal(false) {}
  message_data(const double d) : date(d), isVal(false) {}
  message_data(const std::string &value) : date(Core::undefinedIndex()), value_str(value), isVal(false) {}
  message_data(const unitval &value) : date(Core::undefinedIndex()), value_unitval(value), isVal(true) {}
  message_data(double d, const unitval &val) : date(d), value_unitval(val), isVal(true) {}
  double date; std::string value_str; unitval value_unitval; bool isVal;
};
struct AVisitor {};
struct IModelComponent {
  virtual ~IModelComponent() {}
  virtual std::string getComponentName() const = 0;
  virtual void init(Core *core) = 0;
  virtual unitval sendMessage(const std::string &message,
                              const std::string &datum,
                              const message_data info = message_data()) = 0;
  virtual void setData(const std::string &varName,
                       const message_data &data) = 0;
  virtual void prepareToRun() = 0;
  virtual void run(const double runToDate) = 0;
  virtual bool run_spinup(const int step) { return true; }
  virtual void reset(double time) = 0;
  virtual void shutDown() = 0;
  virtual void accept(AVisitor *visitor) = 0;
 private:
  virtual unitval getData(const std::string &varName, const double date) = 0;
};
struct Traits {
  using Component = IModelComponent; using Core = host::Core; using unitval = host::unitval;
  using message_data = host::message_data; using Visitor = AVisitor;
  static double undefined_index() { return Core::undefinedIndex(); }
  static double value_of(const message_data &d) { return d.isVal ? d.value_unitval.v : std::atof(d.value_str.c_str()); }
  static unitval make(double v, const char *units) { unitval x; x.v = v; x.u = units ? units : ""; return x; }
  static void fail(const std::string &m) { throw std::runtime_error(m); }
};
}  // namespace host

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  try {
    host::Core core;
    hector_amd::BlockComponentT<host::Traits> comp(argv[1], 1, 0, 0);
    host::IModelComponent &c = comp;                       // everything through the interface
    c.init(&core);
    if (core.caps.count("CO2_concentration") == 0 || core.caps["global_tas"] != "hector-amd") return 3;
    comp.recordVariables({"CO2_concentration", "global_tas", "RF_tot"});
    c.setData("S", host::message_data(host::Core::undefinedIndex(), host::unitval(3.5, "degC")));
    for (int y = 2030; y <= 2060; ++y) c.setData("ffi_emissions", host::message_data((double)y, host::unitval(4.0, "Pg C/yr")));
    c.prepareToRun();
    for (int y = 1746; y <= 2100; ++y) {                   // Core::run, src/core.cpp:483-504
      c.run((double)y);
      if (y % 50 == 0 || y == 2100) {
        for (const char *v : {"CO2_concentration", "global_tas", "RF_tot"}) {
          host::unitval x = c.sendMessage("getData", v, host::message_data((double)y));
          std::printf("%d %s %.17g %s\n", y, v, x.value(), x.unitsName().c_str());
        }
      }
    }
    host::unitval s = c.sendMessage("getData", "S");       // undated GETDATA of a parameter
    std::printf("0 S %.17g %s\n", s.value(), s.unitsName().c_str());
    c.reset(1745.0);                                       // Core::reset + rerun
    for (int y = 1746; y <= 1800; ++y) c.run((double)y);
    host::unitval x = c.sendMessage("getData", "CO2_concentration", host::message_data(1800.0));
    std::printf("1800 rerun_CO2_concentration %.17g %s\n", x.value(), x.unitsName().c_str());
    bool threw = false;
    try { c.sendMessage("getData", "no_such_variable", host::message_data(1800.0)); }
    catch (const std::runtime_error &) { threw = true; }
    std::printf("0 unknown_variable_throws %d -\n", threw ? 1 : 0);
    c.shutDown();
  } catch (const std::exception &e) {
    std::fprintf(stderr, "harness: %s\n", e.what());
    return 1;
  }
  return 0;
}
