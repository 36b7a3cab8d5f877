// hector_amd_component.hpp -- an IModelComponent for a reference-shaped Core.
//
// The reference's Core owns components behind `IModelComponent`
// (inst/include/imodel_component.hpp:34-172): init / sendMessage / setData / prepareToRun / run /
// run_spinup / reset / shutDown / accept / getData, and drives them once per model year
// (src/core.cpp:483-504).  A host that keeps that Core -- its message bus, visitors, logging,
// INI reader -- can put the GPU year loop behind the same interface: this adapter stands where
// simpleNbox, ocean, the carbon-cycle solver, temperature, forcing, CH4 / OH / ozone / N2O and the
// halocarbons stood, registers the capabilities they registered, routes setData to
// hx_setvar[_dated], run(runToDate) to hx_run, reset to hx_reset, and answers getData from
// hx_fetchvars with the unit the reference attaches to the variable (hx_var_info).
//
// It is a template over the host's types so that it needs nothing from the reference at compile
// time here (the reference's headers pull in Boost, which this image lacks): with the reference's
// headers on the include path, `HECTOR_AMD_WITH_REFERENCE_HEADERS` defines ReferenceTraits and
// the alias hector_amd::BlockComponent.  tests/adapter/ drives the same template through a
// minimal Core-shaped harness (tests/test_component_adapter.py).
//
//   Traits::Component      the abstract base (Hector::IModelComponent)
//   Traits::Core           needs registerCapability(name, componentName)
//   Traits::unitval        what getData / sendMessage return
//   Traits::message_data   has .date and a numeric value
//   Traits::Visitor        accept() argument (Hector::AVisitor)
//   static double undefined_index();                          // Core::undefinedIndex()
//   static double value_of(const message_data &);             // the number carried by a message
//   static unitval make(double value, const char *units);     // unitval with the named unit
//   static void fail(const std::string &);                    // H_THROW
#ifndef HECTOR_AMD_COMPONENT_HPP
#define HECTOR_AMD_COMPONENT_HPP

#include <cstdio>
#include <string>
#include <vector>

#include "hector_amd.h"

namespace hector_amd {

template <class Traits>
class BlockComponentT : public Traits::Component {
 public:
  using unitval = typename Traits::unitval;
  using message_data = typename Traits::message_data;

  // scenario: the INI file (or scenario pack) the host's Core was configured from; member: which
  // member of an n_members ensemble answers getData (a single-run host uses 1 / 0).
  explicit BlockComponentT(std::string scenario, int n_members = 1, int member = 0, int device = 0)
      : scenario_(std::move(scenario)), n_(n_members), member_(member), device_(device) {}
  ~BlockComponentT() override { shutDown(); }

  std::string getComponentName() const override { return "hector-amd"; }

  // IModelComponent::init: create the ensemble core and register what the replaced components
  // registered (Core::registerCapability, src/core.cpp:640-667)
  void init(typename Traits::Core *core) override {
    core_ = core;
    if (hx_newcore(scenario_.c_str(), n_, device_, &hx_)) Traits::fail(hx_last_error());
    const char *const *names = nullptr;
    int count = 0;
    if (hx_output_capabilities(&names, &count)) Traits::fail(hx_last_error());
    for (int i = 0; i < count; ++i)
      if (names[i] && core_) core_->registerCapability(names[i], getComponentName());
  }

  // sendMessage(M_GETDATA / M_SETDATA, datum, info)  src/core.cpp:716-778
  unitval sendMessage(const std::string &message, const std::string &datum,
                      const message_data info = message_data()) override {
    if (message == "getData") return getData(datum, info.date);
    if (message == "setData") { setData(datum, info); return Traits::make(0.0, ""); }
    Traits::fail("Caller sent unknown message: " + message);
    return Traits::make(0.0, "");
  }

  void setData(const std::string &varName, const message_data &data) override {
    need();
    const double v = Traits::value_of(data);
    if (data.date == Traits::undefined_index()) {
      if (hx_setvar(hx_, varName.c_str(), &v, 1, nullptr)) Traits::fail(hx_last_error());
    } else {
      const int year = (int)data.date;
      if (hx_setvar_dated(hx_, varName.c_str(), &year, &v, 1, nullptr)) Traits::fail(hx_last_error());
    }
  }

  // which variables getData will be asked for (the reference records everything; here recording
  // is opt-in, 8 B per member-year each) -- call before prepareToRun
  void recordVariables(const std::vector<std::string> &vars) {
    need();
    std::vector<const char *> p;
    for (auto &s : vars) p.push_back(s.c_str());
    if (hx_set_outputs(hx_, (int)p.size(), p.data())) Traits::fail(hx_last_error());
  }

  void prepareToRun() override {  // upload + spinup + alkalinity tuning (Core::prepareToRun)
    need();
    std::vector<unsigned> st((size_t)n_);
    if (hx_status(hx_, st.data())) Traits::fail(hx_last_error());
  }

  void run(const double runToDate) override {  // Core::run calls every component once per year
    need();
    if (hx_run(hx_, runToDate) || hx_sync(hx_)) Traits::fail(hx_last_error());
    std::vector<unsigned> st((size_t)n_);
    if (hx_status(hx_, st.data())) Traits::fail(hx_last_error());
    if (st[(size_t)member_]) Traits::fail("hector-amd: model error flags " + std::to_string(st[(size_t)member_]));
  }

  bool run_spinup(const int) override { return true; }  // the ensemble core spins up itself

  void reset(double time) override {
    need();
    if (hx_reset(hx_, time)) Traits::fail(hx_last_error());
  }

  void shutDown() override {
    if (hx_) hx_shutdown(hx_);
    hx_ = nullptr;
  }

  void accept(typename Traits::Visitor *) override {}

  hx_core *handle() { return hx_; }

 private:
  unitval getData(const std::string &varName, const double date) override {
    need();
    int start = 0, end = 0, cur = 0;
    if (hx_dates(hx_, &start, &end, &cur)) Traits::fail(hx_last_error());
    const char *comp = nullptr, *units = nullptr;
    if (hx_var_info(hx_, varName.c_str(), &comp, &units)) Traits::fail(hx_last_error());
    std::vector<double> row((size_t)n_);
    if (date == Traits::undefined_index()) {
      // undated: a parameter (GETDATA without date) or the current value of a state variable
      if (hx_getvar(hx_, varName.c_str(), row.data()) == 0) return Traits::make(row[(size_t)member_], units);
      if (hx_fetchvars(hx_, varName.c_str(), cur, cur, row.data())) Traits::fail(hx_last_error());
    } else {
      const int y = (int)date;
      if (hx_fetchvars(hx_, varName.c_str(), y, y, row.data())) Traits::fail(hx_last_error());
    }
    return Traits::make(row[(size_t)member_], units);
  }
  void need() { if (!hx_) Traits::fail("hector-amd component used before init()"); }

  std::string scenario_;
  int n_, member_, device_;
  hx_core *hx_ = nullptr;
  typename Traits::Core *core_ = nullptr;
};

}  // namespace hector_amd

#ifdef HECTOR_AMD_WITH_REFERENCE_HEADERS
// with -I<hector>/inst/include (and Boost):
#include "core.hpp"
#include "imodel_component.hpp"
namespace hector_amd {
struct ReferenceTraits {
  using Component = Hector::IModelComponent;
  using Core = Hector::Core;
  using unitval = Hector::unitval;
  using message_data = Hector::message_data;
  using Visitor = Hector::AVisitor;
  static double undefined_index() { return Hector::Core::undefinedIndex(); }
  static double value_of(const message_data &d) {
    return d.getUnitval(Hector::U_UNDEFINED).value(Hector::U_UNDEFINED);
  }
  static unitval make(double v, const char *units) {
    char buf[40];
    std::snprintf(buf, sizeof buf, "%.17g", v);  // parse_unitval reads the number back exactly
    return Hector::unitval::parse_unitval(buf, units ? units : "", Hector::U_UNDEFINED);
  }
  static void fail(const std::string &m) { H_THROW(m); }
};
using BlockComponent = BlockComponentT<ReferenceTraits>;
}  // namespace hector_amd
#endif

#endif
