// hx_abi.cpp -- the extern "C" boundary declared in include/hector_amd.h.
// Thin: argument checks, exception -> return code + hx_last_error().
#include <cstring>
#include <stdexcept>
#include <mutex>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/hector_amd.h"
#include "ensemble_core.hpp"
#include "hx_fleet.hpp"

hipError_t hx_launch_unit_csys(int n, const double *Tc, const double *carbon, const double *alk,
                               double inv_vol, double *out, hipStream_t st);
hipError_t hx_launch_doeclim_kernel(const double *diff_row, double *ker, int ns, int count,
                                    int stride, hipStream_t st);
int hx_doeclim_kernel_pad();

#ifndef HX_BACKEND_NAME
#define HX_BACKEND_NAME "hip"
#endif

struct hx_core {
  hx::Fleet *core;  // one shard per GPU of the device list (one for hx_newcore)
};

namespace {
thread_local std::string g_err;
int fail(const std::exception &e) { g_err = e.what(); return 1; }
int fail(const char *msg) { g_err = msg; return 1; }
}  // namespace

#define HX_TRY(body)                                                           \
  if (!core || !core->core) return fail("null hx_core handle");                \
  try { body; return 0; } catch (const std::exception &e) { return fail(e); }  \
  catch (...) { return fail("unknown error"); }

// The HIP the library was built with (hipcc's HIP_VERSION_*) and the runtime it finds itself on
// may differ: torch's wheel bundles its own libamdhip64, an R host links the system's.  A pairing
// across a MAJOR version is reported once, on stderr, by the first call of hx_backend() -- which
// every loader makes first -- and in hx_build_info()'s string.
namespace {
std::string build_info_string() {
  char b[384];
#ifndef HX_HOST_EMULATION
  int rt = 0, drv = 0;
  (void)hipRuntimeGetVersion(&rt);
  (void)hipDriverGetVersion(&drv);
  // (HX_BUILD_TAG: which of the Makefile's builds this is -- the product's code generation flags,
  //  or `make safe`'s defaults)
#ifndef HX_BUILD_TAG
#define HX_BUILD_TAG "flags unknown"
#endif
  std::snprintf(b, sizeof b, "built with HIP %d.%d.%d (%s), gfx950, %s; runtime %d, driver %d",
                HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH, __VERSION__, HX_BUILD_TAG, rt, drv);
#else
  std::snprintf(b, sizeof b, "host emulation (%s)", __VERSION__);
#endif
  return b;
}
void check_runtime_pairing_once() {
#ifndef HX_HOST_EMULATION
  static std::once_flag once;
  std::call_once(once, [] {
    int rt = 0;
    if (hipRuntimeGetVersion(&rt) != hipSuccess || rt <= 0) return;
    const int rt_major = rt / 10000000;   // HIP_VERSION = major * 10^7 + minor * 10^5 + patch
    if (rt_major != HIP_VERSION_MAJOR)
      std::fprintf(stderr, "hector_amd: WARNING: libhector_amd.so was built with HIP %d.%d.%d and runs on HIP "
                           "runtime %d (major version %d): kernels compiled by one major release are not "
                           "guaranteed to load on another -- rebuild with the matching hipcc (make -C "
                           "hector_amd/csrc)\n",
                   HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt, rt_major);
  });
#endif
}
}  // namespace
extern "C" {

const char *hx_backend(void) { check_runtime_pairing_once(); return HX_BACKEND_NAME; }
const char *hx_build_info(void) {
  static const std::string info = build_info_string();
  return info.c_str();
}
const char *hx_last_error(void) { return g_err.c_str(); }

int hx_newcore(const char *scenario, int n_members, int device, hx_core **out) {
  if (!scenario || !out) return fail("hx_newcore: null argument");
  try {
    hx::Fleet *c = new hx::Fleet(scenario, n_members, &device, 1);
    *out = new hx_core{c};
    return 0;
  } catch (const std::exception &e) { return fail(e); }
  catch (...) { return fail("unknown error"); }
}

int hx_newcore_devices(const char *scenario, int n_members, const int *devices, int n_devices,
                       hx_core **out) {
  if (!scenario || !out || !devices) return fail("hx_newcore_devices: null argument");
  try {
    hx::Fleet *c = new hx::Fleet(scenario, n_members, devices, n_devices);
    *out = new hx_core{c};
    return 0;
  } catch (const std::exception &e) { return fail(e); }
  catch (...) { return fail("unknown error"); }
}
int hx_shards(hx_core *core, int *n_shards, int *devices, int *offsets) {
  HX_TRY({
    const int k = core->core->n_shards();
    if (n_shards) *n_shards = k;
    for (int s = 0; s < k; ++s) {
      if (devices) devices[s] = core->core->shard_device(s);
      if (offsets) offsets[s] = core->core->shard_offset(s);
    }
    if (offsets) offsets[k] = core->core->shard_offset(k);
  })
}
int hx_device_var_shard(hx_core *core, int shard, const char *capability, const double **d_ptr,
                        int *npad) {
  if (!capability || !d_ptr) return fail("hx_device_var_shard: null argument");
  HX_TRY(*d_ptr = core->core->device_var(capability, npad, shard))
}
int hx_stream_shard(hx_core *core, int shard, void **stream) {
  if (!stream) return fail("null argument");
  HX_TRY(if (shard < 0 || shard >= core->core->n_shards()) throw std::runtime_error("bad shard index");
         *stream = (void *)core->core->stream(shard))
}
int hx_comm_unique_id(char *id128) {
  if (!id128) return fail("hx_comm_unique_id: null argument");
  try { hx::Fleet::unique_id(id128); return 0; } catch (const std::exception &e) { return fail(e); }
}
int hx_comm_init_rank(hx_core *core, int n_procs, int proc_rank, const char *id128) {
  HX_TRY(core->core->comm_init_rank(n_procs, proc_rank, id128))
}
int hx_comm_info(hx_core *core, int *world, int *first_rank, const char **backend) {
  HX_TRY(if (world) *world = core->core->comm_world();
         if (first_rank) *first_rank = core->core->comm_first_rank();
         if (backend) *backend = core->core->comm_backend())
}
int hx_ensemble_stats(hx_core *core, int nvars, const char *const *capabilities, int year0,
                      int year1, double *out_host, double *d_out) {
  if (nvars < 1 || !capabilities) return fail("hx_ensemble_stats: bad arguments");
  std::vector<std::string> caps;
  for (int i = 0; i < nvars; ++i) caps.push_back(capabilities[i] ? capabilities[i] : "");
  HX_TRY(core->core->ensemble_stats(caps, year0, year1, out_host, d_out))
}

int hx_shutdown(hx_core *core) {
  if (!core) return fail("null hx_core handle");
  delete core->core;
  core->core = nullptr;
  delete core;
  return 0;
}

int hx_setvar(hx_core *core, const char *capability, const double *values, int nvalues,
              const char *units) {
  if (!capability || !values) return fail("hx_setvar: null argument");
  HX_TRY(core->core->setvar(capability, values, nvalues, units))
}
int hx_getvar(hx_core *core, const char *capability, double *out) {
  if (!capability || !out) return fail("hx_getvar: null argument");
  HX_TRY(core->core->getvar(capability, out))
}
int hx_split_biome_of(hx_core *core, const char *old_biome, int n_biomes,
                       const char *const *names, const double *fveg, const double *fdet,
                       const double *fsoil, const double *fpf, const double *fnpp) {
  if (!old_biome || !names || n_biomes < 1) return fail("hx_split_biome_of: bad arguments");
  std::vector<std::string> nm;
  for (int i = 0; i < n_biomes; ++i) nm.push_back(names[i] ? names[i] : "");
  HX_TRY(core->core->split_biome_of(old_biome, nm, fveg, fdet, fsoil, fpf, fnpp))
}
int hx_create_biome(hx_core *core, const char *biome) {
  if (!biome) return fail("hx_create_biome: null name");
  HX_TRY(core->core->create_biome(biome))
}
int hx_delete_biome(hx_core *core, const char *biome) {
  if (!biome) return fail("hx_delete_biome: null name");
  HX_TRY(core->core->delete_biome(biome))
}
int hx_rename_biome(hx_core *core, const char *oldname, const char *newname) {
  if (!oldname || !newname) return fail("hx_rename_biome: null name");
  HX_TRY(core->core->rename_biome(oldname, newname))
}
int hx_split_biome(hx_core *core, int n_biomes, const char *const *names, const double *fveg,
                   const double *fdet, const double *fsoil, const double *fpf,
                   const double *fnpp) {
  if (!names || n_biomes < 1) return fail("hx_split_biome: bad arguments");
  std::vector<std::string> nm;
  for (int i = 0; i < n_biomes; ++i) nm.push_back(names[i] ? names[i] : "");
  HX_TRY(core->core->split_biome(nm, fveg, fdet, fsoil, fpf, fnpp))
}
int hx_set_outputs(hx_core *core, int nvars, const char *const *capabilities) {
  std::vector<std::string> caps;
  for (int i = 0; i < nvars; ++i) caps.push_back(capabilities[i]);
  HX_TRY(core->core->set_outputs(caps))
}
int hx_output_capabilities(const char *const **names, int *count) {
  if (!names) return fail("null argument");
  *names = hx::EnsembleCore::output_capabilities(count);
  return 0;
}
int hx_set_member_sorting(hx_core *core, int on) { HX_TRY(core->core->set_member_sorting(on != 0)) }
int hx_set_lane_calibration(hx_core *core, int on) { HX_TRY(core->core->set_lane_calibration(on != 0)) }
int hx_lane_order_source(hx_core *core, int *source) {
  if (!source) return fail("null argument");
  HX_TRY(*source = core->core->lane_order_source())
}
int hx_set_cost_model(hx_core *core, int on) { HX_TRY(core->core->set_cost_model(on != 0)) }
int hx_cost_models_export(const char *path, int *count) {
  if (!path) return fail("null argument");
  const int n = hx::hx_cost_models_export_file(path);
  if (n < 0) return fail((std::string("hx_cost_models_export: cannot write ") + path).c_str());
  if (count) *count = n;
  return 0;
}
int hx_cost_models_load(const char *path, int *count) {
  if (!path) return fail("null argument");
  const int n = hx::hx_cost_models_load_file(path);
  if (n < 0) return fail((std::string("hx_cost_models_load: cannot read ") + path).c_str());
  if (count) *count = n;
  return 0;
}
int hx_lanes_calibrated(hx_core *core, int *yes) {
  if (!yes) return fail("null argument");
  HX_TRY(*yes = core->core->lanes_calibrated() ? 1 : 0)
}
// ---- unit vectors (function-level parity tests; SURVEY 8c fixture iv) -------------------
int hx_unit_csys(int device, int n, const double *Tc, const double *carbon, const double *alk,
                 double volume, double *out) {
  if (n <= 0 || !Tc || !carbon || !alk || !out) return fail("hx_unit_csys: bad arguments");
  double *d = nullptr;
  try {
    auto ck = [](hipError_t e) { if (e != hipSuccess) throw std::runtime_error(hipGetErrorString(e)); };
    ck(hipSetDevice(device));
    const size_t nb = sizeof(double) * (size_t)n;
    ck(hipMalloc(&d, nb * 7));
    ck(hipMemcpy(d, Tc, nb, hipMemcpyHostToDevice));
    ck(hipMemcpy(d + n, carbon, nb, hipMemcpyHostToDevice));
    ck(hipMemcpy(d + 2 * (size_t)n, alk, nb, hipMemcpyHostToDevice));
    ck(hx_launch_unit_csys(n, d, d + n, d + 2 * (size_t)n, 1.0 / volume, d + 3 * (size_t)n, nullptr));
    ck(hipStreamSynchronize(nullptr));
    ck(hipMemcpy(out, d + 3 * (size_t)n, nb * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
  } catch (const std::exception &e) { if (d) (void)hipFree(d); return fail(e); }
}
int hx_unit_doeclim_kernel(int device, double diff, int ns, double *out) {
  if (ns <= 1 || !out) return fail("hx_unit_doeclim_kernel: bad arguments");
  double *d = nullptr;
  try {
    auto ck = [](hipError_t e) { if (e != hipSuccess) throw std::runtime_error(hipGetErrorString(e)); };
    ck(hipSetDevice(device));
    const int pad = hx_doeclim_kernel_pad();
    ck(hipMalloc(&d, sizeof(double) * ((size_t)ns + 2 * pad + 1)));
    ck(hipMemcpy(d, &diff, sizeof(double), hipMemcpyHostToDevice));
    ck(hx_launch_doeclim_kernel(d, d + 1, ns, 1, 1, nullptr));
    ck(hipStreamSynchronize(nullptr));
    ck(hipMemcpy(out, d + 1 + pad, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
  } catch (const std::exception &e) { if (d) (void)hipFree(d); return fail(e); }
}

int hx_halocarbons(hx_core *core, const char *const **names, int *count) {
  static thread_local std::vector<const char *> ptrs;
  HX_TRY({
    const auto &h = core->core->halocarbon_names();
    ptrs.clear();
    for (auto &n : h) ptrs.push_back(n.c_str());
    if (names) *names = ptrs.data();
    if (count) *count = (int)ptrs.size();
  })
}
int hx_biomes(hx_core *core, const char *const **names, int *count) {
  static thread_local std::vector<const char *> ptrs;
  HX_TRY({
    const auto &b = core->core->biomes();
    ptrs.clear();
    for (auto &n : b) ptrs.push_back(n.c_str());
    if (names) *names = ptrs.data();
    if (count) *count = (int)ptrs.size();
  })
}
int hx_var_info(hx_core *core, const char *capability, const char **component, const char **units) {
  static thread_local std::string comp, un;
  if (!capability) return fail("hx_var_info: null argument");
  HX_TRY({
    core->core->var_info(capability, &comp, &un);
    if (component) *component = comp.c_str();
    if (units) *units = un.c_str();
  })
}
int hx_tracking_pools(hx_core *core, const char *const **names, int *count) {
  static thread_local std::vector<std::string> store;
  static thread_local std::vector<const char *> ptrs;
  HX_TRY({
    store = core->core->tracking_pools();
    ptrs.clear();
    for (auto &n : store) ptrs.push_back(n.c_str());
    if (names) *names = ptrs.data();
    if (count) *count = (int)ptrs.size();
  })
}
int hx_tracking_data(hx_core *core, int member, int year0, int year1, double *values,
                     double *fractions, unsigned long long *source_masks) {
  if (!values || !fractions) return fail("hx_tracking_data: null argument");
  HX_TRY(core->core->tracking_data(member, year0, year1, values, fractions, source_masks))
}
int hx_run_name(hx_core *core, const char **name) {
  static thread_local std::string rn;
  HX_TRY({ rn = core->core->run_name(); if (name) *name = rn.c_str(); })
}
int hx_enable_history(hx_core *core, int on) { HX_TRY(core->core->enable_history(on != 0)) }
int hx_enable_spinup_record(hx_core *core, int on) { HX_TRY(core->core->enable_spinup_record(on != 0)) }
int hx_spinup_record(hx_core *core, int member, const char *const **names, int *nvars, double *values,
                     int max_steps, int *steps) {
  static const std::vector<const char *> cnames = [] {
    std::vector<const char *> v;
    for (const std::string &s : hx::EnsembleCore::spinup_record_vars()) v.push_back(s.c_str());
    return v;
  }();
  if (names) *names = cnames.data();
  if (nvars) *nvars = (int)cnames.size();
  if (!values) return 0;  // (the variable list alone)
  if (!steps) return fail("hx_spinup_record: null argument");
  HX_TRY(*steps = core->core->spinup_record(member, values, max_steps))
}
int hx_setvar_dated(hx_core *core, const char *capability, const int *years, const double *values,
                    int n, const char *units) {
  if (!capability || !years || !values || n < 1) return fail("hx_setvar_dated: bad arguments");
  HX_TRY(core->core->setvar_dated(capability, years, values, n, units))
}
int hx_setvar_dated_members(hx_core *core, const char *capability, const int *years,
                            const double *values, int nyears, const char *units) {
  if (!capability || !years || !values || nyears < 1)
    return fail("hx_setvar_dated_members: bad arguments");
  HX_TRY(core->core->setvar_dated_members(capability, years, values, nyears, units))
}
int hx_lane_of_member(hx_core *core, int *out) {
  if (!out) return fail("null argument");
  HX_TRY(core->core->lane_of_member(out))
}
int hx_reset(hx_core *core, double date) { HX_TRY(core->core->reset(date)) }
int hx_run(hx_core *core, double runtodate) { HX_TRY(core->core->run(runtodate)) }
int hx_sync(hx_core *core) { HX_TRY(core->core->sync()) }
int hx_fetchvars(hx_core *core, const char *capability, int year0, int year1, double *out) {
  if (!capability || !out) return fail("hx_fetchvars: null argument");
  HX_TRY(core->core->fetchvars(capability, year0, year1, out))
}
int hx_device_var(hx_core *core, const char *capability, const double **d_ptr, int *npad) {
  if (!capability || !d_ptr) return fail("hx_device_var: null argument");
  HX_TRY(*d_ptr = core->core->device_var(capability, npad))
}
int hx_stats_device(hx_core *core, const char *capability, int year0, int year1,
                    double *d_stats) {
  if (!capability || !d_stats) return fail("hx_stats_device: null argument");
  HX_TRY(core->core->stats_device(capability, year0, year1, d_stats))
}
int hx_status(hx_core *core, unsigned *out) {
  if (!out) return fail("hx_status: null argument");
  HX_TRY(core->core->status(out))
}
int hx_spinup_steps(hx_core *core, int member, int *steps) {
  if (!steps) return fail("null argument");
  HX_TRY(*steps = core->core->spinup_steps(member))
}
int hx_state_row(hx_core *core, int row, double *out) {
  if (!out) return fail("null argument");
  HX_TRY(core->core->state_row(row, out))
}
int hx_dates(hx_core *core, int *start, int *end, int *current) {
  HX_TRY(if (start) *start = core->core->start_date(); if (end) *end = core->core->end_date();
         if (current) *current = core->core->last_date())
}
int hx_sizes(hx_core *core, int *n_members, int *n_biomes) {
  HX_TRY(if (n_members) *n_members = core->core->n_members();
         if (n_biomes) *n_biomes = core->core->n_biomes())
}
int hx_last_run_ms(hx_core *core, double *ms) {
  HX_TRY(*ms = core->core->last_run_kernel_ms())
}
int hx_last_spinup_ms(hx_core *core, double *ms) { HX_TRY(*ms = core->core->last_spinup_ms()) }
int hx_stream(hx_core *core, void **stream) { HX_TRY(*stream = (void *)core->core->stream()) }
int hx_component_output(hx_core *core, const char *component, int *enabled) {
  HX_TRY(*enabled = core->core->component_output_enabled(component) ? 1 : 0)
}
int hx_set_pair_kernel_limit(hx_core *core, int max_members) {
  HX_TRY(core->core->set_pair_kernel_limit(max_members))
}
int hx_set_prewarm(hx_core *core, int ms) { HX_TRY(core->core->set_prewarm(ms)) }
int hx_last_run_prewarmed(hx_core *core, int *yes) {
  if (!yes) return fail("null argument");
  HX_TRY(*yes = core->core->last_run_prewarmed() ? 1 : 0)
}
int hx_set_two_wave_from(hx_core *core, int min_members) {
  HX_TRY(core->core->set_two_wave_from(min_members))
}
int hx_wave_clock(hx_core *core, int shard, long long *ticks, int cap, int *n_waves) {
  if (!ticks || !n_waves || cap < 0) return fail("hx_wave_clock: bad argument");
  HX_TRY(if (shard < 0 || shard >= core->core->n_shards()) throw std::runtime_error("bad shard index");
         *n_waves = core->core->wave_clock(shard, ticks, cap))
}
int hx_last_run_kernel(hx_core *core, const char **name) { HX_TRY(*name = core->core->last_run_kernel()) }
int hx_last_run_variant(hx_core *core, int *variant) { HX_TRY(*variant = core->core->last_run_variant()) }

}  // extern "C"
