// hx_fleet.cpp -- see hx_fleet.hpp.
#include "hx_fleet.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <thread>

#ifndef HX_HOST_EMULATION
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is loaded at first use
#endif

hipError_t hx_launch_combine_stats(const double *slots, int world, int rows, double *out,
                                   hipStream_t st);

namespace hx {

namespace {

void hip_ck(hipError_t e, const char *what) {
  if (e != hipSuccess)
    throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

#ifndef HX_HOST_EMULATION
// The RCCL entry points the fleet uses, bound once per process.  The copy that is already
// loaded wins (PyTorch-ROCm brings its own librccl.so whose SONAME is librccl.so.1, too): two
// RCCLs in one process would each bring their own topology state and kernels.
struct Rccl {
  void *handle = nullptr;
  std::string origin;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

Rccl &rccl() {
  static Rccl r;
  if (r.handle) return r;
  std::vector<std::pair<std::string, int>> tries;
  if (const char *p = std::getenv("HECTOR_AMD_RCCL")) tries.push_back({p, RTLD_NOW | RTLD_LOCAL});
  tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
  tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
  tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL});
  tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL});
  std::string errs;
  for (auto &t : tries) {
    void *h = dlopen(t.first.c_str(), t.second);
    if (h) { r.handle = h; r.origin = t.first + ((t.second & RTLD_NOLOAD) ? " (already loaded)" : ""); break; }
    if (!(t.second & RTLD_NOLOAD)) { const char *e = dlerror(); errs += std::string(" [") + (e ? e : "?") + "]"; }
  }
  if (!r.handle)
    throw std::runtime_error("hector_amd: RCCL (librccl.so.1) could not be loaded for the multi-GPU "
                             "collective:" + errs + " -- set HECTOR_AMD_RCCL to its path");
  auto sym = [&](const char *name) {
    void *p = dlsym(r.handle, name);
    if (!p) throw std::runtime_error(std::string("hector_amd: RCCL symbol missing: ") + name);
    return p;
  };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
  return r;
}

void nccl_ck(ncclResult_t e, const char *what) {
  if (e != ncclSuccess)
    throw std::runtime_error(std::string("RCCL error in ") + what + ": " + rccl().GetErrorString(e));
}
// ncclGroupStart ... ncclGroupEnd around the per-shard calls of one process.  The group is ALWAYS
// closed: an exception between the two would otherwise leave RCCL's thread-local group open, and
// the library shares the process's RCCL with the host (PyTorch's copy when there is one) -- its
// next collective would be queued into our dangling group and never run.  end() is the checked
// close of the normal path (the calls' results -- communicator handles, queued kernels -- exist
// only after it); the destructor closes what an unwinding exception left open, result ignored.
struct RcclGroup {
  Rccl &r;
  bool open = false;
  RcclGroup(Rccl &rc, bool grouped) : r(rc) {
    if (grouped) { nccl_ck(r.GroupStart(), "ncclGroupStart"); open = true; }
  }
  void end() {
    if (!open) return;
    open = false;
    nccl_ck(r.GroupEnd(), "ncclGroupEnd");
  }
  ~RcclGroup() { if (open) (void)r.GroupEnd(); }
  RcclGroup(const RcclGroup &) = delete;
  RcclGroup &operator=(const RcclGroup &) = delete;
};
#endif

bool rehearsal_allowed() {
#ifdef HX_HOST_EMULATION
  return true;
#else
  const char *e = std::getenv("HECTOR_AMD_FLEET_REHEARSAL");
  return e && std::atoi(e) != 0;
#endif
}

}  // namespace

Fleet::Fleet(const std::string &scenario, int n_members, const int *devices, int n_devices) : n_(n_members) {
  if (n_devices < 1 || !devices) throw std::runtime_error("hx_newcore_devices: empty device list");
  if (n_members < n_devices)
    throw std::runtime_error("hx_newcore_devices: fewer members than devices");
  for (int a = 0; a < n_devices; ++a)
    for (int b = a + 1; b < n_devices; ++b)
      if (devices[a] == devices[b]) duplicates_ = true;
  if (duplicates_ && !rehearsal_allowed())
    throw std::runtime_error(
        "hx_newcore_devices: a device appears twice in the list.  RCCL needs one rank per GPU; for a "
        "rehearsal of the sharded path on a box with fewer GPUs set HECTOR_AMD_FLEET_REHEARSAL=1 "
        "(the statistics are then exchanged by device copies, not by RCCL)");
  // contiguous member blocks, the remainder spread over the first shards
  const int base = n_members / n_devices, rem = n_members % n_devices;
  int off = 0;
  shards_.resize((size_t)n_devices);
  for (int s = 0; s < n_devices; ++s) {
    Shard &sh = shards_[(size_t)s];
    sh.device = devices[s];
    sh.offset = off;
    sh.count = base + (s < rem ? 1 : 0);
    off += sh.count;
    sh.core.reset(new EnsembleCore(scenario, sh.count, sh.device));
  }
}

Fleet::~Fleet() {
  free_stats_buffers();
#ifndef HX_HOST_EMULATION
  for (Shard &s : shards_)
    if (s.comm) { (void)hipSetDevice(s.device); (void)rccl().CommDestroy((ncclComm_t)s.comm); s.comm = nullptr; }
#endif
  for (Shard &s : shards_) { (void)hipSetDevice(s.device); s.core.reset(); }
}

void Fleet::use(const Shard &s) const { hip_ck(hipSetDevice(s.device), "hipSetDevice"); }

EnsembleCore &Fleet::shard(int s) {
  if (s < 0 || s >= n_shards()) throw std::runtime_error("bad shard index");
  use(shards_[(size_t)s]);
  return *shards_[(size_t)s].core;
}

int Fleet::shard_of_member(int member) const {
  if (member < 0 || member >= n_) throw std::runtime_error("bad member index");
  for (int s = n_shards() - 1; s >= 0; --s)
    if (member >= shards_[(size_t)s].offset) return s;
  return 0;
}

void Fleet::check_poison() const {
  if (!poisoned_.empty()) throw std::runtime_error(poisoned_);
}
void Fleet::poison(size_t shard, const char *call, const char *why) {
  poisoned_ = "hector_amd: this sharded core is inconsistent and refuses further calls: " +
              std::string(call) + " failed on shard " + std::to_string(shard) + " of " +
              std::to_string(shards_.size()) + " (" + why + ") after it had been applied to the "
              "shards before it; shut the core down and create it again";
}

// a routed call, shard after shard; a failure after the first shard leaves the shards different
#define HX_EACH(call)                                                              \
  {                                                                                \
    check_poison();                                                                \
    size_t k_ = 0;                                                                 \
    try {                                                                          \
      for (; k_ < shards_.size(); ++k_) { Shard &s = shards_[k_]; use(s); s.core->call; } \
    } catch (const std::exception &e_) {                                           \
      if (k_ > 0) poison(k_, #call, e_.what());                                    \
      throw;                                                                       \
    }                                                                              \
  }

// The verbs that do real host work or wait for a device -- run() (lane order, parameter upload,
// spinup), reset(), sync(), fetchvars() -- take the shards side by side, one host thread per
// shard: done one after the other, GPU k would start k x (host preparation + spinup) late, and
// eight device-to-host copies would share one PCIe link's worth of time.  The first error in shard
// order is the call's error.  (The host-emulation build runs a "kernel" on the calling thread with
// its LDS in globals: shards stay sequential there.)
template <class F>
void Fleet::each_parallel(F &&fn) {
  check_poison();
#ifndef HX_HOST_EMULATION
  if (shards_.size() > 1 && !std::getenv("HECTOR_AMD_FLEET_SEQUENTIAL")) {
    std::vector<std::exception_ptr> err(shards_.size());
    std::vector<std::thread> th;
    auto body = [&](size_t k) {
      try { use(shards_[k]); fn(shards_[k]); } catch (...) { err[k] = std::current_exception(); }
    };
    size_t started = 1;
    try {
      for (; started < shards_.size(); ++started) th.emplace_back(body, started);
    } catch (...) {  // no more threads to be had: the remaining shards on this one
    }
    body(0);
    for (size_t k = started; k < shards_.size(); ++k) body(k);
    for (std::thread &t : th) t.join();
    size_t failed = 0, first = shards_.size();
    for (size_t k = 0; k < err.size(); ++k) if (err[k]) { ++failed; if (first == shards_.size()) first = k; }
    if (failed) {
      if (failed < shards_.size()) {   // some shards went through, some did not
        try { std::rethrow_exception(err[first]); }
        catch (const std::exception &e) { poison(first, "a parallel call (run / reset / sync / fetchvars)", e.what()); }
        catch (...) { poison(first, "a parallel call", "unknown error"); }
      }
      std::rethrow_exception(err[first]);
    }
    return;
  }
#endif
  {
    size_t k = 0;
    try {
      for (; k < shards_.size(); ++k) { use(shards_[k]); fn(shards_[k]); }
    } catch (const std::exception &e) {
      if (k > 0) poison(k, "a sharded call (run / reset / sync / fetchvars)", e.what());
      throw;
    }
  }
}

void Fleet::setvar(const std::string &cap, const double *values, int nvalues, const char *units) {
  if (shards_.size() == 1) { shards_[0].core->setvar(cap, values, nvalues, units); return; }
  if (nvalues != 1 && nvalues != n_)
    throw std::runtime_error("setvar: expected 1 or n_members values");
  check_poison();
  size_t k = 0;
  try {
    for (; k < shards_.size(); ++k) {
      Shard &s = shards_[k];
      use(s);
      if (nvalues == 1) s.core->setvar(cap, values, 1, units);
      else s.core->setvar(cap, values + s.offset, s.count, units);
    }
  } catch (const std::exception &e) {
    if (k > 0) poison(k, "setvar", e.what());
    throw;
  }
}
void Fleet::getvar(const std::string &cap, double *out) {
  for (Shard &s : shards_) { use(s); s.core->getvar(cap, out + s.offset); }
}
void Fleet::split_biome(const std::vector<std::string> &names, const double *fveg, const double *fdet,
                        const double *fsoil, const double *fpf, const double *fnpp) {
  HX_EACH(split_biome(names, fveg, fdet, fsoil, fpf, fnpp))
}
void Fleet::split_biome_of(const std::string &old_biome, const std::vector<std::string> &names,
                           const double *fveg, const double *fdet, const double *fsoil,
                           const double *fpf, const double *fnpp) {
  HX_EACH(split_biome_of(old_biome, names, fveg, fdet, fsoil, fpf, fnpp))
}
void Fleet::create_biome(const std::string &b) { HX_EACH(create_biome(b)) }
void Fleet::delete_biome(const std::string &b) { HX_EACH(delete_biome(b)) }
void Fleet::rename_biome(const std::string &a, const std::string &b) { HX_EACH(rename_biome(a, b)) }
void Fleet::set_outputs(const std::vector<std::string> &caps) { HX_EACH(set_outputs(caps)) }
void Fleet::set_member_sorting(bool on) { HX_EACH(set_member_sorting(on)) }
void Fleet::set_lane_calibration(bool on) { HX_EACH(set_lane_calibration(on)) }
void Fleet::set_cost_model(bool on) { HX_EACH(set_cost_model(on)) }
bool Fleet::lanes_calibrated() const {
  for (const Shard &s : shards_) if (!s.core->lanes_calibrated()) return false;
  return true;
}
void Fleet::enable_history(bool on) { HX_EACH(enable_history(on)) }
void Fleet::enable_spinup_record(bool on) { HX_EACH(enable_spinup_record(on)) }
int Fleet::spinup_record(int member, double *values, int max_steps) {
  Shard &s = shards_[(size_t)shard_of_member(member)];
  use(s);
  return s.core->spinup_record(member - s.offset, values, max_steps);
}
void Fleet::set_pair_kernel_limit(int m) { HX_EACH(set_pair_kernel_limit(m)) }
void Fleet::set_prewarm(int ms) { HX_EACH(set_prewarm(ms)) }
bool Fleet::last_run_prewarmed() const { return shards_.front().core->last_run_prewarmed(); }
void Fleet::set_two_wave_from(int m) { HX_EACH(set_two_wave_from(m)) }   // (members of a shard)
void Fleet::setvar_dated(const std::string &cap, const int *years, const double *values, int n,
                         const char *units) {
  HX_EACH(setvar_dated(cap, years, values, n, units))
}
void Fleet::setvar_dated_members(const std::string &cap, const int *years, const double *values,
                                 int nyears, const char *units) {
  if (shards_.size() == 1) { shards_[0].core->setvar_dated_members(cap, years, values, nyears, units); return; }
  std::vector<double> part;
  for (Shard &s : shards_) {  // values[i * n_members + member] -> this shard's [i][count]
    use(s);
    part.resize((size_t)nyears * (size_t)s.count);
    for (int i = 0; i < nyears; ++i)
      std::memcpy(part.data() + (size_t)i * s.count, values + (size_t)i * n_ + s.offset,
                  sizeof(double) * (size_t)s.count);
    s.core->setvar_dated_members(cap, years, part.data(), nyears, units);
  }
}
void Fleet::lane_of_member(int *out) {
  for (Shard &s : shards_) { use(s); s.core->lane_of_member(out + s.offset); }
}
void Fleet::reset(double date) { each_parallel([&](Shard &s) { s.core->reset(date); }); }
void Fleet::run(double runtodate) { each_parallel([&](Shard &s) { s.core->run(runtodate); }); }
void Fleet::sync() { each_parallel([&](Shard &s) { s.core->sync(); }); }

void Fleet::fetchvars(const std::string &cap, int year0, int year1, double *out_host) {
  if (shards_.size() == 1) { shards_[0].core->fetchvars(cap, year0, year1, out_host); return; }
  const int ny = year1 - year0 + 1;
  if (ny < 1) throw std::runtime_error("fetchvars: year1 < year0");
  // every shard copies its block of members straight into its columns of out_host
  each_parallel([&](Shard &s) { s.core->fetchvars(cap, year0, year1, out_host + s.offset, (size_t)n_); });
}

const double *Fleet::device_var(const std::string &cap, int *npad, int shard_index) {
  if (shard_index < 0) {
    if (shards_.size() > 1)
      throw std::runtime_error("hx_device_var: this core spans several GPUs -- ask for one shard's "
                               "array with hx_device_var_shard");
    shard_index = 0;
  }
  return shard(shard_index).device_var(cap, npad);
}

void Fleet::stats_device(const std::string &cap, int year0, int year1, double *d_stats) {
  if (shards_.size() == 1 && !comm_ready_) { shards_[0].core->stats_device(cap, year0, year1, d_stats); return; }
  ensemble_stats({cap}, year0, year1, nullptr, d_stats);
}

void Fleet::status(unsigned *out) { for (Shard &s : shards_) { use(s); s.core->status(out + s.offset); } }
void Fleet::state_row(int row, double *out) {
  for (Shard &s : shards_) { use(s); s.core->state_row(row, out + s.offset); }
}
int Fleet::spinup_steps(int member) {
  Shard &s = shards_[(size_t)shard_of_member(member)];
  use(s);
  return s.core->spinup_steps(member - s.offset);
}
void Fleet::tracking_data(int member, int year0, int year1, double *values, double *fractions,
                          unsigned long long *source_masks) {
  Shard &s = shards_[(size_t)shard_of_member(member)];
  use(s);
  s.core->tracking_data(member - s.offset, year0, year1, values, fractions, source_masks);
}
int Fleet::wave_clock(int shard_index, long long *ticks, int cap) {
  Shard &s = shards_[(size_t)shard_index];
  use(s);
  return s.core->wave_clock(ticks, cap);
}
double Fleet::last_run_kernel_ms() {
  double ms = 0;
  for (Shard &s : shards_) { use(s); s.core->sync(); ms = std::max(ms, s.core->last_run_kernel_ms()); }
  return ms;
}
double Fleet::last_spinup_ms() {
  double ms = 0;
  for (Shard &s : shards_) ms = std::max(ms, s.core->last_spinup_ms());
  return ms;
}

// ---- the collective ----------------------------------------------------------------------------

void Fleet::unique_id(char *id_out) {
#ifndef HX_HOST_EMULATION
  ncclUniqueId id;
  nccl_ck(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id_out, id.internal, sizeof(id.internal));
#else
  (void)id_out;
  throw std::runtime_error("the host-emulation build has no RCCL");
#endif
}

const char *Fleet::comm_backend() const {
  if (!comm_ready_) return "none";
  if (duplicates_) return "device-copies (rehearsal: a device is listed twice)";
#ifndef HX_HOST_EMULATION
  static thread_local std::string s;
  int v = 0;
  (void)rccl().GetVersion(&v);
  s = "rccl " + std::to_string(v) + " via " + rccl().origin;
  return s.c_str();
#else
  return "none";
#endif
}

void Fleet::comm_init_rank(int n_procs, int proc_rank, const char *id_bytes) {
  if (comm_ready_) throw std::runtime_error("hx_comm_init_rank: this core already has a communicator");
  if (n_procs < 1 || proc_rank < 0 || proc_rank >= n_procs || !id_bytes)
    throw std::runtime_error("hx_comm_init_rank: bad arguments");
  if (duplicates_) {
    if (n_procs != 1)
      throw std::runtime_error("hx_comm_init_rank: a rehearsal core (duplicate devices) cannot join "
                               "other processes");
    world_ = n_shards(); first_rank_ = 0; comm_ready_ = true;
    return;
  }
#ifndef HX_HOST_EMULATION
  Rccl &r = rccl();
  ncclUniqueId id;
  std::memcpy(id.internal, id_bytes, sizeof(id.internal));
  const int L = n_shards();
  world_ = n_procs * L;
  first_rank_ = proc_rank * L;
  // (inside a group ncclCommInitRank writes the handle when the group ends: the slots it writes to
  // must outlive the calls, hence one array for all shards)
  std::vector<ncclComm_t> comms((size_t)L, nullptr);
  try {
    {
      RcclGroup group(r, L > 1);
      for (int s = 0; s < L; ++s) {
        hip_ck(hipSetDevice(shards_[(size_t)s].device), "hipSetDevice");
        nccl_ck(r.CommInitRank(&comms[(size_t)s], world_, id, first_rank_ + s), "ncclCommInitRank");
      }
      group.end();
    }
    for (int s = 0; s < L; ++s)
      if (!comms[(size_t)s]) throw std::runtime_error("ncclCommInitRank returned no communicator");
    for (int s = 0; s < L; ++s) shards_[(size_t)s].comm = comms[(size_t)s];
    free_stats_buffers();  // sized for the old world
  } catch (...) {
    // communicators that did come to life are torn down, not leaked (abort: their peers may never
    // have arrived, a destroy could wait for them)
    for (int s = 0; s < L; ++s)
      if (comms[(size_t)s]) { (void)hipSetDevice(shards_[(size_t)s].device); (void)r.CommAbort(comms[(size_t)s]); }
    for (Shard &s : shards_) s.comm = nullptr;
    world_ = 1; first_rank_ = 0;
    throw;
  }
  comm_ready_ = true;
#else
  throw std::runtime_error("the host-emulation build has no RCCL");
#endif
}

void Fleet::ensure_comm() {
  if (comm_ready_ || shards_.size() == 1) return;
  if (duplicates_) { world_ = n_shards(); first_rank_ = 0; comm_ready_ = true; return; }
  char id[128];
  unique_id(id);  // one process owns every rank: what ncclCommInitAll does
  comm_init_rank(1, 0, id);
}

void Fleet::free_stats_buffers() {
  for (Shard &s : shards_) {
    if (!s.d_local && !s.d_slots && !s.d_result) continue;
    (void)hipSetDevice(s.device);
    if (s.d_local) (void)hipFree(s.d_local);
    if (s.d_slots) (void)hipFree(s.d_slots);
    if (s.d_result) (void)hipFree(s.d_result);
    s.d_local = s.d_slots = s.d_result = nullptr;
  }
  stats_cap_ = 0;
}

void Fleet::ensure_stats_buffers(size_t blk) {
  if (blk <= stats_cap_) return;
  free_stats_buffers();
  for (Shard &s : shards_) {
    use(s);
    hip_ck(hipMalloc(&s.d_local, sizeof(double) * blk), "hipMalloc stats block");
    hip_ck(hipMalloc(&s.d_slots, sizeof(double) * blk * (size_t)world_), "hipMalloc stats slots");
    hip_ck(hipMalloc(&s.d_result, sizeof(double) * blk), "hipMalloc stats result");
  }
  stats_cap_ = blk;
}

void Fleet::ensemble_stats(const std::vector<std::string> &caps, int year0, int year1,
                           double *out_host, double *d_out) {
  if (caps.empty()) throw std::runtime_error("ensemble_stats: no variables");
  const int ny = year1 - year0 + 1;
  if (ny < 1) throw std::runtime_error("ensemble_stats: year1 < year0");
  ensure_comm();
  const int world = comm_ready_ ? world_ : 1;
  const size_t nv = caps.size(), rows = nv * (size_t)ny, blk = rows * 5;
  ensure_stats_buffers(blk);
  // 1. every shard reduces its own members on its own GPU, on its own stream
  for (Shard &s : shards_) {
    use(s);
    for (size_t v = 0; v < nv; ++v)
      s.core->stats_async(caps[v], year0, year1, s.d_local + v * (size_t)ny * 5);
  }
  // 2. the one collective: every rank receives every rank's block
  if (world == 1) {
    Shard &s = shards_[0];
    hip_ck(hipMemcpyAsync(s.d_slots, s.d_local, sizeof(double) * blk, hipMemcpyDeviceToDevice,
                          s.core->stream()), "stats copy");
  } else if (duplicates_) {
    for (Shard &s : shards_) { use(s); hip_ck(hipStreamSynchronize(s.core->stream()), "stats sync"); }
    for (Shard &dst : shards_) {
      use(dst);
      for (size_t r = 0; r < shards_.size(); ++r)
        hip_ck(hipMemcpyAsync(dst.d_slots + r * blk, shards_[r].d_local, sizeof(double) * blk,
                              hipMemcpyDeviceToDevice, dst.core->stream()), "stats exchange");
    }
  } else {
#ifndef HX_HOST_EMULATION
    Rccl &r = rccl();
    RcclGroup group(r, shards_.size() > 1);
    for (Shard &s : shards_) {
      use(s);
      nccl_ck(r.AllGather(s.d_local, s.d_slots, blk, ncclDouble, (ncclComm_t)s.comm, s.core->stream()),
              "ncclAllGather");
    }
    group.end();
#endif
  }
  // 3. combined in rank order on every rank: bit-identical everywhere
  for (Shard &s : shards_) {
    use(s);
    hip_ck(hx_launch_combine_stats(s.d_slots, world, (int)rows, s.d_result, s.core->stream()),
           "combine statistics");
  }
  Shard &s0 = shards_[0];
  use(s0);
  if (d_out)
    hip_ck(hipMemcpyAsync(d_out, s0.d_result, sizeof(double) * blk, hipMemcpyDeviceToDevice,
                          s0.core->stream()), "stats result");
  if (out_host)
    hip_ck(hipMemcpyAsync(out_host, s0.d_result, sizeof(double) * blk, hipMemcpyDeviceToHost,
                          s0.core->stream()), "stats result to host");
  for (Shard &s : shards_) { use(s); hip_ck(hipStreamSynchronize(s.core->stream()), "stats sync"); }
}

}  // namespace hx
