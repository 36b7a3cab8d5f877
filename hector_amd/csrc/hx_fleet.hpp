// hx_fleet.hpp -- one ensemble handle over a LIST of GPUs ("an opaque ensemble
// handle created from ... + device list"; 8e: contiguous member blocks, no exchange during
// integration, one collective for the summary statistics).
//
// The reference keeps any number of independent cores in one process through its registry
// (Core::mkcore / getcore / delcore, inst/include/core.hpp:105-109, src/core.cpp:813-857) and a
// host loops over them; a Fleet is that registry for an ensemble: shard s is an EnsembleCore on
// devices[s] holding the contiguous member block [offset(s), offset(s+1)), every verb of the C
// ABI is routed to the shards (parameters sliced, results concatenated in member order), run()
// queues every device's kernels without waiting, and ensemble_stats() is the ONE collective:
// every shard reduces its block on its GPU (hx_stats_kernel), one ncclAllGather over RCCL hands
// every rank all blocks, and each rank combines them in rank order (bit-identical everywhere).
//
// The communicator spans n_procs x n_shards ranks: a single process that owns 8 GPUs
// (hx_newcore_devices: created on first use, like ncclCommInitAll) or one process per GPU that
// joined with a shared id (hx_comm_init_rank), or any mix.  RCCL is loaded at first use
// (dlopen of librccl.so.1: the copy already in the process if there is one, e.g. PyTorch's), so
// single-GPU hosts never touch it.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "ensemble_core.hpp"

namespace hx {

class Fleet {
 public:
  Fleet(const std::string &scenario, int n_members, const int *devices, int n_devices);
  ~Fleet();
  Fleet(const Fleet &) = delete;
  Fleet &operator=(const Fleet &) = delete;

  int n_shards() const { return (int)shards_.size(); }
  int shard_device(int s) const { return shards_[(size_t)s].device; }
  int shard_offset(int s) const { return s == n_shards() ? n_ : shards_[(size_t)s].offset; }
  EnsembleCore &shard(int s);  // makes its device current
  int shard_of_member(int member) const;

  // ---- the EnsembleCore surface, routed -------------------------------------------------
  int n_members() const { return n_; }
  int n_biomes() const { return shards_[0].core->n_biomes(); }
  int start_date() const { return shards_[0].core->start_date(); }
  int end_date() const { return shards_[0].core->end_date(); }
  int last_date() const { return shards_[0].core->last_date(); }
  const std::vector<std::string> &biomes() const { return shards_[0].core->biomes(); }
  const std::vector<std::string> &halocarbon_names() const { return shards_[0].core->halocarbon_names(); }
  std::string run_name() const { return shards_[0].core->run_name(); }
  void var_info(const std::string &c, std::string *comp, std::string *units) const {
    shards_[0].core->var_info(c, comp, units);
  }
  bool component_output_enabled(const std::string &s) const { return shards_[0].core->component_output_enabled(s); }
  const char *last_run_kernel() const { return shards_[0].core->last_run_kernel(); }
  int last_run_variant() const { return shards_[0].core->last_run_variant(); }
  std::vector<std::string> tracking_pools() const { return shards_[0].core->tracking_pools(); }

  void setvar(const std::string &capability, const double *values, int nvalues, const char *units);
  void getvar(const std::string &capability, double *out);
  void split_biome(const std::vector<std::string> &names, const double *fveg, const double *fdet,
                   const double *fsoil, const double *fpf, const double *fnpp);
  void split_biome_of(const std::string &old_biome, const std::vector<std::string> &names,
                      const double *fveg, const double *fdet, const double *fsoil, const double *fpf,
                      const double *fnpp);
  void create_biome(const std::string &biome);
  void delete_biome(const std::string &biome);
  void rename_biome(const std::string &oldname, const std::string &newname);
  void set_outputs(const std::vector<std::string> &capabilities);
  void set_member_sorting(bool on);
  void set_lane_calibration(bool on);
  bool lanes_calibrated() const;
  int lane_order_source() const { return shards_[0].core->lane_order_source(); }
  void set_cost_model(bool on);
  void enable_history(bool on);
  void enable_spinup_record(bool on);
  int spinup_record(int member, double *values, int max_steps);
  void setvar_dated(const std::string &capability, const int *years, const double *values, int n,
                    const char *units);
  void setvar_dated_members(const std::string &capability, const int *years, const double *values,
                            int nyears, const char *units);
  void lane_of_member(int *out);  // lane inside the member's shard
  void reset(double date);
  void run(double runtodate);  // queues every device, waits for none
  void sync();
  void fetchvars(const std::string &capability, int year0, int year1, double *out_host);
  const double *device_var(const std::string &capability, int *npad, int shard_index = -1);
  void stats_device(const std::string &capability, int year0, int year1, double *d_stats);
  void status(unsigned *out_host);
  void state_row(int row, double *out_host);
  int spinup_steps(int member);
  void tracking_data(int member, int year0, int year1, double *values, double *fractions,
                     unsigned long long *source_masks);
  double last_run_kernel_ms();  // slowest shard
  int wave_clock(int shard_index, long long *ticks, int cap);
  double last_spinup_ms();
  hipStream_t stream(int shard_index = 0) const { return shards_[(size_t)shard_index].core->stream(); }
  void set_pair_kernel_limit(int max_members);
  void set_two_wave_from(int min_members);
  void set_prewarm(int ms);
  bool last_run_prewarmed() const;

  // ---- the collective -----------------------------------------------------------------------
  // Join a communicator of n_procs * n_shards() ranks; this process's shards are ranks
  // proc_rank * n_shards() + s.  id: NCCL_UNIQUE_ID_BYTES from unique_id() of ONE process.
  void comm_init_rank(int n_procs, int proc_rank, const char *id);
  static void unique_id(char *id_out /* 128 bytes */);
  int comm_world() const { return comm_ready_ ? world_ : 0; }  // 0: no communicator yet
  int comm_first_rank() const { return first_rank_; }
  const char *comm_backend() const;
  // Per-year {count, sum, sum of squares, min, max} of nvars outputs over EVERY member of every
  // rank: [nvars][year1-year0+1][5] into out_host and / or the device buffer d_out (memory of
  // shard 0's device), on every rank.  One all-gather; a one-shard core without a communicator
  // does no collective.
  void ensemble_stats(const std::vector<std::string> &capabilities, int year0, int year1,
                      double *out_host, double *d_out);

 private:
  struct Shard {
    std::unique_ptr<EnsembleCore> core;
    int device = 0, offset = 0, count = 0;
    double *d_local = nullptr, *d_slots = nullptr, *d_result = nullptr;  // statistics staging
    void *comm = nullptr;                                                // ncclComm_t
  };
  void use(const Shard &s) const;
  template <class F> void each_parallel(F &&fn);  // fn(shard) on every shard, one host thread each
  void ensure_comm();
  void ensure_stats_buffers(size_t block_doubles);
  void free_stats_buffers();
  std::vector<Shard> shards_;
  int n_ = 0;
  bool duplicates_ = false;  // a device appears twice: rehearsal on a smaller box, copies instead of RCCL
  bool comm_ready_ = false;
  int world_ = 1, first_rank_ = 0;
  size_t stats_cap_ = 0;  // doubles per block the staging buffers hold
  // A routed call that failed on some shards after it went through on others has left the
  // shards with different biome lists, parameters or dates: from then on every call fails with
  // this message (which shard, which call, why) instead of mixing them silently.
  std::string poisoned_;
  void check_poison() const;
  void poison(size_t shard, const char *call, const char *why);
};

}  // namespace hx
