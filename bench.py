#!/usr/bin/env python3
"""bench.py -- ensemble-member simulated years per second, SSP2-4.5 1745-2300.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--gpus N > 1 always drives N GPUs: under torch.distributed.run (WORLD_SIZE set) this process is
one of N ranks; started as plain `python bench.py --gpus N` it launches those N ranks itself
(one process per GPU, rendezvous on 127.0.0.1) and fails if the box has fewer than N devices;
`--single-process` instead shards ONE core over N devices through the C ABI's device list
(hx_newcore_devices).  In every mode the per-year statistics of all members are combined by ONE
RCCL all-gather issued by libhector_amd.so itself (hx_ensemble_stats) on the core's stream.

One "step" = one pass of the hot path over this rank's batch of synthetic
members: reset to the post-spinup state (a device-to-device copy), integrate
1745 -> 2300 (555 model years per member) with the HIP kernels, reduce the
per-year ensemble statistics of CO2 and Tgav on the GPU and (N > 1) all-reduce
them over RCCL.  Inputs (parameters, scenario tables, spun-up state) are
resident in HBM before the timed region; spinup and upload are excluded, as
SURVEY.md 8(d) defines the metric.  Weak scaling: per-GPU members fixed.

The line's `value` is the SAME per-GPU workload at every N -- BASELINE.json configs[2], a
65 536-member perturbed ECS/Q10 ensemble on every MI355X -- so that the values of a
--gpus 1, 2, 4, 8 series form one weak-scaling curve.  BASELINE.json configs[3]'s shape --
131 072 members on every GPU, the named 1 048 576 members at N = 8 -- is timed after it in the
same job (`other_configs`), and both workloads are at the line's top level at every N:
`value_per_gpu_workload` {"65536": ..., "131072": ...} (whole-job member-years/s),
`kernel_ms` and `first_run_kernel_ms` -- the run kernel of a one-shot run (the first run of a
fresh core) next to the steady state of the timed steps.  (--members M: M per GPU at any N,
nothing else.)
A multi-GPU line is only valid if every member of the ensemble is in the gathered statistics,
the collective spans as many ranks as the job has GPUs and it travelled over RCCL: the line says
so (`members_in_statistics`, `collective_world_size`, `collective_backend`) and the process
exits 3 with an `invalid` list in the line otherwise.

Prints ONE JSON line on rank 0 (see the task contract) with three extra objects:
  roofline      -- the BINDING bound: executed fp64 flop/s against the 78.6 TFLOP/s fp64
                   vector peak.  Flops per launch come from the committed SQ-counter profile
                   of this exact kernel source and configuration (profiles/pmc_index.json,
                   written by tools/prof/summarize.py: SQ_INSTS_VALU_FLOPS_FP64 x 64 lanes x
                   lane utilisation); the time is the mean HIP-event duration of the run
                   kernel on the core's stream, measured here.  With it: valu_active_frac
                   (same profile), traffic (HBM bytes per launch from the FETCH_SIZE /
                   WRITE_SIZE passes), hbm_measured_frac = traffic / kernel time / 8 TB/s,
                   design_bytes_per_member_year -- THIS design's own byte model (design_bytes():
                   4 output rows + the SST history once per 32-year block + partial sums,
                   ~128 B) with traffic_over_design_bytes = traffic / it (~1: the kernel moves
                   what it must and little else), and hbm_yardstick -- SURVEY.md 8(d)'s
                   YEAR-STEPPED model (2 680 B per member-year: state round trip + history
                   re-read every year), labelled as not this kernel: the kernel does not move
                   those bytes by design, a frac above 1 there is no evidence of anything.
                   kernel_ms_steps: min / median / max of the timed steps' kernel times.  wave_time: how long the wavefronts of the last timed launch ran
                   (the kernels' own s_memrealtime stamps, hx_wave_clock) -- mean, max and
                   max / mean: with one wavefront per SIMD the launch lasts as long as its
                   costliest wavefront.
  other_configs -- the other single-GPU BASELINE configurations (1 024 members and 32 768, the
                   two-wavefront kernel's range; 131 072
                   members = configs[3]'s per-GPU share; 65 536 members x 4 biomes), 5 steps
                   each, timed the same way after the headline (N = 1 only)
  workload_variants -- the headline ensemble with what users commonly add to it, timed the same
                   way (N = 1 only; not BASELINE configurations): the NPP diagnostic recorded (the
                   plain kernel + diagnostics instantiations), every member its own ocean heat
                   diffusivity (per-member DOECLIM kernel tables: the history contraction on the
                   vector ALU)
  safe_build    -- kernel time of the same sources built WITHOUT the product's code-generation
                   flags (libhector_amd_safe.so) at 65 536 and 131 072 members: what losing them
                   would cost (N = 1 only)
  first_run_*   -- the run kernel of a one-shot run (the first run of a fresh core: lanes by the
                   shipped cost model where the order matters, launched behind the prewarm loop
                   that keeps the chip's clocks up during upload and spinup) next to the steady
                   state, and their ratio
  sustained     -- the headline step 300 times back to back under one clock, OUTSIDE the timed
                   region (N = 1, default workload): ~1.8 s of GPU work an outside utilisation
                   sampler can see, and the throughput held under sustained load
  cpu_baseline  -- the CPU oracle (a scalar C port of the reference loop, pinned to the
                   reference's golden trajectory) on a bounded sample of the same ensemble,
                   all host cores, 555-year loop only
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_MEMBER_YEAR = {1: 2680.0, 4: 3072.0}  # SURVEY.md 8(d) / BASELINE.md 4
# SURVEY.md 8(d), "for context, flops": ~13.8 kflop-equivalent per member-year as the reference
# executes it, ~9.6 k with warm-started Newton iterations (one biome)
SURVEY_FLOPS_PER_MEMBER_YEAR = 9600.0
HBM_PEAK = 8.0e12            # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VECTOR_PEAK = 78.6e12   # fp64 vector: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
YEARS = 555
KERNEL_SOURCES = ["hx_kernels.hip", "hx_dev_chem.h", "hx_dev_const.h", "hx_dev_member.h",
                  "hx_dev_solver.h", "hx_dev_track.h", "hx_dev_math.h", "hx_dev_clock.h", "hx_dev_pair.h",
                  "hx_layout.h", "hx_addrspace.h", "hx_chem_fit.inc", "hx_erfc_fit.inc"]


def kernel_source_hash():
    """sha256 over the device sources: the key of profiles/pmc_index.json, so that counters
    collected on another version of the kernels are never quoted for this one."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "hector_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def compiler_id():
    """The compiler that built (or would build) the kernels: the counters of a profile belong to
    ONE compiler's code -- the register allocation, and with it every figure, moves with it."""
    import subprocess
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True,
                             timeout=60).stdout.splitlines()
        return " | ".join(l.strip() for l in out[:2])
    except Exception:
        return "unknown"


def pmc_entry(members, biomes):
    """-> (entry, stale).  entry: the committed counter figures for THIS kernel source and
    configuration.  If the kernels changed since the last collection: the newest entry of the
    same configuration, flagged stale, and a loud complaint on stderr; (None, True) if there
    is none at all (counters cannot be read from inside the timed run)."""
    cfg = "%dx%d" % (members, biomes)
    try:
        index = json.load(open(os.path.join(ROOT, "profiles", "pmc_index.json")))
    except Exception:
        index = {"entries": {}, "order": []}
    h = kernel_source_hash()
    if cfg in index["entries"].get(h, {}):
        e = index["entries"][h][cfg]
        if e.get("compiler") in (None, compiler_id()):
            return e, False
        sys.stderr.write("bench.py: the PMC profile of kernel source %s, configuration %s was collected "
                         "on a build by another compiler (%s; here %s): flagged stale\n"
                         % (h, cfg, e.get("compiler"), compiler_id()))
        return e, True
    sys.stderr.write("bench.py: NO PMC PROFILE for kernel source %s, configuration %s in "
                     "profiles/pmc_index.json (tools/prof/collect.sh + summarize.py): roofline "
                     "figures that need counters are null or flagged stale\n" % (h, cfg))
    for hh in reversed(index["order"]):  # newest collection first
        if cfg in index["entries"].get(hh, {}):
            return index["entries"][hh][cfg], True
    return None, True


def design_bytes(members, biomes, kernel="run"):
    """HBM bytes per member-year THIS design has to move ("the design's own byte
    model") -- what `traffic` should be compared with, not SURVEY 8(d)'s year-stepped figure:
      writes   4 output rows a year (SST and land temperature: the model's own histories; CO2 and
               Tgav: the north star's outputs) = 32 B, + the block's 32 partial history sums,
               written once per 32-year block = 8 B a year;
      reads    the partial sum of the year (8 B), the land temperature leaving the 200-year Q10
               window (8 B), and the SST history ONCE per 32-year block: block k reads its first
               year index 1 + 32 k rows, sum over the 18 blocks of a 555-year run = 4 914 rows =
               70.8 B a year;
      entry / exit  the state table in and out once per launch: 16 (27 + 7 B) B / 555 years.
    Kernels without the LDS tile of the block's SSTs (the two-wavefront flavour: 20 KB of LDS a
    wavefront; several biomes: the tile's LDS holds the biome arrays) re-read the block's 32 SST
    rows from the output array every year: 256 B more of L2 traffic (`l2_reread_bytes`), which
    reaches HBM only as far as the block's rows fall out of the L2 / MALL between two years -- not
    at all for four biomes at 65 536 members (traffic / design 1.07), about half of it on the
    two-wavefront flavour at 131 072 (2.05).  Small ensembles whose whole SST history stays in the
    L2 (1 024 members: 4.5 MB) move LESS than this figure (0.40)."""
    blocks = range(0, YEARS, 32)
    hist_rows = sum(1 + b for b in blocks)                       # rows 0 .. blk0-1 of every block
    parts = {"output_rows_written": 32.0, "partial_sums_written": 8.0, "partial_sum_read": 8.0,
             "q10_window_read": 8.0, "sst_history_read_once_per_block": 8.0 * hist_rows / YEARS,
             "state_in_and_out": 16.0 * (27 + 7 * biomes) / YEARS}
    total = sum(parts.values())
    if kernel == "run2" or biomes != 1:
        parts["l2_reread_bytes_not_in_the_sum"] = 256.0
    return total, parts


def effective_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(target_seconds=15.0, chunk=32):
    """Oracle on the host cores (ctypes releases the GIL: one thread per usable core),
    on a time-bounded sample of the same seeded ensemble, 555-year loop only: every
    thread keeps taking chunks of members until the deadline."""
    import itertools
    import threading
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_binding
    import hector_amd
    from hector_amd import ensemble
    orc = oracle_binding.Oracle(hector_amd.DEFAULT_SCENARIO)
    cores = effective_cores()
    S, q10 = ensemble.ecs_q10(4)
    orc.run_ecs_q10(S, q10)  # warm (page in libm etc.)
    counter = itertools.count()
    done = [0] * cores
    errs = [0] * cores
    t0 = time.perf_counter()
    deadline = t0 + target_seconds

    def work(t):
        while time.perf_counter() < deadline:
            k = next(counter)
            S, q10 = ensemble.ecs_q10(chunk, offset=k * chunk)
            errs[t] |= orc.run_ecs_q10(S, q10)[2]
            done[t] += chunk
    threads = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    assert not any(errs)
    n = sum(done)
    return n * YEARS / dt, dt, n, cores


def make_core(n, biomes, offset, device, devices=None, lib_path=None):
    """The synthetic perturbed-parameter ensemble of SURVEY.md 8(d) for members
    [offset, offset + n): ECS/Q10 (1 biome) or ECS + per-biome Q10 / warming factor (4)."""
    import hector_amd
    from hector_amd import ensemble
    kw = {"lib_path": lib_path} if lib_path else {}
    core = hector_amd.Core(n_members=n, device=device, devices=devices, **kw)
    if biomes == 1:
        S, q10 = ensemble.ecs_q10(n, offset=offset)
        core.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    else:
        S, q10s, wfs = ensemble.biome4(n, offset=offset)
        names = ["b1", "b2", "b3", "b4"]
        core.split_biome(names)
        core.setvar("S", S, "degC")
        for b, nm in enumerate(names):
            core.setvar(nm + ".q10_rh", q10s[b]).setvar(nm + ".warmingfactor", wfs[b])
    return core


def roofline_object(members, biomes, kernel_ms, kernel="run"):
    """The roofline object of one configuration from its measured kernel time (this run) and
    the committed counter profile of the same kernel source (profiles/pmc_index.json)."""
    entry, stale = pmc_entry(members, biomes)
    bpmy = BYTES_PER_MEMBER_YEAR[biomes]
    alg_bytes = bpmy * members * YEARS  # per launch (one GPU)
    secs = kernel_ms * 1e-3
    yard = alg_bytes / secs / 1e9
    dbytes, dparts = design_bytes(members, biomes, kernel)
    r = {"bound": "fp64-valu", "achieved": None, "peak": FP64_VECTOR_PEAK / 1e12,
         "unit": "TFLOP/s", "frac": None, "traffic": None,
         # this design's own byte model (design_bytes()): what `traffic` is held against
         "design_bytes_per_member_year": dbytes, "design_bytes_parts": dparts,
         "design_bytes_per_launch": dbytes * members * YEARS, "traffic_over_design_bytes": None,
         "kernel": ("hx_pair_kernel" if kernel == "pair" else
                    "hx_run_kernel<HX_B1W2> (two resident wavefronts per SIMD)" if kernel == "run2" else
                    "hx_run_kernel<%d>" % biomes),
         "kernel_ms": kernel_ms,
         "kernel_source_hash": kernel_source_hash()}
    if entry is not None:
        flops = entry["fp64_flops_per_launch"]
        r["achieved"] = flops / secs / 1e12
        r["frac"] = r["achieved"] / r["peak"]
        r["fp64_flops_per_launch"] = flops
        r["fp64_mfma_flops_per_launch"] = entry.get("fp64_mfma_flops_per_launch", 0.0)
        r["fp64_mfma_tflops"] = entry.get("fp64_mfma_flops_per_launch", 0.0) / secs / 1e12
        r["valu_active_frac"] = entry["valu_active_frac"]
        r["traffic"] = entry["traffic_bytes_per_launch"]
        r["hbm_measured_frac"] = entry["traffic_bytes_per_launch"] / secs / HBM_PEAK
        r["traffic_over_design_bytes"] = entry["traffic_bytes_per_launch"] / (dbytes * members * YEARS)
        r["pmc_profile"] = entry["source"]
        r["pmc_profile_stale"] = stale
    r["formula"] = ("achieved = executed fp64 flops per launch (SQ_INSTS_VALU_FLOPS_FP64 x 64 lanes "
                    "x lane utilisation; 2 per FMA, 1 per add / mul / rcp / sqrt; rocprofv3 --pmc "
                    "pass of this kernel source, profiles/) / mean HIP-event kernel time of this "
                    "run; peak = fp64 vector 78.6 TFLOP/s (the DOECLIM history contraction runs on the fp64 "
                    "matrix pipe beside it: fp64_mfma_tflops = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 / time); "
                    "hbm_measured_frac = PMC traffic "
                    "(2 x FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s")
    if biomes == 1:
        yf = SURVEY_FLOPS_PER_MEMBER_YEAR * members * YEARS / secs / 1e12
        r["flops_yardstick"] = {
            "bound": "fp64-valu", "achieved": yf, "peak": FP64_VECTOR_PEAK / 1e12, "unit": "TFLOP/s",
            "frac": yf / (FP64_VECTOR_PEAK / 1e12), "algorithmic_flops_per_member_year": SURVEY_FLOPS_PER_MEMBER_YEAR,
            "note": "SURVEY 8(d)'s flop count of the algorithm AS THE REFERENCE EXECUTES IT (with warm-started "
                    "Newton) over this run's time: a yardstick of the same kind as hbm_yardstick.  The "
                    "kernels execute ~2.9 k flops per member-year, not 9.6 k -- the flux chain, scaled stages, "
                    "coefficient chains, per-year polynomial constants and fitted equilibrium constants "
                    "removed the rest -- so `frac` (EXECUTED flops) FALLS when such a step makes the kernel "
                    "faster: round 6's coefficient chains took 8.5 % of the flops and ~2 % of the time"}
    r["hbm_yardstick"] = {
        "bound": "hbm", "achieved": yard, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": yard / (HBM_PEAK / 1e9), "algorithmic_bytes_per_member_year": bpmy,
        "algorithmic_bytes_per_launch": alg_bytes,
        "note": "SURVEY 8(d)'s YEAR-STEPPED byte model (state round trip through HBM every year + "
                "the whole SST history re-read every year), NOT this kernel: the block-causal "
                "DOECLIM pass reads the history once per 32 years and the state lives in "
                "registers / LDS for the launch, so a frac above 1 here only says the kernel does "
                "not move those bytes.  The figure to hold `traffic` against is "
                "design_bytes_per_member_year (traffic_over_design_bytes)"}
    return r


def spread(ms):
    """min / median / max of the timed steps' kernel times (HIP events on the core's stream)."""
    import numpy as np
    a = np.asarray(ms, dtype=np.float64)
    return {"n": int(a.size), "min": float(a.min()), "median": float(np.median(a)), "max": float(a.max())}


def wave_time(core):
    """The tail of the last year-loop launch from the kernels' own stamps (hx_wave_clock, 100 MHz
    s_memrealtime): how long the wavefronts ran, and the longest over the mean -- the launch lasts
    as long as its last wavefront."""
    import numpy as np
    w = core.wave_clock().astype(np.float64) * 1e-5   # ms
    if len(w) == 0:
        return None
    dur = w[:, 1] - w[:, 0]
    return {"waves": int(len(w)), "mean_ms": float(dur.mean()), "max_ms": float(dur.max()), "min_ms": float(dur.min()),
            "max_over_mean": float(dur.max() / dur.mean()), "last_start_ms": float(w[:, 0].max()),
            "span_ms": float(w[:, 1].max() - w[:, 0].min())}


def time_config(n, biomes, steps, warmup, device, variant=None, lib_path=None):
    """One extra configuration, timed like the headline: -> dict for other_configs.
    variant: None, "npp" (the NPP diagnostic recorded: the plain kernel + diagnostics family) or
    "diff" (every member its own ocean heat diffusivity: per-member DOECLIM kernel tables)."""
    import numpy as np
    import torch
    core = make_core(n, biomes, 0, device, lib_path=lib_path)
    if variant == "npp":
        core.set_outputs(["CO2_concentration", "global_tas", "NPP"])
    elif variant == "diff":
        from hector_amd import ensemble
        core.setvar("diff", 1.2 + 2.2 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 5), "cm2/s")
    start, end = core.strtdate, core.enddate
    stats = torch.zeros((2, end - start + 1, 5), dtype=torch.float64, device="cuda:%d" % device)

    def step():
        core.reset(start)
        core.run(end, wait=False)
        core.stats_device("CO2_concentration", start, end, stats[0].data_ptr())
        core.stats_device("global_tas", start, end, stats[1].data_ptr())
        return core.last_run_ms()
    core.status()
    first_by = core.lane_order_source()               # (the cost model of an earlier core of this process, if any)
    core.run(end)                                     # lane calibration pass (see main) ...
    first_ms = core.last_run_ms()                     # ... and what a one-shot run of this core costs
    first_pw = core.last_run_prewarmed()              # (launched behind the prewarm loop: hx_set_prewarm)
    core.reset(start); core.status()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = [step() for _ in range(steps)]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    bad = int((core.status() != 0).sum())
    which = core.last_run_kernel()
    lanes_by = core.lane_order_source()
    wt = wave_time(core)
    core.shutdown()
    kernel_ms = float(np.mean(kms))
    rf = roofline_object(n, biomes, kernel_ms, which)
    if lib_path:  # (another build of the same sources: time only)
        return {"members": n, "biomes": biomes, "steps": steps, "kernel": which, "kernel_ms": kernel_ms,
                "kernel_ms_steps": spread(kms), "first_run_kernel_ms": first_ms,
                "value": n * YEARS * steps / elapsed, "unit": "member-years/s",
                "members_with_model_errors": bad}
    if variant:   # (no counter profile of these instantiations: time only)
        return {"members": n, "biomes": biomes, "variant": {"npp": "NPP diagnostic recorded",
                                                             "diff": "per-member ocean heat diffusivity"}[variant],
                "steps": steps, "ms_per_step": elapsed / steps * 1e3, "kernel": which, "kernel_ms": kernel_ms,
                "first_run_kernel_ms": first_ms, "value": n * YEARS * steps / elapsed, "unit": "member-years/s",
                "members_with_model_errors": bad}
    return {"members": n, "biomes": biomes, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
            "kernel": rf["kernel"], "kernel_ms": kernel_ms, "kernel_ms_steps": spread(kms),
            "first_run_kernel_ms": first_ms, "first_run_over_steady": first_ms / kernel_ms,
            "first_run_prewarmed": first_pw,
            "lanes_ordered_by": lanes_by, "first_run_lanes_ordered_by": first_by,
            "wave_max_over_mean": wt and wt["max_over_mean"], "wave_mean_ms": wt and wt["mean_ms"],
            "value": n * YEARS * steps / elapsed, "unit": "member-years/s",
            "members_with_model_errors": bad,
            "fp64_valu_frac": rf["frac"], "hbm_yardstick_frac": rf["hbm_yardstick"]["frac"],
            "design_bytes_per_member_year": rf["design_bytes_per_member_year"],
            "traffic_over_design_bytes": rf["traffic_over_design_bytes"],
            "pmc_profile_stale": rf.get("pmc_profile_stale")}


STAT_VARS = ["CO2_concentration", "global_tas"]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per
    GPU over RCCL, like the driver's torch.distributed.run line) and hand their JSON line on."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.dist_backend == "nccl" and have < args.gpus:
        raise SystemExit("bench.py: --gpus %d asked for but this box has %d HIP device(s); RCCL needs "
                         "one GPU per rank (a rehearsal of the multi-process flow on fewer GPUs: "
                         "--dist-backend gloo)" % (args.gpus, have))
    if have < 1:
        raise SystemExit("bench.py needs an MI355X: the integrator has no CPU path")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
           str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


COMM_INIT_TIMEOUT_S = 120.0


def setup_native_collective(core, dist, world, rank, timeout_s=COMM_INIT_TIMEOUT_S):
    """One RCCL communicator inside libhector_amd.so over all ranks (hx_comm_init_rank): rank 0's
    unique id travels through torch.distributed's store.  The call is bounded: it runs on a helper
    thread (ctypes releases the GIL) and a rank whose ncclCommInitRank has not returned after
    timeout_s reports failure instead of hanging the job.  -> (ok, message); every rank agrees."""
    import threading
    import torch
    import hector_amd
    from hector_amd import core as core_mod
    ids = [core_mod.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    res = {}

    def work():
        try:
            core.comm_init_rank(world, rank, ids[0])
            res["ok"] = True
        except hector_amd.HectorAmdError as e:
            res["ok"], res["msg"] = False, str(e)
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    hung = th.is_alive()
    if hung:
        ok, msg = 0, "hx_comm_init_rank did not return within %.0f s" % timeout_s
    else:
        ok, msg = (1, "") if res.get("ok") else (0, res.get("msg", "unknown error"))
    # (min over ranks of "ok", max over ranks of "a helper thread is still inside the call")
    flag = torch.tensor([ok, -int(hung)], dtype=torch.int32,
                        device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag[1].item()) != 0:
        # An abandoned helper thread is still inside hx_comm_init_rank on SOME rank: it goes on
        # mutating that core (communicators, world size, the statistics buffers) while the main
        # thread would run and fetch on it -- no fallback is safe.  Every rank ends the job here.
        if rank == 0:
            print(json.dumps({"invalid": ["hx_comm_init_rank hung on at least one rank (%.0f s): job "
                                          "abandoned, no collective fallback on a core in that state" % timeout_s]}))
        sys.stdout.flush(); sys.stderr.write("bench.py: rank %d: %s; exiting\n" % (rank, msg or "a peer's communicator setup hung"))
        sys.stderr.flush()
        os._exit(3)
    return bool(flag[0].item()), msg


def runtime_versions():
    """HIP runtime / driver this process runs on, next to the compiler that built the kernels: a
    library built by one ROCm and run under another's runtime works today; the line records the
    pairing so that a change of either shows in the figures' provenance."""
    import ctypes
    v = {"compiler": compiler_id()}
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        for name, key in (("hipRuntimeGetVersion", "hip_runtime"), ("hipDriverGetVersion", "hip_driver")):
            x = ctypes.c_int(0)
            if getattr(hip, name)(ctypes.byref(x)) == 0:
                v[key] = x.value
    except OSError:
        pass
    try:
        import torch
        v["torch"] = torch.__version__
        v["torch_hip"] = torch.version.hip
    except Exception:
        pass
    try:
        import hector_amd
        v["library_built_with_hip"] = hector_amd.build_info()
    except Exception:
        pass
    return v


def run_workload(args, ctx, n, steps, warmup, sustain=0):
    """One weak-scaling workload: n members on every GPU, `steps` timed steps between barriers.
    -> dict of this rank's view (elapsed and kernel time already the maximum over ranks)."""
    import numpy as np
    import torch
    from hector_amd.distributed import allreduce_stats
    dist, world, rank, dev = ctx["dist"], ctx["world"], ctx["rank"], ctx["dev"]
    use_dist, shards = ctx["use_dist"], ctx["shards"]
    offset = rank * n  # weak scaling: contiguous member blocks, SURVEY.md 8(e)
    if args.single_process:
        devices = list(range(args.gpus)) if args.dist_backend == "nccl" else [0] * args.gpus
        core = make_core(n * shards, args.biomes, 0, 0, devices=devices)
    else:
        core = make_core(n, args.biomes, offset, ctx["local_rank"])
    start, end = core.strtdate, core.enddate
    nyr = end - start + 1
    stats = torch.zeros((2, nyr, 5), dtype=torch.float64, device=dev)

    # Who combines the statistics across GPUs.  native: the library's own communicator,
    # ncclAllGather on the core's stream.  torch: one all-reduce by torch.distributed on torch's
    # stream, ordered against the core's stream through an ExternalStream.
    collective, fallback = None, None
    if use_dist:
        collective = args.collective if args.dist_backend == "nccl" else "torch"
        if collective == "native":
            ok, msg = setup_native_collective(core, dist, world, rank)
            if not ok:
                sys.stderr.write("bench.py: rank %d: the library's RCCL communicator could not be "
                                 "created (%s); falling back to torch.distributed's\n" % (rank, msg))
                collective, fallback = "torch", msg
    elif shards > 1:
        collective = "native"
    core_stream = torch.cuda.ExternalStream(core.stream(), device=dev) if collective == "torch" else None

    def step():
        core.reset(start)
        core.run(end, wait=False)
        if collective == "torch":
            core.stats_device("CO2_concentration", start, end, stats[0].data_ptr())
            core.stats_device("global_tas", start, end, stats[1].data_ptr())
            cur = torch.cuda.current_stream()
            cur.wait_stream(core_stream)      # statistics written before they are reduced
            allreduce_stats(stats, dist, force=True)
            core_stream.wait_stream(cur)      # ... and reduced before the next step overwrites them
        else:   # local reductions (+ the one all-gather when there are several ranks)
            core.ensemble_stats(STAT_VARS, (start, end), d_out=stats.data_ptr(), host=False)
        return core.last_run_ms()

    core.status()  # upload + spinup + alkalinity tuning, outside every timed region
    # Lane calibration, also outside: one complete pass lets the core measure what every member's
    # solver costs; the reset adopts the lane order by measured cost (hx_set_lane_calibration:
    # wavefronts of like members, the costliest dispatched first) and spins up again.  (Every
    # hx_run_kernel dispatch a profiler sees is a full 555-year one.)
    first_by = core.lane_order_source()
    core.run(end)
    first_ms = core.last_run_ms()   # the one-shot run: lanes in the order a fresh core gives them
    first_pw = core.last_run_prewarmed()
    core.reset(start)
    core.status()
    spin_ms = core.last_spinup_ms()
    for _ in range(warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(steps):
        kern_ms.append(step())
    core.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    km = torch.tensor([float(np.mean(kern_ms)), first_ms], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
    # A sustained leg OUTSIDE the contract's timed region (N = 1, the headline workload only): the
    # same step, `sustain` times back to back under one clock -- a couple of seconds of GPU work
    # that an outside observer's utilisation sampler can see (the K timed steps are ~0.1 s), and
    # the throughput the kernel holds once the chip has settled under load.
    sustained = None
    if sustain and not use_dist:
        stats_host_keep = stats.cpu().numpy()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sk = [step() for _ in range(sustain)]
        core.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        sustained = {"steps": sustain, "seconds": el, "ms_per_step": el / sustain * 1e3,
                     "value": n * shards * YEARS * sustain / el, "unit": "member-years/s",
                     "kernel_ms_steps": spread(sk),
                     "note": "the same step as the timed region, back to back, outside it: long enough "
                             "for an outside utilisation sampler to see the GPU busy"}
        assert np.array_equal(stats_host_keep, stats.cpu().numpy())   # (the same ensemble, the same statistics)
    r = {"n": n, "steps": steps, "warmup": warmup, "sustained": sustained,
         "elapsed": float(t.item()),
         "kernel_ms": float(km[0].item()),   # slowest rank's (and slowest shard's) mean kernel time
         "first_run_kernel_ms": float(km[1].item()),   # ... and its first (one-shot) run's
         "first_run_prewarmed": first_pw,
         "kernel_ms_steps": spread(kern_ms),           # (this rank's)
         "bad": int((core.status() != 0).sum()),
         "stats_host": stats.cpu().numpy(),
         "which_kernel": core.last_run_kernel(),
         "calibrated": core.lanes_calibrated(), "lanes_by": core.lane_order_source(), "first_run_lanes_by": first_by,
         "wave_time": wave_time(core),   # (this rank's first shard, the last timed launch)
         "spin_ms": spin_ms, "collective": collective, "native_fallback": fallback}
    r["comm_world"], _, r["comm_backend"] = core.comm_info()
    core.shutdown()
    return r


def collective_facts(args, ctx, r):
    """-> (backend string, world size of the collective that combined the statistics)."""
    if r["collective"] == "native":
        return "RCCL ncclAllGather issued by libhector_amd.so (%s)" % r["comm_backend"], r["comm_world"]
    if r["collective"] == "torch":
        return (("RCCL all-reduce through torch.distributed (nccl)" if args.dist_backend == "nccl"
                 else "torch.distributed %s (rehearsal, not RCCL)" % args.dist_backend), ctx["world"])
    return None, 1


def workload_name(n, n_gpus, biomes):
    if biomes == 4:
        return ("%d-member perturbed 4-biome ECS/Q10/warmingfactor ensemble per GPU, SSP2-4.5 1745-2300 "
                "(BASELINE configs[4])" % n)
    if n == 131072:
        which = ("BASELINE configs[3]: 1 048 576 members over 8 GPUs" if n_gpus == 8 else
                 "BASELINE configs[3]'s shape, 131 072 members per GPU: %d members over %d GPU%s, "
                 "1 048 576 at 8" % (n * n_gpus, n_gpus, "" if n_gpus == 1 else "s"))
    elif n == 65536:
        which = "BASELINE configs[2] on every GPU" if n_gpus > 1 else "BASELINE configs[2]"
    elif n == 1024:
        which = "BASELINE configs[1]"
    else:
        which = "the kernel of BASELINE configs[2], another ensemble size"
    return "%d-member perturbed ECS/Q10 ensemble per GPU, SSP2-4.5 1745-2300 (%s)" % (n, which)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--members", type=int, default=None,
                    help="members per GPU (default: 65 536 = BASELINE configs[2] on one GPU; "
                         "131 072 = configs[3]'s share of one GPU when --gpus > 1)")
    ap.add_argument("--biomes", type=int, default=1, choices=[1, 4])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain", type=int, default=300,
                    help="N = 1: steps of the sustained leg behind the timed region (0 = none)")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target wall time of the bounded CPU-baseline sample")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                         "multi-process flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--single-process", action="store_true",
                    help="one process, one core over --gpus devices (hx_newcore_devices) instead "
                         "of one process per GPU")
    ap.add_argument("--collective", default="native", choices=["native", "torch"],
                    help="who issues the statistics collective: libhector_amd.so's own RCCL "
                         "communicator (ncclAllGather on the core's stream) or torch.distributed")
    ap.add_argument("--force-collective", action="store_true",
                    help="with one rank: still create the communicator and run the collective "
                         "(world size 1) -- the RCCL path on a one-GPU box")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1 and not args.single_process:
        self_launch(args)   # does not return

    import numpy as np
    import torch
    import torch.distributed as dist
    from hector_amd.distributed import finalize

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and world != args.gpus:
        raise SystemExit("--gpus (%d) must equal WORLD_SIZE (%d)" % (args.gpus, world))
    if args.single_process and world > 1:
        raise SystemExit("--single-process is for one process; do not combine it with a launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the integrator has no CPU path")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and (args.gpus if args.single_process else local_rank + 1) > ndev:
        raise SystemExit("bench.py: %d GPUs needed, %d visible" % (args.gpus, ndev))
    if args.dist_backend != "nccl":  # rehearsal: ranks may share a GPU
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (args.force_collective and not args.single_process)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("NCCL_DEBUG", "WARN")   # RCCL's complaints on stderr, whoever issues the call
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    shards = args.gpus if args.single_process else 1   # GPUs this process drives
    n_gpus = world * shards
    ctx = {"dist": dist, "world": world, "rank": rank, "local_rank": local_rank, "dev": dev,
           "use_dist": use_dist, "shards": shards}
    # The headline workload is the SAME per-GPU work at every N, so that the values of a --gpus
    # 1, 2, 4, 8 series are one weak-scaling curve: BASELINE configs[2], 65 536 members on every
    # GPU.  configs[3]'s shape -- 131 072 members on every GPU, the named 1 048 576 at N = 8 -- is
    # timed after it in the same job (N > 1: other_configs[0]; N = 1: among other_configs), and
    # both are at the top level of every line: value_per_gpu_workload, first_run_kernel_ms.
    n = args.members if args.members else 65536
    r = run_workload(args, ctx, n, args.steps, args.warmup, sustain=(args.sustain if n_gpus == 1 and not args.members else 0))
    second = None
    if n_gpus > 1 and not args.members and not args.no_other_configs:
        # (timed like the headline -- the same steps and warmup -- so that the two
        # workloads of a multi-GPU line are of equal standing)
        second = run_workload(args, ctx, 131072, args.steps, args.warmup)

    failures = []
    if rank == 0:
        def summary(r):
            total_members = r["n"] * n_gpus
            mean, std, mn, mx = finalize(r["stats_host"])
            backend, cworld = collective_facts(args, ctx, r)
            in_stats = int(r["stats_host"][0, -1, 0])
            # what a multi-GPU line must show to count (exit status below)
            if in_stats != total_members:
                failures.append("%d members in the statistics, %d in the ensemble" % (in_stats, total_members))
            if n_gpus > 1:
                if cworld != n_gpus:
                    failures.append("the collective spans %s ranks, the job %d GPUs" % (cworld, n_gpus))
                if args.dist_backend == "nccl" and "RCCL" not in (backend or ""):
                    failures.append("the statistics did not travel over RCCL (%s)" % backend)
            return total_members, mean, backend, cworld, in_stats
        total_members, mean, backend, cworld, in_stats = summary(r)
        elapsed, kernel_ms = r["elapsed"], r["kernel_ms"]
        value = total_members * YEARS * args.steps / elapsed
        out = {
            "metric": "ensemble-member simulated years/sec",
            "value": value,
            "unit": "member-years/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            # which per-GPU workload `value` is: 2 = since round 5, BASELINE configs[2] (65 536
            # members) on every GPU at every N, configs[3]'s shape beside it; lines of rounds 1-4
            # (version 1) quoted 131 072 per GPU at N > 1 and are not comparable with these
            "workload_version": 2,
            "value_workload": "%d members per GPU" % n,
            "scaling_workload": "value: %d members on EVERY GPU at every N (one weak-scaling curve; "
                                "N = 1 is BASELINE configs[2]); BASELINE configs[3]'s shape, 131 072 members "
                                "per GPU = 1 048 576 at N = 8: value_per_gpu_workload['131072']" % n,
            "value_per_gpu_workload": {str(n): value},
            "first_run_kernel_ms": {str(n): r["first_run_kernel_ms"]},
            "first_run_over_steady": {str(n): r["first_run_kernel_ms"] / kernel_ms},
            "first_run_prewarmed": {str(n): r["first_run_prewarmed"]},
            "kernel_ms": {str(n): kernel_ms},
            "kernel_ms_steps": r["kernel_ms_steps"],   # min / median / max of the timed steps (HIP events)
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload_name(n, n_gpus, args.biomes) +
                            "; reference dopri5 adaptive solver + stash/retry logic; spinup and upload excluded",
                "members_per_gpu": n, "global_members": total_members, "years_per_member": YEARS,
                "biomes": args.biomes,
                "parallelism": "member-sharded x%d (%s), one statistics collective per step" %
                               (n_gpus, "one process, device list" if args.single_process
                                else "one process per GPU"),
                "collective_backend": backend,
                "collective_world_size": cworld,
                "native_collective_fallback": r["native_fallback"],
                "rehearsal_not_rccl": bool(use_dist and args.dist_backend != "nccl"),
                # (say so until it has) the builder's boxes have ONE GPU -- a communicator
                # of world size > 1, the grouped all-gather over several shards and per-shard host
                # threads on distinct devices execute for the first time wherever this line is
                # printed with n_gpus > 1; the assertions above (complete statistics, world size,
                # RCCL) are what vouches for that run
                "multi_gpu_path": ("executed here: world size %d" % n_gpus) if n_gpus > 1 else
                                  "not exercised (one GPU)",
                "spinup_ms_excluded": r["spin_ms"], "members_with_model_errors": r["bad"],
                "lanes_ordered_by": r["lanes_by"], "first_run_lanes_ordered_by": r["first_run_lanes_by"],
                "members_in_statistics": in_stats,
                "co2_2300_mean_ppm": float(mean[0, -1]), "tgav_2300_mean_K": float(mean[1, -1]),
                "versions": runtime_versions(),
            },
            # per-rank maximum of the mean kernel time when N > 1
            "roofline": roofline_object(n, args.biomes, kernel_ms, r["which_kernel"]),
        }
        out["roofline"]["wave_time"] = r["wave_time"]
        if r.get("sustained"):
            out["sustained"] = r["sustained"]
        if second is not None:
            tm2, mean2, backend2, cworld2, in2 = summary(second)
            rf2 = roofline_object(second["n"], args.biomes, second["kernel_ms"], second["which_kernel"])
            out["other_configs"] = [{
                "workload": workload_name(second["n"], n_gpus, args.biomes),
                "members_per_gpu": second["n"], "global_members": tm2, "n_gpus": n_gpus,
                "steps": second["steps"], "ms_per_step": second["elapsed"] / second["steps"] * 1e3,
                "value": tm2 * YEARS * second["steps"] / second["elapsed"], "unit": "member-years/s",
                "scaling": "weak", "kernel": rf2["kernel"], "kernel_ms": second["kernel_ms"],
                "collective_backend": backend2, "collective_world_size": cworld2,
                "members_in_statistics": in2, "members_with_model_errors": second["bad"],
                "first_run_kernel_ms": second["first_run_kernel_ms"],
                "first_run_over_steady": second["first_run_kernel_ms"] / second["kernel_ms"],
                "first_run_prewarmed": second["first_run_prewarmed"], "kernel_ms_steps": second["kernel_ms_steps"],
                "lanes_ordered_by": second["lanes_by"], "first_run_lanes_ordered_by": second["first_run_lanes_by"],
                "fp64_valu_frac": rf2["frac"], "pmc_profile_stale": rf2.get("pmc_profile_stale")}]
            k2 = str(second["n"])
            out["value_per_gpu_workload"][k2] = out["other_configs"][0]["value"]
            out["first_run_kernel_ms"][k2] = second["first_run_kernel_ms"]
            out["first_run_over_steady"][k2] = second["first_run_kernel_ms"] / second["kernel_ms"]
            out["first_run_prewarmed"][k2] = second["first_run_prewarmed"]
            out["kernel_ms"][k2] = second["kernel_ms"]
        if n_gpus == 1 and not args.no_other_configs:
            others = []
            for (m2, b2) in ((1024, 1), (32768, 1), (131072, 1), (65536, 4)):
                if (m2, b2) == (n, args.biomes):
                    continue
                others.append(time_config(m2, b2, 5, 1, local_rank))
            out["other_configs"] = others
            # the headline ensemble with what users commonly add to it (not BASELINE configurations)
            out["workload_variants"] = [time_config(n, args.biomes, 5, 1, local_rank, variant=v)
                                        for v in ("npp", "diff")] if args.biomes == 1 else []
            for o in others:   # (one biome: the ensemble sizes of the --gpus N series and configs[1])
                if o["biomes"] == args.biomes:
                    out["value_per_gpu_workload"][str(o["members"])] = o["value"]
                    out["first_run_kernel_ms"][str(o["members"])] = o["first_run_kernel_ms"]
                    out["first_run_over_steady"][str(o["members"])] = o["first_run_over_steady"]
                    out["first_run_prewarmed"][str(o["members"])] = o["first_run_prewarmed"]
                    out["kernel_ms"][str(o["members"])] = o["kernel_ms"]
            # what losing the product's code-generation flags would cost: the same sources built with
            # default code generation (`make safe`: no -disable-machine-licm, the carbonate restart
            # in its select form), timed the same way
            safe = os.path.join(ROOT, "hector_amd", "lib", "libhector_amd_safe.so")
            if os.path.exists(safe):
                try:
                    sb = {str(m2): time_config(m2, 1, 3, 1, local_rank, lib_path=safe) for m2 in (65536, 131072)}
                    for k2, v2 in sb.items():
                        v2["over_product_build"] = v2["kernel_ms"] / out["kernel_ms"][k2] if k2 in out["kernel_ms"] else None
                    out["safe_build"] = dict(sb, note="libhector_amd_safe.so: default code generation, "
                                             "HX_CHEM_SELECT; kernel_ms against the product build's")
                except Exception as e:   # a test-only artefact: its absence or failure is not the bench's
                    out["safe_build"] = {"error": str(e)[:200]}
            else:
                out["safe_build"] = {"error": "hector_amd/lib/libhector_amd_safe.so not built (make -C hector_amd/csrc safe)"}
        if n_gpus == 1 and not args.no_cpu_baseline:
            v, dt, ns, cores = cpu_baseline(args.cpu_seconds)
            out["cpu_baseline"] = {
                "value": v, "unit": "member-years/s", "cores": cores, "kind": "port",
                "sample": "%d members (the first ones of the same seeded ECS/Q10 ensemble) x 555 years, "
                          "oracle/hector_oracle.c (scalar C restatement of the reference loop, "
                          "spinup shared and excluded), %d threads, %.1f s wall" % (ns, cores, dt),
                "provenance": "oracle pinned to the reference's golden trajectory "
                              "tests/testthat/compdata/hector_comp.csv (<= 1e-13 rel CO2, 10 variables x "
                              "556 years; tests/test_oracle_golden.py); not the reference binary (its Boost "
                              "dependency is absent from the image).  The reference's own C++ loop measured "
                              "by the survey with a Boost shim: 600-1 030 member-years/s/core (BASELINE.md "
                              "2), i.e. the oracle is ~100-200x the reference per core",
            }
        if failures:
            out["invalid"] = failures
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.destroy_process_group()
    if failures:
        sys.stderr.write("bench.py: INVALID multi-GPU line: " + "; ".join(failures) + "\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
