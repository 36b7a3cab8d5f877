#!/usr/bin/env python3
"""bench.py -- ensemble-member simulated years per second, SSP2-4.5 1745-2300.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over this rank's batch of synthetic
members: reset to the post-spinup state (a device-to-device copy), integrate
1745 -> 2300 (555 model years per member) with the HIP kernels, reduce the
per-year ensemble statistics of CO2 and Tgav on the GPU and (N > 1) all-reduce
them over RCCL.  Inputs (parameters, scenario tables, spun-up state) are
resident in HBM before the timed region; spinup and upload are excluded, as
SURVEY.md 8(d) defines the metric.  Weak scaling: per-GPU members fixed.

N = 1 workload: BASELINE.json configs[2] -- 65 536-member perturbed ECS/Q10
ensemble on one MI355X (the 1 048 576-member / 8-GPU configs[3] is the same
kernel at 131 072 members per GPU: pass --members 131072).

Prints ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline     -- algorithmic HBM bytes per launch (2 680 B per member-year,
                  SURVEY.md 8d, x members x 555) / mean kernel time by HIP events
                  on the core's stream, against the 8 TB/s HBM3E peak
  cpu_baseline -- the CPU oracle (a scalar C port of the reference loop, validated
                  against the reference's golden trajectory) on a bounded sample of
                  the same ensemble, all host cores, 555-year loop only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_MEMBER_YEAR = {1: 2680.0, 4: 3072.0}  # SURVEY.md 8(d) / BASELINE.md 4
HBM_PEAK = 8.0e12                               # MI355X_MICROARCH.md: 8 TB/s spec
YEARS = 555


def pmc_traffic(members, biomes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (2*FETCH_SIZE +
    WRITE_SIZE, calibrated; profiles/r01_pmc_traffic.json) for this exact workload, or None:
    counters cannot be read from inside the timed run."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        return d["traffic_bytes_per_launch"].get("%dx%d" % (members, biomes))
    except Exception:
        return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=65536, help="members per GPU")
    ap.add_argument("--biomes", type=int, default=1, choices=[1, 4])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target wall time of the bounded CPU-baseline sample")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                         "multi-process flow on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import hector_amd
    from hector_amd import ensemble
    from hector_amd.distributed import allreduce_stats, finalize

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the integrator has no CPU path")
    if args.dist_backend != "nccl":  # rehearsal: ranks may share a GPU
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    n = args.members
    offset = rank * n  # weak scaling: contiguous member blocks, SURVEY.md 8(e)
    core = hector_amd.Core(n_members=n, device=local_rank)
    if args.biomes == 1:
        S, q10 = ensemble.ecs_q10(n, offset=offset)
        core.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    else:
        S, q10s, wfs = ensemble.biome4(n, offset=offset)
        names = ["b1", "b2", "b3", "b4"]
        core.split_biome(names)
        core.setvar("S", S, "degC")
        for b, nm in enumerate(names):
            core.setvar(nm + ".q10_rh", q10s[b]).setvar(nm + ".warmingfactor", wfs[b])
    start, end = core.strtdate, core.enddate
    nyr = end - start + 1
    stats = torch.zeros((2, nyr, 5), dtype=torch.float64, device=dev)

    # the core queues its kernels on its own HIP stream; the collective runs on torch's
    core_stream = torch.cuda.ExternalStream(core.stream(), device=dev) if world > 1 else None

    def step():
        core.reset(start)
        core.run(end, wait=False)
        core.stats_device("CO2_concentration", start, end, stats[0].data_ptr())
        core.stats_device("global_tas", start, end, stats[1].data_ptr())
        if world > 1:
            cur = torch.cuda.current_stream()
            cur.wait_stream(core_stream)      # statistics written before they are reduced
            allreduce_stats(stats, dist)
            core_stream.wait_stream(cur)      # ... and reduced before the next step overwrites them
        return core.last_run_ms()

    core.status()  # upload + spinup + alkalinity tuning, outside every timed region
    # (no run-kernel launch here: every hx_run_kernel dispatch a profiler sees is a full
    # 555-year one, so its average duration is the kernel_ms reported below)
    spin_ms = core.last_spinup_ms()
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        kern_ms.append(step())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    km = torch.tensor([float(np.mean(kern_ms))], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_ms = float(km.item())
    bad = int((core.status() != 0).sum())

    if rank == 0:
        total_members = n * world
        value = total_members * YEARS * args.steps / elapsed
        bpmy = BYTES_PER_MEMBER_YEAR[args.biomes]
        alg_bytes = bpmy * n * YEARS  # per launch (one GPU)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        mean, std, mn, mx = finalize(stats.cpu().numpy())
        out = {
            "metric": "ensemble-member simulated years/sec",
            "value": value,
            "unit": "member-years/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d-member perturbed %s ensemble per GPU, SSP2-4.5 1745-2300 "
                            "(BASELINE configs[%d]); reference dopri5 adaptive solver + "
                            "stash/retry logic; spinup and upload excluded" %
                            (n, "ECS/Q10" if args.biomes == 1 else "4-biome ECS/Q10/warmingfactor",
                             2 if args.biomes == 1 else 4),
                "members_per_gpu": n, "global_members": total_members, "years_per_member": YEARS,
                "biomes": args.biomes, "parallelism": "member-sharded x%d, stats all-reduce" % world,
                "spinup_ms_excluded": spin_ms, "members_with_model_errors": bad,
                "co2_2300_mean_ppm": float(mean[0, -1]), "tgav_2300_mean_K": float(mean[1, -1]),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / (HBM_PEAK / 1e9), "traffic": pmc_traffic(n, args.biomes),
                "kernel": "hx_run_kernel<%d>" % args.biomes, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_member_year": bpmy,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "achieved = algorithmic bytes (SURVEY 8d: state round trip + the yearly "
                        "re-read of the SST history) / kernel time; it can exceed the HBM peak "
                        "because the block-causal DOECLIM pass reads the history once per 32 years "
                        "and the state lives in LDS -- 'traffic' is what the PMC counters saw. "
                        "What actually bounds the kernel is fp64 issue + latency of one resident "
                        "wavefront per SIMD (DESIGN.md section 6).",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            v, dt, ns, cores = cpu_baseline(args.cpu_seconds)
            out["cpu_baseline"] = {
                "value": v, "unit": "member-years/s", "cores": cores, "kind": "port",
                "sample": "%d members (the first ones of the same seeded ECS/Q10 ensemble) x 555 years, "
                          "oracle/hector_oracle.c (scalar C restatement of the reference loop, "
                          "spinup shared and excluded), %d threads, %.1f s wall" % (ns, cores, dt),
            }
        print(json.dumps(out))
    core.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
