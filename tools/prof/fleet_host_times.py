#!/usr/bin/env python3
"""Host-side cost of a sharded core (hx_newcore_devices): wall time of the first run, of a
calibration-loop iteration and of fetchvars with the shards prepared one after the other
(HECTOR_AMD_FLEET_SEQUENTIAL=1) and side by side (one host thread per shard, the default).
On a one-GPU box: HECTOR_AMD_FLEET_REHEARSAL=1 and the device list [0, 0, 0, 0] -- the kernels of the
four shards then share one GPU, so only the HOST part of the difference shows.
    python tools/prof/fleet_host_times.py [members] [shards]"""
import os
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(n, shards):
    import numpy as np
    import torch
    import hector_amd
    from hector_amd import ensemble
    ndev = torch.cuda.device_count()
    devices = list(range(shards)) if ndev >= shards else [0] * shards
    core = hector_amd.Core(n_members=n, devices=devices)
    S, q10 = ensemble.ecs_q10(n)
    core.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    t = {}
    c0 = time.perf_counter(); core.run(2300); core.sync(); t["first run"] = time.perf_counter() - c0
    rng = np.random.default_rng(3)
    c0 = time.perf_counter()
    for _ in range(5):
        core.setvar("S", S * rng.uniform(0.99, 1.01, n), "degC")
        core.reset(1745); core.run(2300); core.sync()
    t["setvar + reset + run"] = (time.perf_counter() - c0) / 5
    out = np.empty((556, n))
    core.fetchvars("global_tas", (1745, 2300), out=out)
    c0 = time.perf_counter(); core.fetchvars("global_tas", (1745, 2300), out=out); t["fetchvars"] = time.perf_counter() - c0
    core.shutdown()
    print(" | ".join("%s %.2f ms" % (k, v * 1e3) for k, v in t.items()), "| devices", devices)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
        sh = int(sys.argv[2]) if len(sys.argv) > 2 else 4
        for seq in ("1", ""):
            env = dict(os.environ, HECTOR_AMD_FLEET_REHEARSAL="1")
            env.pop("HECTOR_AMD_FLEET_SEQUENTIAL", None)
            if seq:
                env["HECTOR_AMD_FLEET_SEQUENTIAL"] = "1"
            print("sequential:" if seq else "side by side:", end=" ", flush=True)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(n), str(sh)], env=env)
