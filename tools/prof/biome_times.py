#!/usr/bin/env python3
"""Kernel time by biome count (unrolled 1-4, looped 5-16):
    python tools/prof/biome_times.py [members] [biome counts ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hector_amd  # noqa: E402
from hector_amd import ensemble  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--lib=")]
    lib = [a[6:] for a in sys.argv[1:] if a.startswith("--lib=")]   # an experiment build (gpuwork/)
    n = int(args[0]) if args else 65536
    counts = [int(x) for x in args[1:]] or [4, 5, 8, 16]
    S = 1.5 + 4.5 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 0)
    for nb in counts:
        c = hector_amd.Core(n_members=n, device=0, **({"lib_path": os.path.abspath(lib[0])} if lib else {}))
        names = ["b%d" % i for i in range(nb)]
        if nb > 1:
            c.split_biome(names)
        c.setvar("S", S, "degC")
        for b, nm in enumerate(names if nb > 1 else [""]):
            pre = nm + "." if nb > 1 else ""
            c.setvar(pre + "q10_rh", 1.0 + 2.0 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 10 + b))
            c.setvar(pre + "warmingfactor", np.full(n, 1.0 + 0.5 * (b % 4)))
        ms = []
        for _ in range(3):
            c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
        bad = int((c.status() != 0).sum())
        co2 = c.fetchvars("CO2_concentration", (2300, 2300))[0]
        print("%6d members x %2d biomes  best %8.3f ms   co2 %.9f  bad %d" % (n, nb, min(ms[1:]), co2.mean(), bad), flush=True)
        c.shutdown()


if __name__ == "__main__":
    main()
