#!/usr/bin/env python3
"""Carbon-tracking runs against the plain kernel: kernel time of run(2300) with trackingDate 1750
(every stash of 550 years moves the origin maps) and without, per ensemble size and biome count.

    python tools/prof/tracking_times.py > gpurun_out/tracking_times.md
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hector_amd  # noqa: E402
from hector_amd import ensemble  # noqa: E402

CASES = ((8192, 1), (65536, 1), (8192, 2), (8192, 4), (1024, 8), (256, 16))


def run(n, nb, date):
    S, q = ensemble.ecs_q10(n)
    c = hector_amd.Core(n_members=n, device=0)
    c.set_pair_kernel_limit(0)   # (like with like: the tracking kernels are flavours of the run kernel)
    if nb > 1:
        c.split_biome(["b%d" % i for i in range(nb)])
    else:
        c.setvar("q10_rh", q)
    c.setvar("S", S, "degC")
    if date:
        c.setvar("trackingDate", [date])
    ms = []
    for _ in range(3):
        c.reset(1745)
        c.run(2300)
        ms.append(c.last_run_ms())
    ok = int((c.status() == 0).sum())
    extra = ""
    if date:
        v, f = c.tracking_data(n - 1, (2300, 2300))
        extra = "%.3e" % np.abs(f.sum(axis=2) - 1.0).max()
    c.shutdown()
    return min(ms), ok, extra


def main():
    print("| members x biomes | pools | plain kernel ms | tracked 1750-2300 ms | ratio | members ok | max abs(sum of a map - 1) in 2300 |")
    print("|---|---|---|---|---|---|---|")
    for n, nb in CASES:
        p, _, _ = run(n, nb, None)
        t, ok, e = run(n, nb, 1750.0)
        print("| %d x %d | %d | %.2f | %.2f | %.1f | %d | %s |" % (n, nb, 6 + 5 * nb, p, t, t / p, ok, e), flush=True)


if __name__ == "__main__":
    main()
