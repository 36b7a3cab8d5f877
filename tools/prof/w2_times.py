#!/usr/bin/env python3
"""Kernel time of the one-biome ensemble on the plain run kernel (one resident wavefront per SIMD)
and on its two-wavefront flavour (hx_run_kernel<HX_B1W2>), same core class, same call:
    python tools/prof/w2_times.py [members ...] [--lib=path]
Best and mean HIP-event time over 4 full 555-year launches after a warm-up pass (which also lets
the core adopt the measured-cost lane order), and checksums."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hector_amd  # noqa: E402


def main():
    sizes, lib, per_member_diff, extended = [], None, False, False
    for a in sys.argv[1:]:
        if a == "--diff":          # every member its own ocean heat diffusivity (KERPM kernels)
            per_member_diff = True
        elif a == "--ext":         # a diagnostic of the extended kernel (CON = 1)
            extended = True
        elif a.startswith("--lib="):
            lib = os.path.abspath(a.split("=", 1)[1])
        else:
            sizes.append(int(a))
    sizes = sizes or [65536, 131072, 262144]
    if lib:
        orig = hector_amd.Core
        hector_amd.Core = lambda *a, **k: orig(*a, **dict(k, lib_path=lib))
    for n in sizes:
        for two_wave in (0, 1):
            c = bench.make_core(n, 1, 0, 0)
            c.set_pair_kernel_limit(0)
            c.set_two_wave_from(two_wave)
            if extended:
                c.set_outputs(["CO2_concentration", "global_tas", "NPP"])
            if per_member_diff:
                from hector_amd import ensemble
                c.setvar("diff", 1.2 + 2.2 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 5), "cm2/s")
            ms = []
            for _ in range(6):
                c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
            ms = ms[2:]
            co2 = c.fetchvars("CO2_concentration", (2300, 2300))[0]
            tg = c.fetchvars("global_tas", (2300, 2300))[0]
            print("%8d members  %-4s  best %8.3f ms  mean %8.3f ms  -> %.3e member-years/s   co2 %.12f tg %.12f bad %d"
                  % (n, c.last_run_kernel(), min(ms), np.mean(ms), n * 555 / (min(ms) * 1e-3),
                     co2.mean(), tg.mean(), int((c.status() != 0).sum())), flush=True)
            c.shutdown()


if __name__ == "__main__":
    main()
