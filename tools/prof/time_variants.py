#!/usr/bin/env python3
"""Kernel time of several builds of the library on the bench configurations:
    python tools/prof/time_variants.py name=path [name=path ...] [--configs 65536x1,65536x4,1024x1]
Prints one line per (build, configuration): best and mean HIP-event time of the run kernel over 4
full 555-year launches, and CO2 / Tgav checksums so that numerically different builds stand out."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hector_amd  # noqa: E402


def main():
    libs, configs = [], [(65536, 1), (65536, 4), (1024, 1), (262144, 1)]
    for a in sys.argv[1:]:
        if a.startswith("--configs"):
            configs = [tuple(int(x) for x in c.split("x")) for c in a.split("=")[1].split(",")]
        else:
            name, path = a.split("=")
            libs.append((name, os.path.abspath(path)))
    orig = hector_amd.Core
    for name, path in libs:
        def core_with_lib(*a, **k):
            k.setdefault("lib_path", path)
            return orig(*a, **k)
        hector_amd.Core = core_with_lib
        for (n, b) in configs:
            try:
                c = bench.make_core(n, b, 0, 0)
                ms = []
                for _ in range(5):
                    c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
                ms = ms[1:]
                co2 = c.fetchvars("CO2_concentration", (2300, 2300))[0]
                tg = c.fetchvars("global_tas", (2300, 2300))[0]
                bad = int((c.status() != 0).sum())
                print("%-10s %7dx%d  best %8.3f ms  mean %8.3f ms  -> %.3e member-years/s   co2 %.12f tg %.12f bad %d"
                      % (name, n, b, min(ms), np.mean(ms), n * 555 / (min(ms) * 1e-3), co2.mean(), tg.mean(), bad),
                      flush=True)
                c.shutdown()
            except Exception as e:  # a variant may lack a configuration
                print("%-10s %7dx%d  FAILED: %s" % (name, n, b, e), flush=True)
    hector_amd.Core = orig


if __name__ == "__main__":
    main()
