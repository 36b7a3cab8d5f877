#!/usr/bin/env python3
"""What small ensembles cost on the small-ensemble kernel and on the run kernels, same call:
    python tools/prof/small_ensembles.py [members]
1 024 members (default): plain, CO2-constrained (concentration-driven), two and four biomes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hector_amd  # noqa: E402
from hector_amd import ensemble  # noqa: E402


def best(c, reps=4):
    ms = []
    for _ in range(reps):
        c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
    return min(ms[1:])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    idx = np.arange(n, dtype=np.uint64)
    S = 1.5 + 4.5 * ensemble.uniform01(idx, 0)
    for what in ("plain", "CO2-constrained", "2 biomes", "4 biomes"):
        row = []
        for limit in (32768, 0):
            c = hector_amd.Core(n_members=n, device=0)
            c.set_pair_kernel_limit(limit)
            c.setvar("S", S, "degC")
            if what.endswith("biomes"):
                nb = int(what[0])
                names = ["b%d" % b for b in range(nb)]
                c.split_biome(names)
                for b, nm in enumerate(names):
                    c.setvar(nm + ".q10_rh", 1.0 + 2.0 * ensemble.uniform01(idx, 10 + b))
                    c.setvar(nm + ".warmingfactor", np.full(n, 1.0 + 0.5 * b))
            else:
                c.setvar("q10_rh", 1.0 + 2.0 * ensemble.uniform01(idx, 1))
            if what == "CO2-constrained":
                c.run(2300)
                yrs = np.arange(1850, 2101)
                c.setvar_dated("CO2_constrain", yrs, c.fetchvars("CO2_concentration", (1850, 2100))[:, 0] * 1.05)
            c.run(2300)
            row.append((c.last_run_kernel(), best(c)))
            c.shutdown()
        print("%6d members, %-16s %s %.3f ms | %s %.3f ms" % (n, what + ":", row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)


if __name__ == "__main__":
    main()
