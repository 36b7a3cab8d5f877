#!/usr/bin/env python3
"""What the extended run kernels cost a large ensemble, feature by feature:
    python tools/prof/constrained_times.py [members ...] [--lib=path]
plain; NPP recorded (CON = -2); CO2-constrained 1850-2100 (concentration-driven); tas-constrained;
a land-ocean warming ratio on every other member; a CO2 constraint series per member.  Best of 3 launches after
two warm-up passes, and which instantiation family ran (hx_last_run_variant)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hector_amd  # noqa: E402
from hector_amd import ensemble  # noqa: E402


def main():
    sizes, lib = [], None
    for a in sys.argv[1:]:
        if a.startswith("--lib="):
            lib = os.path.abspath(a.split("=", 1)[1])
        else:
            sizes.append(int(a))
    for n in sizes or [65536]:
        idx = np.arange(n, dtype=np.uint64)
        S, q10 = ensemble.ecs_q10(n)
        for what in ("plain", "NPP recorded", "CO2-constrained", "tas-constrained", "warming ratio", "FFI per member"):   # (the last: a constraint series per member)
            c = hector_amd.Core(n_members=n, device=0, **({"lib_path": lib} if lib else {}))
            c.set_pair_kernel_limit(0)
            c.setvar("S", S, "degC").setvar("q10_rh", q10)
            outs = ["CO2_concentration", "global_tas"]
            if what == "NPP recorded":
                outs.append("NPP")
            c.set_outputs(outs)
            if what == "CO2-constrained":
                c.run(2300)
                yrs = np.arange(1850, 2101)
                c.setvar_dated("CO2_constrain", yrs, c.fetchvars("CO2_concentration", (1850, 2100))[:, 0] * 1.05)
            elif what == "tas-constrained":
                yrs = np.arange(1850, 2101)
                c.setvar_dated("tas_constrain", yrs, 0.012 * (yrs - 1850), "degC")
            elif what == "warming ratio":
                c.setvar("lo_warming_ratio", np.where(np.arange(n) % 2, 1.6, 0.0))
            elif what == "FFI per member":   # (a CO2 constraint series per member stands in: NaN = none)
                c.run(2300)
                yrs = np.arange(1850, 2101)
                v = np.full((yrs.size, n), np.nan)
                v[:, ::7] = c.fetchvars("CO2_concentration", (1850, 2100))[:, ::7] * 1.02
                c.setvar_dated_members("CO2_constrain", yrs, v, "ppmv CO2")
            ms = []
            for _ in range(5):
                c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
            bad = int((c.status() != 0).sum())
            print("%7d members, %-16s %-4s variant %2d  best %7.3f ms  (bad %d)"
                  % (n, what + ":", c.last_run_kernel(), c.last_run_variant(), min(ms[2:]), bad), flush=True)
            c.shutdown()


if __name__ == "__main__":
    main()
