#!/usr/bin/env python3
"""What a host sees around the year-loop kernel: wall time of every verb of one ensemble job
through the C ABI (ctypes veneer), first job and a second one in the same process.
    python tools/prof/e2e_times.py [members ...]
newcore -> setvar(S, q10_rh) -> run(2300) [upload + spinup + alkalinity + year loop] ->
ensemble_stats -> fetchvars(CO2, Tgav: every member, every year, PCIe) -> shutdown."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import hector_amd  # noqa: E402
from hector_amd import ensemble  # noqa: E402


# E2E_LIB=<path> times another build (the host emulation of tests/emul for a dry run without a GPU)
KW = dict(lib_path=os.environ["E2E_LIB"], allow_emulation=True) if os.environ.get("E2E_LIB") else {}


def job(n, out_bufs=None):
    t = {}
    c0 = time.perf_counter()
    core = hector_amd.Core(n_members=n, device=0, **KW)
    t["newcore"] = time.perf_counter() - c0
    S, q10 = ensemble.ecs_q10(n)
    c0 = time.perf_counter()
    core.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    t["setvar x2"] = time.perf_counter() - c0
    c0 = time.perf_counter()
    core.run(2300)
    t["run (upload+spinup+loop)"] = time.perf_counter() - c0
    t["  of it: year-loop kernel"] = core.last_run_ms() * 1e-3
    t["  of it: spinup kernel"] = core.last_spinup_ms() * 1e-3
    c0 = time.perf_counter()
    core.ensemble_stats(["CO2_concentration", "global_tas"], (1745, 2300))
    t["ensemble_stats (2 vars)"] = time.perf_counter() - c0
    c0 = time.perf_counter()
    if out_bufs is None:
        a = core.fetchvars("CO2_concentration", (1745, 2300))
        b = core.fetchvars("global_tas", (1745, 2300))
        out_bufs = (a, b)
    else:
        core.fetchvars("CO2_concentration", (1745, 2300), out=out_bufs[0])
        core.fetchvars("global_tas", (1745, 2300), out=out_bufs[1])
    t["fetchvars x2 (all members)"] = time.perf_counter() - c0
    c0 = time.perf_counter()
    core.reset(1745)
    core.run(2300)
    t["reset + run again"] = time.perf_counter() - c0
    # the calibration loop: new parameter values for every member, then the whole job again
    rng = np.random.default_rng(7)
    k = 5
    c0 = time.perf_counter()
    for _ in range(k):
        core.setvar("S", S * rng.uniform(0.99, 1.01, n), "degC")
        core.setvar("q10_rh", q10 * rng.uniform(0.99, 1.01, n), "(unitless)")
        core.reset(1745)
        core.run(2300)
        core.ensemble_stats(["CO2_concentration", "global_tas"], (1745, 2300))
    t["setvar x2 + reset + run + stats (mean of %d)" % k] = (time.perf_counter() - c0) / k
    # ... with a parameter the spinup sees (npp_flux0 per member): every member spins up again
    c0 = time.perf_counter()
    for _ in range(k):
        core.setvar("S", S * rng.uniform(0.99, 1.01, n), "degC")
        core.setvar("npp_flux0", rng.uniform(48.0, 60.0, n), "Pg C/yr")
        core.reset(1745)
        core.run(2300)
        core.ensemble_stats(["CO2_concentration", "global_tas"], (1745, 2300))
    t["... with npp_flux0 per member (mean of %d)" % k] = (time.perf_counter() - c0) / k
    t["  of it: spinup kernel (all members)"] = core.last_spinup_ms() * 1e-3
    t["  of it: year-loop kernel (3 rows vary)"] = core.last_run_ms() * 1e-3
    c0 = time.perf_counter()
    core.shutdown()
    t["shutdown"] = time.perf_counter() - c0
    return t, out_bufs


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 65536]
    print("| members | verb | first job ms | second job ms (buffers reused) |")
    print("|---|---|---|---|")
    for n in sizes:
        t1, bufs = job(n)
        t2, _ = job(n, bufs)
        for k in t1:
            print("| %d | %s | %.2f | %.2f |" % (n, k, t1[k] * 1e3, t2[k] * 1e3))
        tot1 = sum(v for k, v in t1.items() if not k.startswith("  "))
        tot2 = sum(v for k, v in t2.items() if not k.startswith("  "))
        print("| %d | total | %.2f | %.2f |" % (n, tot1 * 1e3, tot2 * 1e3))


if __name__ == "__main__":
    main()
