"""Small ensembles on the pair kernel with per-member ocean diffusivity and / or the heat-flux sum:
    python tools/prof/pair_diff_times.py [--lib=path]"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import bench, hector_amd
from hector_amd import ensemble
libs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--lib=")]
if libs:   # an experiment build (gpuwork/)
    _orig = hector_amd.Core
    hector_amd.Core = lambda *a, **k: _orig(*a, **dict(k, lib_path=os.path.abspath(libs[0])))
for n in (1024, 16384):
    for diff in (False, True):
        for hf in (False, True):
            c = bench.make_core(n, 1, 0, 0)
            if diff:
                c.setvar("diff", 1.2 + 2.2 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 5), "cm2/s")
            c.set_outputs(["CO2_concentration", "global_tas"] + (["heatflux"] if hf else []))
            ms = []
            for _ in range(5):
                c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
            print("%6d members diff=%d heatflux=%d: %s best %.3f ms" % (n, diff, hf, c.last_run_kernel(), min(ms[1:])), flush=True)
            c.shutdown()
