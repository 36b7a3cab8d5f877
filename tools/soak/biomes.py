import sys, os, time, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
import test_random_sweep as T
t0 = time.time()
scen = ("picontrol", "ssp119", "ssp126", "ssp245", "ssp370", "ssp434", "ssp460", "ssp534-over", "ssp585")
for seed in (313, 414):
    if "--gpu" in sys.argv:
        w = T.sweep_biomes(R + "/hector_amd/lib/libhector_amd.so", 1024, seed=seed, scenarios=scen, check_every=8, device=0)
    else:
        w = T.sweep_biomes(R + "/tests/emul/libhector_amd_emul.so", 8, seed=seed, scenarios=scen, allow_emulation=True)
    print("seed", seed, {k: "%.2e" % v for k, v in w.items()}, "%.0fs" % (time.time() - t0), flush=True)
