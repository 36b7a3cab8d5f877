import sys, os, time, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
import test_random_sweep as T
T.SCENARIOS = ["picontrol", "ssp119", "ssp126", "ssp245", "ssp370", "ssp434", "ssp460", "ssp534-over", "ssp585"]
t0 = time.time()
for seed in (101, 202):
    gpu = "--gpu" in sys.argv
    if gpu: w = T.sweep(R + "/hector_amd/lib/libhector_amd.so", 2048, seed=seed, check_every=16, device=0,
                        pair="--pair" in sys.argv)  # --pair: the two-wavefront kernel's configuration
    else: w = T.sweep(R + "/tests/emul/libhector_amd_emul.so", 16, seed=seed, allow_emulation=True)
    print("seed", seed, {k: "%.2e" % v for k, v in w.items()}, "%.0fs" % (time.time() - t0), flush=True)
