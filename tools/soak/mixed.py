import sys, os, time, tempfile, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
import test_random_sweep as T
gpu = "--gpu" in sys.argv
t0 = time.time()
tmp = tempfile.mkdtemp(dir=R + "/gpurun_out" if gpu else None)
for seed in ((511, 512, 513) if gpu else (601,)):
    if gpu:
        w = T.sweep_mixed(R + "/hector_amd/lib/libhector_amd.so", 512, seed=seed, rounds=40, tmpdir=tmp, check_every=16, device=0)
    else:
        w = T.sweep_mixed(R + "/tests/emul/libhector_amd_emul.so", 4, seed=seed, rounds=60, tmpdir=tmp, allow_emulation=True)
    print("seed", seed, {k: "%.2e" % v for k, v in w.items()}, "%.0fs" % (time.time() - t0), flush=True)
