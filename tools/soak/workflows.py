import sys, os, time, tempfile
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
import test_workflow_fuzz as W
gpu = "--gpu" in sys.argv
tmp = tempfile.mkdtemp()
t0 = time.time()
for seed in (711, 712):
    if gpu: w = W.workflow_fuzz(R + "/hector_amd/lib/libhector_amd.so", seed, 40, tmp, n=70, device=0)
    else: w = W.workflow_fuzz(R + "/tests/emul/libhector_amd_emul.so", seed, 60, tmp, n=2, allow_emulation=True)
    print("seed", seed, "worst %.2e" % w, "%.0fs" % (time.time() - t0), flush=True)
