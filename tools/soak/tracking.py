# tracking fuzz: random tracking date / biomes / params, run in pieces with resets, vs oracle.run_tracking
import sys, os, time, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
import hector_amd, oracle_binding
gpu = "--gpu" in sys.argv
kw = dict(device=0) if gpu else dict(lib_path=R + "/tests/emul/libhector_amd_emul.so", allow_emulation=True)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 9)
worst = [0.0, 0.0]
for rd in range(int(os.environ.get("ROUNDS", "12"))):
    name = ["ssp119", "ssp245", "ssp585"][rng.integers(3)]
    path = os.path.join(R, "hector_amd", "data", name + ".hxs")
    B = int(rng.choice([1, 2, 4])); n = 3
    T0 = int(rng.integers(1750, 2050)); END = int(rng.integers(T0 + 5, 2301))
    c = hector_amd.Core(path, n, **kw); c.enable_history(True)
    if B > 1: c.split_biome(["b%d" % b for b in range(B)])
    c.setvar("trackingDate", [T0])
    S = rng.uniform(2, 5, n); q10 = rng.uniform(1.2, 2.8, (B, n)); wf = rng.uniform(0.8, 1.6, (B, n))
    c.setvar("S", S, "degC")
    for b in range(B):
        pre = "b%d." % b if B > 1 else ""
        c.setvar(pre + "q10_rh", q10[b]).setvar(pre + "warmingfactor", wf[b])
    # run in pieces with a reset in between
    y1 = int(rng.integers(1746, END + 1)); c.run(y1)
    if y1 > 1750 and rng.uniform() < 0.7:
        c.reset(int(rng.integers(1745, y1)))
    c.run(END)
    assert (c.status() == 0).all()
    o = oracle_binding.Oracle(path)
    for i in range(n):
        p = o.default_params()
        if B > 1: p = o.split_equal(p, B)
        p.S = S[i]
        for b in range(B): p.q10_rh[b] = q10[b][i]; p.warmingfactor[b] = wf[b][i]
        ov, of, _, err = o.run_tracking(p, T0, END); assert err == 0
        gv, gf = c.tracking_data(i, (T0, END))
        k0, k1 = T0 - 1745, END - 1745 + 1
        dv = np.abs(gv - ov[k0:k1]).max() / np.abs(ov).max(); df = np.abs(gf - of[k0:k1]).max()
        worst = [max(worst[0], dv), max(worst[1], df)]
        assert dv < 1e-10 and df < 1e-7, (name, B, T0, END, i, dv, df)
        assert np.abs(gf.sum(2) - 1).max() < 1e-12
print("tracking fuzz ok: worst value dev %.2e fraction dev %.2e" % tuple(worst))
