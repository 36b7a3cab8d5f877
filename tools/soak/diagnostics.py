import sys, os, time, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
import hector_amd, oracle_binding
from test_random_sweep import RANGES, ORACLE_FIELD
from test_diagnostics import KERNEL_VARS, DERIVED_VARS, HOST_VARS, ORACLE_NAME
gpu = "--gpu" in sys.argv
kw = dict(device=0) if gpu else dict(lib_path=R + "/tests/emul/libhector_amd_emul.so", allow_emulation=True)
n = 256 if gpu else 6
step = 8 if gpu else 1
worst = {}
for seed, name in [(1, "ssp245"), (2, "ssp585"), (3, "ssp119"), (4, "ssp534-over"), (5, "picontrol")]:
    rng = np.random.default_rng(seed)
    path = os.path.join(R, "hector_amd", "data", name + ".hxs")
    vals = {k: rng.uniform(lo, hi, n) for k, (lo, hi, _) in RANGES.items()}
    lo_ratio = np.where(rng.uniform(size=n) < 0.3, rng.uniform(1.1, 1.8, n), 0.0)
    c = hector_amd.Core(path, n, **kw)
    for k, (lo, hi, unit) in RANGES.items(): c.setvar(k, vals[k], unit)
    c.setvar("lo_warming_ratio", lo_ratio)
    allv = KERNEL_VARS + DERIVED_VARS
    c.set_outputs(allv + ["RF_tot", "RF_CO2", "global_tas", "CO2_concentration", "land_tas", "sst"]); c.run(2300)
    st = c.status()
    got = {v: c.fetchvars(v, (1746, 2300)) for v in allv + HOST_VARS + ["land_tas", "sst", "CO2_concentration"]}
    o = oracle_binding.Oracle(path)
    for i in range(0, n, step):
        p = o.default_params(); p.lo_warming_ratio = lo_ratio[i]
        for k in RANGES:
            if k in ORACLE_FIELD: setattr(p, ORACLE_FIELD[k], vals[k][i])
            else: getattr(p, k)[0] = vals[k][i]
        r, err, _ = o.run(p)
        assert (err != 0) == (st[i] != 0)
        if err: continue
        for v in got:
            ref = r[ORACLE_NAME.get(v, v)][1:]
            d = np.abs(got[v][:, i] - ref).max() / max(1.0, np.abs(ref).max())
            if d > worst.get(v, (0,))[0]: worst[v] = (d, name, i)
    print(name, "done", flush=True)
for v, w in sorted(worst.items(), key=lambda kv: -kv[1][0])[:12]: print(v, "%.2e" % w[0], w[1:])
