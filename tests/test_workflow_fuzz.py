"""Randomised workflows on one core -- run in pieces, reset to any date, edit emissions and
parameters in between (R/messages.R:107-140 auto-reset; core.cpp:511-549 reset) -- must end on
exactly what a fresh run with the final inputs gives: the oracle reading the final scenario."""
import os

import numpy as np
import pytest

import hector_amd
from conftest import ROOT, edited_pack


def pack_series(path, section, key):
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) > 5 and p[0] == "series" and p[1] == section and p[2] == key:
                return int(p[3]), np.array([float(x) for x in p[5:5 + int(p[4])]])
    raise KeyError(key)


def workflow_fuzz(lib, seed, rounds, tmpdir, n=3, **kw):
    import oracle_binding
    rng = np.random.default_rng(seed)
    names = ["ssp119", "ssp245", "ssp370", "ssp585"]
    worst = 0.0
    for rd in range(rounds):
        name = names[rng.integers(len(names))]
        path = os.path.join(ROOT, "hector_amd", "data", name + ".hxs")
        B = int(rng.choice([1, 2, 4]))
        c = hector_amd.Core(path, n, lib_path=lib, **kw)
        c.enable_history(True)
        if B > 1:
            c.split_biome(["b%d" % b for b in range(B)])
        S = rng.uniform(2.0, 5.0, n); q10 = rng.uniform(1.2, 2.8, (B, n))
        c.setvar("S", S, "degC")
        for b in range(B):
            c.setvar(("b%d." % b if B > 1 else "") + "q10_rh", q10[b])
        outs = ["CO2_concentration", "global_tas", "veg_c", "timesteps"]
        c.set_outputs(outs)
        y0s, ffi = pack_series(path, "simpleNbox", "ffi_emissions")
        _, luc = pack_series(path, "simpleNbox", "luc_emissions")
        log = []
        for op in range(int(rng.integers(3, 7))):
            kind = rng.choice(["run", "reset", "ffi", "luc", "param"])
            if kind == "run":
                y = int(rng.integers(max(c.current_date, 1746), 2301))
                c.run(y); log.append(("run", y))
            elif kind == "reset":
                if c.current_date <= 1746:
                    continue
                y = int(rng.integers(1745, c.current_date + 1))
                c.reset(y); log.append(("reset", y))
            elif kind in ("ffi", "luc"):
                a = int(rng.integers(1760, 2250)); b_ = a + int(rng.integers(1, 50))
                yrs = np.arange(a, b_ + 1)
                ser = ffi if kind == "ffi" else luc
                ser[yrs - y0s] = ser[yrs - y0s] * rng.uniform(0.5, 1.5) + rng.uniform(0, 0.2)
                c.setvar_dated(kind + "_emissions", yrs, ser[yrs - y0s], "Pg C/yr")
                log.append((kind, a, b_))
            else:   # a parameter change invalidates everything (reset to 0 + spinup)
                S = rng.uniform(2.0, 5.0, n)
                c.setvar("S", S, "degC"); log.append(("S",))
        c.run(2300)
        assert (c.status() == 0).all(), log
        allyears = np.arange(y0s, y0s + ffi.size)
        p1 = edited_pack(os.path.join(str(tmpdir), "wf_%d_%d_a.hxs" % (seed, rd)), "simpleNbox",
                         "ffi_emissions", allyears, ffi, base=path)
        p2 = edited_pack(os.path.join(str(tmpdir), "wf_%d_%d_b.hxs" % (seed, rd)), "simpleNbox",
                         "luc_emissions", allyears, luc, base=p1)
        o = oracle_binding.Oracle(p2)
        for i in range(n):
            p = o.default_params()
            if B > 1:
                p = o.split_equal(p, B)
            p.S = S[i]
            for b in range(B):
                p.q10_rh[b] = q10[b][i]
            r, err, _ = o.run(p)
            assert err == 0
            for v in ("CO2_concentration", "global_tas", "veg_c"):
                d = np.abs(c.fetchvars(v, (1745, 2300))[:, i] - r[v]).max() / max(1.0, np.abs(r[v]).max())
                worst = max(worst, d)
                assert d < 2e-8, (name, B, log, i, v, d)
            assert np.array_equal(c.fetchvars("timesteps", (1746, 2300))[:, i], r["timesteps"][1:]), log
    return worst


def test_workflow_fuzz(emul_lib, tmp_path):
    workflow_fuzz(emul_lib, seed=5, rounds=8, tmpdir=tmp_path, allow_emulation=True)


@pytest.mark.gpu
def test_workflow_fuzz_on_gpu(hip_lib, tmp_path):
    print("worst relative deviation:", workflow_fuzz(hip_lib, seed=6, rounds=12, tmpdir=tmp_path, n=70, device=0))
