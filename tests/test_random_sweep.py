"""Randomised parity sweep: every perturbable parameter at once, on every shipped scenario,
member by member against the oracle -- values AND decisions (stash schedule)."""
import os

import numpy as np
import pytest

import hector_amd
from conftest import ROOT

RANGES = {   # capability: (low, high, unit)
    "S": (1.5, 6.0, "degC"), "diff": (0.6, 3.0, "cm2/s"), "aero_scalar": (0.3, 1.7, None),
    "vol_scalar": (0.5, 1.5, None), "C0": (265.0, 290.0, "ppmv CO2"),
    "tt": (5.5e7, 9e7, "m3/s"), "tu": (4e7, 6e7, "m3/s"), "twi": (1e7, 1.6e7, "m3/s"),
    "tid": (1.5e8, 2.5e8, "m3/s"), "beta": (0.2, 0.9, None), "q10_rh": (1.1, 3.0, None),
    "warmingfactor": (0.8, 1.6, None), "f_nppv": (0.3, 0.4, None), "f_nppd": (0.5, 0.6, None),
    "f_litterd": (0.9, 1.0, None), "rh_ch4_frac": (0.01, 0.04, None), "pf_mu": (1.4, 1.9, "degC"),
    "pf_sigma": (0.8, 1.1, "degC"), "fpf_static": (0.6, 0.85, None),
}
ORACLE_FIELD = {"S": "S", "diff": "diff", "aero_scalar": "aero_scalar", "vol_scalar": "vol_scalar",
                "C0": "C0", "tt": "tt", "tu": "tu", "twi": "twi", "tid": "tid"}
SCENARIOS = ["ssp119", "ssp245", "ssp370", "ssp534-over", "ssp585", "picontrol"]


def sweep(lib, n, seed, check_every=1, **kw):
    import oracle_binding
    rng = np.random.default_rng(seed)
    worst = {}
    for name in SCENARIOS:
        path = os.path.join(ROOT, "hector_amd", "data", name + ".hxs")
        vals = {k: rng.uniform(lo, hi, n) for k, (lo, hi, _) in RANGES.items()}
        c = hector_amd.Core(path, n, lib_path=lib, **kw)
        for k, (lo, hi, unit) in RANGES.items():
            c.setvar(k, vals[k], unit)
        outs = ["CO2_concentration", "global_tas", "RF_tot", "NBP", "ocean_c", "timesteps"]
        c.set_outputs(outs); c.run(2300)
        st = c.status()
        got = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
        o = oracle_binding.Oracle(path)
        for i in range(0, n, check_every):
            p = o.default_params()
            for k in RANGES:
                if k in ORACLE_FIELD: setattr(p, ORACLE_FIELD[k], vals[k][i])
                else: getattr(p, k)[0] = vals[k][i]
            r, err, _ = o.run(p)
            assert (err != 0) == (st[i] != 0), (name, i, err, st[i])
            if err:
                continue
            for v, tol in [("CO2_concentration", 2e-8), ("global_tas", 2e-8), ("RF_tot", 2e-8),
                           ("NBP", 2e-7), ("ocean_c", 2e-8)]:
                y0 = 1 if v == "NBP" else 0      # (no NBP is recorded at startDate)
                d = np.abs(got[v][y0:, i] - r[v][y0:]).max() / max(1.0, np.abs(r[v]).max())
                worst[v] = max(worst.get(v, 0.0), d)
                assert d < tol, (name, i, v, d)
            assert np.array_equal(got["timesteps"][1:, i], r["timesteps"][1:]), (name, i)
    return worst


def test_random_parameter_sweep(emul_lib):
    sweep(emul_lib, 6, seed=11, allow_emulation=True)


@pytest.mark.gpu
def test_random_parameter_sweep_on_gpu(hip_lib):
    worst = sweep(hip_lib, 96, seed=12, check_every=3, device=0)
    print("worst relative deviations:", worst)
