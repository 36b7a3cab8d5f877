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


# Tolerances.  The per-variable tolerances below are what the kernels hold against the oracle for
# a member whose trajectory is well conditioned (the north star asks for 1e-6).  Some random
# members are not: with one ocean stash a year (explicit Euler, oceanbox.cpp:240-303) and extreme
# exchange coefficients the high-latitude box amplifies any rounding difference ~1.09x a year for
# decades -- in the reference's own arithmetic just as well; and the alkalinity tuner (Brent on a
# V-shaped objective, "last point evaluated" kept) has branch decisions that are ties up to the
# last bit, after which it ends ~3e-6 of the alkalinity away.  For a member that misses its
# tolerance the oracle is therefore run again with rounding-sized noise injected (every pool
# times 1 +- 1e-13 once a year and every flux the tuner evaluates, hxo_set_rounding_noise): the
# member passes if the kernel is within 50x of what that does to the oracle itself, and never
# beyond HARD_TOL (the tuner's own resolution is 3e-6 of the alkalinity = ~3e-5 in CO2).
HARD_TOL = 1e-4
ROOT_FAILED = 32   # HXO_ERR_ROOT
NOISE = 1e-13


def self_sensitivity(o, p, names):
    """Largest move of the oracle's own trajectory under rounding-sized noise.  Several noise
    amplitudes: some sensitivities are ties broken by the last bit (Brent's parabola test when a
    parabolic step did not improve: the same parabola is fitted again and its minimum IS the
    bracket end), so one noisy run shows them only half of the time."""
    base, err, _ = o.run(p)
    out = {v: 0.0 for v in names}
    try:
        for amp in (1.0, -1.0, 1.7, -2.3, 3.1):
            o.set_rounding_noise(NOISE * amp)
            pert, err2, _ = o.run(p)
            for v in names:
                y0 = 1 if v == "NBP" else 0
                d = np.abs(pert[v][y0:] - base[v][y0:]).max() / max(1.0, np.abs(base[v]).max())
                out[v] = max(out[v], d)
    finally:
        o.set_rounding_noise(0.0)
    return out


def check_member(o, p, d_by_var, tol_by_var, where, ill):
    """d_by_var: {oracle variable: relative deviation of the kernel}."""
    bad = [v for v, d in d_by_var.items() if not d < tol_by_var[v]]
    if not bad:
        return
    sens = self_sensitivity(o, p, bad)
    for v in bad:
        allowed = min(HARD_TOL, max(tol_by_var[v], 50.0 * sens[v]))
        assert d_by_var[v] < allowed, (where, v, d_by_var[v], "oracle self-sensitivity", sens[v])
    ill.append((where, {v: (d_by_var[v], sens[v]) for v in bad}))


def sweep(lib, n, seed, check_every=1, pair=False, **kw):
    """pair=True: the ensemble has to run on the two-wavefront kernel (it records all of these
    outputs; per-member diffusivity included)."""
    import oracle_binding
    rng = np.random.default_rng(seed)
    worst, ill = {}, []
    for name in SCENARIOS:
        path = os.path.join(ROOT, "hector_amd", "data", name + ".hxs")
        vals = {k: rng.uniform(lo, hi, n) for k, (lo, hi, _) in RANGES.items()}
        c = hector_amd.Core(path, n, lib_path=lib, **kw)
        for k, (lo, hi, unit) in RANGES.items():
            c.setvar(k, vals[k], unit)
        outs = ["CO2_concentration", "global_tas", "RF_tot", "NBP", "ocean_c", "timesteps"]
        c.set_outputs(outs); c.run(2300)
        if pair and name != "picontrol":   # (picontrol prescribes its CO2: run kernel)
            assert c.last_run_kernel() == "pair"
        st = c.status()
        got = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
        o = oracle_binding.Oracle(path)
        for i in range(0, n, check_every):
            p = o.default_params()
            for k in RANGES:
                if k in ORACLE_FIELD: setattr(p, ORACLE_FIELD[k], vals[k][i])
                else: getattr(p, k)[0] = vals[k][i]
            r, err, _ = o.run(p)
            if not err and not np.isfinite(r["CO2_concentration"]).all():
                err = ROOT_FAILED   # the oracle ran into NaNs without noticing: the kernels flag that
                assert st[i] != 0, (name, i, "NaN in the oracle, no flag in the kernel")
                continue
            if err & ROOT_FAILED and not st[i]:
                # far outside any plausible state (CO2 of thousands of ppm, surface pH > 11) the
                # reference's root iteration from the polynomial bound can bisect past the root
                # and give up; the kernels' bracketed Newton still finds it and carries on
                continue
            assert (err != 0) == (st[i] != 0), (name, i, err, st[i])
            if err:
                continue
            tols = {"CO2_concentration": 2e-8, "global_tas": 2e-8, "RF_tot": 2e-8, "NBP": 2e-7,
                    "ocean_c": 2e-8}
            tols = {v: t for v, t in tols.items() if v in outs}
            dev = {}
            for v in tols:
                y0 = 1 if v == "NBP" else 0      # (no NBP is recorded at startDate)
                dev[v] = np.abs(got[v][y0:, i] - r[v][y0:]).max() / max(1.0, np.abs(r[v]).max())
                worst[v] = max(worst.get(v, 0.0), dev[v])
            check_member(o, p, dev, tols, (name, i), ill)
            assert np.array_equal(got["timesteps"][1:, i], r["timesteps"][1:]), (name, i)
    worst["ill_conditioned_members"] = len(ill)
    return worst


def test_random_parameter_sweep(emul_lib):
    sweep(emul_lib, 6, seed=11, allow_emulation=True)


@pytest.mark.gpu
def test_random_parameter_sweep_on_gpu(hip_lib):
    worst = sweep(hip_lib, 96, seed=12, check_every=3, device=0)
    print("worst relative deviations:", worst)


@pytest.mark.gpu
def test_random_parameter_sweep_pair_kernel_on_gpu(hip_lib):
    """The same on the two-wavefront kernel: 18 parameters at once (all but the diffusivity,
    which would give every member its own DOECLIM table and the run kernel)."""
    worst = sweep(hip_lib, 192, seed=13, check_every=3, pair=True, device=0)
    print("worst relative deviations (pair kernel):", worst)


BIOME_KEYS = ["beta", "q10_rh", "warmingfactor", "f_nppv", "f_nppd", "f_litterd", "rh_ch4_frac",
              "pf_mu", "pf_sigma", "fpf_static"]
POOLS = ["veg_c", "detritus_c", "soil_c", "permafrost_c", "npp_flux0"]


def sweep_biomes(lib, n, seed, scenarios=("ssp245", "ssp585"), check_every=1, counts=(2, 3, 4), **kw):
    """2, 3 and 4 biomes (or `counts`) with random (per-member) pool splits and every per-biome
    parameter perturbed independently, plus the global ones: member by member against the oracle."""
    import oracle_binding
    rng = np.random.default_rng(seed)
    worst, ill = {}, []
    for name in scenarios:
        path = os.path.join(ROOT, "hector_amd", "data", name + ".hxs")
        o = oracle_binding.Oracle(path)
        for B in counts:
            names = ["b%d" % b for b in range(B)]
            c = hector_amd.Core(path, n, lib_path=lib, **kw)
            base = {k: c.getvar(k)[0] for k in POOLS}
            c.split_biome(names)
            glob = {k: rng.uniform(lo, hi, n) for k, (lo, hi, _) in RANGES.items() if k in ORACLE_FIELD}
            for k, v in glob.items():
                c.setvar(k, v, RANGES[k][2])
            # per-member split of every pool (Dirichlet), independent per pool
            frac = {k: rng.dirichlet(np.full(B, 3.0), n) for k in POOLS}
            per = {k: rng.uniform(RANGES[k][0], RANGES[k][1], (n, B)) for k in BIOME_KEYS}
            for b, nm in enumerate(names):
                for k in POOLS:
                    c.setvar("%s.%s" % (nm, k), base[k] * frac[k][:, b])
                for k in BIOME_KEYS:
                    c.setvar("%s.%s" % (nm, k), per[k][:, b])
            kb = min(B - 1, 3)   # (the oracle reports the pools of its first four biomes)
            outs = ["CO2_concentration", "global_tas", "NBP", "veg_c", "soil_c", "permafrost_c",
                    "timesteps", names[kb] + ".soil_c"]
            c.set_outputs(outs); c.run(2300)
            st = c.status()
            got = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
            for i in range(0, n, check_every):
                p = o.default_params()
                p.nbiome = B
                for k, v in glob.items(): setattr(p, ORACLE_FIELD[k], v[i])
                for b in range(B):
                    for k in POOLS: getattr(p, k)[b] = base[k] * frac[k][i, b]
                    for k in BIOME_KEYS: getattr(p, k)[b] = per[k][i, b]
                r, err, _ = o.run(p)
                assert (err != 0) == (st[i] != 0), (name, B, i, err, st[i])
                if err:
                    continue
                last = "b%d.soil_c" % kb   # the oracle's name of that biome's soil pool
                tols = {"CO2_concentration": 2e-8, "global_tas": 2e-8, "NBP": 2e-7, "veg_c": 2e-8,
                        "soil_c": 2e-8, "permafrost_c": 2e-8, last: 2e-8}
                dev = {}
                for v in tols:
                    mine = got[names[kb] + ".soil_c"] if v == last else got[v]
                    y0 = 1 if v == "NBP" else 0
                    dev[v] = np.abs(mine[y0:, i] - r[v][y0:]).max() / max(1.0, np.abs(r[v]).max())
                    key = "biome.soil_c" if v == last else v
                    worst[key] = max(worst.get(key, 0.0), dev[v])
                check_member(o, p, dev, tols, (name, B, i), ill)
                assert np.array_equal(got["timesteps"][1:, i], r["timesteps"][1:]), (name, B, i)
    worst["ill_conditioned_members"] = len(ill)
    for w in ill:
        print("ill-conditioned member", w)
    return worst


def test_random_biome_sweep(emul_lib):
    sweep_biomes(emul_lib, 3, seed=21, scenarios=("ssp245",), allow_emulation=True)
    # the unrolled five- to eight-biome kernels (lean and slim parks) and the first looped count
    sweep_biomes(emul_lib, 2, seed=22, scenarios=("ssp245",), counts=(5, 7, 8, 9), allow_emulation=True)


@pytest.mark.gpu
def test_random_biome_sweep_on_gpu(hip_lib):
    worst = sweep_biomes(hip_lib, 64, seed=22, check_every=4, device=0)
    print("worst relative deviations:", worst)


CONSTRAINTS = {   # capability: (pack section, unit, value(year index k, rng) )
    "CO2_constrain": ("simpleNbox", "ppmv CO2", lambda k, u: 300.0 + 1.5 * k + 3.0 * u),
    "NBP_constrain": ("simpleNbox", "Pg C/yr", lambda k, u: 0.3 + 0.01 * k + 0.5 * u),
    "tas_constrain": ("temperature", "degC", lambda k, u: 0.4 + 0.02 * k + 0.1 * u),
    "RF_tot_constrain": ("forcing", "W/m2", lambda k, u: 0.5 + 0.03 * k + 0.3 * u),
    "CH4_constrain": ("CH4", "ppbv CH4", lambda k, u: 900.0 + 8.0 * k + 20.0 * u),
}


def sweep_mixed(lib, n, seed, rounds, tmpdir, check_every=1, **kw):
    """Everything at once: a random scenario, 1-4 biomes with per-member pool splits, every
    parameter perturbed, a land-ocean warming ratio for a third of the members and one random
    constraint over a random window -- kernels vs the oracle reading the edited scenario."""
    import oracle_binding
    from conftest import edited_pack
    rng = np.random.default_rng(seed)
    worst, ill = {}, []
    names_all = ["picontrol", "ssp119", "ssp126", "ssp245", "ssp370", "ssp434", "ssp460",
                 "ssp534-over", "ssp585"]
    for rd in range(rounds):
        name = names_all[rng.integers(len(names_all))]
        path = os.path.join(ROOT, "hector_amd", "data", name + ".hxs")
        B = int(rng.integers(1, 5))
        kind = [None] + sorted(CONSTRAINTS)
        kind = kind[rng.integers(len(kind))]
        y0 = int(rng.integers(1800, 2000)); y1 = y0 + int(rng.integers(5, 120))
        years = np.arange(y0, y1 + 1)
        c = hector_amd.Core(path, n, lib_path=lib, **kw)
        base = {k: c.getvar(k)[0] for k in POOLS}
        bn = ["b%d" % b for b in range(B)]
        if B > 1:
            c.split_biome(bn)
        glob = {k: rng.uniform(lo, hi, n) for k, (lo, hi, _) in RANGES.items() if k in ORACLE_FIELD}
        for k, v in glob.items():
            c.setvar(k, v, RANGES[k][2])
        lo_ratio = np.where(rng.uniform(size=n) < 0.33, rng.uniform(1.1, 1.8, n), 0.0)
        c.setvar("lo_warming_ratio", lo_ratio)
        frac = {k: rng.dirichlet(np.full(B, 3.0), n) for k in POOLS}
        per = {k: rng.uniform(RANGES[k][0], RANGES[k][1], (n, B)) for k in BIOME_KEYS}
        for b in range(B):
            pre = (bn[b] + ".") if B > 1 else ""
            for k in POOLS:
                c.setvar(pre + k, base[k] * frac[k][:, b])
            for k in BIOME_KEYS:
                c.setvar(pre + k, per[k][:, b])
        opath = path
        if kind:
            sec, unit, fn = CONSTRAINTS[kind]
            vals = np.array([fn(k, rng.uniform()) for k in range(years.size)])
            c.setvar_dated(kind, years, vals, unit)
            oy, ov = years, vals
            if kind == "RF_tot_constrain":  # holds for every date up to its last one, flat before
                oy = np.arange(1745, y1 + 1)     # its first (forcing_component.cpp:498-505); the
                ov = np.concatenate([np.full(y0 - 1745, vals[0]), vals])  # pack is dense
            opath = edited_pack(os.path.join(str(tmpdir), "mixed_%d_%d.hxs" % (seed, rd)), sec, kind,
                                oy, ov, base=path)
        outs = ["CO2_concentration", "global_tas", "NBP", "land_tas", "timesteps"]
        c.set_outputs(outs); c.run(2300)
        st = c.status()
        got = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
        o = oracle_binding.Oracle(opath)
        for i in range(0, n, check_every):
            p = o.default_params()
            p.nbiome = B
            p.lo_warming_ratio = lo_ratio[i]
            for k, v in glob.items(): setattr(p, ORACLE_FIELD[k], v[i])
            for b in range(B):
                for k in POOLS: getattr(p, k)[b] = base[k] * frac[k][i, b]
                for k in BIOME_KEYS: getattr(p, k)[b] = per[k][i, b]
            r, err, _ = o.run(p)
            where = (name, B, kind, y0, y1, i)
            assert (err != 0) == (st[i] != 0), (where, err, st[i])
            if err:
                continue
            tols = {"CO2_concentration": 2e-8, "global_tas": 2e-8, "NBP": 2e-7, "land_tas": 2e-8}
            dev = {}
            for v in tols:
                y_0 = 1 if v == "NBP" else 0
                dev[v] = np.abs(got[v][y_0:, i] - r[v][y_0:]).max() / max(1.0, np.abs(r[v]).max())
                worst[v] = max(worst.get(v, 0.0), dev[v])
            check_member(o, p, dev, tols, where, ill)
            assert np.array_equal(got["timesteps"][1:, i], r["timesteps"][1:]), where
    worst["ill_conditioned_members"] = len(ill)
    for w in ill:
        print("ill-conditioned member", w)
    return worst


def test_random_mixed_sweep(emul_lib, tmp_path):
    sweep_mixed(emul_lib, 3, seed=31, rounds=6, tmpdir=tmp_path, allow_emulation=True)


@pytest.mark.gpu
def test_random_mixed_sweep_on_gpu(hip_lib, tmp_path):
    worst = sweep_mixed(hip_lib, 48, seed=32, rounds=10, tmpdir=tmp_path, check_every=4, device=0)
    print("worst relative deviations:", worst)
