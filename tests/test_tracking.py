"""Carbon tracking (fluxpool origin maps, inst/include/fluxpool.hpp; SimpleNbox::stashCValues
src/simpleNbox-runtime.cpp:289-540; oceanbox.cpp:240-303; CSVFluxPoolVisitor) on the kernel
source compiled for the host, against the oracle's restatement, plus the properties
tests/testthat/test_tracking.R states.  The GPU suite repeats the comparison through HIP."""
import numpy as np
import pytest

import hector_amd
from conftest import SCENARIO

T0, END = 1770, 2100
FRAC_TOL = 1e-8


def tracked_core(lib, n, date=T0, **kw):
    c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
    c.setvar("trackingDate", [date])
    return c


def check_tracking_vs_oracle(lib, oracle, **kw):
    c = tracked_core(lib, 3, **kw)
    q10 = c.getvar("q10_rh"); q10[1] *= 0.9; q10[2] = 2.6
    S = np.array([3.0, 3.0, 4.5])
    c.setvar("q10_rh", q10).setvar("S", S, "degC")
    POOL_VARS = {"atmos_co2": "atmos_co2", "earth_c": "earth_c", "veg_c": "veg_c",
                 "detritus_c": "detritus_c", "soil_c": "soil_c", "HL": "HL_ocean_c",
                 "LL": "LL_ocean_c", "intermediate": "IO_ocean_c", "deep": "DO_ocean_c"}
    c.set_outputs(["CO2_concentration"] + list(POOL_VARS.values()))
    c.run(END)
    assert (c.status() == 0).all()
    assert c.getvar("trackingDate")[0] == T0
    names = c.tracking_pools()
    assert names == ["atmos_co2", "earth_c", "veg_c", "detritus_c", "soil_c", "permafrost_c",
                     "thawedp_c", "HL", "LL", "intermediate", "deep"]
    for i in range(3):
        p = oracle.default_params(); p.q10_rh[0] = q10[i]; p.S = S[i]
        ov, of, _, err = oracle.run_tracking(p, T0, END)
        assert err == 0
        gv, gf = c.tracking_data(i, (T0, END))
        k0, k1 = T0 - 1745, END - 1745 + 1
        # pools to ~1e-11 relative like every other output; fractions are ratios of sums of
        # fluxes, and in small pools (thawed permafrost, detritus) the 1e-11 differences of the
        # fluxes are amplified: FRAC_TOL absolute on numbers in [0, 1]
        assert np.abs(gv - ov[k0:k1]).max() < 1e-10 * np.abs(ov).max()
        assert np.abs(gf - of[k0:k1]).max() < FRAC_TOL
        # every pool's origins sum to 1 (test_tracking.R:96-101)
        assert np.abs(gf.sum(axis=2) - 1.0).max() < 1e-12
        # first tracked year: mostly itself; the pool values are the model's (test_tracking.R:163-)
        assert gf[0, 1, 1] == 1.0
        # test_tracking.R:164-262 "Pool names are valid": the value tracked for each of the nine
        # pools equals fetchvars(<pool variable>) (the R test allows 1e-3; here every year, exactly)
        # and carries its unit
        for pool, var in POOL_VARS.items():
            assert np.array_equal(gv[:, names.index(pool)], c.fetchvars(var, (T0, END))[:, i]), pool
            assert c.getunits(var) == "Pg C"
    # ... and the source names are the pool names (test_tracking.R:258-262)
    rows = hector_amd.get_tracking_data(c, member=0)
    assert sorted({r[5] for r in rows}) == sorted({r[2] for r in rows}) == sorted(names)
    return c


def test_tracking_vs_oracle(emul_lib, oracle):
    c = check_tracking_vs_oracle(emul_lib, oracle, allow_emulation=True)
    # a changed parameter changes the origins (test_tracking.R:30-50)
    _, f0 = c.tracking_data(0, (T0, END)); _, f1 = c.tracking_data(1, (T0, END))
    assert np.abs(f0 - f1).max() > 1e-4
    rows = hector_amd.get_tracking_data(c, 0)
    years = sorted({r[0] for r in rows})
    assert years == list(range(T0, END + 1))  # test_tracking.R:70-90
    assert {r[1] for r in rows} == {"simpleNbox", "ocean"} and {r[4] for r in rows} == {"Pg C"}


def test_tracking_changes_nothing_else(emul_lib):
    """Tracking only observes: the run itself is the untracked run (to the rounding differences
    between two instantiations of the run kernel)."""
    a = hector_amd.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    b = tracked_core(emul_lib, 2, allow_emulation=True)
    for c in (a, b):
        c.setvar("S", [2.5, 4.0], "degC"); c.run(2100)
    for v in ("CO2_concentration", "global_tas"):
        d = np.abs(a.fetchvars(v, (1745, 2100)) - b.fetchvars(v, (1745, 2100))).max()
        print(v, d)
        assert d < 1e-10


def test_no_tracking_is_empty(emul_lib):
    c = hector_amd.Core(SCENARIO, 1, lib_path=emul_lib, allow_emulation=True)
    c.run(1760)
    assert hector_amd.get_tracking_data(c) == []          # test_tracking.R:12-20
    assert c.getvar("trackingDate")[0] == 9999            # core.cpp:60
    with pytest.raises(hector_amd.HectorAmdError):
        c.tracking_data(0, (1750, 1760))
    # a tracking date that a run can never reach (simpleNbox-runtime.cpp:217: runToDate == tdate)
    c.setvar("trackingDate", [1745]); c.run(1760)
    assert c.getvar("trackingDate")[0] == 1745
    with pytest.raises(hector_amd.HectorAmdError):
        c.tracking_data(0, (1750, 1760))


def test_tracking_with_reset(emul_lib):
    tracking_reset_checks(emul_lib, allow_emulation=True)


def tracking_reset_checks(lib, **kw):
    """test_tracking.R:104-160: reset below the tracking date empties the record, a reset at or
    past it resumes with the maps of that date; a run-reset-run reproduces the straight run."""
    c = tracked_core(lib, 2, date=1760, **kw)
    c.enable_history(True)
    c.setvar("S", [3.0, 4.0], "degC")
    c.run(1800)
    v0, f0 = c.tracking_data(1, (1760, 1800))
    c.reset(1750)
    assert hector_amd.get_tracking_data(c) == []
    assert c.getvar("trackingDate")[0] == 1760
    c.run(1770)
    rows = hector_amd.get_tracking_data(c, 1)
    assert min(r[0] for r in rows) == 1760 and max(r[0] for r in rows) == 1770
    c.reset(1765)                       # >= trackingDate
    c.run(1800)
    v1, f1 = c.tracking_data(1, (1760, 1800))
    assert np.array_equal(v0, v1) and np.array_equal(f0, f1)
    c.reset(1760); c.run(1780); c.run(1800)
    v2, f2 = c.tracking_data(1, (1760, 1800))
    assert np.array_equal(v0, v2) and np.array_equal(f0, f2)
    with pytest.raises(hector_amd.HectorAmdError):
        c.tracking_data(1, (1750, 1800))


def test_tracking_four_biomes(emul_lib, oracle):
    tracking_four_biomes(emul_lib, oracle, allow_emulation=True)


def tracking_four_biomes(lib, oracle, **kw):
    tracking_n_biomes(lib, oracle, 4, **kw)


def tracking_n_biomes(lib, oracle, nb, run_to=2050, **kw):
    """`nb` equal biomes with their own Q10 per member: pools, fractions and which sources are in
    the maps, against the oracle.  5-16 biomes take the looped kernels (run-time pool count), 12
    and more have over 64 pools: two mask words per pool."""
    c = tracked_core(lib, 2, date=1800, **kw)
    names = ["b%d" % (b + 1) for b in range(nb)]
    c.split_biome(names)
    q = [[1.8 + 0.07 * ((5 * b) % 11), 2.6 - 0.05 * ((3 * b) % 13)] for b in range(nb)]
    for b, nm in enumerate(names):
        c.setvar(nm + ".q10_rh", q[b])
    c.run(run_to)
    assert (c.status() == 0).all()
    pools = c.tracking_pools()
    tp = 6 + 5 * nb
    assert len(pools) == tp and pools[2] == "b1.veg_c" and pools[tp - 5] == "b%d.thawedp_c" % nb
    k0, k1 = 1800 - 1745, run_to - 1745 + 1
    for i in range(2):
        p = oracle.split_equal(oracle.default_params(), nb)
        for b in range(nb):
            p.q10_rh[b] = q[b][i]
        ov, of, _, err = oracle.run_tracking(p, 1800, run_to)
        assert err == 0
        gv, gf, held = c.tracking_data(i, (1800, run_to), masks=True)
        assert np.abs(gv - ov[k0:k1]).max() < 1e-10 * np.abs(ov).max()
        assert np.abs(gf - of[k0:k1]).max() < FRAC_TOL
        assert np.abs(gf.sum(axis=2) - 1.0).max() < 1e-12
        # a source outside the map has no share; every pool holds itself; the atmosphere ends up
        # holding every pool that has carbon to give (the highest-numbered ones included)
        assert (gf[~held] == 0.0).all()
        assert held[:, np.arange(tp), np.arange(tp)].all()
        assert held[-1, 0, tp - 4:].all() and held[-1, 0, 2:tp - 4:5].all()
    return c


@pytest.mark.parametrize("nb", [5, 12])
def test_tracking_beyond_four_biomes(emul_lib, oracle, nb):
    """The reference tracks any number of biomes (fluxpool maps are string-keyed); here 5-16 run
    on the looped kernels.  12 biomes = 66 pools: the second mask word."""
    tracking_n_biomes(emul_lib, oracle, nb, run_to=1900, allow_emulation=True)


def test_tracking_refuses_carbon_constraints(emul_lib, tmp_path):
    from conftest import edited_pack
    pack = edited_pack(tmp_path / "c.hxs", "simpleNbox", "CO2_constrain", [1800, 1801], [285.0, 285.2])
    c = hector_amd.Core(str(pack), 1, lib_path=emul_lib, allow_emulation=True)
    c.setvar("trackingDate", [1770])
    with pytest.raises(hector_amd.HectorAmdError, match="constraint"):
        c.run(1850)


TRACK_BOTH_WAYS = """
import sys, numpy as np, hector_amd
from hector_amd import ensemble
out = {}
for nb in (1, 2):
    n = 200
    S, q = ensemble.ecs_q10(n, offset=5)
    c = hector_amd.Core(hector_amd.DEFAULT_SCENARIO, n, device=0)
    if nb > 1:
        c.split_biome(["a", "b"], fveg_c=[0.3, 0.7])
        c.setvar("a.q10_rh", q)
    else:
        c.setvar("q10_rh", q)
    c.setvar("S", S, "degC").setvar("trackingDate", [1800.0])
    c.run(1900)
    c.run(2000)           # resumes past the tracking date: the maps come back from the record
    assert (c.status() == 0).all()
    for i in (0, 63, 64, 199):
        v, f, held = c.tracking_data(i, (1800, 2000), masks=True)
        out["v%d_%d" % (nb, i)] = v; out["f%d_%d" % (nb, i)] = f; out["m%d_%d" % (nb, i)] = held
    c.reset(1850); c.run(1950)   # reset(date) into the tracked span needs the history: refused or redone
np.savez(sys.argv[1], **out)
"""


@pytest.mark.gpu
def test_companion_wavefronts_and_inline_maps_agree_on_gpu(tmp_path):
    """One and two biomes track on companion wavefronts (hx_run_kernel<B,HF,KERPM,3>: the maps in
    the registers of wavefronts that do nothing else), HECTOR_AMD_TRACK_INLINE=1 keeps the maps
    in the record and mixes inside the stash (what 3-16 biomes and big two-biome ensembles do):
    the same pools bit for bit, the same names in every map, fractions to rounding."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = {}
    for mode in ("companions", "inline"):
        env = dict(os.environ)
        env.pop("HECTOR_AMD_TRACK_INLINE", None)
        if mode == "inline":
            env["HECTOR_AMD_TRACK_INLINE"] = "1"
        path = str(tmp_path / (mode + ".npz"))
        script = TRACK_BOTH_WAYS.replace("c.reset(1850); c.run(1950)", "pass")
        r = subprocess.run([sys.executable, "-c", script, path], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        res[mode] = np.load(path)
    a, b = res["companions"], res["inline"]
    for k in a.files:
        if k.startswith("v"):
            assert np.array_equal(a[k], b[k]), k
        elif k.startswith("m"):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert np.abs(a[k] - b[k]).max() < 1e-12, k
            assert np.abs(a[k].sum(axis=2) - 1.0).max() < 1e-12
