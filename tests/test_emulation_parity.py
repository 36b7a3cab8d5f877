"""The kernel SOURCE (hector_amd/csrc/hx_kernels.hip) and the host runtime,
compiled for the host through tests/emul (one lane at a time), against the
oracle and the reference golden trajectory.  CPU-only stand-in for the GPU
parity tests in test_gpu_parity.py: same checks, same tolerances.

Tolerances: the kernels reproduce the reference's decisions exactly (stash
schedule, step counts) but not its arithmetic bit for bit (FMA contraction,
warm-started Newton, grouped sums -- DESIGN.md "numerics"); the dominant term is
the alkalinity tuner, whose result is only defined to 2^-25 relative
(oceanbox.cpp:382-445) and maps to ~1e-10 relative in CO2.
"""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO

REL_CO2 = 2e-8   # relative, CO2 / atmos C / ocean C
ABS_T = 2e-8     # K, temperatures; W/m2 forcings


def mk(emul_lib, n):
    return hector_amd.Core(SCENARIO, n, lib_path=emul_lib, allow_emulation=True)


def test_default_member_vs_golden(emul_lib, golden):
    c = mk(emul_lib, 1)
    c.set_outputs(list(golden.keys()) + ["timesteps", "solver_steps"])
    c.run(2300)
    assert c.status()[0] == 0 and c.spinup_steps(0) == 498
    for var, ref in golden.items():
        got = c.fetchvars(var, (1745, 2300))[:, 0]
        if var in ("CO2_concentration", "atmos_co2", "ocean_c", "permafrost_c", "HL_pH"):
            m = ref != 0
            assert (np.abs(got[m] - ref[m]) / np.abs(ref[m])).max() < REL_CO2, var
        else:
            assert np.abs(got - ref).max() < ABS_T, var
    ts = c.fetchvars("timesteps", (1746, 2300))[:, 0].astype(int)
    assert np.bincount(ts).tolist() == [0, 340, 71, 2, 142]
    assert int(c.fetchvars("solver_steps", (1746, 2300)).sum()) == 1891


def test_perturbed_ensemble_vs_oracle(emul_lib, oracle):
    n = 24
    S, q10 = ensemble.ecs_q10(n)
    c = mk(emul_lib, n).setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert (c.status() == 0).all()
    co2 = c.fetchvars("CO2_concentration", (1745, 2300)).T
    tg = c.fetchvars("global_tas", (1745, 2300)).T
    oco2, otg, err = oracle.run_ecs_q10(S, q10)
    assert err == 0
    assert (np.abs(co2 - oco2) / oco2).max() < REL_CO2
    assert np.abs(tg - otg).max() < ABS_T
    # decisions: identical stash schedule member by member
    ts = c.fetchvars("timesteps", (1746, 2300)).T
    for i in range(0, n, 5):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        o, _, _ = oracle.run(p)
        assert np.array_equal(ts[i], o["timesteps"][1:])


def test_other_parameters_vs_oracle(emul_lib, oracle):
    """beta, diffusivity (per-member DOECLIM kernel table), aerosol / volcanic scale, C0."""
    n = 6
    beta = np.linspace(0.3, 0.9, n); diff = np.linspace(0.8, 2.4, n)
    aero = np.linspace(0.5, 1.5, n); vol = np.linspace(0.7, 1.3, n)
    c = mk(emul_lib, n)
    c.setvar("beta", beta, "(unitless)").setvar("diff", diff, "cm2/s")
    c.setvar("aero_scalar", aero).setvar("vol_scalar", vol)
    c.set_outputs(["CO2_concentration", "global_tas", "RF_tot", "heatflux"])
    c.run(2150)
    assert (c.status() == 0).all()
    for i in range(n):
        p = oracle.default_params()
        p.beta[0] = beta[i]; p.diff = diff[i]; p.aero_scalar = aero[i]; p.vol_scalar = vol[i]
        o, err, _ = oracle.run(p, run_to=2150)
        assert err == 0
        k = 2150 - 1745 + 1
        got = c.fetchvars("CO2_concentration", (1745, 2150))[:, i]
        assert (np.abs(got - o["CO2_concentration"][:k]) / o["CO2_concentration"][:k]).max() < REL_CO2
        for v in ("global_tas", "RF_tot", "heatflux"):
            assert np.abs(c.fetchvars(v, (1745, 2150))[:, i] - o[v][:k]).max() < ABS_T, v


def test_four_biomes_vs_oracle(emul_lib, oracle):
    n = 4
    S, q10s, wfs = ensemble.biome4(n)
    names = ["b1", "b2", "b3", "b4"]
    c = mk(emul_lib, n)
    c.split_biome(names)
    c.setvar("S", S, "degC")
    for b, nm in enumerate(names):
        c.setvar(nm + ".q10_rh", q10s[b]).setvar(nm + ".warmingfactor", wfs[b])
    c.set_outputs(["CO2_concentration", "global_tas", "permafrost_c", "veg_c"])
    c.run(2300)
    assert (c.status() == 0).all()
    for i in range(n):
        p = oracle.split_equal(oracle.default_params(), 4)
        p.S = S[i]
        for b in range(4):
            p.q10_rh[b] = q10s[b][i]; p.warmingfactor[b] = wfs[b][i]
        o, err, _ = oracle.run(p)
        assert err == 0
        for v in ("CO2_concentration", "permafrost_c", "veg_c"):
            got = c.fetchvars(v, (1745, 2300))[:, i]
            assert (np.abs(got - o[v]) / o[v]).max() < REL_CO2, v
        assert np.abs(c.fetchvars("global_tas", (1745, 2300))[:, i] - o["global_tas"]).max() < ABS_T


def test_identical_biome_split_equals_global(emul_lib):
    """test_biome.R:193-256 on the kernel source: an equal split into 2 or 4 identical biomes
    reproduces the single-biome run BIT FOR BIT, like the reference -- biome
    weights are correctly rounded quotients, so they come out as exactly 1/2 and 1/4."""
    S = [2.5, 3.0, 4.5]
    outs = ["CO2_concentration", "global_tas", "veg_c", "soil_c"]
    a = mk(emul_lib, 3).setvar("S", S); a.set_outputs(outs); a.run(2300)
    for nb in (2, 4):
        b = mk(emul_lib, 3).setvar("S", S); b.split_biome(["x%d" % i for i in range(nb)])
        b.set_outputs(outs); b.run(2300)
        for v in outs:
            assert np.array_equal(a.fetchvars(v), b.fetchvars(v)), (nb, v)
    # 3 biomes: thirds are not representable, the property holds to rounding
    b = mk(emul_lib, 3).setvar("S", S); b.split_biome(["x", "y", "z"]); b.set_outputs(outs); b.run(2300)
    for v in outs:
        x, y = a.fetchvars(v), b.fetchvars(v)
        assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(x).max()), v


def test_run_in_segments_equals_one_run(emul_lib):
    """Core::run is resumable (core.cpp:448-509): state carried through HBM."""
    S, q10 = ensemble.ecs_q10(3)
    a = mk(emul_lib, 3).setvar("S", S).setvar("q10_rh", q10).run(2100)
    b = mk(emul_lib, 3).setvar("S", S).setvar("q10_rh", q10)
    for y in (1750, 1751, 1800, 1983, 2100):
        b.run(y)
    for v in ("CO2_concentration", "global_tas", "sst", "land_tas"):
        assert np.array_equal(a.fetchvars(v), b.fetchvars(v)), v


def test_reset_and_rerun_reproduces(emul_lib):
    c = mk(emul_lib, 2).setvar("S", [2.0, 4.5]).run(1900)
    first = c.fetchvars("CO2_concentration")
    c.reset(1745).run(1900)
    assert np.array_equal(first, c.fetchvars("CO2_concentration"))
    c.reset(0).run(1900)
    assert np.array_equal(first, c.fetchvars("CO2_concentration"))
    c.setvar("S", [3.0, 3.0]).run(1900)  # setvar invalidates from date 0
    again = c.fetchvars("CO2_concentration")
    assert np.array_equal(again[:, 0], again[:, 1]) and not np.array_equal(again, first)


def test_error_behaviour_mirrors_reference(emul_lib):
    c = mk(emul_lib, 2)
    with pytest.raises(hector_amd.HectorAmdError, match="Unknown variable"):
        c.setvar("no_such_var", 1.0)
    with pytest.raises(hector_amd.HectorAmdError, match="[Uu]nits"):
        c.setvar("S", 3.0, "W/m2")
    with pytest.raises(hector_amd.HectorAmdError, match="Biome"):
        c.setvar("boreal.q10_rh", 2.0)
    with pytest.raises(hector_amd.HectorAmdError):
        c.setvar("S", [1.0, 2.0, 3.0])
    c.run(1760)
    with pytest.raises(hector_amd.HectorAmdError, match="dates"):
        c.fetchvars("CO2_concentration", (1745, 1800))
    with pytest.raises(hector_amd.HectorAmdError, match="not enabled"):
        c.fetchvars("RF_tot", (1745, 1760))
    with pytest.raises(hector_amd.HectorAmdError, match="unknown variable"):
        c.fetchvars("bogus", (1745, 1760))
    with pytest.raises(hector_amd.HectorAmdError, match="end date"):
        c.run(2400)
    with pytest.raises(hector_amd.HectorAmdError, match="does not exist"):
        hector_amd.Core("/nonexistent.ini", 1, lib_path=emul_lib, allow_emulation=True)
    assert np.array_equal(c.getvar("S"), [3.0, 3.0])


def test_member_sorting_is_transparent(emul_lib):
    """Lane assignment by parameter tiling must not change any member's result."""
    n = 200
    S, q10 = ensemble.ecs_q10(n)
    a = mk(emul_lib, n).setvar("S", S).setvar("q10_rh", q10).run(1900)
    b = mk(emul_lib, n).set_member_sorting(False).setvar("S", S).setvar("q10_rh", q10).run(1900)
    assert not np.array_equal(a.lane_of_member(), np.arange(n))
    assert np.array_equal(b.lane_of_member(), np.arange(n))
    for v in ("CO2_concentration", "global_tas"):
        assert np.array_equal(a.fetchvars(v), b.fetchvars(v))
    assert np.array_equal(a.status(), b.status())
    assert np.array_equal(a.getvar("S"), S)


def test_spinup_relevant_parameters_per_member(emul_lib, oracle):
    """Parameters that enter the spinup (C0, npp_flux0, ocean transports, initial pools,
    NPP partitioning) vary per member -> every lane spins up on its own (no broadcast)."""
    n = 5
    C0 = np.array([270.0, 277.15, 285.0, 277.15, 280.0])
    npp = np.array([50.0, 56.2, 60.0, 56.2, 58.0])
    tt = np.array([7.2e7, 7.2e7, 6.5e7, 8.0e7, 7.2e7])
    fv = np.array([0.35, 0.30, 0.35, 0.40, 0.35])
    veg = np.array([550.0, 500.0, 600.0, 550.0, 520.0])
    c = mk(emul_lib, n)
    c.setvar("C0", C0, "ppmv CO2").setvar("npp_flux0", npp, "Pg C/yr").setvar("tt", tt, "m3/s")
    c.setvar("f_nppv", fv).setvar("veg_c", veg, "Pg C")
    c.set_outputs(["CO2_concentration", "global_tas", "ocean_c", "veg_c"])
    c.run(2100)
    assert (c.status() == 0).all()
    for i in range(n):
        p = oracle.default_params()
        p.C0 = C0[i]; p.npp_flux0[0] = npp[i]; p.tt = tt[i]; p.f_nppv[0] = fv[i]; p.veg_c[0] = veg[i]
        o, err, steps = oracle.run(p, run_to=2100)
        assert err == 0 and c.spinup_steps(i) == steps
        k = 2100 - 1745 + 1
        for v in ("CO2_concentration", "ocean_c", "veg_c"):
            got = c.fetchvars(v, (1745, 2100))[:, i]
            assert (np.abs(got - o[v][:k]) / o[v][:k]).max() < REL_CO2, (v, i)
        assert np.abs(c.fetchvars("global_tas", (1745, 2100))[:, i] - o["global_tas"][:k]).max() < ABS_T


def test_unequal_biome_split_vs_oracle(emul_lib, oracle):
    """split_biome with explicit fractions (R/biome.R:61-130)."""
    fveg = [0.5, 0.3, 0.2]; fdet = [0.2, 0.3, 0.5]; fsoil = [0.4, 0.4, 0.2]
    fpf = [0.0, 0.3, 0.7]; fnpp = [0.6, 0.25, 0.15]
    c = mk(emul_lib, 1)
    c.split_biome(["trop", "temp", "bor"], fveg, fdet, fsoil, fpf, fnpp)
    c.setvar("bor.warmingfactor", 2.0).setvar("trop.q10_rh", 1.6).setvar("temp.beta", 0.4)
    c.set_outputs(["CO2_concentration", "global_tas", "permafrost_c", "soil_c"])
    c.run(2300)
    assert c.status()[0] == 0
    p = oracle.default_params()
    base = dict(v=p.veg_c[0], d=p.detritus_c[0], s=p.soil_c[0], pf=p.permafrost_c[0], n=p.npp_flux0[0])
    oracle.split_equal(p, 3)
    for b in range(3):
        p.veg_c[b] = base["v"] * fveg[b]; p.detritus_c[b] = base["d"] * fdet[b]
        p.soil_c[b] = base["s"] * fsoil[b]; p.permafrost_c[b] = base["pf"] * fpf[b]
        p.npp_flux0[b] = base["n"] * fnpp[b]
    p.warmingfactor[2] = 2.0; p.q10_rh[0] = 1.6; p.beta[1] = 0.4
    o, err, _ = oracle.run(p)
    assert err == 0
    for v in ("CO2_concentration", "permafrost_c", "soil_c"):
        got = c.fetchvars(v, (1745, 2300))[:, 0]
        assert (np.abs(got - o[v]) / o[v]).max() < REL_CO2, v
    assert np.abs(c.fetchvars("global_tas", (1745, 2300))[:, 0] - o["global_tas"]).max() < ABS_T


def test_model_errors_are_flags_not_crashes(emul_lib):
    """A member the reference would abort on (fluxpool / mass-balance / retry asserts)
    sets status bits; its neighbours are unaffected."""
    S = np.array([3.0, 3.0, 3.0]); npp = np.array([56.2, 1e5, 56.2])
    c = mk(emul_lib, 3).setvar("S", S).setvar("npp_flux0", npp).run(1800)
    st = c.status()
    assert st[0] == 0 and st[2] == 0 and st[1] != 0
    a = c.fetchvars("CO2_concentration")
    assert np.array_equal(a[:, 0], a[:, 2]) and np.isfinite(a[:, 0]).all()
    with pytest.raises(hector_amd.HectorAmdError):
        mk(emul_lib, 0)


def test_state_history_and_reset_to_any_date(emul_lib):
    """Core::reset(date) for startDate < date < current date (core.cpp:511-549): every
    component goes back to its recorded state of that year; rerunning reproduces the
    first run bit for bit, and the history does not change any result."""
    n = 5
    S, q10 = ensemble.ecs_q10(n)
    outs = ["CO2_concentration", "global_tas", "ocean_c", "CH4_concentration", "timesteps"]
    a = mk(emul_lib, n).setvar("S", S, "degC").setvar("q10_rh", q10)
    a.set_outputs(outs); a.run(2100)
    ref = {v: a.fetchvars(v, (1745, 2100)) for v in outs}
    b = mk(emul_lib, n).setvar("S", S, "degC").setvar("q10_rh", q10)
    b.enable_history(True); b.set_outputs(outs); b.run(2100)
    for v in outs:
        assert np.array_equal(b.fetchvars(v, (1745, 2100)), ref[v]), v
    for date in (2050, 1790, 1746, 2099):     # mid-block, early, first year, last year
        b.reset(date)
        assert b.current_date == date
        b.run(2100)
        for v in outs:
            assert np.array_equal(b.fetchvars(v, (1745, 2100)), ref[v]), (v, date)
    with pytest.raises(hector_amd.HectorAmdError):
        a.reset(2000)           # no history on this core
    with pytest.raises(hector_amd.HectorAmdError):
        b.reset(2200)           # not computed yet


def test_dated_setvar_emissions_vs_oracle(emul_lib, oracle, tmp_path):
    """setvar(core, dates, FFI_EMISSIONS(), values) after a run (R/messages.R:107-140): the core
    goes back to min(date)-1 and the changed years are recomputed -- compared with the oracle
    reading a scenario whose table holds the new values."""
    import oracle_binding
    from conftest import edited_pack
    years = np.arange(2030, 2061)
    vals = np.linspace(12.0, 2.0, years.size)
    c = mk(emul_lib, 2).setvar("S", np.array([2.5, 4.0]), "degC")
    c.enable_history(True)
    c.set_outputs(["CO2_concentration", "global_tas"])
    c.run(2100)
    before = c.fetchvars("CO2_concentration", (1745, 2100)).copy()
    c.setvar_dated("ffi_emissions", years, vals, "Pg C/yr")
    c.run(2100)
    co2 = c.fetchvars("CO2_concentration", (1745, 2100))
    tg = c.fetchvars("global_tas", (1745, 2100))
    assert np.array_equal(co2[:2030 - 1745], before[:2030 - 1745])
    assert np.abs(co2[2060 - 1745] - before[2060 - 1745]).min() > 1.0
    o = oracle_binding.Oracle(edited_pack(tmp_path / "ffi.hxs", "simpleNbox", "ffi_emissions",
                                          years, vals))
    for i, s in enumerate((2.5, 4.0)):
        p = o.default_params(); p.S = s
        r, err, _ = o.run(p, run_to=2100)
        assert err == 0
        oc, ot = r["CO2_concentration"][:co2.shape[0]], r["global_tas"][:co2.shape[0]]
        assert (np.abs(co2[:, i] - oc) / oc).max() < REL_CO2
        assert np.abs(tg[:, i] - ot).max() < ABS_T
    # the same edit on a core without history (falls back to startDate) and on a fresh core
    d = mk(emul_lib, 2).setvar("S", np.array([2.5, 4.0]), "degC")
    d.set_outputs(["CO2_concentration"]); d.run(2100)
    d.setvar_dated("ffi_emissions", years, vals); d.run(2100)
    e = mk(emul_lib, 2).setvar("S", np.array([2.5, 4.0]), "degC")
    e.set_outputs(["CO2_concentration"]); e.setvar_dated("ffi_emissions", years, vals); e.run(2100)
    assert np.array_equal(d.fetchvars("CO2_concentration", (1745, 2100)), co2)
    assert np.array_equal(e.fetchvars("CO2_concentration", (1745, 2100)), co2)
    with pytest.raises(hector_amd.HectorAmdError):
        c.setvar_dated("ffi_emissions", years, vals, "Tg C/yr")     # unit check
    with pytest.raises(hector_amd.HectorAmdError):
        c.setvar_dated("no_such_series", years, vals)


def unit_vectors_vs_oracle(lib, oracle, device=0):
    """Function-level parity (SURVEY 8c iv): carbonate solve over a grid of (T, DIC, alk) and
    the DOECLIM kernel table for several diffusivities, product functions vs the oracle."""
    import ctypes
    dp = ctypes.POINTER(ctypes.c_double)
    rng = np.random.default_rng(7)
    n = 500
    vol = 3.6e14 * 0.85 * 100.0
    Tc = rng.uniform(-1.0, 30.0, n); carbon = rng.uniform(600.0, 1100.0, n)
    alk = rng.uniform(2100e-6, 2750e-6, n)
    out = np.zeros((n, 4))
    as_p = lambda a: np.ascontiguousarray(a).ctypes.data_as(dp)
    rc = lib.hx_unit_csys(device, n, as_p(Tc), as_p(carbon), as_p(alk), vol, out.ctypes.data_as(dp))
    assert rc == 0, lib.hx_last_error()
    assert (out[:, 3] == 0).all()
    for i in range(n):
        o = oracle.csys(Tc[i], carbon[i], alk[i], vol)      # PCO2o, pH, Tr, K0, h, CO3
        # both root iterations (the reference's from the Fujiwara bound, the kernels' from a
        # guess) converge quadratically: same [H+] to rounding
        assert abs(out[i, 0] - o[0]) < 1e-13 * o[0]
        assert abs(out[i, 1] - o[1]) < 1e-13
        assert abs(out[i, 2] - o[2]) < 1e-13 * o[2]
    for diff in (0.55, 1.16, 2.3, 4.0):
        ker = np.zeros(556)
        rc = lib.hx_unit_doeclim_kernel(device, diff, 556, ker.ctypes.data_as(dp))
        assert rc == 0, lib.hx_last_error()
        ref = oracle.doeclim_kernel(diff, 556)
        assert np.abs(ker - ref).max() < 1e-12 * np.abs(ref).max()


def test_unit_vectors_vs_oracle(emul_lib, oracle):
    import hector_amd._lib as L
    unit_vectors_vs_oracle(L.load(emul_lib, allow_emulation=True), oracle)


def per_member_emissions_vs_oracle(lib, tmp_path, n=4, **kw):
    """vignettes/ex_hector_apply.Rmd: run, go back, give every run its own emissions for a
    period, run again -- all members in one core.  Each member vs the oracle reading a scenario
    that holds that member's series."""
    import oracle_binding
    from conftest import edited_pack
    years = np.arange(2000, 2101)
    S = np.linspace(2.2, 4.6, n)
    c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw).setvar("S", S, "degC")
    c.enable_history(True)
    c.set_outputs(["CO2_concentration", "global_tas", "CH4_concentration"])
    c.run(2150)
    before = c.fetchvars("CO2_concentration", (1745, 2150)).copy()
    base_ffi = c.fetchvars("ffi_emissions", (2000, 2100))          # [year, member]
    scale = np.linspace(0.2, 1.6, n)
    ffi = base_ffi * scale[None, :]
    ch4 = c.fetchvars("CH4_emissions", (2000, 2100)) * scale[::-1][None, :]
    c.setvar_dated_members("ffi_emissions", years, ffi, "Pg C/yr")
    c.setvar_dated_members("CH4_emissions", years, ch4, "Tg CH4")
    assert np.array_equal(c.fetchvars("ffi_emissions", (2000, 2100)), ffi)
    c.run(2150)                                    # auto-reset to 1999, like the R wrapper
    assert (c.status() == 0).all()
    co2 = c.fetchvars("CO2_concentration", (1745, 2150))
    tg = c.fetchvars("global_tas", (1745, 2150))
    m4 = c.fetchvars("CH4_concentration", (1745, 2150))
    assert np.array_equal(co2[:2000 - 1745], before[:2000 - 1745])
    assert (np.diff(co2[2100 - 1745]) > 0).all()   # more emissions (and higher ECS), more CO2
    for i in range(n):
        p1 = edited_pack(tmp_path / ("a%d.hxs" % i), "simpleNbox", "ffi_emissions", years, ffi[:, i])
        p2 = edited_pack(tmp_path / ("b%d.hxs" % i), "CH4", "CH4_emissions", years, ch4[:, i], base=p1)
        o = oracle_binding.Oracle(p2)
        p = o.default_params(); p.S = S[i]
        r, err, _ = o.run(p, run_to=2150)
        assert err == 0
        k = 2150 - 1745 + 1
        assert (np.abs(co2[:, i] - r["CO2_concentration"][:k]) / r["CO2_concentration"][:k]).max() < REL_CO2
        assert np.abs(tg[:, i] - r["global_tas"][:k]).max() < ABS_T
        assert np.abs(m4[:, i] - r["CH4_concentration"][:k]).max() < 1e-6
    with pytest.raises(hector_amd.HectorAmdError):
        c.setvar_dated_members("N2O_emissions", years, ffi)        # a shared (host) gas cycle
    return c


def test_per_member_emissions_vs_oracle(emul_lib, tmp_path):
    c = per_member_emissions_vs_oracle(emul_lib, tmp_path, allow_emulation=True)
    # member sorting must not mix the series up: same result with sorting off
    d = hector_amd.Core(SCENARIO, 4, lib_path=emul_lib, allow_emulation=True)
    d.set_member_sorting(False).setvar("S", np.linspace(2.2, 4.6, 4), "degC")
    d.set_outputs(["CO2_concentration"])
    y = np.arange(2000, 2101)
    # (the same sequence of kernels as `c`: the plain kernel to 2150, then back to 1999 and the
    # extended one from there -- the two instantiations agree to rounding, not bit for bit: the
    # plain kernel advances the thawed-permafrost pool exactly, the extended one integrates it)
    d.enable_history(True)
    d.run(2150)
    d.setvar_dated_members("ffi_emissions", y, c.fetchvars("ffi_emissions", (2000, 2100)))
    d.setvar_dated_members("CH4_emissions", y, c.fetchvars("CH4_emissions", (2000, 2100)))
    d.run(2150)
    assert np.array_equal(d.fetchvars("CO2_concentration", (1745, 2150)),
                          c.fetchvars("CO2_concentration", (1745, 2150)))
