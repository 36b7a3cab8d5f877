"""Constraint branches through the product sources under host emulation:
the reference's own tests/testthat/test_constraints.R restated, plus parity with the oracle
reading a scenario that carries the same constraint.  test_gpu_parity.py repeats the parity
part on the GPU."""
import numpy as np
import pytest

import hector_amd
from conftest import SCENARIO, edited_pack

REL_CO2 = 2e-8
ABS_T = 2e-8
Y0, Y1 = 1745, 2300
ALL = np.arange(Y0, Y1 + 1)


def mk(lib, n=1, path=SCENARIO, **kw):
    kw.setdefault("allow_emulation", True)
    return hector_amd.Core(path, n, lib_path=lib, **kw)


def oracle_for(tmp_path, name, section, key, years, values, **kw):
    import oracle_binding
    return oracle_binding.Oracle(edited_pack(tmp_path / name, section, key, years, values, **kw))


def parity(c, r, i=0, y1=Y1, co2=True):
    n = y1 - Y0 + 1
    if co2:
        a = c.fetchvars("CO2_concentration", (Y0, y1))[:, i]
        assert (np.abs(a - r["CO2_concentration"][:n]) / r["CO2_concentration"][:n]).max() < REL_CO2
    assert np.abs(c.fetchvars("global_tas", (Y0, y1))[:, i] - r["global_tas"][:n]).max() < ABS_T
    assert np.abs(c.fetchvars("RF_tot", (Y0, y1))[:, i] - r["RF_tot"][:n]).max() < ABS_T


OUTS = ["CO2_concentration", "global_tas", "RF_tot", "RF_CO2", "NBP", "CH4_concentration",
        "land_tas", "sst", "timesteps"]


def test_co2_constraint_like_reference_test(emul_lib, tmp_path):
    """test_constraints.R:159-196 'Atmospheric CO2 concentrations can be constrained'."""
    years = np.arange(1850, 2101)
    S = np.array([3.0, 2.2, 4.8])
    hc = mk(emul_lib, 3).setvar("S", S, "degC")
    hc.set_outputs(OUTS); hc.run(Y1)
    base = {v: hc.fetchvars(v, (1850, 2100)).copy() for v in OUTS[:4]}
    for factor in (3.0, 0.5):
        con = base["CO2_concentration"][:, 0] * factor
        hc.setvar_dated("CO2_constrain", years, con, "ppmv CO2")
        hc.reset(Y0); hc.run(Y1)
        assert (hc.status() == 0).all()
        out = {v: hc.fetchvars(v, (1850, 2100)) for v in OUTS[:4]}
        for i in range(3):
            assert np.allclose(out["CO2_concentration"][:, i], con, rtol=1.5e-8, atol=0)
        for v in OUTS[:4]:
            assert ((out[v] >= base[v]) if factor > 1 else (out[v] <= base[v])).all(), v
        o = oracle_for(tmp_path, "co2.hxs", "simpleNbox", "CO2_constrain", years, con)
        for i in range(3):
            p = o.default_params(); p.S = S[i]
            r, err, _ = o.run(p)
            assert err == 0
            parity(hc, r, i)
            assert np.array_equal(hc.fetchvars("timesteps", (Y0 + 1, Y1))[:, i], r["timesteps"][1:])


def test_discontinuous_co2_constraint(emul_lib, tmp_path):
    """test_constraints.R:198-253: two separate constrained periods; no value in between."""
    y1, y2 = np.arange(1850, 1861), np.arange(1870, 1881)
    hc = mk(emul_lib)
    hc.set_outputs(OUTS)
    hc.setvar_dated("CO2_constrain", y1, np.full(y1.size, 278.0), "ppmv CO2")
    hc.setvar_dated("CO2_constrain", y2, np.full(y2.size, 298.0), "ppmv CO2")
    hc.run(Y1)
    rf = hc.fetchvars("RF_CO2", (Y0, Y1))[:, 0]
    ca = hc.fetchvars("CO2_concentration", (Y0, Y1))[:, 0]
    assert (np.abs(np.diff(rf[y1 - Y0])) <= 1e-5).all() and (np.abs(np.diff(rf[y2 - Y0])) <= 1e-5).all()
    assert np.allclose(ca[y1 - Y0], 278.0, rtol=1.5e-8) and np.allclose(ca[y2 - Y0], 298.0, rtol=1.5e-8)
    assert rf[2000 - Y0] > rf[1900 - Y0] and ca[2000 - Y0] > ca[1900 - Y0]
    con = hc.fetchvars("CO2_constrain", (1849, 1881))[:, 0]
    assert np.isnan(con[[0, 12, 32]]).all() and con[1] == 278.0 and con[-2] == 298.0
    o = oracle_for(tmp_path, "co2d.hxs", "simpleNbox", "CO2_constrain",
                   np.concatenate([y1, y2]), [278.0] * 11 + [298.0] * 11)
    r, err, _ = o.run(o.default_params())
    assert err == 0
    parity(hc, r)


def test_tas_constraint_like_reference_test(emul_lib, tmp_path):
    """test_constraints.R:255-273."""
    hc = mk(emul_lib)
    hc.set_outputs(OUTS)
    hc.setvar_dated("tas_constrain", [2000], [2.0], "degC")
    hc.run(Y1)
    assert np.isnan(hc.fetchvars("tas_constrain", (1999, 1999))[0, 0])
    assert hc.fetchvars("tas_constrain", (2000, 2000))[0, 0] == 2.0
    x = hc.fetchvars("global_tas", (1999, 2001))[:, 0]
    assert x[0] < 2.0 and x[1] == 2.0 and x[2] < 2.0
    o = oracle_for(tmp_path, "tas.hxs", "temperature", "tas_constrain", [2000], [2.0])
    r, err, _ = o.run(o.default_params())
    assert err == 0
    parity(hc, r)
    for v in ("land_tas", "sst"):
        assert np.abs(hc.fetchvars(v, (Y0, Y1))[:, 0] - r[v]).max() < ABS_T


def test_tas_constraint_interpolates_between_its_dates(emul_lib, tmp_path):
    """tas_constrain allows interpolation (temperature_component.cpp:112, 510-511)."""
    hc = mk(emul_lib, 2).setvar("S", np.array([2.5, 4.0]), "degC")
    hc.set_outputs(OUTS)
    hc.setvar_dated("tas_constrain", [1990, 2010], [0.5, 1.5])
    hc.run(2100)
    tg = hc.fetchvars("global_tas", (1989, 2011))
    assert np.allclose(tg[1:22, 0], np.linspace(0.5, 1.5, 21), atol=1e-12)
    assert np.array_equal(tg[1:22, 0], tg[1:22, 1]) and tg[0, 0] != tg[0, 1]
    o = oracle_for(tmp_path, "tas2.hxs", "temperature", "tas_constrain",
                   np.arange(1990, 2011), np.linspace(0.5, 1.5, 21))
    p = o.default_params(); p.S = 4.0
    r, err, _ = o.run(p, run_to=2100)
    parity(hc, r, 1, y1=2100)


def test_nbp_constraint_like_reference_test(emul_lib, tmp_path):
    """test_constraints.R:275-294."""
    hc = mk(emul_lib)
    hc.set_outputs(OUTS)
    hc.setvar_dated("NBP_constrain", [2000], [1.0], "Pg C/yr")
    hc.run(Y1)
    assert hc.status()[0] == 0
    x = hc.fetchvars("NBP", (1999, 2001))[:, 0]
    assert x[0] != 1.0 and abs(x[1] - 1.0) < 1e-12 and x[2] != 1.0
    o = oracle_for(tmp_path, "nbp.hxs", "simpleNbox", "NBP_constrain", [2000], [1.0])
    r, err, _ = o.run(o.default_params())
    assert err == 0
    parity(hc, r)
    assert np.abs(hc.fetchvars("NBP", (Y0 + 1, Y1))[:, 0] - r["NBP"][1:]).max() < 1e-8
    assert np.array_equal(hc.fetchvars("timesteps", (Y0 + 1, Y1))[:, 0], r["timesteps"][1:])


def test_nbp_constraint_many_years_vs_oracle(emul_lib, tmp_path):
    years = np.arange(1950, 2051)
    vals = 0.5 + 0.01 * (years - 1950)
    S = np.array([2.0, 3.5, 5.0]); q10 = np.array([1.5, 2.1, 2.8])
    hc = mk(emul_lib, 3).setvar("S", S, "degC").setvar("q10_rh", q10)
    hc.set_outputs(OUTS)
    hc.setvar_dated("NBP_constrain", years, vals)
    hc.run(2150)
    assert (hc.status() == 0).all()
    assert np.abs(hc.fetchvars("NBP", (1950, 2050)) - vals[:, None]).max() < 1e-10
    o = oracle_for(tmp_path, "nbp2.hxs", "simpleNbox", "NBP_constrain", years, vals)
    for i in range(3):
        p = o.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        r, err, _ = o.run(p, run_to=2150)
        assert err == 0
        parity(hc, r, i, y1=2150)
        assert np.array_equal(hc.fetchvars("timesteps", (Y0 + 1, 2150))[:, i], r["timesteps"][1:406])


@pytest.mark.parametrize("gas,section,conc,unit", [
    ("CH4", "CH4", "CH4_concentration", "ppbv CH4"),
    ("N2O", "N2O", "N2O_concentration", "ppbv N2O"),
    ("HFC23", "HFC23_halocarbon", "HFC23_concentration", "pptv"),
])
def test_concentration_forced_gases_like_reference_tests(emul_lib, tmp_path, gas, section, conc, unit):
    """test_constraints.R:5-157: feeding a run its own concentrations as a constraint changes
    nothing; scaled concentrations are honoured and warm (or cool) the run."""
    hc = mk(emul_lib)
    hc.set_outputs(OUTS); hc.run(Y1)
    tas0 = hc.fetchvars("global_tas", (Y0, Y1)).copy()
    c0 = hc.fetchvars(conc, (Y0, Y1))[:, 0].copy()
    hc.setvar_dated(gas + "_constrain", ALL, c0, unit)
    hc.reset(Y0); hc.run(Y1)
    # test_constraints.R:42,89,139: expect_equivalent(emissOut$value, conOut$value, tol = 1e-10),
    # i.e. R's all.equal: mean absolute difference / mean absolute value <= 1e-10, over global_tas
    # and the gas's concentration together
    a = np.concatenate([tas0[:, 0], c0]); b = np.concatenate([hc.fetchvars("global_tas", (Y0, Y1))[:, 0],
                                                              hc.fetchvars(conc, (Y0, Y1))[:, 0]])
    assert np.abs(a - b).mean() / np.abs(a).mean() <= 1e-10
    assert np.abs(hc.fetchvars("global_tas", (Y0, Y1)) - tas0).max() < 1e-9
    assert np.abs(hc.fetchvars(conc, (Y0, Y1))[:, 0] - c0).max() == 0
    c1 = c0 * 1.5
    hc.setvar_dated(gas + "_constrain", ALL, c1, unit)
    hc.reset(0); hc.run(Y1)          # the constraint at startDate replaces the preindustrial value
    assert np.array_equal(hc.fetchvars(conc, (Y0, Y1))[:, 0], c1)
    o = oracle_for(tmp_path, gas + ".hxs", section, gas + "_constrain", ALL, c1)
    r, err, _ = o.run(o.default_params())
    assert err == 0
    parity(hc, r)


def test_ftot_constraint_vs_oracle(emul_lib, tmp_path):
    """forcing_component.cpp:498-505: used for every date up to the constraint's last one."""
    years = np.arange(1800, 2051)
    vals = np.linspace(0.2, 4.0, years.size)
    hc = mk(emul_lib, 2).setvar("S", np.array([2.5, 4.5]), "degC")
    hc.set_outputs(OUTS)
    hc.setvar_dated("RF_tot_constrain", years, vals, "W/m2")
    hc.run(Y1)
    rf = hc.fetchvars("RF_tot", (Y0, Y1))
    # relative to the base year 1750, where the constraint (flat before 1800) is 0.2
    assert np.allclose(rf[1800 - Y0:2051 - Y0, 0], vals - 0.2, atol=1e-12)
    o = oracle_for(tmp_path, "ftot.hxs", "forcing", "RF_tot_constrain",
                   np.arange(Y0, 2051), np.concatenate([np.full(1800 - Y0, 0.2), vals]))
    for i, s in enumerate((2.5, 4.5)):
        p = o.default_params(); p.S = s
        r, err, _ = o.run(p)
        assert err == 0
        parity(hc, r, i)


def test_land_ocean_warming_ratio_per_member(emul_lib, oracle):
    """lo_warming_ratio (temperature_component.cpp:722-739): land and sea temperatures as
    reported and as seen by the carbon cycle follow global tas with the given ratio."""
    lo = np.array([0.0, 1.4, 1.8, 0.0, 1.1])
    S = np.array([3.0, 3.0, 2.4, 4.4, 5.0])
    hc = mk(emul_lib, 5).setvar("S", S, "degC").setvar("lo_warming_ratio", lo, "(unitless)")
    hc.set_outputs(OUTS); hc.run(Y1)
    assert (hc.status() == 0).all()
    tl, sst = hc.fetchvars("land_tas", (Y0, Y1)), hc.fetchvars("sst", (Y0, Y1))
    for i in range(5):
        p = oracle.default_params(); p.S = S[i]; p.lo_warming_ratio = lo[i]
        r, err, _ = oracle.run(p)
        assert err == 0
        parity(hc, r, i)
        assert np.abs(tl[:, i] - r["land_tas"]).max() < ABS_T
        assert np.abs(sst[:, i] - r["sst"]).max() < ABS_T
        if lo[i]:
            assert np.allclose(tl[100:, i] / (1.3 * sst[100:, i]), lo[i], rtol=1e-12)
