"""The R package's capability accessors (src/rcpp_constants.cpp: ECS(), RF_CF4(), PH() ...) as
hector_amd.capabilities, and what they address: aliases, whole-surface ocean values, and the
parameters of the member-independent components (one value per core)."""
import numpy as np
import pytest

import hector_amd
from hector_amd import capabilities as cap
from conftest import SCENARIO, edited_pack


def test_capability_table_is_complete():
    assert len(cap.__all__) == 256
    assert cap.ECS() == "S" and cap.GLOBAL_TAS() == "global_tas" and cap.RF_TOTAL() == "RF_tot"
    assert cap.BETA() == "beta" and cap.BETA("boreal") == "boreal.beta"
    assert cap.EMISSIONS_CF4() == "CF4_emissions" and cap.HFC23_CONSTRAIN() == "HFC23_constrain"
    assert cap.RF_CF4() == "FadjCF4" and cap.PH_HL() == "HL_pH" and cap.BIOME_SPLIT_CHAR() == "."


def capability_checks(lib, tmp_path, **kw):
    import oracle_binding
    c = hector_amd.Core(SCENARIO, 2, lib_path=lib, **kw).setvar(cap.ECS(), np.array([2.5, 4.0]), "degC")
    outs = [cap.PH(), cap.PCO2(), cap.DIC(), cap.CO3(), cap.OCEAN_C_ML(), cap.PH_HL(), cap.PH_LL(),
            cap.OCEAN_C_HL(), cap.OCEAN_C_LL(), cap.RF_CO2(), cap.RF_TOTAL(), cap.GLOBAL_TAS(),
            cap.RF_CF4(), cap.CONCENTRATIONS_CO2()]
    c.set_outputs(outs); c.run(2100)
    f = lambda v: c.fetchvars(v, (1746, 2100))
    assert np.array_equal(f(cap.RF_CF4()), f("RF_CF4"))                       # "FadjCF4" alias
    assert np.allclose(f(cap.PH()), 0.85 * f(cap.PH_LL()) + 0.15 * f(cap.PH_HL()), rtol=1e-15)
    assert np.allclose(f(cap.OCEAN_C_ML()), f(cap.OCEAN_C_LL()) + f(cap.OCEAN_C_HL()), rtol=1e-15)
    assert (f(cap.PCO2()) > 200).all() and (f(cap.DIC()) > 1500).all() and (f(cap.CO3()) > 50).all()
    rf0, cf40 = f(cap.RF_CO2()).copy(), f(cap.RF_CF4()).copy()
    # parameters of the shared components: one value per core
    assert c.getvar(cap.DELTA_CO2())[0] == pytest.approx(0.05)
    c.setvar(cap.DELTA_CO2(), 0.2, "(unitless)")
    c.setvar(cap.RHO_CF4(), 2 * c.getvar(cap.RHO_CF4())[0], "W/m2/pptv")
    c.setvar(cap.PREINDUSTRIAL_CH4(), 700.0, "ppbv CH4")
    c.run(2100)
    # RF = sarf (1 + delta); the CO2 path itself shifts a little with the stronger forcing
    assert np.allclose(f(cap.RF_CO2())[100:] / rf0[100:], 1.2 / 1.05, rtol=3e-2)
    assert np.allclose(f(cap.RF_CF4())[10:], 2 * cf40[10:], rtol=1e-12)
    with pytest.raises(hector_amd.HectorAmdError):
        c.setvar(cap.DELTA_CO2(), np.array([0.1, 0.2]))                        # not per member
    with pytest.raises(hector_amd.HectorAmdError):
        c.setvar(cap.RHO_BC(), 1.0, "degC")
    # the same three edits in the scenario file, through the oracle
    rho = c.getvar(cap.RHO_CF4())[0]
    path = edited_pack(tmp_path / "p.hxs", None, None, [], [], scalars={
        ("forcing", "delta_co2"): 0.2, ("CF4_halocarbon", "rho_CF4"): rho, ("CH4", "M0"): 700.0})
    o = oracle_binding.Oracle(path)
    for i, s in enumerate((2.5, 4.0)):
        p = o.default_params(); p.S = s
        r, err, _ = o.run(p, run_to=2100)
        assert err == 0
        n = 2100 - 1745 + 1
        assert np.abs(c.fetchvars("global_tas", (1745, 2100))[:, i] - r["global_tas"][:n]).max() < 2e-8
        assert np.abs(c.fetchvars("RF_tot", (1745, 2100))[:, i] - r["RF_tot"][:n]).max() < 2e-8
        g = c.fetchvars("CO2_concentration", (1745, 2100))[:, i]
        assert (np.abs(g - r["CO2_concentration"][:n]) / g).max() < 2e-8


def test_capabilities_and_shared_parameters(emul_lib, tmp_path):
    capability_checks(emul_lib, tmp_path, allow_emulation=True)


def test_units_and_set_get_of_every_emission(emul_lib):
    """test_units.R (getunits) and test_set_get_data.R (every emissions series can be set at a
    date and read back) restated."""
    c = hector_amd.Core(SCENARIO, 1, lib_path=emul_lib, allow_emulation=True)
    units = {cap.ECS(): "degC", cap.DIFFUSIVITY(): "cm2/s", cap.BETA(): "(unitless)",
             cap.PREINDUSTRIAL_CO2(): "ppmv CO2", cap.FFI_EMISSIONS(): "Pg C/yr",
             cap.EMISSIONS_SO2(): "Gg S", cap.EMISSIONS_CF4(): "Gg", cap.EMISSIONS_CH4(): "Tg CH4",
             cap.EMISSIONS_NOX(): "Tg N", cap.NAT_EMISSIONS_N2O(): "Tg N", cap.RHO_BC(): "W/m2/Tg",
             cap.RHO_SO2(): "W/m2/Gg", cap.DELTA_CO2(): "(unitless)", cap.TT(): "m3/s",
             cap.OCEAN_PREIND_C_SURF(): "Pg C", cap.CO2_CONSTRAIN(): "ppmv CO2",
             cap.TAS_CONSTRAIN(): "degC", cap.HFC23_CONSTRAIN(): "pptv", cap.GLOBAL_TAS(): "degC",
             cap.RF_TOTAL(): "W/m2", cap.RF_CF4(): "W/m2", cap.PH_HL(): "pH", cap.PCO2_LL(): "uatm",
             cap.CONCENTRATIONS_CH4(): "ppbv CH4", cap.OCEAN_UPTAKE(): "Pg C/yr", cap.NBP(): "Pg C/yr",
             cap.VOLCANIC_SO2(): "W/m2", cap.LIFETIME_SOIL(): "Years", "boreal." + cap.VEG_C(): "Pg C"}
    for v, u in units.items():
        assert c.getunits(v) == u, v
    assert c.component_of(cap.GLOBAL_TAS()) == "temperature" and c.component_of(cap.NBP()) == "simpleNbox"
    with pytest.raises(hector_amd.HectorAmdError):
        c.getunits("no_such_variable")
    emissions = [n for n in cap.__all__ if n.startswith("EMISSIONS_")] + \
        ["FFI_EMISSIONS", "LUC_EMISSIONS", "NAT_EMISSIONS_N2O", "DACCS_UPTAKE", "LUC_UPTAKE"]
    assert len(emissions) >= 37
    rng = np.random.default_rng(3)
    for name in emissions:
        v = getattr(cap, name)()
        val = float(rng.exponential(5.0))
        c.setvar_dated(v, [1800], [val], c.getunits(v))
        assert c.fetchvars(v, (1800, 1800))[0, 0] == val, v
    c.run(1800)
    assert c.status()[0] == 0
    with pytest.raises(hector_amd.HectorAmdError):
        c.setvar_dated(cap.FFI_EMISSIONS(), [1800], [1.0], "boogedyboo")


def test_message_bus_and_na_dates(emul_lib):
    """sendmessage(core, GETDATA/SETDATA, ...) (src/rcpp_hector.cpp:262-350) and fetchvars with
    dates = NA for parameters (R/messages.R:46-88)."""
    import hector_amd as h
    c = h.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    p = h.fetchvars(c, float("nan"), ["beta", "q10_rh"])
    assert np.array_equal(p["beta"], [0.65, 0.65]) and np.array_equal(p["q10_rh"], [1.2, 1.2])
    assert h.fetchvars(c, None, "S")["S"].shape == (2,)
    (year, var, val, unit), = h.sendmessage(c, h.GETDATA, "S")
    assert year is None and var == "S" and unit == "degC" and np.array_equal(val, [3.0, 3.0])
    h.sendmessage(c, h.SETDATA, "S", None, [2.5, 3.5], "degC")
    assert np.array_equal(c.getvar("S"), [2.5, 3.5])
    with pytest.raises(h.HectorAmdError, match="do not match expected"):
        h.sendmessage(c, h.SETDATA, "S", None, [2.5, 3.5], "K")
    h.sendmessage(c, h.SETDATA, "ffi_emissions", [1760, 1761], [0.1, 0.1], "Pg C/yr")
    c.run(1770)
    rows = h.sendmessage(c, h.GETDATA, "global_tas", [1765, 1770])
    assert [r[0] for r in rows] == [1765, 1770] and rows[0][3] == "degC"
    assert np.array_equal(rows[1][2], c.fetchvars("global_tas", (1770, 1770))[0])
    with pytest.raises(h.HectorAmdError):
        h.sendmessage(c, "deepOceanCarbonDump", "x")


def test_r_style_fetchvars_defaults_and_date_filtering(emul_lib):
    """R/messages.R:46-88: vars = NULL -> the default four, dates outside start..current are
    dropped, none left is an error."""
    import hector_amd as h
    c = h.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    c.set_outputs(list(h.core.DEFAULT_FETCHVARS)); c.run(1900)
    r = h.fetchvars(c, [1700, 1800, 1850, 1950])
    assert sorted(r) == sorted(h.core.DEFAULT_FETCHVARS) and r["global_tas"].shape == (2, 2)
    assert np.array_equal(r["global_tas"][1], c.fetchvars("global_tas", (1850, 1850))[0])
    assert h.fetchvars(c, (1890, 1900), "global_tas")["global_tas"].shape == (11, 2)
    with pytest.raises(h.HectorAmdError, match="None of these dates are valid"):
        h.fetchvars(c, [1700, 2000])
    with pytest.raises(h.HectorAmdError, match="all require dates"):
        h.fetchvars(c, None)


def test_remaining_r_level_helpers(emul_lib, golden):
    """NAMESPACE exports: isactive, startdate, enddate, getdate, getname, get_biome_list,
    getunits, getfxn, runscenario (R/hector.R:57-160, R/units.R, R/fxns.R)."""
    import hector_amd as h
    kw = dict(lib_path=emul_lib, allow_emulation=True)
    c = h.newcore(SCENARIO, 2, **kw)
    assert (h.isactive(c), h.startdate(c), h.enddate(c), h.getdate(c)) == (True, 1745, 2300, 1745)
    assert h.getname(c) == "ssp245" and h.get_biome_list(c) == ["global"]
    assert h.getname(h.newcore(SCENARIO, 1, name="my run", **kw)) == "my run"
    assert h.getunits(["beta", "S", "ffi_emissions", "nope"], c) == ["(unitless)", "degC", "Pg C/yr", None]
    assert h.getunits("S", c) == "degC"
    assert h.getfxn(["beta", "q10_rh", "zzz"]) == ["BETA()", "Q10_RH()", None]
    h.shutdown(c)
    assert not h.isactive(c)
    r = h.runscenario(SCENARIO, **kw)
    assert sorted(r) == sorted(h.core.DEFAULT_FETCHVARS) and r["global_tas"].shape == (556, 1)
    assert abs(r["global_tas"][-1, 0] - golden["global_tas"][-1]) < 2e-8


def test_can_fetch_all_variables(emul_lib):
    """test_set_get_data.R 'Can fetch all variables': every name of the reference's ALL_VARS()
    (data/all_vars.rda, kept as tests/golden/all_vars.txt) can be recorded and fetched for a year
    of a run -- 82 distinct variables, finite."""
    import os
    from conftest import ROOT
    names = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "all_vars.txt"))
             if l.strip() and not l.startswith("#")]
    assert len(set(names)) == 82
    c = hector_amd.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    c.setvar("S", [3.0, 4.2], "degC")
    c.set_outputs(names)
    c.run(1850)
    assert (c.status() == 0).all()
    for v in names:
        x = c.fetchvars(v, (1845, 1845))
        assert x.shape == (1, 2) and np.isfinite(x).all(), v
