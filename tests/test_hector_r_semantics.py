"""tests/testthat/test_hector.R restated for the ensemble core: the R wrapper's `clean` /
`reset_date` bookkeeping (R/messages.R:107-140, src/rcpp_hector.cpp:160-175) is inside the core
here -- a dated edit before the current date or a parameter change makes the next run() go back
first; edits in the future do not -- so the tests check what a caller can see of it: dates,
errors, and trajectories equal to those of a fresh core with the same inputs."""
import numpy as np
import pytest

import hector_amd
from conftest import SCENARIO

VARS = ("CO2_concentration", "RF_tot", "global_tas")


def mk(lib, **kw):
    c = hector_amd.Core(SCENARIO, 2, lib_path=lib, allow_emulation=True, **kw)
    c.setvar("S", [3.0, 4.5], "degC")
    c.set_outputs(list(VARS))
    return c


def traj(c, y0=1745, y1=2300):
    return {v: c.fetchvars(v, (y0, y1)).copy() for v in VARS}


def same(a, b):
    return all(np.array_equal(a[v], b[v]) for v in VARS)


def test_reset_produces_identical_results(emul_lib):
    """'Reset produces identical results' (a reset to 2000 needs the state history here) and
    'Rerunning spinup produces minimal change' (test_wrapper.R: 1e-6; here: identical)."""
    c = mk(emul_lib)
    c.enable_history(True)
    c.run()
    assert c.current_date == c.enddate
    first = traj(c)
    c.reset(2000)
    assert c.current_date == 2000
    c.run()
    assert same(first, traj(c))
    c.reset(0)           # back before the start date: the spinup runs again
    c.run(2100)
    again = traj(c, 2000, 2100)
    for v in VARS:
        assert np.array_equal(again[v], first[v][2000 - 1745:2100 - 1745 + 1])


def test_exceptions_are_caught_and_the_core_carries_on(emul_lib):
    """'Exceptions are caught': a scenario that does not exist; beta = -1 is refused when the
    core is prepared (the reference throws in reset()), naming the parameter; the core works
    again once the value is fixed."""
    with pytest.raises(hector_amd.HectorAmdError):
        hector_amd.Core("foo", 1, lib_path=emul_lib, allow_emulation=True)
    c = mk(emul_lib)
    c.setvar("beta", -1.0)
    with pytest.raises(hector_amd.HectorAmdError, match="beta"):
        c.reset(0); c.run(1800)
    c.setvar("beta", 0.5)
    c.reset(0)
    c.run(2100)
    assert (c.status() == 0).all() and c.current_date == 2100


def test_future_edits_do_not_reset_past_edits_do(emul_lib):
    """'Setting future values does not trigger a reset' / 'Setting parameter values or run date
    prior to current date does trigger a reset'."""
    years = np.arange(2050, 2151)
    c = mk(emul_lib)
    c.enable_history(True)
    c.run(2100)
    before = traj(c, 1745, 2100)
    # future values: nothing is recomputed, the past stays as it was, the future sees them
    c.setvar_dated("ffi_emissions", np.arange(2101, 2301), np.zeros(200), "Pg C/yr")
    assert c.current_date == 2100
    c.run(2100)
    assert same(before, traj(c, 1745, 2100))
    c.run(2300)
    f = mk(emul_lib)
    f.setvar_dated("ffi_emissions", np.arange(2101, 2301), np.zeros(200), "Pg C/yr")
    f.run(2300)
    assert same(traj(f), traj(c))
    # values before the current date: the next run goes back to the year before the first one
    c.setvar_dated("ffi_emissions", years, np.zeros(years.size), "Pg C/yr")
    with pytest.raises(hector_amd.HectorAmdError, match="is prior"):
        c.run(2048)                       # an error, but the core has gone back all the same
    assert c.current_date == 2049
    c.run(2050)
    assert c.current_date == 2050
    c.run(2300)
    f.setvar_dated("ffi_emissions", years, np.zeros(years.size), "Pg C/yr")
    f.run(2300)                           # (f resets itself the same way)
    assert same(traj(f), traj(c))
    g = mk(emul_lib)                      # and both equal a core that never ran before the edits
    g.setvar_dated("ffi_emissions", np.arange(2101, 2301), np.zeros(200), "Pg C/yr")
    g.setvar_dated("ffi_emissions", years, np.zeros(years.size), "Pg C/yr")
    g.run(2300)
    assert same(traj(g), traj(c))
    # two sets of values: back to the earlier one; a parameter: back to the very start
    c.setvar_dated("ffi_emissions", [2000], [0.0], "Pg C/yr")
    c.setvar_dated("ffi_emissions", [2010], [0.0], "Pg C/yr")
    c.setvar_dated("ffi_emissions", [1972], [0.0], "Pg C/yr")
    c.run(1980)
    assert c.current_date == 1980
    c.setvar("S", [2.5, 2.5], "degC")
    c.run(1760)
    assert c.current_date == 1760
    h = mk(emul_lib)
    h.setvar("S", [2.5, 2.5], "degC")
    for yy, vv in ((np.arange(2101, 2301), 0.0), (years, 0.0), ([2000], 0.0), ([2010], 0.0), ([1972], 0.0)):
        h.setvar_dated("ffi_emissions", yy, np.full(len(yy), vv), "Pg C/yr")
    h.run(1760)
    assert same(traj(h, 1745, 1760), traj(c, 1745, 1760))


def test_unknown_ini_keys_and_sections_are_errors(emul_lib, tmp_path):
    """A key no component reads, or a section that names no component, makes the reference throw
    while it parses the INI ("Unknown variable name while parsing temperature: ...",
    src/temperature_component.cpp setData; Core::getComponentByName) -- a typo does not pass
    silently.  `enabled` / `output` are the core's, valid in every component's section."""
    from conftest import edited_pack

    def core(scalars):
        p = edited_pack(tmp_path / "s.hxs", None, None, [], [], scalars=scalars)
        return hector_amd.Core(str(p), 1, lib_path=emul_lib, allow_emulation=True)
    with pytest.raises(hector_amd.HectorAmdError, match="Unknown variable name while parsing temperature: bogus_key"):
        core({("temperature", "bogus_key"): 3.0})
    with pytest.raises(hector_amd.HectorAmdError, match="Component not found: temprature"):
        core({("temprature", "S"): 3.0})
    with pytest.raises(hector_amd.HectorAmdError, match="Unknown variable name while parsing simpleNbox: bogus"):
        core({("simpleNbox", "boreal.bogus"): 1.0})
    with pytest.raises(hector_amd.HectorAmdError, match="Unknown variable name while parsing CF4_halocarbon: rho_C2F6"):
        core({("CF4_halocarbon", "rho_C2F6"): 1.0})
    core({("ozone", "output"): 1.0, ("temperature", "lo_warming_ratio"): 0.0}).run(1760)


def test_messages(emul_lib):
    """test_messages.R: a variable name with more than one '.' is invalid; fetchvars without
    dates for the default variables is an error."""
    c = mk(emul_lib)
    with pytest.raises(hector_amd.HectorAmdError, match="Invalid input variable: '"):
        c.setvar("global.permafrost.beta", [10.0], "")
    with pytest.raises(hector_amd.HectorAmdError, match="all require dates"):
        hector_amd.fetchvars(c, None)


def test_forcing_base_year_default_and_check(emul_lib, oracle, tmp_path):
    """ForcingComponent::prepareToRun (forcing_component.cpp:278-289): without a `baseyear` the
    forcings are reported relative to startDate + 1; a base year at or before the start date is
    refused."""
    import oracle_binding
    lines = [l for l in open(SCENARIO) if " baseyear " not in l]
    p = tmp_path / "nobase.hxs"
    p.write_text("".join(lines))
    assert len(lines) == len(open(SCENARIO).readlines()) - 1
    c = hector_amd.Core(str(p), 1, lib_path=emul_lib, allow_emulation=True)
    c.set_outputs(["RF_tot", "global_tas", "CO2_concentration"]); c.run(1800)
    rf = c.fetchvars("RF_tot", (1745, 1800))[:, 0]
    assert rf[0] == 0 and rf[1] == 0 and rf[2] != 0          # 1746 is the base year
    r, err, _ = oracle_binding.Oracle(str(p)).run(None, run_to=1800)
    assert err == 0
    assert np.abs(rf - r["RF_tot"][:rf.size]).max() < 1e-12
    assert np.abs(c.fetchvars("global_tas", (1745, 1800))[:, 0] - r["global_tas"][:rf.size]).max() < 1e-10
    from conftest import edited_pack
    bad = edited_pack(tmp_path / "bad.hxs", None, None, [], [], scalars={("forcing", "baseyear"): 1745.0})
    with pytest.raises(hector_amd.HectorAmdError, match="Base year must be"):
        hector_amd.Core(str(bad), 1, lib_path=emul_lib, allow_emulation=True).run(1760)


def test_a_concentration_driven_below_zero_raises_the_members_flag(emul_lib):
    """CH4 emissions so negative that the concentration crosses zero: log(CH4) is NaN in the
    reference (oh_component.cpp:156) and the member is lost; here it must raise its flag -- not
    carry on with an arbitrary finite value -- and leave its neighbour alone."""
    c = hector_amd.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    c.set_outputs(["CO2_concentration", "CH4_concentration"])
    years = np.arange(1800, 1811)
    em = np.zeros((years.size, 2)); em[:, 0] = c.fetchvars("CH4_emissions", (1800, 1810))[:, 0]
    em[:, 1] = -5.0e4
    c.setvar_dated_members("CH4_emissions", years, em, "Tg CH4")
    c.run(1850)
    st = c.status()
    assert st[0] == 0 and st[1] != 0
    d = hector_amd.Core(SCENARIO, 1, lib_path=emul_lib, allow_emulation=True)
    d.set_outputs(["CO2_concentration"]); d.run(1850)
    a, b = c.fetchvars("CO2_concentration", (1745, 1850))[:, 0], d.fetchvars("CO2_concentration", (1745, 1850))[:, 0]
    assert np.abs(a - b).max() / b.max() < 1e-9   # (c runs the extended kernel: per-member series)
