"""Every scenario the reference ships (inst/input/hector_*.ini, imported to dense packs by
tools/import_scenario.py) and the LUC-pulse input of tests/testthat/test_pulse.R, through
the product sources (host emulation here; test_gpu_parity.py repeats it on the GPU)."""
import os

import numpy as np
import pytest

import hector_amd
from conftest import ROOT

REL_CO2 = 2e-8
ABS_T = 2e-8
SCENARIOS = ["picontrol", "ssp119", "ssp126", "ssp245", "ssp370", "ssp434", "ssp460",
             "ssp534-over", "ssp585"]
REF_INPUT = "/root/reference/inst/input"


def pack(name):
    return os.path.join(ROOT, "hector_amd", "data", name + ".hxs")


def check_scenario_vs_oracle(lib, path, S, q10, **kw):
    import oracle_binding
    o = oracle_binding.Oracle(path)
    n = len(S)
    c = hector_amd.Core(path, n, lib_path=lib, **kw).setvar("S", S, "degC").setvar("q10_rh", q10)
    c.set_outputs(["CO2_concentration", "global_tas", "RF_tot", "timesteps"])
    c.run(o.end)
    assert (c.status() == 0).all()
    co2 = c.fetchvars("CO2_concentration", (o.start, o.end))
    tg = c.fetchvars("global_tas", (o.start, o.end))
    rf = c.fetchvars("RF_tot", (o.start, o.end))
    ts = c.fetchvars("timesteps", (o.start + 1, o.end))
    for i in range(n):
        p = o.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        r, err, _ = o.run(p)
        assert err == 0
        assert (np.abs(co2[:, i] - r["CO2_concentration"]) / r["CO2_concentration"]).max() < REL_CO2
        assert np.abs(tg[:, i] - r["global_tas"]).max() < ABS_T
        assert np.abs(rf[:, i] - r["RF_tot"]).max() < ABS_T
        assert np.array_equal(ts[:, i], r["timesteps"][1:])
    return c


@pytest.mark.parametrize("name", SCENARIOS)
def test_shipped_scenarios_vs_oracle(emul_lib, oracle, name):
    check_scenario_vs_oracle(emul_lib, pack(name), np.array([3.0, 2.0, 5.5]),
                             np.array([2.2, 1.3, 2.9]), allow_emulation=True)


@pytest.mark.skipif(not os.path.isdir(REF_INPUT), reason="needs the reference's input files")
@pytest.mark.parametrize("name", ["ssp119", "ssp585", "picontrol"])
def test_ini_reader_equals_pack(emul_lib, name):
    """The C++ INI + csv: reader (hx_scenario.cpp; ini_to_core_reader.cpp:100-180,
    csv_table_reader.cpp:115-198) gives the core the same inputs as the imported pack."""
    out = []
    for path in (os.path.join(REF_INPUT, "hector_%s.ini" % name), pack(name)):
        c = hector_amd.Core(path, 1, lib_path=emul_lib, allow_emulation=True)
        c.set_outputs(["CO2_concentration", "global_tas", "RF_tot", "CH4_concentration"])
        c.run(2300)
        out.append([c.fetchvars(v, (1745, 2300)) for v in
                    ("CO2_concentration", "global_tas", "RF_tot", "CH4_concentration")])
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_luc_pulse_like_reference_test(emul_lib, oracle):
    """tests/testthat/test_pulse.R: zero emissions, Q10 = 1, beta = 0 -- flat vegetation
    carbon after spinup, and flat again after the 1800 LUC pulse."""
    path = os.path.join(ROOT, "tests", "golden", "luc_pulse.hxs")
    c = check_scenario_vs_oracle(emul_lib, path, np.array([3.0]), np.array([1.0]),
                                 allow_emulation=True)
    c.set_outputs(["veg_c"])
    c.run(1850)
    v = c.fetchvars("veg_c", (1745, 1850))[:, 0]
    y = np.arange(1745, 1851)
    assert (np.diff(v[(y >= 1750) & (y <= 1799)]) < 1e-6).all()
    assert (np.diff(v[(y >= 1801) & (y <= 1850)]) < 1e-6).all()
    luc = c.fetchvars("luc_emissions", (1745, 1850))[:, 0]
    assert luc[1800 - 1745] > 0 and np.count_nonzero(luc) == 1
    assert abs((v[1799 - 1745] - v[1801 - 1745])) > 0.01     # the pulse did hit the pool


def test_do_spinup_0(emul_lib, tmp_path):
    """[core] do_spinup=0 (core.cpp:378-384): the run starts from the INI pools as they are."""
    from conftest import edited_pack, SCENARIO
    import oracle_binding
    path = edited_pack(tmp_path / "nospin.hxs", None, None, [], [], scalars={("core", "do_spinup"): 0})
    c = check_scenario_vs_oracle(emul_lib, path, np.array([3.0, 4.2]), np.array([2.0, 2.6]),
                                 allow_emulation=True)
    assert c.spinup_steps(0) == 0
    d = hector_amd.Core(SCENARIO, 1, lib_path=emul_lib, allow_emulation=True)
    d.set_outputs(["CO2_concentration"]); d.run(1800)
    c.set_outputs(["CO2_concentration"]); c.run(1800)
    assert abs(c.fetchvars("CO2_concentration", (1800, 1800))[0, 0] -
               d.fetchvars("CO2_concentration", (1800, 1800))[0, 0]) > 1e-3     # it matters
