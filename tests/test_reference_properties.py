"""The reference's own science checks (tests/testthat/test_ocean.R, test_atmosphere.R,
test_parameters.R, test_inis.R 'picontrol', test_hector.R) restated for the ensemble core: each
perturbation the R tests apply by re-running one core is one member here.  Host emulation of the
product sources; test_gpu_parity.py runs the same function on the GPU."""
import os

import numpy as np
import pytest

import hector_amd
from conftest import ROOT, SCENARIO


def science_checks(lib, **kw):
    mk = lambda n, path=SCENARIO: hector_amd.Core(path, n, lib_path=lib, **kw)
    # ---- test_ocean.R ---------------------------------------------------------------
    hc = mk(1)
    ocean = ["ocean_c", "HL_ocean_c", "LL_ocean_c", "IO_ocean_c", "DO_ocean_c", "ocean_uptake",
             "HL_ocean_uptake", "LL_ocean_uptake", "HL_pH", "LL_pH", "HL_PCO2", "LL_PCO2",
             "HL_sst", "LL_sst", "HL_CO3", "LL_CO3", "HL_DIC"]
    hc.set_outputs(ocean); hc.run(1900)
    o = {v: hc.fetchvars(v, (1850, 1900))[:, 0] for v in ocean}
    assert np.allclose(o["ocean_c"], o["HL_ocean_c"] + o["LL_ocean_c"] + o["IO_ocean_c"] + o["DO_ocean_c"],
                       rtol=1.5e-8)                                   # "Checking carbon pools"
    assert np.allclose(o["ocean_uptake"], o["HL_ocean_uptake"] + o["LL_ocean_uptake"], rtol=1.5e-8)
    for a, b in [("HL_ocean_c", "LL_ocean_c"), ("HL_pH", "LL_pH"), ("HL_PCO2", "LL_PCO2"),
                 ("HL_ocean_uptake", "LL_ocean_uptake"), ("HL_sst", "LL_sst"), ("HL_CO3", "LL_CO3")]:
        assert o[a].mean() != o[b].mean()                              # HL and LL boxes differ
    # "Read and writing ocean parameters": each parameter times 1.1 moves the 1850-1900 mean of
    # every ocean variable by more than the test's 1e-10; a parameter has no dates
    params = ["tt", "tu", "twi", "tid", "preind_surface_c", "preind_interdeep_c"]
    with pytest.raises(hector_amd.HectorAmdError):
        hc.fetchvars("tt", (1850, 1900))
    pc = mk(1 + len(params))
    for k, p in enumerate(params):
        v = np.full(1 + len(params), pc.getvar(p)[0]); v[1 + k] *= 1.1
        pc.setvar(p, v)
        assert np.array_equal(pc.getvar(p), v)
    ocean_vars = ["ocean_uptake", "ocean_c", "HL_pH", "HL_PCO2", "HL_DIC", "HL_sst", "HL_CO3"]
    pc.set_outputs(ocean_vars); pc.run(1900)
    assert (pc.status() == 0).all()
    for v in ocean_vars:
        x = pc.fetchvars(v, (1850, 1900)).mean(axis=0)
        for k in range(len(params)):
            assert abs(x[1 + k] - x[0]) > 1e-10, (v, params[k])
    # ---- test_atmosphere.R "Check Temp" ----------------------------------------------
    hc = mk(1); hc.set_outputs(["land_tas", "ocean_tas", "global_tas", "gmst"]); hc.run(2100)
    land, oc = hc.fetchvars("land_tas", (2020, 2100)), hc.fetchvars("ocean_tas", (2020, 2100))
    tas, gmst = hc.fetchvars("global_tas", (2020, 2100)), hc.fetchvars("gmst", (2020, 2100))
    assert np.allclose(tas, 0.29 * land + oc * (1 - 0.29), atol=1e-5)
    assert (tas > gmst).all()
    # ---- test_inis.R "picontrol" ------------------------------------------------------
    pi = mk(1, os.path.join(ROOT, "hector_amd", "data", "picontrol.hxs"))
    pi.set_outputs(["global_tas", "RF_tot", "CH4_concentration"]); pi.run(2300)
    for v in ["global_tas", "RF_tot", "CH4_concentration"]:
        assert pi.fetchvars(v, (1750, 2100)).std() <= 1e-4, v
    # ---- test_parameters.R: one member per perturbation -------------------------------
    names = ["default", "C0", "S", "q10_rh", "diff", "aero", "vol", "f_nppv", "f_nppd", "f_litterd",
             "beta"]
    n = len(names)
    hc = mk(n)
    base = {p: hc.getvar(p)[0] for p in ["C0", "S", "q10_rh", "diff", "aero_scalar", "vol_scalar",
                                         "f_nppv", "f_nppd", "f_litterd", "beta"]}
    def member(i, p, value, unit=None):
        v = np.full(n, base[p]); v[i] = value
        hc.setvar(p, v, unit)
    member(1, "C0", 250.0, "ppmv CO2"); member(2, "S", base["S"] / 2, "degC")
    member(3, "q10_rh", base["q10_rh"] * 2); member(4, "diff", base["diff"] / 2, "cm2/s")
    member(5, "aero_scalar", base["aero_scalar"] / 2); member(6, "vol_scalar", base["vol_scalar"] * 2)
    member(7, "f_nppv", base["f_nppv"] / 2); member(8, "f_nppd", base["f_nppd"] / 2)
    member(9, "f_litterd", base["f_litterd"] / 2); member(10, "beta", base["beta"] * 2)
    hc.set_outputs(["CO2_concentration", "global_tas", "NPP", "RF_vol"]); hc.run(2100)
    assert (hc.status() == 0).all()
    co2 = hc.fetchvars("CO2_concentration", (1745, 2100)); tas = hc.fetchvars("global_tas", (1745, 2100))
    npp = hc.fetchvars("NPP", (1746, 2100)); rfv = hc.fetchvars("RF_vol", (1745, 2100))
    yr = lambda y: y - 1745
    assert co2[0, 0] == pytest.approx(base["C0"], rel=1e-12) and co2[0, 1] == pytest.approx(250.0, rel=1e-12)
    assert (co2[1:, 1] < co2[1:, 0]).all()                 # lowering initial CO2 lowers CO2
    # ... by about the change in the initial concentration, growing with time (test_parameters.R:88:
    # expect_lt(max(diff), -26.0) over 1750:2100)
    assert (co2[yr(1750):, 1] - co2[yr(1750):, 0]).max() < -26.0
    late = slice(yr(2000), yr(2100) + 1)
    assert (tas[late, 2] < tas[late, 0]).all()             # lowering ECS lowers temperature
    assert (co2[late, 3] > co2[late, 0]).all()             # raising Q10 increases CO2
    assert (tas[late, 4] > tas[late, 0]).all()             # lowering diffusivity increases temperature
    assert (tas[late, 5] > tas[late, 0]).all()             # weaker (negative) aerosol forcing warms
    for y in (1960, 1965):                                 # volcanic scaling only acts in eruptions
        assert tas[yr(y), 6] != tas[yr(y), 0] and abs(rfv[yr(y), 6] - rfv[yr(y), 0]) > 0
    for k in (7, 8, 9):                                    # NPP / litter fractions: downstream impacts
        assert (np.abs(co2[yr(1850):, k] - co2[yr(1850):, 0]) > 0).all()
        assert (np.abs(tas[yr(1850):, k] - tas[yr(1850):, 0]) > 0).all()
    assert (npp[yr(1850):, 10] - npp[yr(1850):, 0]).min() > 0   # more CO2 fertilisation, more NPP
    # "land ocean warming ratio"
    keep = np.floor(np.linspace(1850, 2100, 30)).astype(int)
    lo = mk(2)
    assert lo.getvar("lo_warming_ratio")[0] == 0
    lo.setvar("lo_warming_ratio", np.array([0.0, 3.0]), "(unitless)")
    lo.set_outputs(["land_tas", "ocean_tas", "global_tas", "sst"]); lo.run(2100)
    lt, ot = lo.fetchvars("land_tas", (1745, 2100)), lo.fetchvars("ocean_tas", (1745, 2100))
    emergent = lt[keep - 1745, 0] / ot[keep - 1745, 0]
    assert np.unique(emergent).size == emergent.size
    ratio = lt[keep - 1745, 1] / ot[keep - 1745, 1]
    assert (np.abs(3.0 - ratio) <= 1e-5).all() and np.unique(np.round(ratio, 3)).size == 1
    # the global mean barely moves, land and ocean air temperatures do (test_parameters.R:422-435)
    gt = lo.fetchvars("global_tas", (1745, 2100))
    assert np.abs(gt[keep - 1745, 0] - gt[keep - 1745, 1]).mean() < 1e-1
    assert np.abs(lt[keep - 1745, 0] - lt[keep - 1745, 1]).mean() > 1e-1
    assert np.abs(ot[keep - 1745, 0] - ot[keep - 1745, 1]).mean() > 1e-1
    # ---- test_hector.R / test_messages.R: errors --------------------------------------
    with pytest.raises(hector_amd.HectorAmdError):
        hc.fetchvars("no_such_variable", (1800, 1801))
    with pytest.raises(hector_amd.HectorAmdError):
        hc.setvar("beta", 0.5, "degC")                     # wrong unit string
    hc.shutdown()
    with pytest.raises(hector_amd.HectorAmdError):
        hc.run(2100)                                       # inactive core


def test_reference_science_checks(emul_lib):
    science_checks(emul_lib, allow_emulation=True)
