"""The two-wavefront flavour of the one-biome run kernel (hx_run_kernel<HX_B1W2>,
hector_amd/csrc/hx_dev_member.h) in the host build of the kernel source (tests/emul).

The flavour changes WHERE a member's values wait between their uses (carbonate constants parked in
LDS through the step loop, constants read from the tables instead of the park, the block's SSTs
re-read from the output array, nothing requested a phase ahead) and nothing else: the host build
has no FMA contraction, so there it must reproduce the plain kernel BIT FOR BIT -- every output,
in pieces or in one launch, through the state history.  (On the GPU the two are separate
instantiations with independently contracted multiply-adds: tests/test_gpu_two_wave.py holds
them to rounding and both to the oracle.)"""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO

OUTS = ["CO2_concentration", "global_tas", "timesteps", "solver_steps", "RF_tot", "RF_CO2",
        "atmos_co2", "ocean_c", "veg_c", "detritus_c", "soil_c", "permafrost_c", "thawedp_c",
        "earth_c", "NBP", "ocean_uptake", "HL_pH", "LL_pH", "CH4_concentration", "O3_concentration",
        "sst", "land_tas", "heatflux"]


def mk(emul_lib, n, two_wave, outs, **params):
    c = hector_amd.Core(SCENARIO, n, lib_path=emul_lib, allow_emulation=True)
    c.set_two_wave_from(1 if two_wave else 0)
    for k, v in params.items():
        c.setvar(k, v)
    c.set_outputs(outs)
    return c


@pytest.mark.parametrize("outs", [["CO2_concentration", "global_tas"], OUTS],
                         ids=["plain-outputs", "every-output-and-heatflux"])
def test_two_wave_flavour_is_bitwise_the_plain_kernel_in_the_host_build(emul_lib, outs):
    n = 24
    S, q10 = ensemble.ecs_q10(n)
    a = mk(emul_lib, n, True, outs, S=S, q10_rh=q10)
    b = mk(emul_lib, n, False, outs, S=S, q10_rh=q10)
    a.run(2300); b.run(2300)
    assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run"
    assert (a.status() == 0).all()
    for v in outs:
        assert np.array_equal(a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))), v
    for row in range(27 + 7):   # the state table a later launch (of either kernel) starts from
        assert np.array_equal(a.state_row(row), b.state_row(row)), row


def test_two_wave_flavour_with_per_member_biome_constants_and_aerosol_scaling(emul_lib):
    """Rows that differ between members are read from the member's table rows, uniform ones through
    scalar loads: both paths against the plain kernel."""
    n = 16
    S, q10 = ensemble.ecs_q10(n)
    rng = np.random.default_rng(7)
    par = {"S": S, "q10_rh": q10, "beta": 0.3 + 0.4 * rng.random(n), "aero_scalar": 0.5 + rng.random(n),
           "vol_scalar": 0.5 + rng.random(n), "npp_flux0": 50.0 + 10.0 * rng.random(n),
           "warmingfactor": 1.0 + 0.5 * rng.random(n), "f_nppv": 0.3 + 0.1 * rng.random(n),
           "pf_mu": 1.6 + 0.2 * rng.random(n), "tt": 6.5e7 + 1e7 * rng.random(n)}
    outs = ["CO2_concentration", "global_tas", "timesteps", "permafrost_c"]
    a = mk(emul_lib, n, True, outs, **par)
    b = mk(emul_lib, n, False, outs, **par)
    a.run(2300); b.run(2300)
    assert a.last_run_kernel() == "run2"
    for v in outs:
        assert np.array_equal(a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))), v


def test_two_wave_flavour_in_pieces_through_history_and_across_kernels(emul_lib):
    n = 8
    S, q10 = ensemble.ecs_q10(n)
    outs = ["CO2_concentration", "global_tas", "timesteps"]
    ref = mk(emul_lib, n, False, outs, S=S, q10_rh=q10)
    ref.run(2300)
    want = {v: ref.fetchvars(v, (1745, 2300)) for v in outs}
    c = mk(emul_lib, n, True, outs, S=S, q10_rh=q10)
    c.enable_history()
    for y in (1760, 1761, 1850, 2300):   # DOECLIM blocks start where the launches do
        c.run(y)
    for v in outs:
        assert np.array_equal(want[v], c.fetchvars(v, (1745, 2300))), v
    c.reset(1901); c.run(2300)          # from the state history the two-wave kernel wrote
    for v in outs:
        assert np.array_equal(want[v], c.fetchvars(v, (1745, 2300))), v
    c.reset(1745); c.run(1950)          # two-wave kernel ...
    c.set_two_wave_from(0); c.run(2300)  # ... then the plain one on its state
    assert c.last_run_kernel() == "run"
    for v in outs:
        assert np.array_equal(want[v], c.fetchvars(v, (1745, 2300))), v


def test_two_wave_flavour_with_per_member_diffusivity(emul_lib):
    """Members that differ in ocean heat diffusivity (each its own DOECLIM kernel table; the
    flavour's history pass sweeps the block in four groups of 8 years instead of two of 16: the
    same sums in the same order), with and without the heat-flux output."""
    n = 12
    S, q10 = ensemble.ecs_q10(n)
    diff = np.linspace(1.2, 3.4, n)
    for outs in (["CO2_concentration", "global_tas", "sst"], ["CO2_concentration", "global_tas", "heatflux"]):
        a = mk(emul_lib, n, True, outs, S=S, q10_rh=q10, diff=diff)
        b = mk(emul_lib, n, False, outs, S=S, q10_rh=q10, diff=diff)
        a.run(2300); b.run(2300)
        assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run"
        for v in outs:
            assert np.array_equal(a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))), v


def test_two_wave_flavour_of_the_extended_kernel(emul_lib, tmp_path):
    """Constraints, a land-ocean warming ratio and the diagnostics written from inside the stash:
    the extended instantiation in the two-wavefront flavour against the plain one, bit for bit."""
    from conftest import edited_pack
    n = 8
    S, q10 = ensemble.ecs_q10(n)
    years = np.arange(1950, 2011)
    path = edited_pack(tmp_path / "tas.hxs", "temperature", "tas_constrain", years, 0.3 + 0.01 * (years - 1950))
    outs = ["CO2_concentration", "global_tas", "sst", "land_tas", "gmst", "NPP", "RH", "HL_ocean_uptake",
            "f_frozen", "timesteps"]
    res = []
    for two_wave in (True, False):
        c = hector_amd.Core(path, n, lib_path=emul_lib, allow_emulation=True)
        c.set_two_wave_from(1 if two_wave else 0)
        c.setvar("S", S).setvar("q10_rh", q10).setvar("lo_warming_ratio", np.where(np.arange(n) % 2, 1.6, 0.0))
        c.set_outputs(outs)
        c.run(2300)
        assert c.last_run_kernel() == ("run2" if two_wave else "run") and (c.status() == 0).all()
        res.append({v: c.fetchvars(v, (1745, 2300)) for v in outs})
    for v in outs:
        assert np.array_equal(res[0][v], res[1][v]), v


def test_two_wave_flavour_is_not_taken_where_it_does_not_apply(emul_lib, monkeypatch):
    monkeypatch.delenv("HECTOR_AMD_TWO_WAVE_FROM", raising=False)
    n = 4
    c = hector_amd.Core(SCENARIO, n, lib_path=emul_lib, allow_emulation=True)
    c.set_outputs(["CO2_concentration"])
    c.run(1750)
    assert c.last_run_kernel() == "run"          # default: only beyond one wavefront per SIMD
    c.shutdown()
    c = hector_amd.Core(SCENARIO, n, lib_path=emul_lib, allow_emulation=True)
    c.set_two_wave_from(1)
    c.split_biome(["a", "b"])
    c.set_outputs(["CO2_concentration"])
    c.run(1750)
    assert c.last_run_kernel() == "run"          # one biome only
