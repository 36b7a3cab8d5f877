"""The two-wavefronts-per-SIMD flavour of the one-biome run kernel (hx_run_kernel<HX_B1W2>,
hector_amd/csrc/hx_dev_member.h) on the MI355X.

Ensembles of more wavefronts than the GPU has SIMDs take it by default (`hx_set_two_wave_from`);
the tests force it on smaller ones too.  Same criterion as the other parity tests: the HIP path
against the CPU oracle, tolerance 2e-8 (north star 1e-6), identical stash schedules; against the
plain kernel the decisions must be identical and the trajectories agree to rounding (the two are
separate instantiations whose multiply-add pairs the compiler contracts independently)."""
import json
import os

import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import ROOT, SCENARIO

pytestmark = pytest.mark.gpu

REL_CO2 = 2e-8
ABS_T = 2e-8
OUTS = ["CO2_concentration", "global_tas", "timesteps", "solver_steps", "RF_tot", "atmos_co2",
        "ocean_c", "veg_c", "detritus_c", "soil_c", "permafrost_c", "thawedp_c", "earth_c", "NBP",
        "ocean_uptake", "HL_pH", "LL_pH", "CH4_concentration", "O3_concentration", "sst", "land_tas"]


def _core(hip_lib, n, two_wave, outs=OUTS):
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    assert c.backend == "hip"
    c.set_pair_kernel_limit(0)
    c.set_two_wave_from(1 if two_wave else 0)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(outs)
    return c, S, q10


def test_two_wave_kernel_vs_plain_kernel_and_oracle(hip_lib, oracle):
    n = 4096
    a, S, q10 = _core(hip_lib, n, True)
    b, _, _ = _core(hip_lib, n, False)
    a.run(2300); b.run(2300)
    assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run"
    assert (a.status() == 0).all() and (b.status() == 0).all()
    worst = {}
    for v in OUTS:
        x, y = a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))
        if v in ("timesteps", "solver_steps"):
            assert np.array_equal(x, y), v     # the same decisions, member by member
            continue
        # (annual fluxes are differences of the pools: the pools' absolute noise, ~1e-9 of the
        # atmosphere's ~1000 Pg C for the worst member in thousands, shows in them undiminished)
        scale = 1000.0 if v in ("NBP", "ocean_uptake") else np.maximum(np.abs(y), 1.0)
        worst[v] = float((np.abs(x - y) / scale).max())
    print("two-wave against plain kernel, worst scaled differences:", json.dumps(worst))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_two_wave_vs_plain_4096x1.json"), "w") as f:
        json.dump({"members": n, "years": 555, "worst_scaled_difference_by_variable": worst,
                   "scale": "max(|plain kernel's value|, 1); annual fluxes (NBP, ocean_uptake): 1000 Pg C", "tolerance": REL_CO2,
                   "stash_schedules_and_step_counts": "identical"}, f, indent=1)
    # (the same bar as against the oracle: both kernels are held to it there)
    for v, w in worst.items():
        assert w < REL_CO2, (v, w)
    # ... and against the oracle, a sample of the members
    idx = np.arange(0, n, 32)
    oco2, otg, err = oracle.run_ecs_q10(S[idx], q10[idx])
    assert err == 0
    co2 = a.fetchvars("CO2_concentration", (1745, 2300))[:, idx].T
    tg = a.fetchvars("global_tas", (1745, 2300))[:, idx].T
    assert (np.abs(co2 - oco2) / oco2).max() < REL_CO2
    assert np.abs(tg - otg).max() < ABS_T
    a.shutdown(); b.shutdown()


def test_two_wave_kernel_in_pieces_equals_one_launch(hip_lib):
    n = 1024
    a, _, _ = _core(hip_lib, n, True, ["CO2_concentration", "global_tas", "timesteps"])
    b, _, _ = _core(hip_lib, n, True, ["CO2_concentration", "global_tas", "timesteps"])
    a.run(2300)
    for y in (1760, 1761, 1850, 1999, 2100, 2300):   # block starts move with the launches
        b.run(y)
    assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run2"
    for v in ("CO2_concentration", "global_tas", "timesteps"):
        assert np.array_equal(a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))), v
    # reset(startDate) + run reproduces the run
    ref = a.fetchvars("CO2_concentration", (1745, 2300))
    a.reset(1745); a.run(2300)
    assert np.array_equal(ref, a.fetchvars("CO2_concentration", (1745, 2300)))
    a.shutdown(); b.shutdown()


def test_two_wave_kernel_with_state_history_and_kernel_changes(hip_lib):
    """reset(date) from the state history the two-wave kernel writes, and a run that changes
    kernels between launches (the state table is the interface)."""
    n = 512
    outs = ["CO2_concentration", "global_tas"]
    a, _, _ = _core(hip_lib, n, True, outs)
    a.enable_history()
    a.run(2300)
    ref = {v: a.fetchvars(v, (1745, 2300)) for v in outs}
    a.reset(1900); a.run(2300)
    for v in outs:
        assert np.array_equal(ref[v], a.fetchvars(v, (1745, 2300))), v
    # plain kernel to 1900, two-wave kernel from there: agrees with the plain kernel throughout
    b, _, _ = _core(hip_lib, n, False, outs)
    b.run(1900)
    assert b.last_run_kernel() == "run"
    b.set_two_wave_from(1)
    b.run(2300)
    assert b.last_run_kernel() == "run2"
    c, _, _ = _core(hip_lib, n, False, outs)
    c.run(2300)
    for v in outs:
        x, y = b.fetchvars(v, (1745, 2300)), c.fetchvars(v, (1745, 2300))
        assert (np.abs(x - y) / np.maximum(np.abs(y), 1.0)).max() < REL_CO2, v
    a.shutdown(); b.shutdown(); c.shutdown()


def test_two_wave_kernel_is_the_default_beyond_one_wavefront_per_simd(hip_lib):
    n = 65536 + 64
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    S, q10 = ensemble.ecs_q10(n)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration"])
    c.run(1760)
    assert c.last_run_kernel() == "run2"
    c.shutdown()
    c = hector_amd.Core(SCENARIO, 65536, device=0, lib_path=hip_lib)
    c.set_outputs(["CO2_concentration"])
    c.run(1760)
    assert c.last_run_kernel() == "run"
    c.shutdown()


def test_config4_share_of_one_gpu_every_member_vs_oracle(hip_lib, oracle):
    """131 072 members -- what one GPU holds of BASELINE configs[3]'s 1 048 576 over eight -- on the
    two-wave kernel, EVERY member against the oracle."""
    from test_gpu_fullsize import _compare
    n = 131072
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert c.last_run_kernel() == "run2"
    assert (c.status() == 0).all()

    def mp(i):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        return p
    _compare("config4_share_131072x1_two_wave", c, oracle, mp, np.arange(n), n,
             {"ensemble": "S ~ U(1.5, 6), q10_rh ~ U(1, 3), seed 20260928 (hector_amd/ensemble.py)",
              "kernel": c.last_run_kernel()})
    c.shutdown()


def test_two_wave_kernel_heat_flux_instantiation_vs_plain_kernel_and_oracle(hip_lib, oracle):
    """hx_run_kernel<HX_B1W2, HF>: the second history sum (16 accumulator tiles in VGPRs) behind
    the `heatflux` output -- against the plain kernel's heat-flux instantiation and the oracle."""
    n = 2048
    outs = ["CO2_concentration", "global_tas", "heatflux", "timesteps"]
    a, S, q10 = _core(hip_lib, n, True, outs)
    b, _, _ = _core(hip_lib, n, False, outs)
    a.run(2300); b.run(2300)
    assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run"
    assert (a.status() == 0).all()
    assert np.array_equal(a.fetchvars("timesteps", (1745, 2300)), b.fetchvars("timesteps", (1745, 2300)))
    for v in ("CO2_concentration", "global_tas", "heatflux"):
        x, y = a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))
        assert (np.abs(x - y) / np.maximum(np.abs(y), 1.0)).max() < REL_CO2, v
    for i in (0, 777, 2047):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        r, err, _ = oracle.run(p)
        assert err == 0
        assert np.abs(a.fetchvars("heatflux", (1745, 2300))[:, i] - r["heatflux"]).max() < ABS_T
        assert np.abs(a.fetchvars("global_tas", (1745, 2300))[:, i] - r["global_tas"]).max() < ABS_T
    a.shutdown(); b.shutdown()


def test_two_wave_kernel_with_rows_that_differ_between_members(hip_lib, oracle):
    """Every table row the flavour reads through a scalar load when the members share it --
    biome constants, warming factor, aerosol / volcanic scaling, C0, ocean transports -- perturbed
    per member here: the member's own rows are read instead.  Against the plain kernel and, for a
    few members, the oracle."""
    n = 1024
    S, q10 = ensemble.ecs_q10(n)
    rng = np.random.default_rng(11)
    par = {"beta": 0.3 + 0.4 * rng.random(n), "aero_scalar": 0.5 + rng.random(n),
           "vol_scalar": 0.5 + rng.random(n), "npp_flux0": 50.0 + 10.0 * rng.random(n),
           "warmingfactor": 1.0 + 0.5 * rng.random(n), "f_nppv": 0.3 + 0.1 * rng.random(n),
           "pf_mu": 1.6 + 0.2 * rng.random(n), "tt": 6.5e7 + 1e7 * rng.random(n),
           "C0": 270.0 + 15.0 * rng.random(n)}
    outs = ["CO2_concentration", "global_tas", "timesteps"]
    cores = []
    for two_wave in (True, False):
        c, _, _ = _core(hip_lib, n, two_wave, outs)
        for k, v in par.items():
            c.setvar(k, v)
        c.run(2300)
        assert (c.status() == 0).all()
        cores.append(c)
    a, b = cores
    assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run"
    assert np.array_equal(a.fetchvars("timesteps", (1745, 2300)), b.fetchvars("timesteps", (1745, 2300)))
    for v in ("CO2_concentration", "global_tas"):
        x, y = a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))
        assert (np.abs(x - y) / np.maximum(np.abs(y), 1.0)).max() < REL_CO2, v
    for i in (0, 513, 1023):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        p.beta[0] = par["beta"][i]; p.aero_scalar = par["aero_scalar"][i]; p.vol_scalar = par["vol_scalar"][i]
        p.npp_flux0[0] = par["npp_flux0"][i]; p.warmingfactor[0] = par["warmingfactor"][i]
        p.f_nppv[0] = par["f_nppv"][i]; p.pf_mu[0] = par["pf_mu"][i]; p.tt = par["tt"][i]; p.C0 = par["C0"][i]
        r, err, _ = oracle.run(p)
        assert err == 0
        ref = r["CO2_concentration"]
        assert (np.abs(a.fetchvars("CO2_concentration", (1745, 2300))[:, i] - ref) / ref).max() < REL_CO2
        assert np.abs(a.fetchvars("global_tas", (1745, 2300))[:, i] - r["global_tas"]).max() < ABS_T
    a.shutdown(); b.shutdown()


def test_measured_cost_lane_order_pairs_costly_with_cheap_wavefronts(hip_lib):
    """Beyond one wavefront per SIMD the measured-cost lane order puts the costliest wavefronts
    first (one per SIMD) and the next batch in ASCENDING cost, so that the wavefront that joins the
    costliest one on its SIMD is the cheapest: same results bit for bit under either order."""
    n = 131072
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration", "global_tas", "solver_steps", "timesteps"])
    c.run(2300)
    assert c.last_run_kernel() == "run2" and not c.lanes_calibrated()
    ref = {v: c.fetchvars(v, (2290, 2300)).copy() for v in ("CO2_concentration", "global_tas")}
    cost = 4 * c.fetchvars("solver_steps", (1746, 2300)).sum(0) + 5 * c.fetchvars("timesteps", (1746, 2300)).sum(0)
    ms0 = c.last_run_ms()
    c.reset(1745)
    assert c.lanes_calibrated()
    by_wave = cost[np.argsort(c.lane_of_member())].reshape(-1, 64).mean(1)   # 2048 wavefronts
    simds = 1024
    assert (np.diff(by_wave[:simds]) <= 1e-9).all()       # the first batch: costliest first
    assert (np.diff(by_wave[simds:]) >= -1e-9).all()      # the second: cheapest first
    assert by_wave[simds - 1] >= by_wave[-1] - 1e-9
    c.run(2300)
    for v, x in ref.items():
        assert np.array_equal(c.fetchvars(v, (2290, 2300)), x), v
    print("kernel ms, parameter key / measured cost paired: %.3f / %.3f" % (ms0, c.last_run_ms()))
    c.shutdown()


@pytest.mark.parametrize("heatflux", [False, True])
def test_two_wave_kernel_with_per_member_diffusivity(hip_lib, oracle, heatflux):
    """hx_run_kernel<HX_B1W2, HF, KERPM>: every member its own DOECLIM kernel table, the history
    pass on the vector ALU in four sweeps of 8 block years -- against the plain kernel's
    instantiation and the oracle."""
    n = 1024
    outs = ["CO2_concentration", "global_tas", "timesteps"] + (["heatflux"] if heatflux else [])
    S, q10 = ensemble.ecs_q10(n)
    diff = 1.2 + 2.2 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 5)
    cores = []
    for two_wave in (True, False):
        c, _, _ = _core(hip_lib, n, two_wave, outs)
        c.setvar("diff", diff, "cm2/s")
        c.run(2300)
        assert (c.status() == 0).all()
        cores.append(c)
    a, b = cores
    assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run"
    assert np.array_equal(a.fetchvars("timesteps", (1745, 2300)), b.fetchvars("timesteps", (1745, 2300)))
    for v in outs[:2] + outs[3:]:
        x, y = a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))
        assert (np.abs(x - y) / np.maximum(np.abs(y), 1.0)).max() < REL_CO2, v
    for i in (0, 500, 1023):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]; p.diff = diff[i]
        r, err, _ = oracle.run(p)
        assert err == 0
        ref = r["CO2_concentration"]
        assert (np.abs(a.fetchvars("CO2_concentration", (1745, 2300))[:, i] - ref) / ref).max() < REL_CO2
        assert np.abs(a.fetchvars("global_tas", (1745, 2300))[:, i] - r["global_tas"]).max() < ABS_T
        if heatflux:
            assert np.abs(a.fetchvars("heatflux", (1745, 2300))[:, i] - r["heatflux"]).max() < ABS_T
    a.shutdown(); b.shutdown()


def test_two_wave_flavour_of_the_extended_kernel_on_gpu(hip_lib, oracle, tmp_path):
    """hx_run_kernel<HX_B1W2, HF, KERPM, 1>: a tas constraint, a land-ocean warming ratio for every
    other member and diagnostics written from inside the stash -- against the plain extended
    kernel (same decisions) and, for a few members, the oracle reading the same scenario."""
    import oracle_binding
    from conftest import edited_pack
    n = 1024
    S, q10 = ensemble.ecs_q10(n)
    years = np.arange(1950, 2011)
    path = edited_pack(tmp_path / "tas.hxs", "temperature", "tas_constrain", years, 0.3 + 0.01 * (years - 1950))
    lo = np.where(np.arange(n) % 2, 1.6, 0.0)
    outs = ["CO2_concentration", "global_tas", "sst", "land_tas", "gmst", "NPP", "RH", "timesteps"]
    cores = []
    for two_wave in (True, False):
        c = hector_amd.Core(path, n, device=0, lib_path=hip_lib)
        c.set_pair_kernel_limit(0)
        c.set_two_wave_from(1 if two_wave else 0)
        c.setvar("S", S, "degC").setvar("q10_rh", q10).setvar("lo_warming_ratio", lo)
        c.set_outputs(outs)
        c.run(2300)
        assert (c.status() == 0).all()
        cores.append(c)
    a, b = cores
    assert a.last_run_kernel() == "run2" and b.last_run_kernel() == "run"
    assert np.array_equal(a.fetchvars("timesteps", (1745, 2300)), b.fetchvars("timesteps", (1745, 2300)))
    for v in outs[:-1]:
        x, y = a.fetchvars(v, (1745, 2300)), b.fetchvars(v, (1745, 2300))
        scale = 1000.0 if v in ("NPP", "RH") else np.maximum(np.abs(y), 1.0)
        assert (np.abs(x - y) / scale).max() < REL_CO2, v
    o = oracle_binding.Oracle(path)
    for i in (0, 1, 511, 1022):
        p = o.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]; p.lo_warming_ratio = lo[i]
        r, err, _ = o.run(p)
        assert err == 0
        ref = r["CO2_concentration"]
        assert (np.abs(a.fetchvars("CO2_concentration", (1745, 2300))[:, i] - ref) / ref).max() < REL_CO2
        for v in ("global_tas", "sst", "land_tas"):
            assert np.abs(a.fetchvars(v, (1745, 2300))[:, i] - r[v]).max() < ABS_T, (v, i)
    a.shutdown(); b.shutdown()


def test_wave_clock_and_cost_model_on_gpu(hip_lib, monkeypatch):
    """hx_wave_clock: every wavefront of the last launch with its start and end (100 MHz ticks);
    the launch lasts as long as its last wavefront.  And the fitted cost model on the GPU: a core's
    measured costs order the FIRST run of a later core of the same study (HECTOR_AMD_SIMDS: the
    lane-order logic of a GPU with 8 SIMDs, so that 4 096 members are 'more wavefronts than
    SIMDs'), results bit for bit those of the parameter-key order."""
    monkeypatch.setenv("HECTOR_AMD_SIMDS", "8")
    n = 4096
    S, q = ensemble.ecs_q10(n)
    a = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    a.set_pair_kernel_limit(0)
    a.set_cost_model(False)    # (an earlier test of this process may have left a model for this key)
    a.setvar("S", S, "degC").setvar("q10_rh", q)
    assert len(a.wave_clock()) == 0                       # nothing has run yet
    a.run(2300)
    assert a.last_run_kernel() == "run2" and a.lane_order_source() == "parameter key"
    a.set_cost_model(True)     # its measured costs make (or replace) the model at the reset below
    w = a.wave_clock()
    assert w.shape == (n // 64, 2) and (w[:, 1] > w[:, 0]).all() and w[:, 0].min() == 0
    span_ms = (w[:, 1].max() - w[:, 0].min()) * 1e-5
    assert 0.5 * a.last_run_ms() < span_ms <= 1.05 * a.last_run_ms() + 0.05
    ref = a.fetchvars("CO2_concentration", (1745, 2300)).copy()
    a.reset(1745)
    assert a.lane_order_source() == "measured cost"
    S2, q2 = ensemble.ecs_q10(n, offset=20000)
    b = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    b.set_pair_kernel_limit(0)
    b.setvar("S", S2, "degC").setvar("q10_rh", q2)
    b.status()
    assert b.lane_order_source() == "cost model"
    b.run(2300)
    off = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    off.set_pair_kernel_limit(0).set_cost_model(False)
    off.setvar("S", S2, "degC").setvar("q10_rh", q2)
    off.run(2300)
    assert off.lane_order_source() == "parameter key"
    assert not np.array_equal(off.lane_of_member(), b.lane_of_member())
    for v in ("CO2_concentration", "global_tas"):
        assert np.array_equal(off.fetchvars(v), b.fetchvars(v)), v
    # the pair kernel stamps two wavefronts per 64 members
    c = hector_amd.Core(SCENARIO, 256, device=0, lib_path=hip_lib)
    c.setvar("S", S[:256], "degC").run(1800)
    assert c.last_run_kernel() == "pair" and c.wave_clock().shape == (8, 2)
    a.run(2300)
    assert np.array_equal(a.fetchvars("CO2_concentration", (1745, 2300)), ref)
    for x in (a, b, off, c):
        x.shutdown()


@pytest.mark.gpu
def test_prewarm_and_shipped_cost_model_on_gpu(hip_lib, monkeypatch):
    """Round 6, the one-shot run: (i) the prewarm loop (hx_set_prewarm) that keeps the chip's clocks
    up while run()'s preparation uploads and spins up is transparent -- bit-identical results with
    and without it, hx_last_run_prewarmed says which launch was behind it, a second run of a warm
    core is not; (ii) a core whose workload the SHIPPED models know (hector_amd/data/cost_models.txt:
    SSP2-4.5, ECS x Q10) orders its lanes by the model from its first upload on, with no earlier
    core to learn from (HECTOR_AMD_SIMDS: the lane-order logic of a GPU with 8 SIMDs), and one the
    file does not know (another uniform beta: another key) keeps the parameter key."""
    n = 4096
    S, q = ensemble.ecs_q10(n)
    res = {}
    for ms in (0, 50):
        c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
        c.set_pair_kernel_limit(0).set_prewarm(ms)
        c.setvar("S", S, "degC").setvar("q10_rh", q)
        c.run(2300)
        assert c.last_run_prewarmed() == (ms > 0)
        res[ms] = (c.fetchvars("CO2_concentration", (1745, 2300)).copy(), c.fetchvars("global_tas", (1745, 2300)).copy())
        assert (c.status() == 0).all()
        c.reset(1745); c.run(2300)                 # a warm core, nothing to prepare: no loop
        assert not c.last_run_prewarmed()
        c.shutdown()
    assert np.array_equal(res[0][0], res[50][0]) and np.array_equal(res[0][1], res[50][1])
    models = os.path.join(os.path.dirname(os.path.dirname(hip_lib)), "data", "cost_models.txt")
    assert os.path.exists(models) and open(models).read().count("\nmodel ") >= 9
    monkeypatch.setenv("HECTOR_AMD_SIMDS", "8")
    a = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    a.set_pair_kernel_limit(0)
    a.setvar("S", S, "degC").setvar("q10_rh", q)
    a.status()
    assert a.lane_order_source() == "cost model"
    a.set_two_wave_from(0)     # (the same kernel flavour as above: bit-identical whatever the lane order)
    a.run(2300)
    assert a.last_run_kernel() == "run"
    cost = a.lane_of_member()
    assert sorted(cost) == list(range(n))
    assert np.array_equal(a.fetchvars("CO2_concentration", (1745, 2300)), res[0][0])   # whatever the order
    b = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    b.set_pair_kernel_limit(0)
    b.setvar("S", S, "degC").setvar("q10_rh", q).setvar("beta", np.full(n, 0.5))
    b.status()
    assert b.lane_order_source() == "parameter key"
    a.shutdown(); b.shutdown()
