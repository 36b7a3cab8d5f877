"""The small-ensemble kernel (two wavefronts per 64 members, hector_amd/csrc/hx_dev_pair.h) against
the oracle and against the one-wavefront run kernel.

Ensembles of up to 32 768 members (one biome, no constraints but scenario-wide CO2 / tas / RF_tot / CH4
ones, the usual outputs) take it by
default; `set_pair_kernel_limit(0)` forces the run kernel.  Same criterion as the other parity
tests (test_gpu_parity.py); the per-year stash schedule ("timesteps": every retry and
reduced-timestep decision of the reference) has to agree with the oracle member by
member -- both wavefronts of a pair take those decisions independently from the same inputs."""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO

pytestmark = pytest.mark.gpu

REL_CO2 = 2e-8
ABS_T = 2e-8


def mk(hip_lib, n, S, q10, limit=None):
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    assert c.backend == "hip"
    if limit is not None:
        c.set_pair_kernel_limit(limit)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    return c


@pytest.mark.parametrize("n", [1024, 32768])
def test_pair_kernel_vs_oracle_config2(hip_lib, oracle, n):
    """BASELINE configs[1] (1 024 members with perturbed ECS and Q10) and the largest ensemble the
    kernel serves (32 768: every SIMD busy), every member, 555 years."""
    from test_gpu_fullsize import _oracle_all
    S, q10 = ensemble.ecs_q10(n)
    c = mk(hip_lib, n, S, q10)
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert c.last_run_kernel() == "pair"
    assert (c.status() == 0).all()

    def mp(i):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        return p
    oco2, otg, ots, oerr = _oracle_all(oracle, mp, n)
    assert (oerr == 0).all()
    co2 = c.fetchvars("CO2_concentration", (1745, 2300)).T
    tg = c.fetchvars("global_tas", (1745, 2300)).T
    ts = c.fetchvars("timesteps", (1745, 2300)).T
    assert (np.abs(co2 - oco2) / oco2).max() < REL_CO2
    assert np.abs(tg - otg).max() < ABS_T
    assert int((ts.astype(np.int64) != ots.astype(np.int64)).any(axis=1).sum()) == 0


@pytest.mark.parametrize("n", [1, 64, 100, 4096])
def test_pair_kernel_equals_run_kernel(hip_lib, n):
    """Both kernels on the same ensemble (ragged sizes too): trajectories to 1e-8 relative,
    identical stash schedules, identical status."""
    S, q10 = ensemble.ecs_q10(n)
    out = {}
    for name, limit in (("pair", None), ("run", 0)):
        c = mk(hip_lib, n, S, q10, limit)
        c.set_outputs(["CO2_concentration", "global_tas", "timesteps", "RF_tot", "RF_CO2"])
        c.run(2300)
        assert c.last_run_kernel() == name
        out[name] = (c.status(), c.fetchvars("CO2_concentration"), c.fetchvars("global_tas"),
                     c.fetchvars("sst"), c.fetchvars("land_tas"), c.fetchvars("timesteps"),
                     c.fetchvars("RF_tot"), c.fetchvars("RF_CO2"))
    p, r = out["pair"], out["run"]
    assert np.array_equal(p[0], r[0]) and (p[0] == 0).all()
    assert (np.abs(p[1] - r[1]) / r[1]).max() < 1e-8
    for k in (2, 3, 4, 6, 7):
        assert np.abs(p[k] - r[k]).max() < 1e-8
    assert np.array_equal(p[5], r[5])


POOLS = ["atmos_co2", "ocean_c", "ocean_uptake", "HL_pH", "LL_pH", "CH4_concentration",
         "O3_concentration", "NBP", "veg_c", "detritus_c", "soil_c", "permafrost_c", "thawedp_c",
         "earth_c", "NPP", "RH", "rh_det", "rh_soil", "rh_ch4", "f_frozen", "gmst"]


@pytest.mark.parametrize("n", [100, 2048])
def test_pair_kernel_pools_and_fluxes_equal_run_kernel(hip_lib, n):
    """The carbon pools, NBP, ocean uptake, surface pH and the CH4/O3 concentrations are recorded
    by the pair kernel too (each by the wavefront that holds them): same values as the run kernel,
    and switching them on does not change the trajectory."""
    S, q10 = ensemble.ecs_q10(n)
    out = {}
    for name, limit in (("pair", None), ("run", 0)):
        c = mk(hip_lib, n, S, q10, limit)
        c.set_outputs(["CO2_concentration", "global_tas", "timesteps"] + POOLS)
        c.run(2300)
        assert c.last_run_kernel() == name
        assert (c.status() == 0).all()
        out[name] = {v: c.fetchvars(v) for v in ["CO2_concentration", "global_tas", "timesteps"] + POOLS}
    p, r = out["pair"], out["run"]
    assert np.array_equal(p["timesteps"], r["timesteps"])
    for v in POOLS + ["CO2_concentration", "global_tas"]:
        if v in ("NBP", "ocean_uptake", "NPP", "RH", "rh_det", "rh_soil"):
            # a year's flux is the change of a 600-40 000 Pg C pool that the two kernels hold to
            # ~1e-10 relative: absolute criterion, on the pool's scale
            assert np.abs(p[v] - r[v]).max() < 2e-6, v
            continue
        scale = np.maximum(np.abs(r[v]), 1.0)
        assert (np.abs(p[v] - r[v]) / scale).max() < 1e-8, v
    plain = mk(hip_lib, n, S, q10).run(2300)
    assert plain.last_run_kernel() == "pair"
    assert np.array_equal(plain.fetchvars("CO2_concentration"), p["CO2_concentration"])
    assert np.array_equal(plain.fetchvars("global_tas"), p["global_tas"])


def test_pair_kernel_pools_vs_oracle(hip_lib, oracle):
    """The same outputs against the oracle's own trajectory of the default member."""
    c = hector_amd.Core(SCENARIO, 64, device=0, lib_path=hip_lib)
    c.set_outputs(["CO2_concentration"] + POOLS)
    c.run(2300)
    assert c.last_run_kernel() == "pair"
    r, err, _ = oracle.run(oracle.default_params())
    assert err == 0
    checked = 0
    for v in POOLS:
        if v not in r:
            continue
        got = c.fetchvars(v, (1746, 2300))[:, 0]
        ref = np.asarray(r[v])[1:]
        assert np.abs(got - ref).max() < 2e-8 * max(1.0, np.abs(ref).max()), v
        checked += 1
    assert checked == len(POOLS)


@pytest.mark.parametrize("n", [96, 4096])
def test_pair_kernel_with_per_member_diffusivity(hip_lib, oracle, n):
    """Members that differ in ocean heat diffusivity have their own DOECLIM kernel table (the
    history pass runs per lane instead of on the matrix pipe): against the run kernel and against
    the oracle, every member, across several 32-year history blocks."""
    S, q10 = ensemble.ecs_q10(n)
    diff = np.linspace(0.6, 3.0, n)
    out = {}
    for name, limit in (("pair", None), ("run", 0)):
        c = mk(hip_lib, n, S, q10, limit)
        c.setvar("diff", diff, "cm2/s")
        c.set_outputs(["CO2_concentration", "global_tas", "sst", "timesteps", "RF_tot"])
        c.run(2300)
        assert c.last_run_kernel() == name
        assert (c.status() == 0).all()
        out[name] = {v: c.fetchvars(v, (1745, 2300)) for v in
                     ("CO2_concentration", "global_tas", "sst", "timesteps", "RF_tot")}
    p, r = out["pair"], out["run"]
    assert np.array_equal(p["timesteps"], r["timesteps"])
    assert (np.abs(p["CO2_concentration"] - r["CO2_concentration"]) / r["CO2_concentration"]).max() < 1e-8
    for v in ("global_tas", "sst", "RF_tot"):
        assert np.abs(p[v] - r[v]).max() < 1e-8, v
    # the same run in pieces (history blocks restart at every launch): bit for bit
    c = mk(hip_lib, n, S, q10)
    c.setvar("diff", diff, "cm2/s")
    c.set_outputs(["CO2_concentration", "global_tas", "sst", "timesteps", "RF_tot"])
    for y in (1746, 1777, 1778, 1900, 2107, 2300):
        c.run(y)
        assert c.last_run_kernel() == "pair"
    for v in ("CO2_concentration", "global_tas", "sst"):
        assert np.array_equal(c.fetchvars(v, (1745, 2300)), p[v]), v
    from test_gpu_fullsize import _oracle_all

    def mp(i):
        q = oracle.default_params(); q.S = S[i]; q.q10_rh[0] = q10[i]; q.diff = diff[i]
        return q
    oco2, otg, ots, oerr = _oracle_all(oracle, mp, n)          # every member
    assert (oerr == 0).all()
    assert (np.abs(p["CO2_concentration"].T - oco2) / oco2).max() < REL_CO2
    assert np.abs(p["global_tas"].T - otg).max() < ABS_T
    assert int((p["timesteps"].T.astype(np.int64) != ots.astype(np.int64)).any(axis=1).sum()) == 0


def test_pair_kernel_in_segments_reset_and_handover(hip_lib):
    """A run in pieces, a reset, and a hand-over between the two kernels in the middle of a run
    (the state table is common) give the one-launch trajectory."""
    n = 200
    S, q10 = ensemble.ecs_q10(n)
    a = mk(hip_lib, n, S, q10).run(2300)
    assert a.last_run_kernel() == "pair"
    first = {v: a.fetchvars(v) for v in ("CO2_concentration", "global_tas", "sst", "land_tas")}
    b = mk(hip_lib, n, S, q10)
    for y in (1746, 1777, 1778, 1900, 2107, 2300):
        b.run(y)
        assert b.last_run_kernel() == "pair"
    for v, x in first.items():
        assert np.array_equal(x, b.fetchvars(v)), v
    a.reset(0).run(2300)
    assert np.array_equal(first["CO2_concentration"], a.fetchvars("CO2_concentration"))
    # first half on one kernel, second half on the other
    c = mk(hip_lib, n, S, q10).run(1990)
    c.set_pair_kernel_limit(0)
    c.run(2300)
    assert c.last_run_kernel() == "run"
    d = mk(hip_lib, n, S, q10, 0).run(1990)
    d.set_pair_kernel_limit(32768)
    d.run(2300)
    assert d.last_run_kernel() == "pair"
    for x in (c, d):
        assert (x.status() == 0).all()
        assert (np.abs(x.fetchvars("CO2_concentration") - first["CO2_concentration"]) /
                first["CO2_concentration"]).max() < 1e-8
        assert np.abs(x.fetchvars("global_tas") - first["global_tas"]).max() < 1e-8


@pytest.mark.parametrize("own_diffusivity", [False, True], ids=["shared-kernel-table", "per-member-table"])
def test_pair_kernel_records_the_ocean_heat_flux(hip_lib, oracle, own_diffusivity):
    """"heatflux" (the R package's HEAT_FLUX) needs DOECLIM's second history sum: both
    instantiations that carry it, against the run kernel (all members) and the oracle (a sample),
    in one launch and in pieces."""
    n = 500
    S, q10 = ensemble.ecs_q10(n)
    diff = np.linspace(0.6, 3.0, n) if own_diffusivity else None
    outs = ["CO2_concentration", "global_tas", "heatflux", "timesteps"]
    out = {}
    for name, limit in (("pair", None), ("run", 0)):
        c = mk(hip_lib, n, S, q10, limit)
        if own_diffusivity:
            c.setvar("diff", diff, "cm2/s")
        c.set_outputs(outs)
        c.run(2300)
        assert c.last_run_kernel() == name
        assert (c.status() == 0).all()
        out[name] = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
    p, r = out["pair"], out["run"]
    assert np.array_equal(p["timesteps"], r["timesteps"])
    assert np.abs(p["heatflux"] - r["heatflux"]).max() < 1e-8
    assert np.abs(p["heatflux"]).max() > 0.5          # (W/m2: a real signal, not zeros)
    assert np.abs(p["global_tas"] - r["global_tas"]).max() < 1e-8
    for i in range(0, n, 50):
        q = oracle.default_params(); q.S = S[i]; q.q10_rh[0] = q10[i]
        if own_diffusivity:
            q.diff = diff[i]
        o, err, _ = oracle.run(q)
        assert err == 0
        assert np.abs(p["heatflux"][1:, i] - o["heatflux"][1:]).max() < 2e-8, i
        assert np.abs(p["global_tas"][:, i] - o["global_tas"]).max() < ABS_T
    c = mk(hip_lib, n, S, q10)
    if own_diffusivity:
        c.setvar("diff", diff, "cm2/s")
    c.set_outputs(outs)
    for y in (1746, 1777, 1778, 1900, 2107, 2300):
        c.run(y)
        assert c.last_run_kernel() == "pair"
    assert np.array_equal(c.fetchvars("heatflux", (1745, 2300)), p["heatflux"])


def test_pair_kernel_state_history_and_reset_to_any_date(hip_lib):
    """Core::reset(date), core.cpp:511-549, on the two-wavefront kernel: each wavefront writes its
    rows of the year's state slab; the rerun from any computed year is bit-identical, the history
    changes no result, and slabs written by one kernel restart the other."""
    n = 1000
    S, q10 = ensemble.ecs_q10(n)
    outs = ["CO2_concentration", "global_tas", "ocean_c", "CH4_concentration", "timesteps", "NBP"]
    a = mk(hip_lib, n, S, q10)
    a.set_outputs(outs); a.run(2300)
    assert a.last_run_kernel() == "pair"
    ref = {v: a.fetchvars(v, (1745, 2300)) for v in outs}
    b = mk(hip_lib, n, S, q10)
    b.enable_history(True); b.set_outputs(outs); b.run(2300)
    assert b.last_run_kernel() == "pair"
    for v in outs:
        assert np.array_equal(b.fetchvars(v, (1745, 2300)), ref[v]), v
    for date in (2050, 1790, 1746, 2299):
        b.reset(date)
        assert b.current_date == date
        b.run(2300)
        assert b.last_run_kernel() == "pair"
        for v in outs:
            assert np.array_equal(b.fetchvars(v, (1745, 2300)), ref[v]), (v, date)
    # a dated edit: the run restarts from the edited year's slab, as a fresh core with the edit
    b.setvar_dated("ffi_emissions", [2030, 2031], [0.5, 0.5], "Pg C/yr")
    b.run(2300)
    f = mk(hip_lib, n, S, q10)
    f.set_outputs(outs); f.setvar_dated("ffi_emissions", [2030, 2031], [0.5, 0.5], "Pg C/yr"); f.run(2300)
    for v in outs:
        assert np.array_equal(b.fetchvars(v, (1745, 2300)), f.fetchvars(v, (1745, 2300))), v
    # slabs of one kernel, rerun on the other (the slab is the run kernel's state table)
    r = mk(hip_lib, n, S, q10, 0)
    r.enable_history(True); r.set_outputs(outs); r.run(2300)
    r.set_pair_kernel_limit(32768)
    r.reset(1900); r.run(2300)
    assert r.last_run_kernel() == "pair"
    b2 = mk(hip_lib, n, S, q10)
    b2.enable_history(True); b2.set_outputs(outs); b2.run(2300)
    b2.set_pair_kernel_limit(0)
    b2.reset(1900); b2.run(2300)
    assert b2.last_run_kernel() == "run"
    for x in (r, b2):
        assert (x.status() == 0).all()
        assert (np.abs(x.fetchvars("CO2_concentration", (1745, 2300)) - ref["CO2_concentration"]) /
                ref["CO2_concentration"]).max() < 1e-8
        assert np.array_equal(x.fetchvars("timesteps", (1745, 2300)), ref["timesteps"])
    # a member that fails keeps its flag in the year's status slab and loses it on a reset before
    npp = np.full(130, 56.2); npp[77] = 1e5
    e = hector_amd.Core(SCENARIO, 130, device=0, lib_path=hip_lib)
    e.enable_history(True); e.setvar("npp_flux0", npp); e.run(1800)
    assert e.last_run_kernel() == "pair"
    st = e.status(); assert st[77] != 0 and (np.delete(st, 77) == 0).all()
    e.reset(1799); assert e.status()[77] != 0


def test_pair_kernel_is_only_taken_where_it_applies(hip_lib, tmp_path):
    n = 128
    S, q10 = ensemble.ecs_q10(n)
    c = mk(hip_lib, n, S, q10)
    c.set_outputs(["CO2_concentration", "HL_PCO2"])          # an ocean-chemistry diagnostic: run kernel
    assert c.run(1800).last_run_kernel() == "run"
    c = mk(hip_lib, n, S, q10)
    c.split_biome(["a", "b"])                                 # two to four biomes: served (round 5)
    assert c.run(1800).last_run_kernel() == "pair"
    c = mk(hip_lib, n, S, q10)
    c.split_biome(["a", "b", "c", "d", "e"])                  # five: the run kernels
    assert c.run(1800).last_run_kernel() == "run"
    c = mk(hip_lib, n, S, q10)
    c.split_biome(["a", "b"]); c.set_outputs(["CO2_concentration", "NPP"])   # a split core's NPP: run kernel
    assert c.run(1800).last_run_kernel() == "run"
    c = mk(hip_lib, n, S, q10)
    c.setvar_dated_members("ffi_emissions", [1800], np.linspace(0.0, 1.0, n)[None, :])  # per-member series
    assert c.run(1810).last_run_kernel() == "run"
    c = mk(hip_lib, 40000, *ensemble.ecs_q10(40000))          # above the limit
    assert c.run(1760).last_run_kernel() == "run"
    c = mk(hip_lib, n, S, q10)
    assert c.run(1800).last_run_kernel() == "pair"


def test_pair_kernel_error_flags(hip_lib):
    """A member that leaves the model's domain is flagged by whichever wavefront notices, stops
    being integrated (no hang: both wavefronts must leave their loops together), and does not
    disturb its neighbours."""
    S = np.full(130, 3.0); npp = np.full(130, 56.2); npp[77] = 1e5
    c = hector_amd.Core(SCENARIO, 130, device=0, lib_path=hip_lib)
    c.setvar("S", S).setvar("npp_flux0", npp).run(1800)
    assert c.last_run_kernel() == "pair"
    st = c.status()
    r = hector_amd.Core(SCENARIO, 130, device=0, lib_path=hip_lib)
    r.set_pair_kernel_limit(0)
    r.setvar("S", S).setvar("npp_flux0", npp).run(1800)
    assert st[77] != 0 and (np.delete(st, 77) == 0).all()
    assert st[77] == r.status()[77]
    a = c.fetchvars("CO2_concentration")
    assert np.array_equal(a[:, 0], a[:, 129]) and np.isfinite(a[:, 0]).all()


@pytest.mark.parametrize("name", ["picontrol", "ssp119", "ssp126", "ssp370", "ssp434", "ssp460",
                                  "ssp534-over", "ssp585"])
def test_pair_kernel_shipped_scenarios_vs_oracle(hip_lib, name):
    """Every scenario the reference ships, 16 members each, on the two-wavefront kernel
    (test_gpu_parity.py runs them on the one-wavefront kernel: it records RF_tot)."""
    import oracle_binding
    from test_scenarios import pack
    path = pack(name)
    o = oracle_binding.Oracle(path)
    S, q10 = ensemble.ecs_q10(16, offset=1000)
    c = hector_amd.Core(path, 16, device=0, lib_path=hip_lib)
    c.setvar("S", S, "degC").setvar("q10_rh", q10)
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps", "RF_tot", "RF_CO2"])
    c.run(o.end)
    # (picontrol prescribes its CO2 -- a scenario-wide constraint: the CONS instantiation, round 5)
    assert c.last_run_kernel() == "pair"
    assert (c.status() == 0).all()
    co2 = c.fetchvars("CO2_concentration", (o.start, o.end))
    tg = c.fetchvars("global_tas", (o.start, o.end))
    ts = c.fetchvars("timesteps", (o.start + 1, o.end))
    for i in range(16):
        p = o.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        r, err, _ = o.run(p)
        assert err == 0
        assert (np.abs(co2[:, i] - r["CO2_concentration"]) / r["CO2_concentration"]).max() < REL_CO2
        assert np.abs(tg[:, i] - r["global_tas"]).max() < ABS_T
        assert np.abs(c.fetchvars("RF_tot", (o.start, o.end))[:, i] - r["RF_tot"]).max() < ABS_T
        assert np.abs(c.fetchvars("RF_CO2", (o.start, o.end))[:, i] - r["RF_CO2"]).max() < ABS_T
        assert np.array_equal(ts[:, i], r["timesteps"][1:])


@pytest.mark.timeout(300)
def test_pair_kernel_runaway_member_is_flagged_and_does_not_hang(hip_lib):
    """The member of test_host_logic.runaway_member_checks (SSP1-1.9, a carbon cycle that runs
    away around 2130) on the two-wavefront kernel: flagged and retired by both wavefronts
    together, neighbours untouched."""
    import os
    from conftest import ROOT
    from test_host_logic import RUNAWAY
    path = os.path.join(ROOT, "hector_amd", "data", "ssp119.hxs")
    out = {}
    for which, limit in (("pair", 32768), ("run", 0)):
        c = hector_amd.Core(path, 3, device=0, lib_path=hip_lib)
        c.set_pair_kernel_limit(limit)
        for k, v in RUNAWAY.items():
            base = c.getvar(k)[0]
            c.setvar(k, [base, v, base])
        c.set_outputs(["CO2_concentration", "timesteps"])
        c.run(2300)
        assert c.last_run_kernel() == which
        out[which] = (c.status(), c.fetchvars("CO2_concentration", (1745, 2300)), c.fetchvars("timesteps", (1745, 2300)))
    st, co2, ts = out["pair"]
    assert np.array_equal(st, out["run"][0])
    assert st[0] == 0 and st[2] == 0
    assert np.isfinite(co2).all()
    assert np.array_equal(co2[:, 0], co2[:, 2])
    ok = [0, 2] if st[1] else [0, 1, 2]
    assert (np.abs(co2[:, ok] - out["run"][1][:, ok]) / out["run"][1][:, ok]).max() < 1e-8
    assert np.array_equal(ts[:, ok], out["run"][2][:, ok])


@pytest.mark.parametrize("nb", [2, 3, 4])
def test_pair_kernel_with_biomes_vs_oracle(hip_lib, nb):
    """Two to four heterogeneous biomes on the small-ensemble kernel (the land wavefront owns the
    biome loops; hx_pair_kernel<false, false, false, NB>): every 9th of 128 members against the
    oracle, stash schedules identical, and the pools' totals against the run kernel's."""
    import oracle_binding
    n = 128
    idx = np.arange(n, dtype=np.uint64)
    S = 1.5 + 4.5 * ensemble.uniform01(idx, 0)
    q10 = [1.0 + 2.0 * ensemble.uniform01(idx, 10 + b) for b in range(nb)]
    wf = [np.full(n, 1.0 + 0.5 * b) for b in range(nb)]
    names = ["b%d" % b for b in range(nb)]
    outs = ["CO2_concentration", "global_tas", "timesteps", "veg_c", "soil_c", "permafrost_c", "thawedp_c", "f_frozen"]
    res = {}
    for limit, which in ((32768, "pair"), (0, "run")):
        c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
        c.set_pair_kernel_limit(limit)
        c.split_biome(names)
        c.setvar("S", S, "degC")
        for b, nm in enumerate(names):
            c.setvar(nm + ".q10_rh", q10[b]).setvar(nm + ".warmingfactor", wf[b])
        c.set_outputs(outs)
        c.run(2300)
        assert c.last_run_kernel() == which and (c.status() == 0).all()
        res[which] = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
        c.shutdown()
    o = oracle_binding.Oracle(SCENARIO)
    for i in range(0, n, 9):
        p = o.split_equal(o.default_params(), nb)
        p.S = S[i]
        for b in range(nb):
            p.q10_rh[b] = q10[b][i]; p.warmingfactor[b] = wf[b][i]
        r, err, _ = o.run(p)
        assert err == 0
        g = res["pair"]
        assert (np.abs(g["CO2_concentration"][:, i] - r["CO2_concentration"]) / r["CO2_concentration"]).max() < REL_CO2
        assert np.abs(g["global_tas"][:, i] - r["global_tas"]).max() < ABS_T
        assert np.array_equal(g["timesteps"][1:, i], r["timesteps"][1:])
        for v in ("veg_c", "soil_c", "permafrost_c", "thawedp_c"):
            assert np.abs(g[v][1:, i] - r[v][1:]).max() <= 2e-8 * max(1.0, np.abs(r[v]).max()), v
    for v in outs:   # and the two kernels agree with each other
        a, b = res["pair"][v], res["run"][v]
        assert np.abs(a - b).max() <= 5e-9 * max(1.0, np.abs(b).max()), v


def test_pair_kernel_constrained_split_ensemble_vs_run_kernel(hip_lib):
    """A CO2-constrained (concentration-driven) 4-biome ensemble of 256 members: the pair kernel's
    <CONS, NB = 4> instantiation against the extended run kernel."""
    n, nb = 256, 4
    idx = np.arange(n, dtype=np.uint64)
    S = 1.5 + 4.5 * ensemble.uniform01(idx, 0)
    names = ["b%d" % b for b in range(nb)]
    yrs = np.arange(1850, 2101)
    res = {}
    for limit, which in ((32768, "pair"), (0, "run")):
        c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
        c.set_pair_kernel_limit(limit)
        c.split_biome(names)
        c.setvar("S", S, "degC")
        for b, nm in enumerate(names):
            c.setvar(nm + ".q10_rh", 1.0 + 2.0 * ensemble.uniform01(idx, 10 + b)).setvar(nm + ".warmingfactor", np.full(n, 1.0 + 0.5 * b))
        c.set_outputs(["CO2_concentration", "global_tas", "timesteps", "soil_c", "ocean_c"])
        c.setvar_dated("CO2_constrain", yrs, 285.0 + 0.004 * (yrs - 1850.0) ** 2)
        c.run(2300)
        assert c.last_run_kernel() == which and (c.status() == 0).all()
        res[which] = {v: c.fetchvars(v, (1745, 2300)) for v in ("CO2_concentration", "global_tas", "timesteps", "soil_c", "ocean_c")}
        c.shutdown()
    assert np.array_equal(res["pair"]["timesteps"], res["run"]["timesteps"])
    co2 = res["pair"]["CO2_concentration"]
    assert np.abs(co2[1850 - 1745:2100 - 1745 + 1, :] - (285.0 + 0.004 * (yrs - 1850.0) ** 2)[:, None]).max() < 1e-9   # pinned
    for v in ("CO2_concentration", "global_tas", "soil_c", "ocean_c"):
        a, b = res["pair"][v], res["run"][v]
        assert np.abs(a - b).max() <= 5e-9 * max(1.0, np.abs(b).max()), v
