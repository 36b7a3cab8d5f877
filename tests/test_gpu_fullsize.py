"""Every-member parity at BASELINE.json's full sizes.

The HIP path against the CPU oracle on EVERY member of configs[2] (65 536-member ECS/Q10
ensemble) and of configs[4] (65 536 members, 4 biomes, per-biome Q10 and warming factor), plus
4 096 members spread over the 1 048 576-member grid of configs[3].  Criterion as in the
reference's own old-new test (tests/testthat/test_old-new.R:11 compares trajectories year by
year), tolerance 2e-8 (north star: 1e-6).  The number of members whose per-year stash schedule
("timesteps": the reference's retry / reduced-timestep decisions) differs from the
oracle's is counted and must be zero: SURVEY.md 7 expected "a tiny flip rate" at this size.

The oracle runs on all host cores (ctypes releases the GIL); figures go to
gpurun_out/parity_fullsize_*.json (copied to profiles/ by the builder)."""
import json
import os
import threading
import time

import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import ROOT, SCENARIO

pytestmark = pytest.mark.gpu

REL_CO2 = 2e-8
ABS_T = 2e-8


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def _oracle_all(oracle, make_params, n, chunk=64):
    """Oracle trajectories of members [0, n): co2[n, ns], tg[n, ns], timesteps[n, ns], errs[n]."""
    ns = oracle.ns
    co2 = np.empty((n, ns)); tg = np.empty((n, ns)); ts = np.empty((n, ns), dtype=np.uint8)
    errs = np.zeros(n, dtype=np.int32)
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                i0 = nxt[0]
                nxt[0] += chunk
            if i0 >= n:
                return
            i1 = min(n, i0 + chunk)
            c, t, s, e = oracle.run_ensemble([make_params(i) for i in range(i0, i1)])
            co2[i0:i1] = c; tg[i0:i1] = t; ts[i0:i1] = s; errs[i0:i1] = e
    th = [threading.Thread(target=work) for _ in range(_cores())]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return co2, tg, ts, errs


def _fetch_members(core, var, members, years_per_piece=40):
    """[len(members), 556] of one variable, fetched in pieces of a few dozen years so that the
    host never holds a whole [556][n] array of a million-member run."""
    out = np.empty((len(members), 2300 - 1745 + 1))
    for y0 in range(1745, 2301, years_per_piece):
        y1 = min(2300, y0 + years_per_piece - 1)
        out[:, y0 - 1745:y1 - 1745 + 1] = core.fetchvars(var, (y0, y1))[:, members].T
    return out


def _compare(tag, core, oracle, make_params, members, total, extra, ill_conditioned_members=()):
    """members: indices (into the core's ensemble) to check against the oracle.
    ill_conditioned_members: the members -- named by index, not by count -- that may miss the
    tolerance IF the oracle's own answer for them moves as far under rounding-sized noise
    (hxo_set_rounding_noise: a controller decision on a tie, which no implementation pins --
    SURVEY.md 7 expected "a tiny flip rate on 1 M members"); each is listed in the report.  Any
    other member outside the tolerance fails the test."""
    t0 = time.time()
    oco2, otg, ots, oerr = _oracle_all(oracle, lambda k: make_params(int(members[k])), len(members))
    t_or = time.time() - t0
    assert (oerr == 0).all()
    co2, tg, ts = (_fetch_members(core, v, members) for v in
                   ("CO2_concentration", "global_tas", "timesteps"))
    rel = np.abs(co2 - oco2) / oco2
    dt = np.abs(tg - otg)
    flips = (ts.astype(np.int64) != ots.astype(np.int64)).any(axis=1)
    rep = {
        "config": tag, "members_in_run": int(total), "members_checked": int(len(members)),
        "years_per_member": 555,
        "max_rel_dCO2": float(rel.max()), "median_of_member_max_rel_dCO2": float(np.median(rel.max(1))),
        "max_abs_dTgav_K": float(dt.max()), "median_of_member_max_abs_dTgav_K": float(np.median(dt.max(1))),
        "members_over_1e-9_rel_CO2": int((rel.max(1) > 1e-9).sum()),
        "members_with_a_different_stash_schedule": int(flips.sum()),
        "tolerance": {"rel_CO2": REL_CO2, "abs_Tgav_K": ABS_T},
        "oracle_seconds": round(t_or, 1), "oracle_threads": _cores(),
        "kernel_ms": core.last_run_ms(),
    }
    rep.update(extra)
    bad = np.nonzero((rel.max(1) >= REL_CO2) | (dt.max(1) >= ABS_T) | flips)[0]
    rep["members_outside_the_tolerance"] = int(bad.size)
    rep["north_star_members_over_1e-6"] = int((rel.max(1) > 1e-6).sum())
    known = set(int(x) for x in ill_conditioned_members)
    unexpected = [int(members[k]) for k in bad if int(members[k]) not in known]
    rep["members_outside_the_tolerance_not_on_the_list_of_known_ties"] = unexpected
    if bad.size and not unexpected:
        from test_random_sweep import self_sensitivity
        rep["ill_conditioned_members"] = []
        for k in bad:
            p = make_params(int(members[k]))
            sens = self_sensitivity(oracle, p, ["CO2_concentration"])["CO2_concentration"]
            d = np.nonzero(ts[k].astype(np.int64) != ots[k].astype(np.int64))[0]
            rep["ill_conditioned_members"].append({
                "member": int(members[k]), "S": float(p.S), "q10_rh": float(p.q10_rh[0]),
                "rel_dCO2": float(rel[k].max()),
                "first_year_of_a_different_schedule": int(1745 + d[0]) if d.size else None,
                "oracle_moves_under_1e-13_noise_by": float(sens)})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_fullsize_%s.json" % tag), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    assert not unexpected, rep
    for m in rep.get("ill_conditioned_members", []):
        # the oracle does not pin this member either: its own answer moves as far
        assert m["rel_dCO2"] < 50.0 * m["oracle_moves_under_1e-13_noise_by"], m
    if not bad.size:
        assert rel.max() < REL_CO2 and dt.max() < ABS_T and flips.sum() == 0, rep


def test_config3_every_member_vs_oracle(hip_lib, oracle):
    n = 65536
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    assert c.backend == "hip"
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert (c.status() == 0).all()

    def mp(i):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        return p
    _compare("config3_65536x1", c, oracle, mp, np.arange(n), n,
             {"ensemble": "S ~ U(1.5, 6), q10_rh ~ U(1, 3), seed 20260928 (hector_amd/ensemble.py)"})


def _biome4_params(oracle, S, q10s, wfs):
    def mp(i):
        p = oracle.default_params()
        oracle.split_equal(p, 4)
        p.S = S[i]
        for b in range(4):
            p.q10_rh[b] = q10s[b][i]; p.warmingfactor[b] = wfs[b][i]
        return p
    return mp


def test_config5_every_member_vs_oracle(hip_lib, oracle):
    """configs[4]: 65 536 members x 4 biomes run at size, every member checked against the oracle
    (~1 min of oracle time on the box's 16 cores); status clean and replicas bit-identical."""
    n = 65536
    S, q10s, wfs = ensemble.biome4(n)
    S[50000:50064] = S[:64]
    for b in range(4):
        q10s[b][50000:50064] = q10s[b][:64]
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    names = ["b1", "b2", "b3", "b4"]
    c.split_biome(names)
    c.setvar("S", S, "degC")
    for b, nm in enumerate(names):
        c.setvar(nm + ".q10_rh", q10s[b]).setvar(nm + ".warmingfactor", wfs[b])
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert (c.status() == 0).all()
    co2 = c.fetchvars("CO2_concentration", (2200, 2300))
    assert np.isfinite(co2).all()
    assert np.array_equal(co2[:, :64], co2[:, 50000:50064])  # replicas in other wavefronts
    del co2
    members = np.arange(n)
    _compare("config5_65536x4", c, oracle, _biome4_params(oracle, S, q10s, wfs), members, n,
             {"ensemble": "4 equal biomes, S ~ U(1.5, 6), q10_rh ~ U(1, 3) per biome, "
                          "warmingfactor 1, 1.5, 2, 2.5 (hector_amd/ensemble.py biome4)"})


@pytest.mark.parametrize("nb", [6, 8, 12])
def test_unrolled_many_biome_kernels_every_member_vs_oracle(hip_lib, oracle, nb):
    """The unrolled kernels of five to eight biomes (round 3; six: the lean park, eight: the slim
    one with f_frozen / f_new_thaw in HBM, its tables read as row address + lane offset) and the
    looped kernel (twelve: four pools in the park, the thawed pool, tempferts, f_frozen in their
    state rows, the year's per-biome values in scratch rows): 16 384 members (twelve biomes: 8 192),
    every one against the oracle -- S, a Q10 per biome, warming factors 1 ... 2.5 -- with identical
    stash schedules."""
    n = 16384 if nb <= 8 else 8192
    idx = np.arange(n, dtype=np.uint64)
    S = 1.5 + 4.5 * ensemble.uniform01(idx, 0)
    q10s = [1.0 + 2.0 * ensemble.uniform01(idx, 10 + b) for b in range(nb)]
    wfs = [1.0 + 0.5 * (b % 4) for b in range(nb)]
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    names = ["b%d" % b for b in range(nb)]
    c.split_biome(names)
    c.setvar("S", S, "degC")
    for b, nm in enumerate(names):
        c.setvar(nm + ".q10_rh", q10s[b]).setvar(nm + ".warmingfactor", wfs[b])
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert (c.status() == 0).all()

    def mp(i):
        p = oracle.default_params()
        oracle.split_equal(p, nb)
        p.S = S[i]
        for b in range(nb):
            p.q10_rh[b] = q10s[b][i]; p.warmingfactor[b] = wfs[b]
        return p
    _compare(("unrolled_%dx%d" if nb <= 8 else "looped_%dx%d") % (n, nb), c, oracle, mp, np.arange(n), n,
             {"ensemble": "%d equal biomes, S ~ U(1.5, 6), q10_rh ~ U(1, 3) per biome, warmingfactor "
                          "1, 1.5, 2, 2.5 repeating (tools/prof/biome_times.py)" % nb})


def test_config4_million_member_grid_sample_vs_oracle(hip_lib, oracle):
    """configs[3]'s 1 048 576 members on one GPU (the 8-GPU job shards exactly this grid): 65 536
    members spread evenly over it (every 16th, among them member 394 646, the one member of the
    million that round 2's every-member pass found outside the tolerance) against the oracle.
    The statement: none of them misses 2e-8 except, possibly, member 394 646 (S = 5.385,
    Q10 = 1.807) -- whitelisted BY INDEX: its last-year stash decision is a tie of the ocean's
    timestep controller that the oracle itself does not pin (it flips under +-1e-13 noise, which
    the test then demonstrates)."""
    n = 1 << 20
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert (c.status() == 0).all()
    members = np.arange(65536) * 16 + 6
    assert 394646 in members

    def mp(i):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        return p
    _compare("config4_1048576x1_sample", c, oracle, mp, members, n,
             {"ensemble": "S ~ U(1.5, 6), q10_rh ~ U(1, 3), seed 20260928", "kernel": c.last_run_kernel()},
             ill_conditioned_members=(394646,))
