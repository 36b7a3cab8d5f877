#!/usr/bin/env python3
"""Soak / fuzz harnesses: larger versions of the randomised parity tests in tests/ (same code:
tests/test_random_sweep.py, tests/test_workflow_fuzz.py, tests/test_diagnostics.py), for a GPU box.

    python tools/soak/soak.py <harness> [--gpu] [--pair]      harness: one of HARNESSES, or "all"
    gpurun -- 'python tools/soak/soak.py all --gpu'

Without --gpu they run the host build of the kernel source (tests/emul/) at smaller sizes.
--pair (all_parameters): the two-wavefront kernel's configuration.  README.md says what each
harness draws and what it compares with the oracle."""
import os
import sys
import tempfile
import time

import numpy as np

R = os.environ.get("GRAFT_REPO_ROOT") or os.path.normpath(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
HIP = R + "/hector_amd/lib/libhector_amd.so"
EMUL = R + "/tests/emul/libhector_amd_emul.so"
ALL_SCENARIOS = ["picontrol", "ssp119", "ssp126", "ssp245", "ssp370", "ssp434", "ssp460", "ssp534-over", "ssp585"]


def _fmt(w):
    return {k: "%.2e" % v for k, v in w.items()}


def all_parameters(gpu, argv):
    """19 parameters at once, 9 scenarios, 2 048 members each."""
    import test_random_sweep as T
    T.SCENARIOS = list(ALL_SCENARIOS)
    t0 = time.time()
    for seed in (101, 202):
        if gpu:
            w = T.sweep(HIP, 2048, seed=seed, check_every=16, device=0, pair="--pair" in argv)
        else:
            w = T.sweep(EMUL, 16, seed=seed, allow_emulation=True)
        print("seed", seed, _fmt(w), "%.0fs" % (time.time() - t0), flush=True)


def biomes(gpu, argv):
    """2-4 biomes, per-member pool splits, every per-biome parameter."""
    import test_random_sweep as T
    t0 = time.time()
    for seed in (313, 414):
        if gpu:
            w = T.sweep_biomes(HIP, 1024, seed=seed, scenarios=tuple(ALL_SCENARIOS), check_every=8, device=0)
        else:
            w = T.sweep_biomes(EMUL, 8, seed=seed, scenarios=tuple(ALL_SCENARIOS), allow_emulation=True)
        print("seed", seed, _fmt(w), "%.0fs" % (time.time() - t0), flush=True)


def biomes_many(gpu, argv):
    """5-9 biomes (the unrolled five- to eight-biome kernels, the looped kernel for nine), then
    10-16 (the looped kernel's chunks of four with every tail length), per-member pool splits,
    every per-biome parameter."""
    import test_random_sweep as T
    for seed in range(3):
        if gpu:
            w = T.sweep_biomes(HIP, 512, seed=100 + seed, scenarios=("ssp245", "ssp585", "ssp119"),
                               check_every=8, counts=(5, 6, 7, 8, 9) if seed < 2 else (10, 11, 13, 14, 15, 16),
                               device=0)
        else:
            w = T.sweep_biomes(EMUL, 4, seed=100 + seed, scenarios=("ssp245",), counts=(5, 6, 7, 8, 9),
                               allow_emulation=True)
        print("biomes_many seed %d: %s" % (seed, _fmt(w)), flush=True)


def mixed(gpu, argv):
    """scenario x biomes x one random constraint window x land-ocean warming ratio."""
    import test_random_sweep as T
    t0 = time.time()
    tmp = tempfile.mkdtemp()
    for seed in ((511, 512, 513) if gpu else (601,)):
        if gpu:
            w = T.sweep_mixed(HIP, 512, seed=seed, rounds=40, tmpdir=tmp, check_every=16, device=0)
        else:
            w = T.sweep_mixed(EMUL, 4, seed=seed, rounds=60, tmpdir=tmp, allow_emulation=True)
        print("seed", seed, _fmt(w), "%.0fs" % (time.time() - t0), flush=True)


def workflows(gpu, argv):
    """random run / reset(date) / emission edits / parameter changes: final trajectory = fresh run."""
    import test_workflow_fuzz as W
    tmp = tempfile.mkdtemp()
    t0 = time.time()
    for seed in (711, 712):
        if gpu:
            w = W.workflow_fuzz(HIP, seed, 40, tmp, n=70, device=0)
        else:
            w = W.workflow_fuzz(EMUL, seed, 60, tmp, n=2, allow_emulation=True)
        print("seed", seed, "worst %.2e" % w, "%.0fs" % (time.time() - t0), flush=True)


def diagnostics(gpu, argv):
    """all ~60 output-stream variables under full perturbation."""
    import hector_amd
    import oracle_binding
    from test_random_sweep import RANGES, ORACLE_FIELD
    from test_diagnostics import KERNEL_VARS, DERIVED_VARS, HOST_VARS, ORACLE_NAME
    kw = dict(device=0) if gpu else dict(lib_path=EMUL, allow_emulation=True)
    n = 256 if gpu else 6
    step = 8 if gpu else 1
    worst = {}
    for seed, name in [(1, "ssp245"), (2, "ssp585"), (3, "ssp119"), (4, "ssp534-over"), (5, "picontrol")]:
        rng = np.random.default_rng(seed)
        path = os.path.join(R, "hector_amd", "data", name + ".hxs")
        vals = {k: rng.uniform(lo, hi, n) for k, (lo, hi, _) in RANGES.items()}
        lo_ratio = np.where(rng.uniform(size=n) < 0.3, rng.uniform(1.1, 1.8, n), 0.0)
        c = hector_amd.Core(path, n, **kw)
        for k, (lo, hi, unit) in RANGES.items():
            c.setvar(k, vals[k], unit)
        c.setvar("lo_warming_ratio", lo_ratio)
        allv = KERNEL_VARS + DERIVED_VARS
        c.set_outputs(allv + ["RF_tot", "RF_CO2", "global_tas", "CO2_concentration", "land_tas", "sst"])
        c.run(2300)
        st = c.status()
        got = {v: c.fetchvars(v, (1746, 2300)) for v in allv + HOST_VARS + ["land_tas", "sst", "CO2_concentration"]}
        o = oracle_binding.Oracle(path)
        for i in range(0, n, step):
            p = o.default_params(); p.lo_warming_ratio = lo_ratio[i]
            for k in RANGES:
                if k in ORACLE_FIELD:
                    setattr(p, ORACLE_FIELD[k], vals[k][i])
                else:
                    getattr(p, k)[0] = vals[k][i]
            r, err, _ = o.run(p)
            assert (err != 0) == (st[i] != 0)
            if err:
                continue
            for v in got:
                ref = r[ORACLE_NAME.get(v, v)][1:]
                d = np.abs(got[v][:, i] - ref).max() / max(1.0, np.abs(ref).max())
                if d > worst.get(v, (0,))[0]:
                    worst[v] = (d, name, i)
        print(name, "done", flush=True)
    for v, w in sorted(worst.items(), key=lambda kv: -kv[1][0])[:12]:
        print(v, "%.2e" % w[0], w[1:])


def shared_parameters(gpu, argv):
    """the member-independent components' parameters (M0, N0, lifetimes, delta and rho of every agent)."""
    import hector_amd
    import oracle_binding
    from conftest import edited_pack
    kw = dict(device=0) if gpu else dict(lib_path=EMUL, allow_emulation=True)
    SH = {"M0": "CH4", "Tsoil": "CH4", "Tstrat": "CH4", "N0": "N2O", "PO3": "ozone", "TOH0": "OH",
          "delta_co2": "forcing", "delta_ch4": "forcing", "delta_n2o": "forcing", "rho_bc": "forcing",
          "rho_oc": "forcing", "rho_so2": "forcing", "rho_nh3": "forcing"}
    rng = np.random.default_rng(3); tmp = tempfile.mkdtemp(); worst = 0
    outs = ["CO2_concentration", "global_tas", "RF_tot", "CH4_concentration", "RF_CH4", "RF_N2O"]
    for rd in range(10):
        name = ["ssp119", "ssp245", "ssp585", "ssp370"][rd % 4]
        path = os.path.join(R, "hector_amd", "data", name + ".hxs")
        c = hector_amd.Core(path, 2, **kw); c.setvar("S", [2.5, 4.0], "degC")
        sc = {}
        for k, sec in SH.items():
            v0 = c.getvar(k)[0]; v = v0 * rng.uniform(0.7, 1.3) if v0 != 0 else rng.uniform(-0.2, 0.2)
            c.setvar(k, [v]); sc[(sec, k)] = v
        for h in c.halocarbons()[:6]:
            for pre in ("rho_", "delta_"):
                v0 = c.getvar(pre + h)[0]; v = v0 * rng.uniform(0.7, 1.3) if v0 != 0 else rng.uniform(-0.2, 0.2)
                c.setvar(pre + h, [v]); sc[(h + "_halocarbon", pre + h)] = v
        c.set_outputs(outs); c.run(2300)
        assert (c.status() == 0).all()
        o = oracle_binding.Oracle(edited_pack(os.path.join(tmp, "s%d.hxs" % rd), None, None, [], [], base=path, scalars=sc))
        for i, S in enumerate((2.5, 4.0)):
            p = o.default_params(); p.S = S
            r, err, _ = o.run(p); assert err == 0
            for v in outs:
                d = np.abs(c.fetchvars(v, (1745, 2300))[:, i] - r[v]).max() / max(1.0, np.abs(r[v]).max())
                worst = max(worst, d)
                assert d < 2e-8, (name, v, d)
    print("shared-parameter fuzz ok, worst %.2e" % worst)


def tracking(gpu, argv):
    """carbon tracking: random tracking date / biomes / parameters, run in pieces with a reset."""
    import hector_amd
    import oracle_binding
    kw = dict(device=0) if gpu else dict(lib_path=EMUL, allow_emulation=True)
    seeds = [int(a) for a in argv[1:] if a.isdigit()]
    rng = np.random.default_rng(seeds[0] if seeds else 9)
    worst = [0.0, 0.0]
    for rd in range(int(os.environ.get("ROUNDS", "12"))):
        name = ["ssp119", "ssp245", "ssp585"][rng.integers(3)]
        path = os.path.join(R, "hector_amd", "data", name + ".hxs")
        B = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 8, 12])); n = 3   # (5+: looped kernels; 12: two mask words)
        T0 = int(rng.integers(1750, 2050)); END = int(rng.integers(T0 + 5, 2301))
        c = hector_amd.Core(path, n, **kw); c.enable_history(True)
        if B > 1:
            c.split_biome(["b%d" % b for b in range(B)])
        c.setvar("trackingDate", [T0])
        S = rng.uniform(2, 5, n); q10 = rng.uniform(1.2, 2.8, (B, n)); wf = rng.uniform(0.8, 1.6, (B, n))
        no_pf = B <= 4 and rng.uniform() < 0.3   # no permafrost: pools of zero, maps shared equally
        if no_pf:
            for b in range(B):
                c.setvar(("b%d." % b if B > 1 else "") + "permafrost_c", [0.0])
        c.setvar("S", S, "degC")
        for b in range(B):
            pre = "b%d." % b if B > 1 else ""
            c.setvar(pre + "q10_rh", q10[b]).setvar(pre + "warmingfactor", wf[b])
        y1 = int(rng.integers(1746, END + 1)); c.run(y1)   # in pieces, with a reset in between
        if y1 > 1750 and rng.uniform() < 0.7:
            c.reset(int(rng.integers(1745, y1)))
        c.run(END)
        assert (c.status() == 0).all()
        o = oracle_binding.Oracle(path)
        for i in range(n):
            p = o.default_params()
            if B > 1:
                p = o.split_equal(p, B)
            p.S = S[i]
            for b in range(B):
                p.q10_rh[b] = q10[b][i]; p.warmingfactor[b] = wf[b][i]
                if no_pf:
                    p.permafrost_c[b] = 0.0
            ov, of, _, err = o.run_tracking(p, T0, END); assert err == 0
            gv, gf = c.tracking_data(i, (T0, END))
            k0, k1 = T0 - 1745, END - 1745 + 1
            dv = np.abs(gv - ov[k0:k1]).max() / np.abs(ov).max(); df = np.abs(gf - of[k0:k1]).max()
            worst = [max(worst[0], dv), max(worst[1], df)]
            assert dv < 1e-10 and df < 1e-7, (name, B, T0, END, i, dv, df)
            assert np.abs(gf.sum(2) - 1).max() < 1e-12
    print("tracking fuzz ok: worst value dev %.2e fraction dev %.2e" % tuple(worst))


def million(gpu, argv):
    """BASELINE configs[3]: 1 048 576 members on one GPU, EVERY member against the oracle (the
    driver-run test checks 4 096 of them), in slabs of 65 536; ~15 min of oracle time on 16 cores.
    Writes gpurun_out/parity_every_member_1048576.json."""
    import json
    import hector_amd
    import oracle_binding
    from hector_amd import ensemble
    sys.path.insert(0, R + "/tests")
    from test_gpu_fullsize import _oracle_all, _cores
    assert gpu, "needs the GPU"
    n = 1 << 20
    scen = os.path.join(R, "hector_amd", "data", "ssp245.hxs")
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(scen, n, device=0)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
    c.run(2300)
    assert (c.status() == 0).all()
    o = oracle_binding.Oracle(scen)
    rep = {"config": "configs[3] 1048576x1, every member", "members": n, "years_per_member": 555,
           "max_rel_dCO2": 0.0, "max_abs_dTgav_K": 0.0, "members_over_1e-9_rel_CO2": 0,
           "members_with_a_different_stash_schedule": 0, "oracle_threads": _cores(), "kernel_ms": c.last_run_ms()}
    t0 = time.time()
    slab = 32768
    for m0 in range(0, n, slab):
        def mp(k):
            p = o.default_params(); p.S = S[m0 + k]; p.q10_rh[0] = q10[m0 + k]
            return p
        oco2, otg, ots, oerr = _oracle_all(o, mp, slab)
        assert (oerr == 0).all()
        co2 = np.empty((slab, 556)); tg = np.empty((slab, 556)); ts = np.empty((slab, 556))
        for y0 in range(1745, 2301, 80):
            y1 = min(2300, y0 + 79)
            co2[:, y0 - 1745:y1 - 1744] = c.fetchvars("CO2_concentration", (y0, y1))[:, m0:m0 + slab].T
            tg[:, y0 - 1745:y1 - 1744] = c.fetchvars("global_tas", (y0, y1))[:, m0:m0 + slab].T
            ts[:, y0 - 1745:y1 - 1744] = c.fetchvars("timesteps", (y0, y1))[:, m0:m0 + slab].T
        rel = np.abs(co2 - oco2) / oco2
        rep["max_rel_dCO2"] = max(rep["max_rel_dCO2"], float(rel.max()))
        rep["max_abs_dTgav_K"] = max(rep["max_abs_dTgav_K"], float(np.abs(tg - otg).max()))
        rep["members_over_1e-9_rel_CO2"] += int((rel.max(1) > 1e-9).sum())
        flip = (ts.astype(np.int64) != ots.astype(np.int64)).any(axis=1)
        rep["members_with_a_different_stash_schedule"] += int(flip.sum())
        for k in np.nonzero((rel.max(1) > 1e-9) | flip)[0]:   # who they are: (member, S, q10, max rel dCO2, first year the schedules differ)
            d = np.nonzero(ts[k].astype(np.int64) != ots[k].astype(np.int64))[0]
            rep.setdefault("members_to_look_at", []).append(
                [int(m0 + k), float(S[m0 + k]), float(q10[m0 + k]), float(rel[k].max()), int(1745 + d[0]) if d.size else None])
        rep["members_checked"] = m0 + slab
        rep["oracle_seconds"] = round(time.time() - t0, 1)
        print(json.dumps(rep), flush=True)
        os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
        json.dump(rep, open(os.path.join(R, "gpurun_out", "parity_every_member_1048576.json"), "w"), indent=1)
    # A member beyond the tolerance has to be one whose trajectory the ORACLE ITSELF does not pin:
    # rounding-sized noise (every pool times 1 +- 1e-13 once a year, tests/test_random_sweep.py)
    # must move the oracle's own answer as far -- a controller decision sitting on a tie.
    from test_random_sweep import self_sensitivity
    rep["ill_conditioned_members"] = []
    for (i, Si, qi, dev, flip_year) in rep.get("members_to_look_at", []):
        if dev < 2e-8 and flip_year is None:
            continue
        p = o.default_params(); p.S = Si; p.q10_rh[0] = qi
        sens = self_sensitivity(o, p, ["CO2_concentration"])["CO2_concentration"]
        rep["ill_conditioned_members"].append({"member": i, "S": Si, "q10_rh": qi, "rel_dCO2": dev,
                                                "first_year_of_a_different_schedule": flip_year,
                                                "oracle_moves_under_1e-13_noise_by": sens})
        assert dev < 50.0 * sens, (i, dev, sens)
    json.dump(rep, open(os.path.join(R, "gpurun_out", "parity_every_member_1048576.json"), "w"), indent=1)
    print("ill-conditioned members:", rep["ill_conditioned_members"])


def every_member(gpu, argv):
    """EVERY member of a perturbed-parameter ensemble against the oracle on scenarios other than
    SSP2-4.5: `ssp585` / `ssp119` (any shipped scenario name) -- 65 536 ECS x
    Q10 members, the scenarios where the retry controller works hardest and least -- and `multi`
    -- 16 384 SSP2-4.5 members with S, q10_rh, beta, diff, aero_scalar and npp_flux0 perturbed
    together.  Each writes gpurun_out/parity_every_member_<name>.json: max rel CO2, max |dTgav|,
    the count of different stash schedules, and every member outside 2e-8 by index with what the
    ORACLE's own answer moves by under 1e-13 rounding noise (hxo_set_rounding_noise).
        python tools/soak/soak.py every_member ssp585 ssp119 multi --gpu"""
    import json
    import hector_amd
    import oracle_binding
    from hector_amd import ensemble
    from test_gpu_fullsize import _oracle_all, _cores
    from test_random_sweep import NOISE
    assert gpu, "needs the GPU"
    names = [a for a in argv[2:] if not a.startswith("--")] or ["ssp585", "ssp119", "multi"]

    def sensitivity(o, p):
        """Largest move of the oracle's own CO2 trajectory under rounding-sized noise (every pool
        times 1 +- amp x 1e-13 once a year).  More amplitudes than tests/test_random_sweep.py's five:
        a tie of Brent's minimiser (the alkalinity tuner returns its LAST evaluated point) shows
        under some amplitudes only -- members 1286 / 9225 / 10876 of the `multi` ensemble move by
        4.6e-6 / 5.6e-5 / 2.5e-5 under amplitudes 0.3, -7 and 5 and by 1e-11 under the usual five."""
        base, _, _ = o.run(p)
        worst = 0.0
        try:
            for amp in (1.0, -1.0, 1.7, -2.3, 3.1, 0.3, -0.5, 5.0, -7.0, 11.0, 0.1, -0.13, 23.0):
                o.set_rounding_noise(NOISE * amp)
                pert, _, _ = o.run(p)
                worst = max(worst, float((np.abs(pert["CO2_concentration"] - base["CO2_concentration"]) /
                                          base["CO2_concentration"]).max()))
        finally:
            o.set_rounding_noise(0.0)
        return worst
    for name in names:
        multi = name == "multi"
        n = 16384 if multi else 65536
        scen_name = name
        if "@" in name:     # "<scenario>@<members>": e.g. ssp585@131072, the two-wave flavour on another scenario
            scen_name, n = name.split("@")[0], int(name.split("@")[1])
        scen = os.path.join(R, "hector_amd", "data", ("ssp245" if multi else scen_name) + ".hxs")
        idx = np.arange(n, dtype=np.uint64)
        S, q10 = ensemble.ecs_q10(n)
        c = hector_amd.Core(scen, n, device=0)
        c.set_pair_kernel_limit(0)
        c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
        extra = {}
        if multi:
            extra = {"beta": 0.25 + 0.5 * ensemble.uniform01(idx, 21), "diff": 1.2 + 2.2 * ensemble.uniform01(idx, 22),
                     "aero_scalar": 0.5 + 1.0 * ensemble.uniform01(idx, 23),
                     "npp_flux0": 45.0 + 20.0 * ensemble.uniform01(idx, 24)}
            c.setvar("beta", extra["beta"]).setvar("diff", extra["diff"], "cm2/s")
            c.setvar("aero_scalar", extra["aero_scalar"]).setvar("npp_flux0", extra["npp_flux0"], "Pg C/yr")
        c.set_outputs(["CO2_concentration", "global_tas", "timesteps"])
        c.run(2300)
        bad = int((c.status() != 0).sum())
        o = oracle_binding.Oracle(scen)

        def mp(k):
            p = o.default_params(); p.S = S[k]; p.q10_rh[0] = q10[k]
            if multi:
                p.beta[0] = extra["beta"][k]; p.diff = extra["diff"][k]
                p.aero_scalar = extra["aero_scalar"][k]; p.npp_flux0[0] = extra["npp_flux0"][k]
            return p
        t0 = time.time()
        oco2, otg, ots, oerr = _oracle_all(o, mp, n)
        ny = oco2.shape[1]
        y_end = 1745 + ny - 1
        co2 = np.empty((n, ny)); tg = np.empty((n, ny)); ts = np.empty((n, ny))
        for y0 in range(1745, y_end + 1, 80):
            y1 = min(y_end, y0 + 79)
            co2[:, y0 - 1745:y1 - 1744] = c.fetchvars("CO2_concentration", (y0, y1)).T
            tg[:, y0 - 1745:y1 - 1744] = c.fetchvars("global_tas", (y0, y1)).T
            ts[:, y0 - 1745:y1 - 1744] = c.fetchvars("timesteps", (y0, y1)).T
        rel = np.abs(co2 - oco2) / oco2
        flip = (ts.astype(np.int64) != ots.astype(np.int64)).any(axis=1)
        rep = {"config": ("%d ECS x Q10 members, %s" % (n, name)) if not multi else
               "16384 members, ssp245: S, q10_rh, beta, diff, aero_scalar, npp_flux0 perturbed together",
               "members": n, "members_checked": n, "years_per_member": ny - 1, "kernel": c.last_run_kernel(),
               "kernel_ms": c.last_run_ms(), "members_with_model_errors_gpu": bad,
               "members_with_model_errors_oracle": int((oerr != 0).sum()),
               "max_rel_dCO2": float(rel.max()), "median_of_member_max_rel_dCO2": float(np.median(rel.max(1))),
               "max_abs_dTgav_K": float(np.abs(tg - otg).max()),
               "members_over_1e-9_rel_CO2": int((rel.max(1) > 1e-9).sum()),
               "members_over_2e-8_rel_CO2": int((rel.max(1) > 2e-8).sum()),
               "members_with_a_different_stash_schedule": int(flip.sum()),
               "north_star_members_over_1e-6": int((rel.max(1) > 1e-6).sum()),
               "stashes_per_member_year_mean": float(ots[:, 1:].mean()),
               "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": _cores(), "ill_conditioned_members": []}
        # a member beyond 2e-8 or with another schedule must be one the ORACLE ITSELF does not pin
        unexpected = 0
        for k in np.nonzero((rel.max(1) > 2e-8) | flip)[0][:128]:
            d = np.nonzero(ts[k].astype(np.int64) != ots[k].astype(np.int64))[0]
            sens = sensitivity(o, mp(int(k)))
            dev = float(rel[k].max())
            tie = dev < 50.0 * sens
            unexpected += 0 if tie else 1
            rep["ill_conditioned_members"].append(
                {"member": int(k), "S": float(S[k]), "q10_rh": float(q10[k]), "rel_dCO2": dev,
                 "first_year_of_a_different_schedule": int(1745 + d[0]) if d.size else None,
                 "oracle_moves_under_1e-13_noise_by": sens, "a_tie_the_oracle_does_not_pin": bool(tie)})
        rep["unexpected_members"] = unexpected
        os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
        json.dump(rep, open(os.path.join(R, "gpurun_out", "parity_every_member_%s.json" % name.replace("@", "_")), "w"), indent=1)
        print(json.dumps(rep), flush=True)
        c.shutdown()
        assert unexpected == 0 and bad == int((oerr != 0).sum()), (name, unexpected, bad)


HARNESSES = {"all_parameters": all_parameters, "biomes": biomes, "biomes_many": biomes_many, "mixed": mixed, "workflows": workflows,
             "diagnostics": diagnostics, "shared_parameters": shared_parameters, "tracking": tracking,
             "million": million, "every_member": every_member}

if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
    if not which or (which[0] != "all" and which[0] not in HARNESSES):
        sys.exit(__doc__ + "\nHARNESSES: " + " ".join(HARNESSES))
    for name in ([h for h in HARNESSES if h not in ("million", "every_member")] if which[0] == "all" else [which[0]]):
        print("==", name, flush=True)
        HARNESSES[name]("--gpu" in sys.argv, sys.argv)
