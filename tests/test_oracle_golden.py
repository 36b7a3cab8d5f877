"""The oracle (oracle/hector_oracle.c) against the reference's own golden
trajectory tests/testthat/compdata/hector_comp.csv (10 variables x 556 years,
15 significant digits; the reference's tolerance for it is 1e-10,
tests/testthat/test_old-new.R:11)."""
import numpy as np

# golden values carry 15 significant digits; allow a few units of that
TOL_REL = {"CO2_concentration": 5e-13, "atmos_co2": 5e-13, "ocean_c": 5e-14,
           "permafrost_c": 1e-12, "HL_pH": 1e-12}
TOL_ABS = {"global_tas": 5e-12, "sst": 5e-12, "RF_tot": 5e-12, "RF_CO2": 5e-12,
           "heatflux": 1e-11}


def test_oracle_matches_reference_golden_trajectory(oracle, golden):
    out, err, steps = oracle.run()
    assert err == 0
    for var, ref in golden.items():
        got = out[var]
        assert got.shape == ref.shape
        if var in TOL_REL:
            m = ref != 0
            rel = np.abs(got[m] - ref[m]) / np.abs(ref[m])
            assert rel.max() < TOL_REL[var], (var, rel.max())
            assert np.abs(got[~m]).max(initial=0) == 0
        else:
            assert np.abs(got - ref).max() < TOL_ABS[var], (var, np.abs(got - ref).max())


def test_reference_tolerance_1e10_all_variables(oracle, golden):
    """test_old-new.R's own criterion."""
    out, _, _ = oracle.run()
    for var, ref in golden.items():
        assert np.allclose(out[var], ref, rtol=1e-10, atol=1e-10), var


def test_spinup_and_stash_schedule_counters(oracle):
    """Work counters of the default member (measured on the
    reference): 498 spinup steps; 1056 stashes = 340x1 + 71x2 + 2x3 + 142x4;
    1891 accepted dopri5 steps, none rejected."""
    out, err, steps = oracle.run()
    assert steps == 498
    ts = out["timesteps"][1:].astype(int)
    assert ts.sum() == 1056
    assert np.bincount(ts).tolist() == [0, 340, 71, 2, 142]
    assert int(out["solver_steps"].sum()) == 1891
    # the controller first reduces max_timestep at the 1784 stash (App. C-2), so the
    # first multi-stash year is 1785
    assert 1746 + int(np.argmax(ts > 1)) == 1785
    assert out["max_timestep"][1784 - 1745] == 0.5 and out["max_timestep"][1783 - 1745] == 1.0


def test_partial_run_is_prefix(oracle):
    full, _, _ = oracle.run()
    part, _, _ = oracle.run(run_to=1900)
    k = 1900 - 1745 + 1
    for v in ("CO2_concentration", "global_tas", "ocean_c"):
        assert np.array_equal(full[v][:k], part[v][:k])
        assert np.all(part[v][k:] == 0)


def test_shared_spinup_ensemble_equals_per_member_runs(oracle):
    """hxo_run_ensemble_ecs_q10 spins up once (S, q10 do not enter the spinup);
    that must equal an independent spinup + run per member, bit for bit."""
    S = np.array([1.7, 3.0, 5.9]); q10 = np.array([1.1, 1.2, 2.8])
    co2, tg, err = oracle.run_ecs_q10(S, q10)
    assert err == 0
    for i in range(3):
        p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        o, e, _ = oracle.run(p)
        assert e == 0
        assert np.array_equal(co2[i], o["CO2_concentration"])
        assert np.array_equal(tg[i], o["global_tas"])
