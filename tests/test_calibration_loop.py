"""The calibration loop -- new values of a few parameters for every member, reset, run, again and
again on ONE core (the reference's workflow: R/messages.R setvar + reset + run on a core kept
alive) -- takes three short cuts on the host (ensemble_core.cpp: upload_params, assign_lanes,
prepare): only the rows that moved are sent, the lanes are ordered by a radix sort, and a shared
spinup whose inputs did not change is not run again.  None of them may show in the results:
every case here is compared bit for bit with a fresh core that was given the final parameters
directly.  CPU: the host-emulation build of the product sources; GPU: the product library."""
import numpy as np
import pytest

import hector_amd


def _core(n, lib, **kw):
    if lib is None:
        return hector_amd.Core(n_members=n, device=0, **kw)
    return hector_amd.Core(n_members=n, lib_path=lib, allow_emulation=True, **kw)


def _results(core, run_to):
    core.run(run_to)
    co2 = core.fetchvars("CO2_concentration", (1745, run_to))
    tg = core.fetchvars("global_tas", (1745, run_to))
    return co2, tg, core.status().copy(), [core.spinup_steps(m) for m in (0, core.n_members - 1)]


def _params(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(1.5, 6.0, n), rng.uniform(1.0, 3.0, n), rng.uniform(0.2, 0.8, n),
            rng.uniform(1.0, 3.5, n))


def _loop_equals_fresh(lib, n, run_to):
    S1, q1, b1, d1 = _params(n, 1)
    S2, q2, b2, d2 = _params(n, 2)
    a = _core(n, lib)
    a.setvar("S", S1, "degC").setvar("q10_rh", q1, "(unitless)")
    _results(a, run_to)
    assert a.last_spinup_ms() > 0
    # iteration 2: other values of the same rows; the spinup sees neither
    a.setvar("S", S2, "degC").setvar("q10_rh", q2, "(unitless)")
    a.reset(1745)
    r2 = _results(a, run_to)
    assert a.last_spinup_ms() == 0.0, "the shared spinup was run again for parameters it does not see"
    f = _core(n, lib)
    f.setvar("S", S2, "degC").setvar("q10_rh", q2, "(unitless)")
    rf = _results(f, run_to)
    for x, y in zip(r2[:3], rf[:3]):
        assert np.array_equal(x, y)
    assert r2[3] == rf[3]
    # iteration 3: a third row joins (lane order now by three keys), one goes back to uniform
    a.setvar("beta", b2, "(unitless)").setvar("q10_rh", 2.0, "(unitless)").setvar("diff", d2, "cm2/s")
    a.reset(1745)
    r3 = _results(a, run_to)
    assert a.last_spinup_ms() == 0.0
    f.shutdown()
    f = _core(n, lib)
    f.setvar("S", S2, "degC").setvar("beta", b2, "(unitless)").setvar("diff", d2, "cm2/s")
    f.setvar("q10_rh", 2.0, "(unitless)")
    rf = _results(f, run_to)
    for x, y in zip(r3[:3], rf[:3]):
        assert np.array_equal(x, y)
    # iteration 4: a parameter the spinup DOES see -- uniform: the shared spinup runs again;
    # per member: every member spins up on its own
    a.setvar("npp_flux0", 52.0, "Pg C/yr")
    a.reset(1745)
    r4 = _results(a, run_to)
    assert a.last_spinup_ms() > 0
    f.shutdown()
    f = _core(n, lib)
    f.setvar("S", S2, "degC").setvar("beta", b2, "(unitless)").setvar("diff", d2, "cm2/s")
    f.setvar("npp_flux0", 52.0, "Pg C/yr").setvar("q10_rh", 2.0, "(unitless)")
    rf = _results(f, run_to)
    for x, y in zip(r4[:3], rf[:3]):
        assert np.array_equal(x, y)
    assert r4[3] == rf[3]
    npp = np.linspace(48.0, 58.0, n)
    a.setvar("npp_flux0", npp, "Pg C/yr")
    a.reset(1745)
    r5 = _results(a, run_to)
    assert a.last_spinup_ms() > 0
    a.setvar("S", S1, "degC")       # ... and a per-member spinup is never reused
    a.reset(1745)
    r6 = _results(a, run_to)
    assert a.last_spinup_ms() > 0
    f.shutdown()
    f = _core(n, lib)
    f.setvar("S", S1, "degC").setvar("beta", b2, "(unitless)").setvar("diff", d2, "cm2/s")
    f.setvar("npp_flux0", npp, "Pg C/yr").setvar("q10_rh", 2.0, "(unitless)")
    rf = _results(f, run_to)
    for x, y in zip(r6[:3], rf[:3]):
        assert np.array_equal(x, y)
    assert r6[3] == rf[3] and not np.array_equal(r5[0], r6[0])
    a.shutdown()
    f.shutdown()


def test_calibration_loop_equals_fresh_cores(emul_lib):
    _loop_equals_fresh(emul_lib, 200, 1800)


# every parameter the spinup is said not to see (kParams: spinup = false), with a range for it
NOT_SEEN_BY_SPINUP = {
    "S": (1.5, 6.0), "diff": (0.6, 3.0), "qco2": (3.4, 4.1), "aero_scalar": (0.3, 1.7),
    "vol_scalar": (0.5, 1.5), "lo_warming_ratio": (1.2, 1.9), "beta": (0.2, 0.9), "q10_rh": (1.1, 3.0),
    "warmingfactor": (0.8, 1.6), "rh_ch4_frac": (0.01, 0.04), "pf_mu": (1.4, 1.9),
    "pf_sigma": (0.8, 1.1), "fpf_static": (0.6, 0.85),
}


def test_the_spinup_really_does_not_see_those_parameters(emul_lib):
    """The reuse of a shared spinup rests on the table of which parameters the spinup reads.  If one
    marked `spinup = false` did enter it, a core that spun up under OTHER values of it would differ
    from a fresh one: all thirteen at once, per member, then one at a time (uniform values)."""
    n, run_to = 70, 1790
    rng = np.random.default_rng(11)
    new = {k: rng.uniform(lo, hi, n) for k, (lo, hi) in NOT_SEEN_BY_SPINUP.items()}
    a = _core(n, emul_lib)
    a.setvar("lo_warming_ratio", 1.5)         # (its first non-zero value adds an output array: new buffers)
    _results(a, run_to)                       # spun up under the scenario's defaults
    for k, v in new.items():
        a.setvar(k, v)
    a.reset(1745)
    ra = _results(a, run_to)
    assert a.last_spinup_ms() == 0.0
    f = _core(n, emul_lib)
    for k, v in new.items():
        f.setvar(k, v)
    rf = _results(f, run_to)
    for x, y in zip(ra[:3], rf[:3]):
        assert np.array_equal(x, y)
    assert ra[3] == rf[3]
    a.shutdown(); f.shutdown()
    for k, (lo, hi) in NOT_SEEN_BY_SPINUP.items():
        a = _core(4, emul_lib)
        a.setvar("lo_warming_ratio", 1.5)
        _results(a, 1760)
        a.setvar(k, hi)
        a.reset(1745)
        ra = _results(a, 1760)
        assert a.last_spinup_ms() == 0.0, k
        f = _core(4, emul_lib)
        f.setvar("lo_warming_ratio", 1.5).setvar(k, hi)
        rf = _results(f, 1760)
        assert np.array_equal(ra[0], rf[0]) and np.array_equal(ra[1], rf[1]) and ra[3] == rf[3], k
        a.shutdown(); f.shutdown()


def _reference_order(rows, n, wave=64):
    """The lane order as DESIGN.md 4 states it, with numpy's stable sorts: by the first varying
    parameter; then, inside sqrt(n / 64) bins of that order, by the second (more than two: the
    sum of the standardised others)."""
    order = np.argsort(rows[0], kind="stable")
    if len(rows) > 1:
        q = np.zeros(n)
        for r in rows[1:]:
            mean = np.sum(r) / n      # (plain left-to-right sums in the library; n is small here)
            sd = np.sqrt(np.sum((r - mean) ** 2) / n)
            if sd > 0:
                q += (r - mean) / sd
        nbins = max(1, int(round(np.sqrt(n / wave))))
        per = (n + nbins - 1) // nbins
        for b0 in range(0, n, per):
            seg = order[b0:b0 + per]
            order[b0:b0 + per] = seg[np.argsort(q[seg], kind="stable")]
    lane = np.empty(n, dtype=np.int64)
    lane[order] = np.arange(n)
    return lane


def test_radix_lane_order_is_the_stable_sort_order(emul_lib):
    n = 1000
    rng = np.random.default_rng(5)
    # ties, negative values, both zeros, denormals, huge: what a stable comparison sort orders
    S = rng.choice(np.array([1.5, 2.0, 2.0, 3.25, 4.0, 5.999]), n)
    lo = rng.choice(np.array([-0.0, 0.0, 1e-310, -1e-310, 0.5, -0.5, 1e150, -1e150, 1.0]), n)
    c = _core(n, emul_lib)
    c.setvar("S", S, "degC")
    assert np.array_equal(c.lane_of_member(), _reference_order([S], n))
    c.setvar("lo_warming_ratio", lo, "(unitless)")
    # row order of the table: S before lo_warming_ratio
    assert np.array_equal(c.lane_of_member(), _reference_order([S, lo], n))
    q = rng.uniform(1, 3, n)
    c.setvar("q10_rh", q, "(unitless)")
    lane3 = c.lane_of_member()
    assert sorted(lane3) == list(range(n))
    c.shutdown()


@pytest.mark.gpu
def test_calibration_loop_equals_fresh_cores_on_gpu(hip_lib):
    _loop_equals_fresh(None, 4096 + 37, 1900)
