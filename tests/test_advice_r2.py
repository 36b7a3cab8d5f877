"""Regression tests for the round-2 advisor findings (host runtime + kernel source through the
test-only emulation build).

1. (medium) setvar(dates) of an INTERPOLATING constraint (tas_constrain, RF_tot_constrain) on a core
   that already has per-member series of it: the years between the points follow, like the
   shared series (temperature_component.cpp:112, forcing_component.cpp:112: allowInterp(true)).
2. (low) sparse per-member points of those two constraints interpolate per member, RF_tot_constrain
   also holds flat before a member's first point (forcing_component.cpp:498).
4. (low) hx_sqrt(0) = 0, hx_exp(-inf) = 0, hx_log(+inf) = +inf like libm: a zero concentration
   (e.g. an N2O constraint of 0) keeps a member finite.
"""
import numpy as np

import hector_amd
from conftest import SCENARIO


def mk(emul_lib, n):
    return hector_amd.Core(SCENARIO, n, lib_path=emul_lib, allow_emulation=True)


def test_shared_points_after_member_series_interpolate(emul_lib):
    n = 2
    ref = mk(emul_lib, n)           # shared series only
    ref.setvar_dated("tas_constrain", [1960, 1990], [1.0, 2.0])
    ref.run(2000)
    c = mk(emul_lib, n)
    yrs = list(range(1800, 1811))   # dense per-member values somewhere else
    per = np.tile(np.linspace(0.1, 0.2, len(yrs))[:, None], (1, n))
    per[:, 1] += 0.05
    c.setvar_dated_members("tas_constrain", yrs, per)
    c.setvar_dated("tas_constrain", [1960, 1990], [1.0, 2.0])
    c.run(2000)
    got = c.fetchvars("tas_constrain", (1974, 1976))
    np.testing.assert_allclose(got[:, 0], [1.0 + 14 / 30, 1.5, 1.0 + 16 / 30], rtol=1e-15)
    tas = c.fetchvars("global_tas", (1960, 1990))
    np.testing.assert_allclose(tas, ref.fetchvars("global_tas", (1960, 1990)), atol=1e-12)
    np.testing.assert_allclose(tas[15], 1.5, atol=1e-12)
    # between the member points and the shared ones (1811..1959) the series interpolates as
    # well: one tseries per member, like the reference's
    mid = c.fetchvars("tas_constrain", (1885, 1885))[0]
    np.testing.assert_allclose(mid, per[-1] + (1.0 - per[-1]) * (1885 - 1810) / (1960 - 1810), rtol=1e-14)
    # removing a point (NaN) moves the interpolation again
    c.setvar_dated("tas_constrain", [1960], [np.nan])
    v = c.fetchvars("tas_constrain", (1900, 1900))[0]
    np.testing.assert_allclose(v, per[-1] + (2.0 - per[-1]) * (1900 - 1810) / (1990 - 1810), rtol=1e-14)
    ref.shutdown(); c.shutdown()


def test_sparse_member_points_interpolate_per_member(emul_lib):
    n = 3
    c = mk(emul_lib, n)
    pts = np.array([[0.5, 0.6, np.nan], [1.0, 1.4, np.nan]])   # member 2: unconstrained
    c.setvar_dated_members("tas_constrain", [1900, 1950], pts)
    c.run(1960)
    tas = c.fetchvars("global_tas", (1925, 1925))[0]
    np.testing.assert_allclose(tas[:2], [0.75, 1.0], atol=1e-12)
    free = mk(emul_lib, 1)
    free.run(1960)
    np.testing.assert_allclose(tas[2], free.fetchvars("global_tas", (1925, 1925))[0, 0], atol=1e-12)
    con = c.fetchvars("tas_constrain", (1899, 1951))
    assert np.isnan(con[0]).all() and np.isnan(con[-1]).all() and np.isnan(con[:, 2]).all()
    np.testing.assert_allclose(con[1:-1, 1], np.linspace(0.6, 1.4, 51), rtol=1e-14)
    # RF_tot_constrain: flat before a member's first point, nothing after its last
    f = mk(emul_lib, 2)
    f.set_outputs(["RF_tot", "global_tas"])
    f.setvar_dated_members("RF_tot_constrain", [1800, 1850], np.array([[0.2, 0.3], [0.7, 0.3]]))
    f.run(1860)
    con = f.fetchvars("RF_tot_constrain", (1746, 1860))
    np.testing.assert_array_equal(con[0], [0.2, 0.3])                       # 1746: back-filled
    np.testing.assert_allclose(con[1825 - 1746], [0.45, 0.3], rtol=1e-15)   # interpolated
    assert np.isnan(con[1851 - 1746:]).all()                                # free again
    # RF_tot is reported relative to the base year 1750 (forcing_component.cpp:512-530), whose
    # forcing is the back-filled constraint
    rf = f.fetchvars("RF_tot", (1746, 1860))
    np.testing.assert_allclose(rf[1825 - 1746], [0.25, 0.0], atol=1e-13)
    np.testing.assert_allclose(rf[1790 - 1746], [0.0, 0.0], atol=1e-13)
    assert abs(rf[1855 - 1746, 0] - 0.5) > 1e-3
    c.shutdown(); free.shutdown(); f.shutdown()


def test_zero_concentration_keeps_a_member_finite(emul_lib):
    c = mk(emul_lib, 1)
    c.set_outputs(["RF_tot", "global_tas", "CO2_concentration"])
    c.setvar_dated("N2O_constrain", [1800, 1801], [0.0, 0.0])   # sqrt(N2O) in the forcing
    c.run(1810)
    assert np.isfinite(c.fetchvars("RF_tot", (1746, 1810))).all()
    assert np.isfinite(c.fetchvars("global_tas", (1746, 1810))).all()
    assert (c.status() == 0).all()
    c.shutdown()
