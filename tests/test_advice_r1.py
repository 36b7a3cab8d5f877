"""Regression tests for the round-1 advisor findings (host runtime + kernel source through the
test-only emulation build; the GPU variants of the two workflow cases are in test_gpu_parity.py).

1. reset(date) through the state history restores the members' status bits of that date
   (the reference's reset()+run() recovers after the inputs are fixed, core.cpp:511-549).
2. setvar(dates) on a variable that already has per-member series reaches every member.
3. a scenario pack whose series do not cover startDate..endDate is rejected.
4. a shared spinup neither loses nor spreads the per-member DOECLIM-singular flag.
5. "<biome>.<variable>" outputs stay with their biome when biomes are deleted / split.
"""
import numpy as np
import pytest

import hector_amd
from conftest import SCENARIO


def mk(emul_lib, n, scenario=SCENARIO):
    return hector_amd.Core(scenario, n, lib_path=emul_lib, allow_emulation=True)


def test_reset_through_history_restores_status(emul_lib):
    n = 3
    c = mk(emul_lib, n)
    c.enable_history()
    years = list(range(1900, 1911))
    # member 1 gets absurd emissions for a decade: it leaves the model's domain and is flagged
    ffi = np.tile(c_fetch_ffi(c, years)[:, None], (1, n))
    bad = ffi.copy()
    bad[:, 1] = 5e4
    c.setvar_dated_members("ffi_emissions", years, bad, "Pg C/yr")
    c.run(1950)
    st = c.status()
    assert st[1] != 0 and st[0] == 0 and st[2] == 0
    # fix the inputs: the core resets itself to 1899 through the history and reruns
    c.setvar_dated_members("ffi_emissions", years, ffi, "Pg C/yr")
    c.run(1950)
    assert (c.status() == 0).all()
    co2 = c.fetchvars("CO2_concentration", (1745, 1950))
    assert np.array_equal(co2[:, 1], co2[:, 0]) and np.array_equal(co2[:, 1], co2[:, 2])
    # ... and equals a core that never saw the bad values
    f = mk(emul_lib, n)
    f.run(1950)
    assert np.abs(f.fetchvars("CO2_concentration", (1745, 1950)) - co2).max() < 1e-9


def c_fetch_ffi(c, years):
    return c.fetchvars("ffi_emissions", (years[0], years[-1]))[:, 0]


def test_setvar_dated_after_member_series_reaches_every_member(emul_lib):
    n = 2
    c = mk(emul_lib, n)
    yrs = list(range(1850, 1900))
    base = np.tile(c_fetch_ffi(c, yrs)[:, None], (1, n))
    per = base.copy()
    per[:, 1] += 0.5
    c.setvar_dated_members("ffi_emissions", yrs, per, "Pg C/yr")
    c.run(1900)
    before = c.fetchvars("CO2_concentration", (1900, 1900))[0]
    # the same value for every member, through the shared-series API
    c.setvar_dated("ffi_emissions", yrs, np.full(len(yrs), 20.0), "Pg C/yr")
    got = c.fetchvars("ffi_emissions", (1850, 1899))
    assert (got == 20.0).all()
    c.run(1900)
    after = c.fetchvars("CO2_concentration", (1900, 1900))[0]
    assert (after > before + 100).all() and after[0] == after[1]
    # same result as a core that only ever had the shared series
    f = mk(emul_lib, n)
    f.setvar_dated("ffi_emissions", yrs, np.full(len(yrs), 20.0), "Pg C/yr")
    f.run(1900)
    assert np.abs(f.fetchvars("CO2_concentration", (1900, 1900))[0] - after).max() < 1e-9


def test_pack_series_must_cover_the_run(tmp_path, emul_lib):
    lines = open(SCENARIO).read().splitlines()
    out, done = [], False
    for ln in lines:
        p = ln.split()
        if not done and len(p) > 5 and p[0] == "series" and p[2] == "ffi_emissions":
            ln = " ".join(p[:3] + ["1900", "10"] + p[5:15])   # truncated, starts elsewhere
            done = True
        out.append(ln)
    assert done
    bad = tmp_path / "short.hxs"
    bad.write_text("\n".join(out) + "\n")
    with pytest.raises(hector_amd.HectorAmdError, match="does not cover"):
        hector_amd.Core(str(bad), 1, lib_path=emul_lib, allow_emulation=True)
    # negative length
    out2 = [(" ".join(l.split()[:4] + ["-3"]) if l.startswith("series simpleNbox luc_emissions") else l)
            for l in lines]
    bad2 = tmp_path / "neg.hxs"
    bad2.write_text("\n".join(out2) + "\n")
    with pytest.raises(hector_amd.HectorAmdError, match="bad length"):
        hector_amd.Core(str(bad2), 1, lib_path=emul_lib, allow_emulation=True)


def test_shared_spinup_keeps_per_member_derive_flags(emul_lib):
    """Every member shares the spinup (only S differs); status bits found while deriving a
    member's own DOECLIM constants must stay with that member, whichever lane it is."""
    n = 4
    c = mk(emul_lib, n)
    c.set_member_sorting(False)
    c.setvar("S", np.array([3.0, 2.5, 4.0, 3.5]), "degC")
    assert (c.status() == 0).all()     # (spinup done) a healthy lane 0 does not flag anybody
    rows = c.state_row(0)
    assert np.all(rows == rows[0])     # shared spinup state
    c.run(1760)
    assert (c.status() == 0).all()


def test_biome_outputs_follow_their_biome(emul_lib):
    c = mk(emul_lib, 2)
    c.split_biome(["a", "b", "c"])
    c.set_outputs(["b.veg_c"])
    c.delete_biome("a")
    c.run(1760)
    v = c.fetchvars("b.veg_c", (1745, 1760))
    assert np.isfinite(v).all() and (v > 0).all()
    with pytest.raises(hector_amd.HectorAmdError, match="not enabled"):
        c.fetchvars("c.veg_c", (1745, 1760))
    # splitting the remaining biome b: its outputs go away with it, c's stay off
    c.set_outputs(["c.soil_c"])
    hector_amd.split_biome(c, "b", ["b1", "b2"])
    c.run(1750)
    assert (c.fetchvars("c.soil_c", (1745, 1750)) > 0).all()
    with pytest.raises(hector_amd.HectorAmdError, match="not enabled"):
        c.fetchvars("b1.soil_c", (1745, 1750))
