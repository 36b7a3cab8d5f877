"""A core over a device LIST (hx_newcore_devices): every verb routed to the shards.

CPU part: the host-emulation build of the product sources with the device list [0, 0, 0] (the
emulation has one device; duplicate devices exchange their statistics by copies, the rehearsal
path of hx_fleet.cpp) against ONE core over the same members: every routed call must give the
same answer bit for bit -- members are independent.  GPU part: tests/test_gpu_fleet.py.
"""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble

N = 7           # 3 + 2 + 2: ragged blocks
RUN_TO = 1790


def _cores(emul_lib, n=N, **kw):
    one = hector_amd.Core(n_members=n, lib_path=emul_lib, allow_emulation=True, **kw)
    many = hector_amd.Core(n_members=n, devices=[0, 0, 0], lib_path=emul_lib, allow_emulation=True, **kw)
    return one, many


def test_shards_are_contiguous_ragged_blocks(emul_lib):
    one, many = _cores(emul_lib)
    assert one.shards() == ([0], [0, N])
    assert many.shards() == ([0, 0, 0], [0, 3, 5, 7])
    one.shutdown(); many.shutdown()


def test_fewer_members_than_devices_is_an_error(emul_lib):
    with pytest.raises(hector_amd.HectorAmdError, match="fewer members than devices"):
        hector_amd.Core(n_members=2, devices=[0, 0, 0], lib_path=emul_lib, allow_emulation=True)


def test_every_routed_verb_equals_the_single_core(emul_lib, tmp_path):
    one, many = _cores(emul_lib)
    S, q10 = ensemble.ecs_q10(N)
    years = np.arange(1760, 1770)
    ffi = 0.1 + 0.01 * np.arange(years.size * N).reshape(years.size, N)
    for c in (one, many):
        c.set_outputs(["CO2_concentration", "global_tas", "RF_tot", "HL_pH", "LL_pH"])
        c.setvar("S", S, "degC").setvar("q10_rh", q10).setvar("beta", [0.5])
        c.enable_history(True)
        c.setvar_dated_members("ffi_emissions", years, ffi)
        c.run(RUN_TO)
    np.testing.assert_array_equal(many.getvar("S"), S)
    np.testing.assert_array_equal(many.getvar("beta"), np.full(N, 0.5))
    # (on the device; a per-member input; combined on the host from two arrays; answered on the
    #  host from the scenario: each shard writes its columns of the whole ensemble's array)
    for v in ("CO2_concentration", "global_tas", "RF_tot", "ffi_emissions", "pH", "luc_emissions", "RF_SO2"):
        a, b = one.fetchvars(v, (1745, RUN_TO)), many.fetchvars(v, (1745, RUN_TO))
        np.testing.assert_array_equal(a, b, err_msg=v)
    np.testing.assert_array_equal(one.status(), many.status())
    np.testing.assert_array_equal(one.state_row(4), many.state_row(4))
    assert [one.spinup_steps(m) for m in range(N)] == [many.spinup_steps(m) for m in range(N)]
    assert many.current_date == RUN_TO and many.biomes() == ["global"]
    # reset(date) + a dated edit + run again: every shard follows
    for c in (one, many):
        c.setvar_dated("ffi_emissions", [1775, 1776], [3.0, 3.5])
        c.run(RUN_TO)
    np.testing.assert_array_equal(one.fetchvars("CO2_concentration", (1745, RUN_TO)),
                                  many.fetchvars("CO2_concentration", (1745, RUN_TO)))
    # biome verbs
    for c in (one, many):
        c.split_biome(["a", "b"], fveg_c=[0.3, 0.7])
        c.setvar("a.q10_rh", q10)
        c.run(1760)
    assert many.biomes() == ["a", "b"]
    np.testing.assert_array_equal(one.fetchvars("global_tas", (1745, 1760)),
                                  many.fetchvars("global_tas", (1745, 1760)))
    with pytest.raises(hector_amd.HectorAmdError, match="several GPUs"):
        many.device_var("global_tas")
    p, npad = many.device_var_shard(2, "global_tas")
    assert p and npad == 64
    one.shutdown(); many.shutdown()


def test_per_member_values_need_the_full_length(emul_lib):
    _, many = _cores(emul_lib)
    with pytest.raises(hector_amd.HectorAmdError, match="1 or n_members"):
        many.setvar("S", [3.0, 3.1, 3.2])
    many.shutdown()


def test_tracking_data_is_routed_to_the_members_shard(emul_lib):
    one, many = _cores(emul_lib, n=4)
    S, _ = ensemble.ecs_q10(4)
    for c in (one, many):
        c.setvar("S", S).setvar("trackingDate", [1750.0])
        c.run(1755)
    for m in (0, 3):
        va, fa = one.tracking_data(m, (1750, 1755))
        vb, fb = many.tracking_data(m, (1750, 1755))
        np.testing.assert_array_equal(va, vb)
        np.testing.assert_array_equal(fa, fb)
    one.shutdown(); many.shutdown()


def test_product_library_refuses_duplicate_devices_without_the_rehearsal_switch(hip_lib, monkeypatch):
    """No GPU here: the product library must fail -- either on the duplicate (checked first) or,
    with the switch, on the missing device; never fall back to anything."""
    monkeypatch.delenv("HECTOR_AMD_FLEET_REHEARSAL", raising=False)
    with pytest.raises(hector_amd.HectorAmdError, match="appears twice"):
        hector_amd.Core(n_members=128, devices=[0, 0], lib_path=hip_lib)


def test_c_host_example_builds_and_runs_over_a_device_list(emul_lib, tmp_path):
    """examples/multi_gpu_host.c: a plain C99 host on the C ABI alone -- what an R / C++ embedding
    compiles against -- over the device list [0, 0] of the host build."""
    import os
    import subprocess
    from conftest import ROOT, SCENARIO
    exe = str(tmp_path / "multi_gpu_host")
    libdir = os.path.dirname(emul_lib)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "multi_gpu_host.c"), "-o", exe,
                           "-L", libdir, "-lhector_amd_emul", "-Wl,-rpath," + libdir, "-lm"])
    r = subprocess.run([exe, SCENARIO, "6", "0,0", "1800"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "shard 0: GPU 0, members 0..2" in r.stdout and "shard 1: GPU 0, members 3..5" in r.stdout
    assert "members with model errors: 0" in r.stdout
    one = hector_amd.Core(n_members=6, lib_path=emul_lib, allow_emulation=True)
    n = 6
    one.setvar("S", 1.5 + 4.5 * (np.arange(n) + 0.5) / n).setvar("q10_rh", 1.0 + 2.0 * np.fmod(np.arange(n) * 0.6180339887498949, 1.0))
    one.run(1800)
    tas = one.fetchvars("global_tas", (1800, 1800))[0]
    line = [l for l in r.stdout.splitlines() if l.startswith("global_tas(1800)")][0]
    got = [float(x) for x in line.split(":")[1].split()]
    np.testing.assert_allclose(got, [tas[0], tas[3], tas[5]], atol=5e-7)
    one.shutdown()


def test_a_call_that_fails_on_a_later_shard_poisons_the_core(emul_lib):
    """A routed call that went through on shard 0 and failed on shard 1 has left the shards
    different: the core says which shard and why, and refuses every later call
    instead of mixing them silently."""
    _, many = _cores(emul_lib)
    fv = np.full(N, 0.35)
    fv[4] = 1.7                       # member 4 lives on shard 1: f_nppv > 1 fails its parameter check
    many.setvar("f_nppv", fv)
    with pytest.raises(hector_amd.HectorAmdError, match="f_nppv"):
        many.run(1760)                # shard 0 ran, shard 1 refused
    for call in (lambda: many.run(1770), lambda: many.setvar("beta", [0.4]), lambda: many.reset(1745),
                 lambda: many.fetchvars("CO2_concentration", (1745, 1750))):
        with pytest.raises(hector_amd.HectorAmdError, match="inconsistent and refuses further calls.*shard 1 of 3"):
            call()
    many.shutdown()
    # a failure on the FIRST shard leaves nothing half-done: the core stays usable
    _, many = _cores(emul_lib)
    fv = np.full(N, 0.35); fv[0] = 1.7
    many.setvar("f_nppv", fv)
    with pytest.raises(hector_amd.HectorAmdError, match="f_nppv"):
        many.run(1760)
    many.setvar("f_nppv", np.full(N, 0.35))
    many.run(1760)
    assert (many.status() == 0).all()
    many.shutdown()
