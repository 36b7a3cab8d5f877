"""The spinup as the reference's output stream sees it (CSVOutputStreamVisitor is visited after
every spinup step with spinup = 1: src/core.cpp:402-408, src/csv_outputstream_visitor.cpp:86-95):
hx_enable_spinup_record / hx_spinup_record against the oracle's restatement of the spinup loop
(carbon-cycle-solver.cpp:313-370), step by step."""
import numpy as np
import pytest

import hector_amd
from conftest import SCENARIO

# capability -> the oracle's name for it is the same
REL = 1e-10


def spinup_record_checks(lib, oracle, **kw):
    # (1) a shared spinup: no member differs in a parameter the spinup sees
    c = hector_amd.Core(SCENARIO, 3, lib_path=lib, **kw)
    c.enable_spinup_record()
    c.set_outputs(["CO2_concentration", "atmos_co2"])
    c.setvar("S", [2.0, 3.0, 4.5], "degC")
    want, err, steps = oracle.run_spinup()
    assert err == 0 and steps == 498
    for member in (0, 2):
        got = c.spinup_record(member)
        assert len(got) == 21 and all(len(v) == steps for v in got.values())
        for name, v in got.items():
            scale = max(np.abs(want[name]).max(), 1e-30)
            assert np.abs(v - want[name]).max() <= REL * scale, name
    # the run that follows starts from the recorded end state
    c.run(1750)
    assert c.fetchvars("atmos_co2", (1745, 1745))[0, 0] == got["atmos_co2"][-1]
    c.shutdown()
    # (2) every member its own spinup (initial pools and NPP differ), lanes reordered
    c = hector_amd.Core(SCENARIO, 3, lib_path=lib, **kw)
    c.enable_spinup_record()
    npp0 = np.array([50.0, 56.2, 61.0]); veg = np.array([500.0, 550.0, 620.0])
    c.setvar("npp_flux0", npp0).setvar("veg_c", veg)
    for i in range(3):
        p = oracle.default_params(); p.npp_flux0[0] = npp0[i]; p.veg_c[0] = veg[i]
        want, err, steps = oracle.run_spinup(p)
        assert err == 0 and steps == c.spinup_steps(i)
        got = c.spinup_record(i)
        for name, v in got.items():
            scale = max(np.abs(want[name]).max(), 1e-30)
            assert len(v) == steps and np.abs(v - want[name]).max() <= REL * scale, (i, name)
    c.shutdown()
    # (3) four biomes: the stream's global rows are sums over the biomes
    c = hector_amd.Core(SCENARIO, 1, lib_path=lib, **kw)
    c.enable_spinup_record()
    c.split_biome(["a", "b", "c", "d"], fveg_c=[0.1, 0.2, 0.3, 0.4])
    c.setvar("b.q10_rh", [2.6])
    p = oracle.split_equal(oracle.default_params(), 4)
    for b, f in enumerate([0.1, 0.2, 0.3, 0.4]):
        p.veg_c[b] = 550.0 * f
    p.q10_rh[1] = 2.6
    want, err, steps = oracle.run_spinup(p)
    got = c.spinup_record(0)
    assert err == 0 and steps == c.spinup_steps(0)
    for name, v in got.items():
        scale = max(np.abs(want[name]).max(), 1e-30)
        assert np.abs(v - want[name]).max() <= 1e-9 * scale, name
    c.shutdown()


def test_spinup_record_vs_oracle(emul_lib, oracle):
    spinup_record_checks(emul_lib, oracle, allow_emulation=True)


def test_spinup_record_is_off_by_default_and_routed_over_a_device_list(emul_lib, oracle):
    c = hector_amd.Core(SCENARIO, 2, lib_path=emul_lib, allow_emulation=True)
    with pytest.raises(hector_amd.HectorAmdError, match="spinup record is off"):
        c.spinup_record(0)
    c.shutdown()
    many = hector_amd.Core(SCENARIO, 5, devices=[0, 0], lib_path=emul_lib, allow_emulation=True)
    many.enable_spinup_record()
    want, _, steps = oracle.run_spinup()
    got = many.spinup_record(4)
    assert len(got["NBP"]) == steps and np.abs(got["soil_c"] - want["soil_c"]).max() < 1e-9
    many.shutdown()


@pytest.mark.gpu
def test_spinup_record_on_gpu(hip_lib, oracle):
    spinup_record_checks(hip_lib, oracle, device=0)
