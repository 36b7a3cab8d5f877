"""include/hector_amd_component.hpp: the IModelComponent adapter (north star: "keeping the
Core::run / IModelComponent ... API surface"; inst/include/imodel_component.hpp:34-172),
compiled and driven like the reference's Core drives a component -- one run() per model year,
GETDATA messages, setData with and without dates, reset -- and compared with the same run made
through the C ABI directly."""
import os
import subprocess

import numpy as np
import pytest

import hector_amd
from conftest import ROOT, SCENARIO, EMUL_LIB, HIP_LIB


def build_harness(tmp_path, lib):
    exe = str(tmp_path / "adapter_harness")
    libdir, libname = os.path.dirname(lib), os.path.basename(lib)[3:-3]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "adapter", "harness.cpp"), "-o", exe,
                           "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir])
    return exe


def adapter_checks(tmp_path, lib, **kw):
    exe = build_harness(tmp_path, lib)
    r = subprocess.run([exe, SCENARIO], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = {}
    for line in r.stdout.splitlines():
        y, var, val, unit = line.split(None, 3)
        got[(int(y), var)] = (float(val), unit)
    c = hector_amd.Core(SCENARIO, 1, lib_path=lib, **kw)
    c.set_outputs(["CO2_concentration", "global_tas", "RF_tot"])
    c.setvar("S", 3.5)
    c.setvar_dated("ffi_emissions", np.arange(2030, 2061), np.full(31, 4.0))
    c.run(2100)
    units = {"CO2_concentration": "ppmv CO2", "global_tas": "degC", "RF_tot": "W/m2"}
    n = 0
    for (y, var), (val, unit) in got.items():
        if var in units:
            ref = c.fetchvars(var, (y, y))[0, 0]
            assert val == ref, (y, var)          # year-by-year run() == one run(): bit for bit
            assert unit == units[var]
            n += 1
    assert n == 3 * 8   # 1750, 1800 ... 2100
    assert got[(0, "S")] == (3.5, "degC")
    assert got[(1800, "rerun_CO2_concentration")][0] == c.fetchvars("CO2_concentration", (1800, 1800))[0, 0]
    assert got[(0, "unknown_variable_throws")][0] == 1.0


def test_component_adapter(emul_lib, tmp_path):
    adapter_checks(tmp_path, emul_lib, allow_emulation=True)


@pytest.mark.gpu
def test_component_adapter_on_gpu(hip_lib, tmp_path):
    adapter_checks(tmp_path, hip_lib, device=0)
