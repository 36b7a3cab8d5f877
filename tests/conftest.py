import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SCENARIO = os.path.join(ROOT, "hector_amd", "data", "ssp245.hxs")
GOLDEN = os.path.join(ROOT, "tests", "golden", "hector_comp_ssp245.txt")
EMUL_LIB = os.path.join(ROOT, "tests", "emul", "libhector_amd_emul.so")
HIP_LIB = os.path.join(ROOT, "hector_amd", "lib", "libhector_amd.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _make(path):
    subprocess.check_call(["make", "-s", "-C", path])


@pytest.fixture(scope="session")
def oracle():
    _make(os.path.join(ROOT, "oracle"))
    import oracle_binding
    return oracle_binding.Oracle(SCENARIO)


@pytest.fixture(scope="session")
def emul_lib():
    """Test-only host build of the product sources (tests/emul/README)."""
    _make(os.path.join(ROOT, "tests", "emul"))
    return EMUL_LIB


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    g = {}
    with open(GOLDEN) as f:
        for line in f:
            if line.startswith("#"):
                continue
            p = line.split()
            g[p[0]] = np.array([float(x) for x in p[3:]])
    return g


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests must load THIS, never the emulation."""
    if not os.path.exists(HIP_LIB):
        _make(os.path.join(ROOT, "hector_amd", "csrc"))
    return HIP_LIB


def edited_pack(path, section, key, years, values, base=None, scalars=None):
    """Write a copy of a scenario pack with `section.key` changed at `years` -- what a
    reference user gets from setvar(core, dates, var, values) -- for the oracle to read.
    A series the pack does not hold yet (a constraint) is added, NaN where it has no value.
    `scalars`: {(section, key): value} to add or replace."""
    out, found, y0, n = [], False, None, None
    with open(base or SCENARIO) as f:
        for line in f:
            p = line.split()
            if len(p) > 5 and p[0] == "series":
                y0, n = int(p[3]), int(p[4])
                if p[1] == section and p[2] == key:
                    found = True
                    v = p[5:5 + n]
                    for y, x in zip(years, values):
                        v[int(y) - y0] = repr(float(x))
                    line = " ".join(p[:5] + v) + "\n"
            if len(p) > 3 and p[0] == "scalar" and scalars and (p[1], p[2]) in scalars:
                continue
            out.append(line)
    if not found and section:
        v = ["nan"] * n
        for y, x in zip(years, values):
            v[int(y) - y0] = repr(float(x))
        out.append(" ".join(["series", section, key, str(y0), str(n)] + v) + "\n")
    for (sec, k), val in (scalars or {}).items():
        out.append("scalar %s %s %r\n" % (sec, k, float(val)))
    with open(path, "w") as f:
        f.writelines(out)
    return str(path)
