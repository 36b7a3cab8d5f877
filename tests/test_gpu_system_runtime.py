"""The library on the HIP runtime it was BUILT for (second half).

The product library is compiled by ROCm 7.2's hipcc; in a process that could also import PyTorch
it runs on the wheel's bundled runtime (HIP 7.0.x, `hector_amd/_lib.py: _share_torch_hip_runtime`)
so that both can share one runtime.  Where a system runtime of the build's major.minor version is
installed, `HECTOR_AMD_NO_TORCH_HIP=1` makes the library take THAT one: this test runs
`__graft_entry__.smoke()` (all three year-loop kernels against the oracle) in a fresh process that
way and checks which runtime answered.  Skipped where no such runtime exists."""
import glob
import os
import re
import subprocess
import sys

import pytest

import hector_amd
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _built_with(hip_lib):
    m = re.search(r"built with HIP (\d+)\.(\d+)", hector_amd.build_info(hip_lib))
    return (int(m.group(1)), int(m.group(2))) if m else None


def _system_runtime(major, minor):
    """Path of an installed libamdhip64.so.<major>.<minor>.*, or None."""
    hits = []
    for d in ("/opt/rocm/lib", "/opt/rocm/lib64", "/usr/lib/x86_64-linux-gnu"):
        hits += glob.glob(os.path.join(d, "libamdhip64.so.%d.%d.*" % (major, minor)))
    return hits[0] if hits else None


def test_smoke_on_the_system_runtime_of_the_build(hip_lib):
    built = _built_with(hip_lib)
    assert built is not None, hector_amd.build_info(hip_lib)
    rt = _system_runtime(*built)
    if rt is None:
        pytest.skip("no system HIP runtime %d.%d installed" % built)
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g, hector_amd\n"
        "g.smoke()\n"
        "assert 'torch' not in sys.modules\n"
        "print('BUILD_INFO', hector_amd.build_info())\n"
        "print('MAPS', sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))\n"
        % ROOT)
    env = dict(os.environ, HECTOR_AMD_NO_TORCH_HIP="1", HECTOR_AMD_VERBOSE="1")
    env["LD_LIBRARY_PATH"] = os.path.dirname(rt) + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert p.stdout.count("smoke ok") == 3, p.stdout
    maps = [l for l in p.stdout.splitlines() if l.startswith("MAPS")][0]
    # exactly one HIP runtime in the process, and it is the system's -- not the wheel's copy
    assert os.path.realpath(rt) in maps and "dist-packages/torch" not in maps, maps
    info = [l for l in p.stdout.splitlines() if l.startswith("BUILD_INFO")][0]
    m = re.search(r"runtime (\d+)", info)
    assert m and int(m.group(1)) // 10000000 == built[0] and (int(m.group(1)) // 100000) % 100 == built[1], info
