"""libhector_amd.so and PyTorch-ROCm in one process, in either order: one HIP runtime
(hector_amd/_lib.py: _share_torch_hip_runtime).  Each order in a fresh interpreter."""
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

LIB_FIRST = """
import numpy as np, hector_amd
c = hector_amd.Core(hector_amd.DEFAULT_SCENARIO, 64, device=0)
assert c.backend == "hip"
c.run(1800)
import torch
x = torch.ones(8, dtype=torch.float64, device="cuda:0")            # torch's device init after ours
ptr, rows = c.device_var("CO2_concentration")
assert abs(float(x.sum()) - 8.0) == 0 and rows > 0
c.run(1850)                                                       # and ours still runs
assert np.isfinite(c.fetchvars("CO2_concentration", (1746, 1850))).all()
rt = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
assert len(rt) == 1, rt
print("ok", rt[0])
"""

TORCH_FIRST = """
import torch
x = torch.ones(8, dtype=torch.float64, device="cuda:0")
import numpy as np, hector_amd
c = hector_amd.Core(hector_amd.DEFAULT_SCENARIO, 64, device=0)
c.run(1850)
assert np.isfinite(c.fetchvars("CO2_concentration", (1746, 1850))).all()
rt = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
assert len(rt) == 1, rt
print("ok", rt[0])
"""


@pytest.mark.parametrize("script", [LIB_FIRST, TORCH_FIRST], ids=["library-first", "torch-first"])
def test_one_hip_runtime_in_either_import_order(script):
    r = subprocess.run([sys.executable, "-c", script], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("ok")
