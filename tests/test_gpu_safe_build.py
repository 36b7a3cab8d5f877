"""Correctness does not hang on the product build's compiler flags.

`make -C hector_amd/csrc safe` builds the same kernel sources with default code generation (no
`-mllvm -disable-machine-licm`) and the carbonate restart in its select form (`-DHX_CHEM_SELECT`,
the form ROCm 7.2's register allocator cannot miscompile); tools/check_isa.py passes on its
assembly or the build fails.  Here that library runs what the product library runs: the smoke
check of all three year-loop kernels and the ECS/Q10 and four-biome ensembles against the oracle.
Slower is fine; wrong is not."""
import os
import subprocess

import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import ROOT

pytestmark = pytest.mark.gpu
SAFE_LIB = os.path.join(ROOT, "hector_amd", "lib", "libhector_amd_safe.so")


@pytest.fixture(scope="module")
def safe_lib():
    if not os.path.exists(SAFE_LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hector_amd", "csrc"), "safe"])
    return SAFE_LIB


def test_safe_build_is_another_build(safe_lib, hip_lib):
    a, b = hector_amd.build_info(safe_lib), hector_amd.build_info(hip_lib)
    assert "safe build" in a and "HX_CHEM_SELECT" in a and "disable-machine-licm" not in a
    assert "product build" in b and "disable-machine-licm" in b


def test_smoke_of_the_three_year_loop_kernels_on_the_safe_build(safe_lib, oracle):
    n, run_to = 64, 1800
    S, q10 = ensemble.ecs_q10(n)
    oco2, otg, err = oracle.run_ecs_q10(S[:8], q10[:8], run_to)
    assert err == 0
    k = run_to - 1745 + 1
    for limit, two_wave, which in ((32768, 0, "pair"), (0, 0, "run"), (0, 1, "run2")):
        c = hector_amd.Core(n_members=n, device=0, lib_path=safe_lib)
        c.set_pair_kernel_limit(limit).set_two_wave_from(two_wave)
        c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
        c.run(run_to)
        assert c.last_run_kernel() == which
        co2 = c.fetchvars("CO2_concentration", (1745, run_to))
        tg = c.fetchvars("global_tas", (1745, run_to))
        assert (c.status() == 0).all()
        assert (np.abs(co2[:, :8].T - oco2[:, :k]) / oco2[:, :k]).max() < 1e-8, which
        assert np.abs(tg[:, :8].T - otg[:, :k]).max() < 1e-8, which
        c.shutdown()


def test_ecs_q10_ensemble_vs_oracle_on_the_safe_build(safe_lib, oracle):
    import test_gpu_parity
    test_gpu_parity.test_ecs_q10_ensemble_vs_oracle(safe_lib, oracle)


def test_four_biome_ensemble_vs_oracle_on_the_safe_build(safe_lib, oracle):
    import test_gpu_parity
    test_gpu_parity.test_four_biome_ensemble_vs_oracle(safe_lib, oracle)
