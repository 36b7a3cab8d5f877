"""The extended run kernel without the NBP machinery (CON = -1, hx_dev_solver.h: hx_nbp): taken for
diagnostics, constraints other than NBP and the warming ratio -- five solver variables and one set
of interval constants like the plain kernel, where the instantiation that can hold an NBP
constraint (HECTOR_AMD_EXTENDED_NBP=1 forces it) integrates the thawed pool as a sixth variable
and selects between two interval sets at every stage."""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO, edited_pack

REL = 2e-8


def _run(lib, scen, n, outs, monkeypatch, force_nbp, lo=False, **kw):
    if force_nbp:
        monkeypatch.setenv("HECTOR_AMD_EXTENDED_NBP", "1")
    else:
        monkeypatch.delenv("HECTOR_AMD_EXTENDED_NBP", raising=False)
    c = hector_amd.Core(scen, n, lib_path=lib, **kw)
    c.set_pair_kernel_limit(0)   # (the run kernels: small CO2- / tas-constrained ensembles would take the pair kernel)
    S, q10 = ensemble.ecs_q10(n)
    c.setvar("S", S, "degC").setvar("q10_rh", q10)
    if lo:
        c.setvar("lo_warming_ratio", np.where(np.arange(n) % 2, 1.6, 0.0))
    c.set_outputs(outs)
    c.run(2300)
    assert (c.status() == 0).all()
    r = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
    c.shutdown()
    return r


def _both(lib, tmp_path, monkeypatch, n, **kw):
    years = np.arange(1950, 2011)
    tas = edited_pack(tmp_path / "tas.hxs", "temperature", "tas_constrain", years, 0.3 + 0.01 * (years - 1950))
    co2 = edited_pack(tmp_path / "co2.hxs", "simpleNbox", "CO2_constrain", years, 310.0 + 1.2 * (years - 1950))
    for scen, outs, lo in ((SCENARIO, ["CO2_concentration", "global_tas", "NPP", "RH", "permafrost_c", "thawedp_c"], False),
                           (tas, ["CO2_concentration", "global_tas", "sst", "land_tas", "NPP"], True),
                           (co2, ["CO2_concentration", "global_tas", "NBP", "thawedp_c"], False)):
        a = _run(lib, scen, n, outs, monkeypatch, False, lo, **kw)
        b = _run(lib, scen, n, outs, monkeypatch, True, lo, **kw)
        for v in outs:
            scale = np.abs(b[v]).max() + 1e-30
            assert np.abs(a[v] - b[v]).max() / scale < REL, (scen, v)
    # diagnostics alone: the carbon cycle is the plain kernel's, operation by operation
    outs = ["CO2_concentration", "global_tas", "timesteps"]
    plain = _run(lib, SCENARIO, n, outs, monkeypatch, False, **kw)
    ext = _run(lib, SCENARIO, n, outs + ["NPP"], monkeypatch, False, **kw)
    assert np.array_equal(plain["timesteps"], ext["timesteps"])
    for v in ("CO2_concentration", "global_tas"):
        assert np.abs(plain[v] - ext[v]).max() / np.abs(plain[v]).max() < REL, v


def test_extended_kernel_without_nbp_in_the_host_build(emul_lib, tmp_path, monkeypatch):
    _both(emul_lib, tmp_path, monkeypatch, 8, allow_emulation=True)


@pytest.mark.gpu
def test_extended_kernel_without_nbp_on_gpu(hip_lib, tmp_path, monkeypatch):
    _both(hip_lib, tmp_path, monkeypatch, 2048, device=0)
