"""tests/golden/oracle_vectors.npz (tools/make_regression_vectors.py): vectors of the oracle --
64 ECS x Q10 members, 16 four-biome members, 200 carbonate solves, two DOECLIM kernel tables --
committed so that neither the oracle nor the kernels can drift unnoticed (SURVEY 8c ii-iv; the
reference itself cannot be built here, its own golden member is tests/test_oracle_golden.py)."""
import ctypes
import os

import numpy as np
import pytest

import hector_amd
from conftest import ROOT, SCENARIO

V = np.load(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"))
# the oracle against its own vectors: another compiler / libm may move the last digits
ORACLE_TOL = 1e-11
# the kernels against the vectors: the parity tolerance of the other tests
REL_CO2, ABS_T = 2e-8, 2e-8


def test_oracle_reproduces_its_vectors(oracle):
    for i in range(0, 64, 7):
        p = oracle.default_params(); p.S = V["ecs_S"][i]; p.q10_rh[0] = V["ecs_q10"][i]
        r, err, _ = oracle.run(p)
        assert err == 0
        assert (np.abs(r["CO2_concentration"] - V["ecs_co2"][i]) / V["ecs_co2"][i]).max() < ORACLE_TOL
        assert np.abs(r["global_tas"] - V["ecs_tgav"][i]).max() < ORACLE_TOL
        assert np.array_equal(r["timesteps"].astype(np.int8), V["ecs_stashes"][i])
    for i in range(0, 16, 5):
        p = oracle.split_equal(oracle.default_params(), 4); p.S = V["b4_S"][i]
        for b in range(4):
            p.q10_rh[b] = V["b4_q10"][b][i]; p.warmingfactor[b] = V["b4_wf"][b][i]
        r, err, _ = oracle.run(p)
        assert err == 0
        assert (np.abs(r["CO2_concentration"] - V["b4_co2"][i]) / V["b4_co2"][i]).max() < ORACLE_TOL
        assert np.abs(r["permafrost_c"] - V["b4_permafrost"][i]).max() < ORACLE_TOL * 1e3
    vol = float(V["csys_vol"])
    for i in range(200):
        o = oracle.csys(V["csys_T"][i], V["csys_carbon"][i], V["csys_alk"][i], vol)[:3]
        assert np.allclose(o, V["csys_out"][i], rtol=1e-12, atol=0)
    for k, d in enumerate(V["ker_diff"]):
        assert np.abs(oracle.doeclim_kernel(float(d), 556) - V["ker"][k]).max() < 1e-15


def kernels_vs_vectors(lib_path, loaded_lib, **kw):
    c = hector_amd.Core(SCENARIO, 64, lib_path=lib_path, **kw)
    c.setvar("S", V["ecs_S"], "degC").setvar("q10_rh", V["ecs_q10"])
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps"]); c.run(2300)
    assert (c.status() == 0).all()
    co2 = c.fetchvars("CO2_concentration", (1745, 2300)).T
    assert (np.abs(co2 - V["ecs_co2"]) / V["ecs_co2"]).max() < REL_CO2
    assert np.abs(c.fetchvars("global_tas", (1745, 2300)).T - V["ecs_tgav"]).max() < ABS_T
    assert np.array_equal(c.fetchvars("timesteps", (1746, 2300)).T.astype(np.int8), V["ecs_stashes"][:, 1:])
    b = hector_amd.Core(SCENARIO, 16, lib_path=lib_path, **kw)
    names = ["b1", "b2", "b3", "b4"]
    b.split_biome(names); b.setvar("S", V["b4_S"], "degC")
    for k, nm in enumerate(names):
        b.setvar(nm + ".q10_rh", V["b4_q10"][k]).setvar(nm + ".warmingfactor", V["b4_wf"][k])
    b.set_outputs(["CO2_concentration", "global_tas", "permafrost_c"]); b.run(2300)
    assert (b.status() == 0).all()
    assert (np.abs(b.fetchvars("CO2_concentration", (1745, 2300)).T - V["b4_co2"]) / V["b4_co2"]).max() < REL_CO2
    assert np.abs(b.fetchvars("global_tas", (1745, 2300)).T - V["b4_tgav"]).max() < ABS_T
    assert np.abs(b.fetchvars("permafrost_c", (1745, 2300)).T - V["b4_permafrost"]).max() < 2e-8 * 865
    dp = ctypes.POINTER(ctypes.c_double)
    as_p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    out = np.zeros((200, 4))
    dev = kw.get("device", 0)
    rc = loaded_lib.hx_unit_csys(dev, 200, as_p(V["csys_T"]), as_p(V["csys_carbon"]), as_p(V["csys_alk"]),
                                 float(V["csys_vol"]), out.ctypes.data_as(dp))
    assert rc == 0 and (out[:, 3] == 0).all()
    assert np.allclose(out[:, :3], V["csys_out"], rtol=1e-12, atol=0)
    for k, d in enumerate(V["ker_diff"]):
        ker = np.zeros(556)
        assert loaded_lib.hx_unit_doeclim_kernel(dev, float(d), 556, ker.ctypes.data_as(dp)) == 0
        assert np.abs(ker - V["ker"][k]).max() < 1e-12 * np.abs(V["ker"][k]).max()


def test_kernels_reproduce_the_vectors(emul_lib):
    import hector_amd._lib as L
    kernels_vs_vectors(emul_lib, L.load(emul_lib, allow_emulation=True), allow_emulation=True)


@pytest.mark.gpu
def test_kernels_reproduce_the_vectors_on_gpu(hip_lib):
    import hector_amd._lib as L
    kernels_vs_vectors(hip_lib, L.load(hip_lib), device=0)
