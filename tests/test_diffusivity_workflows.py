"""Members with their own ocean heat diffusivity (per-member DOECLIM kernel tables) through the
workflows that restart the history blocks: a run in pieces that end in the middle of a block, a
reset to a date inside a block, the heat-flux sum and a diagnostic recorded -- bit for bit the
one-launch run, and the oracle's trajectories.  Round 5 changed where such kernels get their kernel
entries from (requested before the solver; the pass's window in halves, a chunk ahead:
hx_kernels.hip), and a launch that starts anywhere makes its first block as long as it likes."""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO

OUTS = ["CO2_concentration", "global_tas", "heatflux", "sst", "NPP", "timesteps"]


def _core(lib, n, biomes, **kw):
    c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
    c.set_pair_kernel_limit(0)
    S, q10 = ensemble.ecs_q10(n)
    c.setvar("S", S, "degC")
    if biomes > 1:
        names = ["b%d" % i for i in range(biomes)]
        c.split_biome(names)
        for b, nm in enumerate(names):
            c.setvar(nm + ".q10_rh", q10 + 0.05 * b)
    else:
        c.setvar("q10_rh", q10)
    diff = 1.2 + 2.2 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 5)
    c.setvar("diff", diff, "cm2/s")
    c.set_outputs(OUTS)
    c.enable_history(True)
    return c, S, q10, diff


def diffusivity_workflows(lib, oracle, n, n_oracle, **kw):
    for biomes in (1, 3):
        a, S, q10, diff = _core(lib, n, biomes, **kw)
        a.run(2300)
        ref = {v: a.fetchvars(v, (1745, 2300)) for v in OUTS}
        assert (a.status() == 0).all()
        # in pieces: launches that start in the middle of what was a block, and one-year launches
        b, _, _, _ = _core(lib, n, biomes, **kw)
        for y in (1746, 1747, 1790, 1811, 1843, 1844, 1900, 2077, 2300):
            b.run(y)
        for v in OUTS:
            assert np.array_equal(ref[v], b.fetchvars(v, (1745, 2300))), (biomes, "pieces", v)
        # back to a date inside a block, and on in two launches
        b.reset(1861); b.run(1950); b.run(2300)
        for v in OUTS:
            assert np.array_equal(ref[v], b.fetchvars(v, (1745, 2300))), (biomes, "reset", v)
        if biomes == 1:   # ... and what the numbers are: the oracle's
            for i in range(n_oracle):
                p = oracle.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]; p.diff = diff[i]
                r, err, _ = oracle.run(p)
                assert err == 0
                rel = np.abs(ref["CO2_concentration"][:, i] - r["CO2_concentration"]) / r["CO2_concentration"]
                assert rel.max() < 2e-8, (i, rel.max())
                assert np.abs(ref["global_tas"][:, i] - r["global_tas"]).max() < 2e-8, i
                assert np.abs(ref["heatflux"][1:, i] - r["heatflux"][1:]).max() < 2e-7, i
                assert np.array_equal(ref["timesteps"][1:, i], r["timesteps"][1:]), i
        a.shutdown(); b.shutdown()


def test_diffusivity_workflows_in_the_host_build(emul_lib, oracle):
    diffusivity_workflows(emul_lib, oracle, 5, 2, allow_emulation=True)


@pytest.mark.gpu
def test_diffusivity_workflows_on_gpu(hip_lib, oracle):
    diffusivity_workflows(hip_lib, oracle, 192, 4, device=0)
