"""The plain kernel + diagnostics (CON = -2, hx_dev_solver.h: hx_cons): what a run takes that merely
RECORDS one of the extended kernel's outputs -- NPP, RH and its parts, the ocean boxes' carbon, pCO2
and uptake, gmst ... (csv_outputstream_visitor.cpp:126-365) -- while no constraint, land-ocean
warming ratio or per-member series exists anywhere.  It carries none of their code (65 536 members
with NPP recorded: 6.75 -> 6.3 ms).  Held here against the extended kernel proper
(HECTOR_AMD_EXTENDED_CONS=1 keeps a run on it), against the plain kernel's carbon cycle, and --
in tests/test_diagnostics.py, which now runs on it -- against the oracle."""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO, edited_pack

DIAG = ["NPP", "RH", "rh_det", "rh_soil", "rh_ch4", "f_frozen", "atmos_c_residual", "gmst",
        "HL_ocean_uptake", "LL_ocean_uptake", "ocean_uptake", "HL_ocean_c", "LL_ocean_c",
        "IO_ocean_c", "DO_ocean_c", "HL_downwelling", "HL_PCO2", "LL_PCO2", "HL_pH", "TAU_OH",
        "permafrost_c", "thawedp_c", "NBP", "RF_tot"]
REL = 2e-8   # (what tests/test_diagnostics.py holds every diagnostic to against the oracle: two instantiations
             # contract different multiply-adds -- 7e-11 in CO2, 1.2e-8 in a box's pCO2, whose carbonate
             # solve stops at that size; identical step sequences are asserted)
# air-sea fluxes are (CO2 - pCO2) x a large exchange coefficient: a 7e-11 difference in CO2 shows
# as 1e-7 of the flux (2.3e-7 PgC/yr seen on the GPU); NBP is a small difference of NPP and RH
FLUX = {v: 50.0 for v in ("HL_ocean_uptake", "LL_ocean_uptake", "ocean_uptake", "NBP")}


def _run(lib, n, outs, monkeypatch, cons, biomes=1, two_wave=0, diff=False, scen=SCENARIO, lo=False, **kw):
    if cons:
        monkeypatch.setenv("HECTOR_AMD_EXTENDED_CONS", "1")
    else:
        monkeypatch.delenv("HECTOR_AMD_EXTENDED_CONS", raising=False)
    c = hector_amd.Core(scen, n, lib_path=lib, **kw)
    c.set_pair_kernel_limit(0).set_two_wave_from(two_wave)
    S, q10 = ensemble.ecs_q10(n)
    c.setvar("S", S, "degC")
    if biomes > 1:
        names = ["b%d" % i for i in range(biomes)]
        c.split_biome(names, fveg_c=[1.0 / biomes] * biomes)
        for b, nm in enumerate(names):
            c.setvar(nm + ".q10_rh", q10 + 0.1 * b)
            c.setvar(nm + ".warmingfactor", np.full(n, 1.0 + 0.15 * b))
        outs = outs + ["b1.NPP", "b0.RH", "b%d.veg_c" % (biomes - 1)]
    else:
        c.setvar("q10_rh", q10)
    if diff:
        c.setvar("diff", 1.2 + 2.2 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 5), "cm2/s")
    if lo:
        c.setvar("lo_warming_ratio", np.where(np.arange(n) % 2, 1.6, 0.0))
    c.set_outputs(outs)
    c.run(2300)
    assert (c.status() == 0).all()
    r = {v: c.fetchvars(v, (1745, 2300)) for v in outs}
    variant, kernel = c.last_run_variant(), c.last_run_kernel()
    c.shutdown()
    return r, variant, kernel


def _check(lib, monkeypatch, n, tmp_path, **kw):
    base = ["CO2_concentration", "global_tas", "timesteps"]
    for biomes, two_wave, diff, outs in ((1, 0, False, base + DIAG), (1, 0, True, base + DIAG + ["heatflux"]),
                                         (1, 1, False, base + DIAG), (1, 1, True, base + ["NPP", "heatflux_mixed"]),
                                         (4, 0, False, base + DIAG), (2, 0, False, base + ["NPP", "heatflux"]),
                                         (6, 0, False, base + ["NPP", "RH"])):
        a, va, ka = _run(lib, n, outs, monkeypatch, False, biomes, two_wave, diff, **kw)
        b, vb, kb = _run(lib, n, outs, monkeypatch, True, biomes, two_wave, diff, **kw)
        # (the family is built for one to four biomes: more take the extended kernel either way)
        assert (va, vb) == ((-2 if biomes <= 4 else -1), -1) and ka == kb == ("run2" if two_wave else "run"), (biomes, va, vb, ka, kb)
        assert np.array_equal(a["timesteps"], b["timesteps"])
        for v in a:
            scale = np.abs(b[v]).max() + 1e-30
            assert np.abs(a[v] - b[v]).max() / scale < REL * FLUX.get(v, 1.0), (biomes, two_wave, diff, v)
    # the carbon cycle is the plain kernel's
    p, vp, _ = _run(lib, n, base, monkeypatch, False, **kw)
    d, vd, _ = _run(lib, n, base + ["NPP"], monkeypatch, False, **kw)
    assert (vp, vd) == (0, -2)
    assert np.array_equal(p["timesteps"], d["timesteps"])
    for v in ("CO2_concentration", "global_tas"):
        assert np.abs(p[v] - d[v]).max() / np.abs(p[v]).max() < REL, v
    # anything the diagnostics-only kernel does not carry keeps a run on the extended one
    years = np.arange(1950, 2011)
    tas = edited_pack(tmp_path / "tas.hxs", "temperature", "tas_constrain", years, 0.3 + 0.01 * (years - 1950))
    assert _run(lib, n, base + ["NPP"], monkeypatch, False, scen=tas, **kw)[1] == -1
    assert _run(lib, n, base + ["NPP"], monkeypatch, False, lo=True, **kw)[1] == -1


def test_diagnostics_only_kernel_in_the_host_build(emul_lib, tmp_path, monkeypatch):
    _check(emul_lib, monkeypatch, 8, tmp_path, allow_emulation=True)


@pytest.mark.gpu
def test_diagnostics_only_kernel_on_gpu(hip_lib, tmp_path, monkeypatch):
    _check(hip_lib, monkeypatch, 1024, tmp_path, device=0)
