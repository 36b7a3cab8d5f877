"""More than four biomes: the reference creates any number of biomes
(src/simpleNbox.cpp:864-1124, biome_list.push_back at :928; tests/testthat/test_biome.R:127-300);
1-8 run fully unrolled kernels, 9-32 the looped kernels (template tag HX_DYN: the biomes' pools
in the LDS park, loops to the core's biome count; HX_BDYN = 32 since round 5 -- 17 to 32 biomes
take 41-71 KB of LDS a wavefront, two wavefronts a CU).  Heterogeneous splits into 5 ... 32
biomes -- unequal pool fractions, per-biome Q10 / warming factor / beta, per member -- against
the oracle; the identical-split property; per-biome outputs; create / delete keep working."""
import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from conftest import SCENARIO

Y0 = 1745


def many_biome_checks(lib, oracle, counts=(5, 8, 16), n=6, run_to=2300, **kw):
    for nb in counts:
        names = ["b%02d" % i for i in range(nb)]
        r = np.random.default_rng(100 + nb)
        fr = r.random(nb) + 0.3; fr /= fr.sum()
        fs = r.random(nb) + 0.3; fs /= fs.sum()
        c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
        hector_amd.split_biome(c, "global", names, fveg_c=fr, fdetritus_c=fr, fsoil_c=fs,
                               fpermafrost_c=fs, fnpp_flux0=fr)
        assert c.biomes() == names
        S = 2.0 + 3.0 * ensemble.uniform01(np.arange(n), 70, seed=nb)
        q10 = {b: 1.2 + 1.6 * ensemble.uniform01(np.arange(n), 71 + i, seed=nb) for i, b in enumerate(names)}
        wf = {b: 0.8 + 1.4 * r.random() for b in names}
        beta = {b: 0.3 + 0.5 * r.random() for b in names}
        c.setvar("S", S, "degC")
        for b in names:
            c.setvar(b + ".q10_rh", q10[b]).setvar(b + ".warmingfactor", wf[b]).setvar(b + ".beta", beta[b])
        last = names[-1]
        c.set_outputs(["CO2_concentration", "global_tas", "veg_c", "soil_c", "permafrost_c", "timesteps",
                       last + ".veg_c", names[0] + ".soil_c", last + ".NPP"])
        c.run(run_to)
        assert (c.status() == 0).all(), nb
        got = {v: c.fetchvars(v, (Y0, run_to)) for v in ("CO2_concentration", "global_tas", "veg_c", "soil_c",
                                                        "permafrost_c", last + ".veg_c", names[0] + ".soil_c",
                                                        last + ".NPP")}
        ts = c.fetchvars("timesteps", (Y0 + 1, run_to))
        nk = run_to - Y0 + 1
        for i in range(n):
            p = oracle.default_params()
            d = {k: getattr(p, k)[0] for k in ("veg_c", "detritus_c", "soil_c", "permafrost_c", "npp_flux0")}
            p.nbiome = nb
            p.S = S[i]
            for j, b in enumerate(names):
                for k in ("f_nppv", "f_nppd", "f_litterd", "rh_ch4_frac", "pf_mu", "pf_sigma", "fpf_static"):
                    getattr(p, k)[j] = getattr(p, k)[0]
                p.veg_c[j] = d["veg_c"] * fr[j]; p.detritus_c[j] = d["detritus_c"] * fr[j]
                p.soil_c[j] = d["soil_c"] * fs[j]; p.permafrost_c[j] = d["permafrost_c"] * fs[j]
                p.npp_flux0[j] = d["npp_flux0"] * fr[j]
                p.q10_rh[j] = q10[b][i]; p.warmingfactor[j] = wf[b]; p.beta[j] = beta[b]
            ref, err, _ = oracle.run(p, run_to=run_to)
            assert err == 0
            rc = ref["CO2_concentration"][:nk]
            assert (np.abs(got["CO2_concentration"][:, i] - rc) / rc).max() < 2e-8, (nb, i)
            assert np.abs(got["global_tas"][:, i] - ref["global_tas"][:nk]).max() < 2e-8, (nb, i)
            for v in ("veg_c", "soil_c", "permafrost_c"):
                assert np.abs(got[v][:, i] - ref[v][:nk]).max() < 2e-8 * np.abs(ref[v]).max(), (nb, v, i)
            assert np.array_equal(ts[:, i], ref["timesteps"][1:nk]), (nb, i)
        assert (got[last + ".veg_c"] > 0).all() and (got[names[0] + ".soil_c"] > 0).all()
        assert np.isfinite(got[last + ".NPP"]).all()
        c.shutdown()


def identical_split_equals_global(lib, nb=7, **kw):
    """test_biome.R: an equal split with identical parameters reproduces the single-biome climate."""
    one = hector_amd.Core(SCENARIO, 2, lib_path=lib, **kw).setvar("S", [2.5, 4.0]).run(2100)
    many = hector_amd.Core(SCENARIO, 2, lib_path=lib, **kw).setvar("S", [2.5, 4.0])
    many.split_biome(["x%d" % i for i in range(nb)])
    many.run(2100)
    for v in ("CO2_concentration", "global_tas"):
        a, b = one.fetchvars(v, (Y0, 2100)), many.fetchvars(v, (Y0, 2100))
        assert np.abs(a - b).max() < 1e-9 * max(1.0, np.abs(a).max()), v


def unrolled_equals_looped(lib, monkeypatch, n, run_to, tol=1e-11, **kw):
    """Five to eight biomes run unrolled kernels (round 3), nine to sixteen the looped ones;
    HECTOR_AMD_LOOPED_BIOMES_FROM=5 sends 5-8 to the looped kernels too: the same ensemble on
    both -- plain, with the heat-flux output and per-member diffusivity, and with a constraint
    (the extended kernels) -- must agree to rounding: the sums over biomes run in the same order,
    only the compiler's contraction of multiply-adds differs (1e-11 on the host build to 1900;
    on the GPU to 2300 the last bits grow to ~1e-10, the same size as against the oracle)."""
    for nb in (5, 6, 7, 8):
        res = []
        for looped in (False, True):
            if looped:
                monkeypatch.setenv("HECTOR_AMD_LOOPED_BIOMES_FROM", "5")
            else:
                monkeypatch.delenv("HECTOR_AMD_LOOPED_BIOMES_FROM", raising=False)
            r = np.random.default_rng(nb)
            c = hector_amd.Core(SCENARIO, n, lib_path=lib, **kw)
            names = ["k%d" % i for i in range(nb)]
            fr = r.random(nb) + 0.3; fr /= fr.sum()
            hector_amd.split_biome(c, "global", names, fveg_c=fr, fdetritus_c=fr, fsoil_c=fr,
                                   fpermafrost_c=fr, fnpp_flux0=fr)
            c.setvar("S", 2.0 + 3.0 * ensemble.uniform01(np.arange(n), 3, seed=nb), "degC")
            c.setvar("diff", 1.5 + 1.5 * ensemble.uniform01(np.arange(n), 4, seed=nb), "cm2/s")
            for i, b in enumerate(names):
                c.setvar(b + ".q10_rh", 1.2 + 1.6 * ensemble.uniform01(np.arange(n), 5 + i, seed=nb))
                c.setvar(b + ".warmingfactor", 0.8 + 0.2 * i)
            outs = ["CO2_concentration", "global_tas", "heatflux", "permafrost_c", names[-1] + ".soil_c", "NPP"]
            c.set_outputs(outs)
            c.run(run_to)
            got = [c.fetchvars(v, (Y0, run_to)) for v in outs]
            # ... and the constrained kernels: global tas pinned from 1850 on
            yrs = np.arange(1850, run_to + 1)
            c.setvar_dated("tas_constrain", yrs, 0.01 * (yrs - 1850), "degC")
            c.reset(Y0); c.run(run_to)
            got += [c.fetchvars(v, (Y0, run_to)) for v in outs]
            assert (c.status() == 0).all()
            res.append(got)
            c.shutdown()
        monkeypatch.delenv("HECTOR_AMD_LOOPED_BIOMES_FROM", raising=False)
        for a, b in zip(*res):
            assert np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), nb


def test_many_biomes_vs_oracle(emul_lib, oracle, monkeypatch):
    many_biome_checks(emul_lib, oracle, counts=(5, 8, 11, 21), n=2, run_to=2100, allow_emulation=True)
    identical_split_equals_global(emul_lib, allow_emulation=True)
    unrolled_equals_looped(emul_lib, monkeypatch, 3, 1900, allow_emulation=True)


def test_many_biomes_api(emul_lib):
    c = hector_amd.Core(SCENARIO, 1, lib_path=emul_lib, allow_emulation=True)
    c.split_biome(["a", "b", "c", "d", "e", "f"])
    c.delete_biome("c")
    c.create_biome("g")
    assert c.biomes() == ["a", "b", "d", "e", "f", "g"]
    c.setvar("g.veg_c", 10.0).setvar("g.soil_c", 50.0).setvar("g.npp_flux0", 1.0)
    c.run(1800)
    assert c.status()[0] == 0
    # carbon tracking on the looped kernels (tests/test_tracking.py holds it to the oracle)
    t = hector_amd.Core(SCENARIO, 1, lib_path=emul_lib, allow_emulation=True)
    t.split_biome(["a", "b", "c", "d", "e"]); t.setvar("trackingDate", [1800.0]); t.run(1810)
    v, f = t.tracking_data(0, (1800, 1810))
    assert v.shape == (11, 31) and abs(f.sum(axis=2) - 1.0).max() < 1e-12
    # the caps: 32 biomes (HX_BDYN); carbon tracking up to 24 (128 pools = two mask words a pool)
    big = hector_amd.Core(SCENARIO, 1, lib_path=emul_lib, allow_emulation=True)
    big.split_biome(["n%02d" % i for i in range(32)])
    with pytest.raises(hector_amd.HectorAmdError, match="at most 32 biomes"):
        big.create_biome("one_more")
    big.run(1760)
    assert big.status()[0] == 0
    big.setvar("trackingDate", [1750.0])
    with pytest.raises(hector_amd.HectorAmdError, match="at most 24 biomes"):
        big.reset(1745); big.run(1760)


@pytest.mark.gpu
def test_many_biomes_vs_oracle_on_gpu(hip_lib, oracle):
    many_biome_checks(hip_lib, oracle, counts=(5, 6, 8, 9, 16, 17, 32), n=6, device=0)
    identical_split_equals_global(hip_lib, device=0)


@pytest.mark.gpu
def test_unrolled_and_looped_kernels_agree_on_gpu(hip_lib, monkeypatch):
    unrolled_equals_looped(hip_lib, monkeypatch, 200, 2300, tol=2e-8, device=0)
