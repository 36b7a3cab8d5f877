#!/usr/bin/env python3
"""Regression vectors of the ORACLE (oracle/hector_oracle.c), not of the reference: the reference
cannot be built in this image (Boost), so SURVEY 8(c)'s fixtures (ii)-(iv) are pinned here from
the oracle, whose default member is itself pinned to the reference's hector_comp.csv.  They
guard against silent drift of the oracle AND the kernels (tests/test_regression_vectors.py).

  tests/golden/oracle_vectors.npz
    ecs_q10: 64 members (S, q10_rh from hector_amd.ensemble.ecs_q10): CO2, Tgav, stashes per year
    biome4:  16 members, 4 biomes (ensemble.biome4): CO2, Tgav, permafrost_c
    csys:    200 carbonate solves (T, carbon, alk) -> pCO2, pH, Tr
    ker:     DOECLIM kernel table for diff = 1.042 and 2.3
"""
import os, sys
import numpy as np
ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_binding
from hector_amd import ensemble
SC = os.path.join(ROOT, "hector_amd", "data", "ssp245.hxs")
o = oracle_binding.Oracle(SC)
out = {}
n = 64
S, q10 = ensemble.ecs_q10(n)
co2 = np.zeros((n, o.ns)); tg = np.zeros((n, o.ns)); st = np.zeros((n, o.ns), dtype=np.int8)
for i in range(n):
    p = o.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
    r, err, _ = o.run(p); assert err == 0
    co2[i] = r["CO2_concentration"]; tg[i] = r["global_tas"]; st[i] = r["timesteps"].astype(np.int8)
out.update(ecs_S=S, ecs_q10=q10, ecs_co2=co2, ecs_tgav=tg, ecs_stashes=st)
n = 16
S, q10s, wfs = ensemble.biome4(n)
co2 = np.zeros((n, o.ns)); tg = np.zeros((n, o.ns)); pf = np.zeros((n, o.ns))
for i in range(n):
    p = o.split_equal(o.default_params(), 4); p.S = S[i]
    for b in range(4): p.q10_rh[b] = q10s[b][i]; p.warmingfactor[b] = wfs[b][i]
    r, err, _ = o.run(p); assert err == 0
    co2[i] = r["CO2_concentration"]; tg[i] = r["global_tas"]; pf[i] = r["permafrost_c"]
out.update(b4_S=S, b4_q10=np.array(q10s), b4_wf=np.array(wfs), b4_co2=co2, b4_tgav=tg, b4_permafrost=pf)
rng = np.random.default_rng(20260928)
m = 200; vol = 3.6e14 * 0.85 * 100.0
Tc = rng.uniform(-1.0, 30.0, m); carbon = rng.uniform(600.0, 1000.0, m); alk = rng.uniform(2200e-6, 2750e-6, m)
cs = np.array([o.csys(Tc[i], carbon[i], alk[i], vol)[:3] for i in range(m)])
assert (cs[:, 0] > 0).all()
out.update(csys_T=Tc, csys_carbon=carbon, csys_alk=alk, csys_vol=np.array(vol), csys_out=cs)
out.update(ker_diff=np.array([1.042, 2.3]), ker=np.array([o.doeclim_kernel(d, 556) for d in (1.042, 2.3)]))
dst = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
