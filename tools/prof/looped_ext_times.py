import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import hector_amd
from hector_amd import ensemble
lib = os.path.abspath(sys.argv[1])
n = 65536
S = 1.5 + 4.5 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 0)
for nb in (9, 12):
    for ext in (0, 1, 2):
        c = hector_amd.Core(n_members=n, device=0, lib_path=lib)
        names = ["b%d" % i for i in range(nb)]
        c.split_biome(names)
        c.setvar("S", S, "degC")
        for b, nm in enumerate(names):
            c.setvar(nm + ".q10_rh", 1.0 + 2.0 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 10 + b))
            c.setvar(nm + ".warmingfactor", np.full(n, 1.0 + 0.5 * (b % 4)))
        outs = ["CO2_concentration", "global_tas"] + (["NPP"] if ext else [])
        c.set_outputs(outs)
        if ext == 2:
            yrs = np.arange(1850, 2101)
            c.setvar_dated("tas_constrain", yrs, 0.012 * (yrs - 1850), "degC")
        ms = []
        for _ in range(3):
            c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
        print("%s: %d biomes, %s: best %.3f ms  co2 %.9f bad %d" % (os.path.basename(lib), nb, ("plain", "NPP", "NPP + tas constraint")[ext], min(ms[1:]),
              c.fetchvars("CO2_concentration", (2300, 2300))[0].mean(), int((c.status() != 0).sum())), flush=True)
        c.shutdown()
