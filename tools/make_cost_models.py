#!/usr/bin/env python3
"""hector_amd/data/cost_models.txt -- the lane-cost models shipped with the scenarios.

A one-shot run (fresh process, one core) has no measured costs and no earlier core to learn from;
where the lane order matters -- more wavefronts than SIMDs, e.g. the 131 072 members one GPU holds
of BASELINE configs[3] -- the library orders the lanes by a model  cost ~ quadratic in the
standardised perturbed parameters  (EnsembleCore::fit_cost_model).  This script makes the models
a fresh process finds: for every shipped scenario it runs the perturbed ECS x Q10 ensemble of
SURVEY.md 8(d) (and, for SSP2-4.5, the four-biome ensemble of configs[4]) ONCE on the GPU with
the product library, lets the core fit its model to the per-member solver work the run kernel
counted (4 x dopri5 steps + 5 x stashes), and writes the process's registry through
hx_cost_models_export.  Runs on an MI355X (no CPU path):

    python tools/make_cost_models.py [members]        # default 16 384
"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HECTOR_AMD_COST_MODELS"] = ""            # start from an empty registry
os.environ["HECTOR_AMD_CALIBRATE_ALWAYS"] = "1"      # (fit even where the order would not matter)
import bench  # noqa: E402
import hector_amd  # noqa: E402
from hector_amd import core as core_mod, ensemble  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    out = os.path.join(ROOT, "hector_amd", "data", "cost_models.txt")
    scenarios = sorted(glob.glob(os.path.join(ROOT, "hector_amd", "data", "*.hxs")))
    for sc in scenarios:
        S, q10 = ensemble.ecs_q10(n)
        c = hector_amd.Core(sc, n_members=n, device=0)
        c.set_pair_kernel_limit(0)
        c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
        c.run(c.enddate)
        c.reset(c.strtdate)          # the complete run's costs -> the model
        bad = int((c.status() != 0).sum())
        print("%-28s ECS x Q10, %d members: kernel %.2f ms, %d members with model errors"
              % (os.path.basename(sc), n, c.last_run_ms(), bad), flush=True)
        c.shutdown()
    c = bench.make_core(n, 4, 0, 0)   # SSP2-4.5, four biomes (BASELINE configs[4])
    c.set_pair_kernel_limit(0)
    c.run(c.enddate); c.reset(c.strtdate); c.status()
    print("ssp245 four-biome ensemble, %d members: kernel %.2f ms" % (n, c.last_run_ms()), flush=True)
    c.shutdown()
    k = core_mod.cost_models_export(out)
    print("%d model(s) -> %s" % (k, out))


if __name__ == "__main__":
    main()
