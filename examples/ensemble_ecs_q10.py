#!/usr/bin/env python3
"""A perturbed-parameter ensemble the way an R user would write it for one run -- but for all
members at once.  Needs an MI355X:  python examples/ensemble_ecs_q10.py [n_members]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hector_amd                                    # noqa: E402
from hector_amd.capabilities import (ECS, Q10_RH, CONCENTRATIONS_CO2, GLOBAL_TAS, RF_TOTAL,  # noqa: E402
                                     FFI_EMISSIONS)


def main(n=10000, **core_kwargs):
    rng = np.random.default_rng(1)
    core = hector_amd.newcore(None, n_members=n, **core_kwargs)   # packaged SSP2-4.5, or an INI path
    hector_amd.setvar(core, None, ECS(), rng.uniform(1.5, 6.0, n), "degC")
    hector_amd.setvar(core, None, Q10_RH(), rng.uniform(1.0, 3.0, n), "(unitless)")
    core.enable_history(True)                        # so that we can go back to 2020 below
    core.set_outputs([CONCENTRATIONS_CO2(), GLOBAL_TAS(), RF_TOTAL()])
    hector_amd.run(core, 2100)
    tas = hector_amd.fetchvars(core, (2100, 2100), GLOBAL_TAS())[GLOBAL_TAS()][0]
    print("2100 warming: median %.2f K, 5-95%% %.2f-%.2f K  (%d members, %.1f ms on the GPU)"
          % (np.median(tas), *np.percentile(tas, [5, 95]), n, core.last_run_ms()))

    # vignettes/ex_hector_apply.Rmd: halve fossil emissions from 2021 on and re-run from there
    years = np.arange(2021, 2101)
    ffi = hector_amd.fetchvars(core, (2021, 2100), FFI_EMISSIONS())[FFI_EMISSIONS()][:, 0]
    hector_amd.setvar(core, years, FFI_EMISSIONS(), 0.5 * ffi, "Pg C/yr")
    hector_amd.run(core, 2100)                       # resets itself to 2020, like the R wrapper
    tas2 = hector_amd.fetchvars(core, (2100, 2100), GLOBAL_TAS())[GLOBAL_TAS()][0]
    print("with halved fossil emissions after 2020: median %.2f K (%.2f K less)"
          % (np.median(tas2), np.median(tas - tas2)))
    hector_amd.shutdown(core)
    return tas, tas2


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10000)
