#!/usr/bin/env python3
"""Quick parity of an experiment build's small-ensemble kernel against the oracle:
    python tools/prof/pair_check.py gpuwork/libX.so [members]
CO2 / Tgav of every 16th member and the stash counts per year, 1745-2300."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hector_amd  # noqa: E402
from hector_amd import ensemble  # noqa: E402
import oracle_binding  # noqa: E402


def main():
    lib = os.path.abspath(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(n_members=n, device=0, lib_path=lib)
    c.set_pair_kernel_limit(32768)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
    c.set_outputs(["CO2_concentration", "global_tas", "timesteps", "soil_c", "veg_c", "NBP"])
    c.run(2300)
    assert c.last_run_kernel() == "pair"
    sel = np.arange(0, n, 16)
    orc = oracle_binding.Oracle(hector_amd.DEFAULT_SCENARIO)
    co2 = c.fetchvars("CO2_concentration", (1745, 2300)); tg = c.fetchvars("global_tas", (1745, 2300))
    ts = c.fetchvars("timesteps", (1746, 2300)); soil = c.fetchvars("soil_c", (1745, 2300))
    worst = [0, 0, 0, 0]
    for i in sel:
        p = orc.default_params(); p.S = S[i]; p.q10_rh[0] = q10[i]
        o, err, _ = orc.run(p)
        worst[0] = max(worst[0], (np.abs(co2[:, i] - o["CO2_concentration"]) / o["CO2_concentration"]).max())
        worst[1] = max(worst[1], np.abs(tg[:, i] - o["global_tas"]).max())
        worst[2] += int((ts[:, i] != o["timesteps"][1:]).sum())
        worst[3] = max(worst[3], (np.abs(soil[1:, i] - o["soil_c"][1:]) / o["soil_c"][1:]).max())
    print("%s: %d members on the pair kernel, %.3f ms; vs oracle (%d members): max rel dCO2 %.2e, max |dTgav| %.2e, "
          "rel dsoil %.2e, years with another stash count %d, bad %d"
          % (os.path.basename(lib), n, c.last_run_ms(), len(sel), worst[0], worst[1], worst[3], worst[2],
             int((c.status() != 0).sum())), flush=True)


if __name__ == "__main__":
    main()
