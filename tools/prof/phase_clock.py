#!/usr/bin/env python3
"""Where a wavefront's time goes: run the -DHX_PHASE_CLOCK build of the library
(tools/prof/build_variant.sh clk -DHX_PHASE_CLOCK) and print s_memtime ticks per section per
model year -- mean over the wavefronts and for the slowest one.

    python tools/prof/phase_clock.py [members] [biomes] [lib] > gpurun_out/phase_clock.json
The stamps themselves cost ~10-15 % (they serialise the schedule and go through LDS), so the
shares matter, not the absolute figures."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SECTIONS = {0: "A: loop top", 1: "A: park reads + OH/CH4/O3", 2: "A: chem constants (2 boxes)",
            3: "A: year-start carbonate solve", 4: "A: slow parameters",
            5: "B: first interval constants", 6: "B: dopri5 attempts",
            7: "B: stash - park constants + carbonate solve", 8: "B: stash - boxes + land pools",
            9: "B: stash - next interval constants", 10: "B: loop control / idle segments",
            11: "C: DOECLIM history pass", 12: "C: loads + forcing",
            13: "C: DOECLIM in-block sum + step", 14: "C: outputs"}


def main():
    ext = "--ext" in sys.argv     # record NPP: the extended kernel (build with -DHX_MINIMAL_EXT)
    if ext:
        sys.argv.remove("--ext")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    biomes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpuwork", "libclk.so")
    import hector_amd
    orig = hector_amd.Core

    def core_with_lib(*a, **k):
        k.setdefault("lib_path", lib)
        return orig(*a, **k)
    hector_amd.Core = core_with_lib
    if biomes in (1, 4):
        c = bench.make_core(n, biomes, 0, 0)
    else:   # any other count: the ensemble of tools/prof/biome_times.py (S, per-biome Q10, warming factors)
        from hector_amd import ensemble
        c = hector_amd.Core(n_members=n, device=0)
        names = ["b%d" % i for i in range(biomes)]
        c.split_biome(names)
        c.setvar("S", 1.5 + 4.5 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 0), "degC")
        for b, nm in enumerate(names):
            c.setvar(nm + ".q10_rh", 1.0 + 2.0 * ensemble.uniform01(np.arange(n, dtype=np.uint64), 10 + b))
            c.setvar(nm + ".warmingfactor", np.full(n, 1.0 + 0.5 * (b % 4)))
    if ext:
        c.set_pair_kernel_limit(0)
        c.set_outputs(["CO2_concentration", "global_tas", "NPP"])
    c.run(2300)
    ms = c.last_run_ms()
    t = c.fetchvars("global_tas", (1745, 1745 + 23))      # [24][members]: per-wave values
    lane = c.lane_of_member()
    inv = np.argsort(lane)
    t = t[:, inv][:, ::64]                                  # one column per wavefront
    tot = t[:15].sum(0)
    slow = int(np.argmax(tot))
    out = {"members": n, "biomes": biomes, "kernel_ms_with_stamps": ms, "years": 555,
           "carbonate_newton_iterations_per_year": {"mean": float(t[18].mean() / 555), "slowest": float(t[18, slow] / 555)},
           "carbonate_safeguarded_restarts_per_year": {"mean": float(t[19].mean() / 555), "slowest": float(t[19, slow] / 555)},
           "ticks_per_year_mean_total": float(tot.mean() / 555),
           "ticks_per_year_slowest_total": float(tot[slow] / 555),
           "step_loop_iterations_per_year": {"mean": float(t[16].mean() / 555), "slowest": float(t[16, slow] / 555)},
           "segments_per_year": {"mean": float(t[17].mean() / 555), "slowest": float(t[17, slow] / 555)},
           "sections": {}}
    for k, name in SECTIONS.items():
        out["sections"][name] = {"mean_ticks_per_year": float(t[k].mean() / 555),
                                 "share_of_mean": float(t[k].mean() / tot.mean()),
                                 "slowest_wave_ticks_per_year": float(t[k, slow] / 555)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
