#!/usr/bin/env python3
"""The tail of a year-loop launch, wavefront by wavefront (hx_wave_clock: s_memrealtime stamps of
the shipped kernels, no profiling build):
    python tools/prof/wave_tail.py [members[xbiomes] ...] [--json=path]
For every configuration: one cold first run, then the steady state with the lanes in parameter-key
order, with HECTOR_AMD_KEY_ORDER variants where two wavefronts share a SIMD, and with the
measured-cost order; per run the kernel's HIP-event time and, from the stamps, when the wavefronts
started, how long they ran (min / mean / max; max over mean = the tail) and a histogram."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

TICK_MS = 1e-5   # 100 MHz


def facts(c):
    w = c.wave_clock().astype(np.float64) * TICK_MS
    start, end = w[:, 0], w[:, 1]
    dur = end - start
    hist, edges = np.histogram(dur, bins=12)
    return {"kernel_ms": c.last_run_ms(), "kernel": c.last_run_kernel(), "waves": int(len(w)),
            "start_ms": {"min": float(start.min()), "p50": float(np.median(start)), "max": float(start.max())},
            "end_ms": {"min": float(end.min()), "mean": float(end.mean()), "max": float(end.max())},
            "wave_ms": {"min": float(dur.min()), "mean": float(dur.mean()), "max": float(dur.max())},
            "max_over_mean": float(dur.max() / dur.mean()),
            "span_ms": float(end.max() - start.min()),
            "histogram_wave_ms": {"edges": [round(float(e), 3) for e in edges], "counts": [int(h) for h in hist]}}


def steady(c, reps=4):
    best = None
    for _ in range(reps):
        c.reset(1745); c.run(2300)
        f = facts(c)
        if best is None or f["kernel_ms"] < best["kernel_ms"]:
            best = f
    return best


def main():
    cfgs, out_path = [], None
    for a in sys.argv[1:]:
        if a.startswith("--json="):
            out_path = a.split("=", 1)[1]
        else:
            m, _, b = a.partition("x")
            cfgs.append((int(m), int(b or 1)))
    cfgs = cfgs or [(65536, 1), (131072, 1)]
    res = {}
    for n, b in cfgs:
        r = {}
        modes = [0] + ([1, 2] if n > 65536 and b == 1 else [])
        for mode in modes:
            os.environ["HECTOR_AMD_KEY_ORDER"] = str(mode)
            c = bench.make_core(n, b, 0, 0)
            c.set_lane_calibration(False)
            c.status()
            c.run(2300)
            tag = "key_order_%d" % mode
            r[tag + "_first_run"] = facts(c)
            r[tag + "_steady"] = steady(c)
            if mode == 0:
                c.set_lane_calibration(True)
                c.reset(1745); c.status()
                if c.lanes_calibrated():
                    r["measured_cost_steady"] = steady(c)
            c.shutdown()
        os.environ.pop("HECTOR_AMD_KEY_ORDER", None)
        res["%dx%d" % (n, b)] = r
        for k, v in r.items():
            print("%8dx%d %-24s %-5s kernel %7.3f ms  waves %5d  wave ms min/mean/max %6.3f %6.3f %6.3f  max/mean %.3f  "
                  "starts <= %.3f ms" % (n, b, k, v["kernel"], v["kernel_ms"], v["waves"], v["wave_ms"]["min"],
                                         v["wave_ms"]["mean"], v["wave_ms"]["max"], v["max_over_mean"],
                                         v["start_ms"]["max"]), flush=True)
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
