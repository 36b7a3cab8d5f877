#!/usr/bin/env python3
"""The one-shot run of a large ensemble with and without the fitted cost model:
    python tools/prof/first_run.py [members]
A 65 536-member core runs once (its measured costs make the model: ensemble_core.cpp,
fit_cost_model); then fresh cores of `members` (default 131 072) members run for the first time --
lanes by the parameter key (HECTOR_AMD_COST_MODEL=0), by the model, and, after their own complete
run, by their measured costs.  Kernel time by HIP events; first = the cold first launch, steady =
best of 4 right behind each other (the first launch after an idle gap runs at ramping clocks:
tools/prof/cold_run.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def steady(c, reps=4):
    ms = []
    for _ in range(reps):
        c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
    return min(ms)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    a = bench.make_core(65536, 1, 0, 0)
    a.run(2300); a.reset(1745); a.status()
    print("65536-member core: first %.3f ms, steady %.3f ms (%s)" % (a.last_run_ms(), steady(a), a.lane_order_source()), flush=True)
    for model in (0, 1, 0, 1):
        c = bench.make_core(n, 1, 0, 0)
        c.set_cost_model(bool(model))
        c.set_lane_calibration(True)
        c.status()
        src = c.lane_order_source()
        c.run(2300)
        first = c.last_run_ms()
        c.set_lane_calibration(False)   # (keep this order for the steady figure)
        st = steady(c)
        c.set_lane_calibration(True)
        c.reset(1745); c.run(2300); c.reset(1745); c.status()
        src2 = c.lane_order_source()
        st2 = steady(c)
        print("%d members, model %d: lanes by %-13s first run %.3f ms, steady %.3f ms | then by %s: steady %.3f ms"
              % (n, model, src, first, st, src2, st2), flush=True)
        c.shutdown()


if __name__ == "__main__":
    main()
