#!/usr/bin/env python3
"""How much busy time ahead of a fresh core's first run brings it to the steady state?
    python tools/prof/prewarm_curve.py [members]
Fresh cores of `members` members (lanes by the shipped cost model, calibration off so that the
order stays), prepared (status()), then k back-to-back runs of ANOTHER warm core (65 536 members,
~6 ms of full-chip work each, launched without waiting), then the first run; steady = best of 6."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    a = bench.make_core(65536, 1, 0, 0)
    a.set_lane_calibration(False)
    for _ in range(3):
        a.reset(1745); a.run(2300)
    a.status()
    for warm in (0, 2, 5, 10, 20, 40, 0, 10, 40):
        c = bench.make_core(n, 1, 0, 0)
        c.set_lane_calibration(False)
        c.status()
        src = c.lane_order_source()
        for _ in range(warm):
            a.reset(1745); a.run(2300, wait=False)
        a.sync()
        c.run(2300)
        first = c.last_run_ms()
        ms = []
        for _ in range(6):
            c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
        print("%d members (%s): %2d warm-up runs (~%3d ms busy) -> first %.3f ms; then %s; first / best %.3f"
              % (n, src, warm, 6 * warm, first, " ".join("%.3f" % x for x in ms), first / min(ms)), flush=True)
        c.shutdown()


if __name__ == "__main__":
    main()
