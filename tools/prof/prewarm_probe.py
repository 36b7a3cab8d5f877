#!/usr/bin/env python3
"""Is the slow first run of a fresh core the GPU's clocks?  (tools/prof/first_run.py, cold_run.py)
    python tools/prof/prewarm_probe.py [members]
A fresh core's first run() right after its prepare (status()), and the same with k complete runs of
ANOTHER, warm core (6 ms of full-chip work each) launched right before it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    a = bench.make_core(65536, 1, 0, 0)
    for _ in range(3):
        a.reset(1745); a.run(2300)
    a.status()
    for warm in (0, 1, 3, 0, 1, 3):
        c = bench.make_core(n, 1, 0, 0)
        c.status()
        for _ in range(warm):
            a.reset(1745); a.run(2300)
        c.run(2300)
        first = c.last_run_ms()
        ms = []
        for _ in range(4):
            c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
        print("%d members: %d warm-up run(s) of another core before the first run: first %.3f ms, then %s"
              % (n, warm, first, " ".join("%.3f" % x for x in ms)), flush=True)
        c.shutdown()


if __name__ == "__main__":
    main()
