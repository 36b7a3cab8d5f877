#!/usr/bin/env python3
"""Why is the first run of a fresh core slower than its later ones?
    python tools/prof/cold_run.py [members ...]
Kernel time (HIP events) of: a fresh core's run() with everything queued back to back (one-shot);
its steady state; a run after the GPU idled 0.3 s; a second fresh core prepared first (status())
and run after an idle gap / right behind another core's run (GPU busy until the launch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [65536, 131072]
    for n in sizes:
        c = bench.make_core(n, 1, 0, 0)
        c.set_lane_calibration(False)
        c.run(2300)
        one_shot = c.last_run_ms()
        st = []
        for _ in range(4):
            c.reset(1745); c.run(2300); st.append(c.last_run_ms())
        c.reset(1745); c.sync(); time.sleep(0.3); c.run(2300)
        idle = c.last_run_ms()
        c.reset(1745); c.run(2300)
        after = c.last_run_ms()
        c2 = bench.make_core(n, 1, 0, 0)
        c2.set_lane_calibration(False)
        c2.status(); time.sleep(0.3)
        c2.run(2300)
        fresh_idle = c2.last_run_ms()
        c3 = bench.make_core(n, 1, 0, 0)
        c3.set_lane_calibration(False)
        c3.status()
        c.reset(1745); c.run(2300, wait=False)     # the GPU is busy right up to c3's launch
        c3.run(2300)
        fresh_hot = c3.last_run_ms()
        c3.reset(1745); c3.run(2300)
        fresh_second = c3.last_run_ms()
        print("%7d members (%s): one-shot run() of a fresh core %.3f ms | steady %.3f (best of 4: %s) | after 0.3 s idle %.3f, "
              "then %.3f | fresh core, prepared, idle, run %.3f | fresh core, prepared, run behind another core's launch %.3f, "
              "its second run %.3f" % (n, c.last_run_kernel(), one_shot, min(st), " ".join("%.3f" % x for x in st), idle, after,
                                       fresh_idle, fresh_hot, fresh_second), flush=True)
        c.shutdown(); c2.shutdown(); c3.shutdown()


if __name__ == "__main__":
    main()
