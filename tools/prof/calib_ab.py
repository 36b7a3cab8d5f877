#!/usr/bin/env python3
"""Kernel time with the lane order by parameter key vs by measured cost, same core, same call:
    python tools/prof/calib_ab.py [65536x1 131072x1 ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfgs = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(65536, 1), (131072, 1), (262144, 1), (65536, 4)]
    for n, b in cfgs:
        c = bench.make_core(n, b, 0, 0)
        c.set_lane_calibration(False)
        ms = []
        for _ in range(4):
            c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
        key = min(ms[1:])
        c.set_lane_calibration(True)
        c.reset(1745)          # adopts the measured order (the last run was complete)
        assert c.lanes_calibrated()
        ms = []
        for _ in range(4):
            c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
        print("%7dx%d  parameter key %8.3f ms   measured cost %8.3f ms   (%.1f %%)" %
              (n, b, key, min(ms[1:]), 100 * (min(ms[1:]) / key - 1)), flush=True)
        c.shutdown()


if __name__ == "__main__":
    main()
