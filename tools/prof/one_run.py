#!/usr/bin/env python3
"""One full 555-year launch of a given build / configuration (a target for rocprofv3):
    python tools/prof/one_run.py [members] [biomes] [lib] [launches]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hector_amd  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = sys.argv[3] if len(sys.argv) > 3 else hector_amd.DEFAULT_LIB
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
orig = hector_amd.Core
hector_amd.Core = lambda *a, **k: orig(*a, **dict(k, lib_path=os.path.abspath(lib)))
c = bench.make_core(n, b, 0, 0)
for _ in range(reps):
    c.reset(1745); c.run(2300)
print("kernel ms", c.last_run_ms())
