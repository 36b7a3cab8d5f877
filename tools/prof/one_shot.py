#!/usr/bin/env python3
"""What a genuine one-shot run costs: a FRESH process creates one core, runs it once and reports
the run kernel's time (HIP events), which order its lanes had and whether the launch was behind
the prewarm loop -- for each setting of HECTOR_AMD_PREWARM_MS / HECTOR_AMD_COST_MODEL given --
next to the steady state of the same ensemble (best of 6 back-to-back runs of a warm core).
    python tools/prof/one_shot.py [members ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import bench
n = int(sys.argv[1])
t0 = time.perf_counter()
c = bench.make_core(n, 1, 0, 0)
c.run(2300)
first = c.last_run_ms()
wall = time.perf_counter() - t0
out = {"first_ms": first, "order": c.lane_order_source(), "prewarmed": c.last_run_prewarmed(),
       "kernel": c.last_run_kernel(), "wall_newcore_to_end_ms": wall * 1e3}
if os.environ.get("ONE_SHOT_STEADY"):
    c.set_lane_calibration(False)
    ms = []
    for _ in range(6):
        c.reset(1745); c.run(2300); ms.append(c.last_run_ms())
    out["steady_same_order_ms"] = min(ms)
print(json.dumps(out))
''' % ROOT


def child(n, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD, str(n)], env=e, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [131072, 65536]
    for n in sizes:
        for rep in range(2):
            for pw in (0, 20, 40):
                for cm in ((1,) if n <= 65536 else (1, 0)):
                    r = child(n, HECTOR_AMD_PREWARM_MS=pw, HECTOR_AMD_COST_MODEL=cm, ONE_SHOT_STEADY=1)
                    print("%7d members, prewarm %2d ms, cost model %d: %s" % (n, pw, cm, json.dumps(r)), flush=True)


if __name__ == "__main__":
    main()
