import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import hector_amd
from hector_amd import ensemble
SCEN = os.path.join(os.getcwd(), "hector_amd", "data", "ssp245.hxs")
lib = os.path.join(os.getcwd(), "gpuwork", "libpairclk.so")
os.environ["HECTOR_AMD_PAIR_MAX_MEMBERS"] = str(1 << 30)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S, q10 = ensemble.ecs_q10(n)
c = hector_amd.Core(SCEN, n, device=0, lib_path=lib)
c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)")
c.run(2300)
print("ms", c.last_run_ms())
t = c.fetchvars("global_tas", (1746, 1746 + 45))[:, ::64]
O = ["consts (both boxes)", "wait A0", "solve HL", "wait A", "interval consts", "step loop", "stash", "wait S1", "post-stash + solve", "wait S2", "year end (forcing, DOECLIM)", "wait C", "f:attempt", "f:wait step", "f:control"]
L = ["Tland factors + flows", "wait A0", "solve LL", "wait A", "(setup)", "step loop", "stash", "wait S1", "post-stash + solve", "wait S2", "next year's gas + sums", "wait C", "f:attempt", "f:wait step", "f:control"]
tot = t[:23].sum(0)
slow = int(np.argmax(tot))
print("ocean: total/yr mean %.0f  slowest wg %.0f" % (tot.mean() / 555, tot[slow] / 555))
for k, nm in enumerate(O): print("  %-30s %8.0f  slowest %8.0f" % (nm, t[k].mean() / 555, t[k, slow] / 555))
print("land: total/yr mean %.0f" % (t[23:46].sum(0).mean() / 555))
for k, nm in enumerate(L): print("  %-30s %8.0f  slowest %8.0f" % (nm, t[23 + k].mean() / 555, t[23 + k, slow] / 555))
