"""The counters bench.py quotes must belong to the kernels it runs: profiles/pmc_index.json has
an entry for the current kernel-source hash and every configuration of the bench line (VERDICT
r1: 'fail loudly, rather than printing a stale traffic').  After editing a kernel: one
`tools/prof/collect.sh` on the GPU box + `tools/prof/summarize.py` (about a minute).  Named to
run last."""
import json
import os

from conftest import ROOT


def test_pmc_index_matches_the_kernel_sources():
    import bench
    index = json.load(open(os.path.join(ROOT, "profiles", "pmc_index.json")))
    h = bench.kernel_source_hash()
    assert h in index["entries"], (
        "profiles/pmc_index.json has no counters for kernel source %s: run tools/prof/collect.sh "
        "on the GPU box and tools/prof/summarize.py" % h)
    for cfg in ("65536x1", "1024x1", "131072x1", "65536x4"):
        e = index["entries"][h].get(cfg)
        assert e, "no counters for configuration %s of kernel source %s" % (cfg, h)
        assert e["traffic_bytes_per_launch"] > 0 and e["fp64_flops_per_launch"] > 0
        assert os.path.exists(os.path.join(ROOT, e["source"]))
        assert e.get("compiler") == bench.compiler_id(), (
            "the counters of %s were collected on a build by %r, this tree builds with %r"
            % (cfg, e.get("compiler"), bench.compiler_id()))
