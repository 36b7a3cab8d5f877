#!/usr/bin/env python3
"""gpurun_out/<tag>_<members>x<biomes>/ (tools/prof/collect.sh) -> profiles/:
   profiles/<tag>_pmc_<members>x<biomes>.json       counters of the 555-year dispatch + derived figures
   profiles/<tag>_kernel_stats_<members>x<biomes>.csv   rocprofv3 --kernel-trace --stats summary
   profiles/pmc_index.json                          what bench.py looks up: kernel-source hash ->
                                                    configuration -> traffic, flops, VALU-active share

    python tools/prof/summarize.py r02 65536x1 65536x4 1024x1

HBM traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB as reported): gfx950's FETCH_SIZE counts half of
coalesced reads; the factor is re-measured in every collection on hx_stats_kernel, which reads
one [556][npad] fp64 array exactly once (MI355X_MICROARCH.md, HBM / rocprofv3 section).
Executed fp64 flops = SQ_INSTS_VALU_FLOPS_FP64 (per wave-instruction: 2 per FMA, 1 per
add / mul / transcendental) x 64 lanes x lane utilisation (SQ_THREAD_CYCLES_VALU /
(64 SQ_ACTIVE_INST_VALU))."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash, compiler_id  # noqa: E402

G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
YEARS = 555


def counters(d, kernel=("hx_run_kernel", "hx_pair_kernel"), name_out=None):
    """counter -> value of the largest dispatch of `kernel` (summed over its XCD rows); small
    one-biome ensembles are run by hx_pair_kernel (two wavefronts per 64 members)."""
    if isinstance(kernel, str):
        kernel = (kernel,)
    acc = {}
    path = os.path.join(d, "p_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if any(kn in r["Kernel_Name"] for kn in kernel):
            if name_out is not None:
                name_out.add("hx_pair_kernel" if "hx_pair_kernel" in r["Kernel_Name"] else "hx_run_kernel")
            k = (r["Dispatch_Id"], r["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    best = {}
    for (_, cn), v in acc.items():
        best[cn] = max(best.get(cn, 0.0), v)
    return best


def main():
    tag = sys.argv[1]
    index_path = os.path.join(P, "pmc_index.json")
    index = json.load(open(index_path)) if os.path.exists(index_path) else {"entries": {}, "order": []}
    h = kernel_source_hash()
    if h in index["order"]:
        index["order"].remove(h)
    index["order"].append(h)  # order of collection, newest last
    for cfg in sys.argv[2:]:
        members, biomes = (int(x) for x in cfg.split("x"))
        d = os.path.join(G, "%s_%s" % (tag, cfg))
        c = {}
        names = set()
        for sub in ("fetch", "write", "sq_a", "sq_b", "sq_c"):
            c.update(counters(os.path.join(d, sub), name_out=names))
        kname = "hx_pair_kernel" if names == {"hx_pair_kernel"} else "hx_run_kernel<%d,...>" % biomes
        cal = counters(os.path.join(d, "fetch"), "hx_stats_kernel").get("FETCH_SIZE")
        npad = (members + 63) // 64 * 64
        known_kib = 556 * npad * 8 / 1024.0
        factor = known_kib / cal if cal else None
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        waves = c["SQ_WAVES"]
        lane_util = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
        flops = c["SQ_INSTS_VALU_FLOPS_FP64"] * 64.0 * lane_util
        # DOECLIM history contraction on the matrix pipe: one v_mfma_f64_16x16x4_f64 = 4 MOPS of
        # 512 flops (2 x 16 x 16 x 4); not part of the VALU figure
        mfma_flops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
        bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
        der = {
            "VALU_insts_per_wave_year": c["SQ_INSTS_VALU"] / waves / YEARS,
            "SALU_insts_per_wave_year": c["SQ_INSTS_SALU"] / waves / YEARS,
            "fp64_FMA_per_wave_year": c["SQ_INSTS_VALU_FMA_F64"] / waves / YEARS,
            "fp64_ADD_per_wave_year": c["SQ_INSTS_VALU_ADD_F64"] / waves / YEARS,
            "fp64_MUL_per_wave_year": c["SQ_INSTS_VALU_MUL_F64"] / waves / YEARS,
            "fp64_TRANS_per_wave_year": c["SQ_INSTS_VALU_TRANS_F64"] / waves / YEARS,
            "branches_per_wave_year": c["SQ_INSTS_BRANCH"] / waves / YEARS,
            "VMEM_reads_per_wave_year": c["SQ_INSTS_VMEM_RD"] / waves / YEARS,
            "LDS_insts_per_wave_year": c["SQ_INSTS_LDS"] / waves / YEARS,
            "lane_utilisation": lane_util,
            "valu_active_frac": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
            "any_active_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            "wait_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "mean_wave_cycles_per_year": 4 * c["SQ_WAVE_CYCLES"] / waves / YEARS,
            "executed_fp64_flops_per_launch": flops,
            "executed_fp64_mfma_flops_per_launch": mfma_flops,
            "executed_fp64_flops_per_member_year": flops / (members * YEARS),
            "hbm_traffic_bytes_per_launch": traffic,
            "fetch_size_correction_measured": factor,
        }
        # the bench line was printed BEFORE this collection was reduced (the run could only find an
        # older profile): its counter-derived figures are re-derived here from this collection
        if isinstance(bench, dict) and isinstance(bench.get("roofline"), dict) and bench["roofline"].get("kernel_ms"):
            r = bench["roofline"]
            secs = r["kernel_ms"] * 1e-3
            r.update({"achieved": flops / secs / 1e12, "frac": flops / secs / 1e12 / r["peak"],
                      "fp64_flops_per_launch": flops, "fp64_mfma_flops_per_launch": mfma_flops,
                      "fp64_mfma_tflops": mfma_flops / secs / 1e12,
                      "valu_active_frac": der["valu_active_frac"], "traffic": traffic,
                      "hbm_measured_frac": traffic / secs / 8.0e12,
                      "pmc_profile": "profiles/%s_pmc_%s.json" % (tag, cfg), "pmc_profile_stale": False,
                      "kernel_source_hash": h,
                      "rederived": "counter figures of this line re-derived from this collection by "
                                   "tools/prof/summarize.py (the run itself preceded it)"})
        json.dump({"kernel": kname, "kernel_source_hash": h,
                   "workload": "%d members x 555 years, %d biome(s), one launch" % (members, biomes),
                   "counters_of_the_555_year_dispatch": c, "derived": der,
                   "bench_line_same_build": bench,
                   "note": "SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count units of 4 clocks; "
                           "traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB, factor 2 re-measured on "
                           "hx_stats_kernel in the same pass (fetch_size_correction_measured)"},
                  open(os.path.join(P, "%s_pmc_%s.json" % (tag, cfg)), "w"), indent=1)
        st = os.path.join(d, "stats", "s_kernel_stats.csv")
        if os.path.exists(st):
            shutil.copy(st, os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, cfg)))
        index["entries"].setdefault(h, {})[cfg] = {
            "traffic_bytes_per_launch": traffic,
            "fp64_flops_per_launch": flops,
            "fp64_mfma_flops_per_launch": mfma_flops,
            "valu_active_frac": der["valu_active_frac"],
            "lane_utilisation": lane_util,
            "source": "profiles/%s_pmc_%s.json" % (tag, cfg),
            "compiler": compiler_id(),   # bench.py refuses the figures for another compiler's build
        }
        print(cfg, json.dumps(der, indent=1))
    json.dump(index, open(index_path, "w"), indent=1, sort_keys=True)
    # The default bench line of the same call (tools/prof/gpu_round.sh) was printed on the GPU box
    # before this collection existed there: its counter-derived figures (headline roofline and the
    # other configurations' fractions) are re-derived from the entries just written.
    line_path = os.path.join(G, "%s_bench_default_line.json" % tag)
    if os.path.exists(line_path):
        try:
            line = json.load(open(line_path))
        except ValueError:
            line = None
        ent = index["entries"].get(h, {})
        if isinstance(line, dict):
            def refresh(r, cfg, secs):
                e = ent.get(cfg)
                if not e:
                    return False
                fl = e["fp64_flops_per_launch"]
                r.update({"achieved": fl / secs / 1e12, "frac": fl / secs / 1e12 / 78.6,
                          "fp64_flops_per_launch": fl,
                          "fp64_mfma_flops_per_launch": e["fp64_mfma_flops_per_launch"],
                          "fp64_mfma_tflops": e["fp64_mfma_flops_per_launch"] / secs / 1e12,
                          "valu_active_frac": e["valu_active_frac"],
                          "traffic": e["traffic_bytes_per_launch"],
                          "hbm_measured_frac": e["traffic_bytes_per_launch"] / secs / 8.0e12,
                          "pmc_profile": e["source"], "pmc_profile_stale": False,
                          "kernel_source_hash": h})
                return True
            cfgm = line.get("config", {})
            r = line.get("roofline")
            if isinstance(r, dict) and r.get("kernel_ms"):
                cfg = "%dx%d" % (cfgm.get("members_per_gpu", 0), cfgm.get("biomes", 1))
                if refresh(r, cfg, r["kernel_ms"] * 1e-3):
                    r["rederived"] = ("counter figures re-derived from the counter passes of the same gpurun "
                                      "call by tools/prof/summarize.py (the line itself preceded them)")
            for o in line.get("other_configs", []) or []:
                e = ent.get("%dx%d" % (o.get("members", 0), o.get("biomes", 1)))
                if e and o.get("kernel_ms"):
                    o["fp64_valu_frac"] = e["fp64_flops_per_launch"] / (o["kernel_ms"] * 1e-3) / 78.6e12
                    o["pmc_profile_stale"] = False
            json.dump(line, open(os.path.join(P, "%s_bench_default_line.json" % tag), "w"))


if __name__ == "__main__":
    main()
