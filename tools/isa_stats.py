#!/usr/bin/env python3
"""Static figures of the kernels in the compiler's gfx950 assembly (hector_amd/build/*.s):
registers, scratch, LDS and an instruction histogram per kernel -- what the register allocator
made of an edit, without a GPU.

    python tools/isa_stats.py [file.s] [--kernel SUBSTRING] [--loops]

--loops: also the histogram of every innermost-loop-ish block range (label to backward branch).
"""
import collections
import re
import sys

CLASSES = [
    ("fp64", re.compile(r"^v_(fma|fmac|add|mul|max|min)_f64")),
    ("fp64-trans", re.compile(r"^v_(rcp|rsq|sqrt|ldexp|frexp_mant|frexp_exp_i32|rndne|trunc|floor|fract|div_\w+)_f64")),
    ("cvt", re.compile(r"^v_cvt_")),
    ("accvgpr", re.compile(r"^v_accvgpr_")),
    ("v_mov", re.compile(r"^v_mov_")),
    ("cndmask", re.compile(r"^v_cndmask_")),
    ("v_cmp", re.compile(r"^v_cmpx?_")),
    ("mfma", re.compile(r"^v_mfma_")),
    ("lane", re.compile(r"^v_(readlane|writelane|readfirstlane)_")),
    ("valu-other", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("vmem", re.compile(r"^(global|flat|buffer)_")),
    ("scratch", re.compile(r"^scratch_")),
    ("smem", re.compile(r"^s_(load|buffer_load)_")),
    ("waitcnt", re.compile(r"^s_waitcnt")),
    ("branch", re.compile(r"^s_(cbranch|branch)")),
    ("salu", re.compile(r"^s_")),
]


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other"


def parse(path):
    """-> ({kernel: [(opcode | "label", text)]}, {kernel: metadata dict})"""
    kernels, meta = collections.OrderedDict(), {}
    cur, in_meta, entry = None, False, None
    name_rx = re.compile(r"^(_Z\w+):")
    key_rx = re.compile(r"^  (?:- | ) \.(\w+):\s*(\S+)")
    for line in open(path, errors="replace"):
        if line.startswith("amdhsa.kernels:"):
            in_meta, cur = True, None
            continue
        if in_meta:
            if line.startswith("amdhsa.") or line.startswith("..."):
                in_meta = False
                continue
            if line.startswith("  - "):
                entry = {}
            m = key_rx.match(line)
            if m and entry is not None:
                entry[m.group(1)] = m.group(2)
                if m.group(1) == "name":
                    meta[m.group(2)] = entry
            continue
        m = name_rx.match(line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        s = line.strip()
        if s.startswith(".section") or s.startswith(".end_amdhsa_kernel"):
            cur = None
            continue
        if cur is None or not s or s.startswith(";"):
            continue
        if s.startswith(".LBB"):
            kernels[cur].append(("label", s.split(":")[0]))
            continue
        op = s.split()[0]
        if re.match(r"^[vs]_|^ds_|^global_|^flat_|^buffer_|^scratch_", op):
            kernels[cur].append((op, s))
    return kernels, meta


def hist(instrs):
    h = collections.Counter()
    for op, _ in instrs:
        if op != "label":
            h[classify(op)] += 1
    return h


def fmt(h):
    order = [c for c, _ in CLASSES] + ["other"]
    return "  ".join("%s %d" % (c, h[c]) for c in order if h[c])


def main():
    argv = sys.argv[1:]
    sub = None
    if "--kernel" in argv:
        i = argv.index("--kernel")
        sub = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    path = args[0] if args else "hector_amd/build/hx_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
    kernels, meta = parse(path)
    for k, ins in kernels.items():
        if sub and sub not in k:
            continue
        if k not in meta:
            continue
        md = meta[k]
        h = hist(ins)
        print("%s\n  vgpr %s agpr %s sgpr %s  scratch %s B  lds %s B  spills s%s v%s  static instr %d" % (
            k, md.get("vgpr_count"), md.get("agpr_count"), md.get("sgpr_count"),
            md.get("private_segment_fixed_size"), md.get("group_segment_fixed_size"),
            md.get("sgpr_spill_count"), md.get("vgpr_spill_count"), sum(h.values())))
        print("  " + fmt(h))
        if "--loops" in sys.argv:
            labels = {}
            for i, (op, s) in enumerate(ins):
                if op == "label":
                    labels[s] = i
            seen = set()
            for i, (op, s) in enumerate(ins):
                if op.startswith("s_cbranch") or op == "s_branch":
                    tgt = s.split()[-1]
                    if tgt in labels and labels[tgt] < i and (labels[tgt], i) not in seen:
                        seen.add((labels[tgt], i))
                        hh = hist(ins[labels[tgt]:i + 1])
                        n = sum(hh.values())
                        if n >= 40:
                            print("    loop %s..+%d: %d instr: %s" % (tgt, i - labels[tgt], n, fmt(hh)))


if __name__ == "__main__":
    main()
