#!/usr/bin/env python3
"""In-order issue model of ONE wavefront on a gfx950 SIMD, on the compiler's assembly.

A lone wavefront (one per SIMD: the 65 536-member launch) issues its instructions in program
order; what a stretch of code costs is its issue slots plus the stalls of instructions whose
operands are not there yet.  This script walks a line range of a kernel's assembly as straight-line
code (branches not taken, labels ignored), keeps a scoreboard of when every register becomes
available, and reports the clocks the stretch takes, how many of them are issue and how many are
stalls, and the instructions that stall longest -- the nearest thing to a per-instruction profile
this pool allows (rocprofv3's PC sampling is not supported by the boxes' driver, ATT has no decoder
here: profiles/r06_pc_sampling_unavailable.txt).

    python tools/isa_sim.py file.s --kernel SUBSTR --from LABEL_OR_LINE --to LABEL_OR_LINE [--top N]
    python tools/isa_sim.py file.s --kernel SUBSTR --loop LABEL        # a loop body: label .. its back edge

Latencies (core clocks, issue of the producer -> earliest issue of a consumer) from the
microbenchmarks of earlier rounds (tools/prof/issue_rate.hip, smem_latency.hip;
profiles/r02_issue_rate.txt): fp64 FMA / add / mul / max 9, issue cost 4.3; v_rcp / v_rsq / v_sqrt
f64 (quarter rate) 21, issue cost 16; fp32 and integer VALU 9 / 4.3 (v_log / v_exp f32 quarter rate);
SALU 4; an LDS read 120; a scalar load 60 (K$ hit); a vector load 1 500 (HBM; 800 if it hits L2).
It is a MODEL: it says where a schedule leaves the pipe idle, not the last clock."""
import argparse
import collections
import re
import sys

ISSUE = {"valu": 4.3, "trans64": 16.0, "trans32": 16.0, "salu": 4.0, "smem": 4.0, "lds": 4.0, "vmem": 4.0,
         "branch": 4.0, "wait": 0.0, "mfma": 8.0, "other": 4.0}
LAT = {"valu": 9.0, "trans64": 21.0, "trans32": 21.0, "salu": 4.0, "smem": 60.0, "lds": 120.0, "vmem": 1500.0,
       "mfma": 72.0, "other": 4.0}

REG = re.compile(r"\b([vsa])\[(\d+):(\d+)\]|\b([vsa])(\d+)\b|\b(vcc|exec|scc|m0)\b")


def regs_of(text):
    out = []
    for m in REG.finditer(text):
        if m.group(1):
            out.extend("%s%d" % (m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4):
            out.append("%s%s" % (m.group(4), m.group(5)))
        else:
            out.append(m.group(6))
    return out


def classify(op):
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("v_mfma"):
        return "mfma"
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "trans64"
    if re.match(r"v_(rcp|rsq|sqrt|log|exp|sin|cos)_f(32|16)", op):
        return "trans32"
    if op.startswith("v_"):
        return "valu"
    return "other"


def kernel_lines(path, sub):
    lines = open(path, errors="replace").read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w+:", l) and sub in l:
            start = i
        elif start is not None and re.match(r"^\.Lfunc_end\d+:", l):
            return lines[start:i]
    raise SystemExit("kernel not found: " + sub)


def find(lines, key):
    if key.isdigit():
        return int(key) - 1
    for i, l in enumerate(lines):
        if l.startswith(key + ":"):
            return i
    raise SystemExit("label not found: " + key)


def simulate(lines, i0, i1, top=15, quiet=False):
    ready = collections.defaultdict(float)   # register -> clock its value is there
    t = 0.0                                  # clock the next instruction may issue (in order)
    lgkm, vm = [], []                        # completion clocks of outstanding smem+lds / vector memory ops
    issue_total = 0.0
    stalls = []
    counts = collections.Counter()
    n = 0
    for ln in range(i0, i1 + 1):
        s = lines[ln].strip()
        if not s or s[0] in ";." or s.endswith(":") or s.startswith("//"):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        op = s.split()[0]
        cls = classify(op)
        counts[cls] += 1
        n += 1
        operands = s[len(op):]
        if cls == "wait":
            need = t
            m = re.search(r"lgkmcnt\((\d+)\)", operands)
            if m and lgkm:
                k = int(m.group(1))
                # scalar loads return out of order: anything but 0 cannot be relied on -- treat as 0
                pend = sorted(lgkm)
                need = max(need, pend[-1] if k == 0 else pend[max(0, len(pend) - 1 - k)])
                lgkm = [x for x in lgkm if x > need]
            m = re.search(r"vmcnt\((\d+)\)", operands)
            if m and vm:
                k = int(m.group(1))
                if len(vm) > k:
                    need = max(need, vm[len(vm) - 1 - k])
                    vm = vm[len(vm) - k:] if k else []
            if need > t:
                stalls.append((need - t, ln + 1, s))
            t = max(t, need)
            continue
        parts = [p.strip() for p in operands.split(",")]
        dst_txt, src_txt = (parts[0], ",".join(parts[1:])) if parts else ("", "")
        dsts = regs_of(dst_txt)
        srcs = regs_of(src_txt)
        if cls in ("vmem", "lds") and ("store" in op or "write" in op):
            srcs = regs_of(operands); dsts = []
        if op.startswith(("v_fmac", "v_mac", "v_cndmask_b32_e32", "v_writelane")):
            srcs += dsts                           # read-modify-write / implicit vcc below
        if op.startswith("v_cndmask_b32_e32") or op.endswith("_co_u32_e32") and "addc" in op:
            srcs.append("vcc")
        if op.startswith("v_cmp") and "_e32" in op:
            dsts = ["vcc"]; srcs = regs_of(operands.replace("vcc", ""))
        if "saveexec" in op or op.startswith("s_") and "exec" in dst_txt:
            pass
        start = max([t] + [ready[r] for r in srcs])
        if start > t + 1e-9:
            stalls.append((start - t, ln + 1, s))
        lat = LAT.get(cls, 4.0)
        if cls == "vmem" and ("store" in op):
            lat = 800.0
        done = start + lat
        for r in dsts:
            ready[r] = done
        if op.startswith("v_cmp") and "_e64" in op:
            pass
        if cls == "smem" or cls == "lds":
            lgkm.append(done)
        if cls == "vmem":
            vm.append(done)
        issue_total += ISSUE.get(cls, 4.0)
        t = start + ISSUE.get(cls, 4.0)
    stall_total = sum(x[0] for x in stalls)
    if not quiet:
        print("lines %d..%d: %d instructions %s" % (i0 + 1, i1 + 1, n, dict(counts)))
        print("  model: %.0f clocks = %.0f issue + %.0f stalled (%.0f %%)" %
              (t, issue_total, stall_total, 100.0 * stall_total / max(t, 1)))
        for d, ln, s in sorted(stalls, reverse=True)[:top]:
            print("    %7.0f  line %5d  %s" % (d, ln, s[:110]))
    return t, issue_total, stall_total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--from", dest="frm")
    ap.add_argument("--to")
    ap.add_argument("--loop")
    ap.add_argument("--top", type=int, default=15)
    a = ap.parse_args()
    lines = kernel_lines(a.file, a.kernel)
    if a.loop:
        i0 = find(lines, a.loop)
        lab = a.loop
        i1 = None
        for i in range(i0 + 1, len(lines)):
            if re.search(r"s_(cbranch\w*|branch)\s+%s\b" % re.escape(lab), lines[i]):
                i1 = i
        if i1 is None:
            raise SystemExit("no back edge to " + lab)
    else:
        i0, i1 = find(lines, a.frm), find(lines, a.to)
    simulate(lines, i0, i1, a.top)


if __name__ == "__main__":
    main()
