#!/usr/bin/env python3
"""Lint of the compiler's gfx950 assembly for one miscompilation this code base has met.

Under register pressure ROCm 7.2's allocator parks live VGPR values in AGPRs (v_accvgpr_write)
or scratch.  At the top of the block where a divergent `if` rejoins it sometimes places such a
save BEFORE the instruction that restores the execution mask (s_or_b64 exec, exec, s[..]) --
seen where SGPR spill code (v_writelane) already sits in front of that restore.  The save then
runs for the lanes of the branch only -- for none at all when the block was reached through the
`s_cbranch_execz` that skips an empty branch -- while the reload after the join runs for every
lane, which reads whatever the AGPR held before.  Two sightings: the two-wavefront kernel lost 22
values (land temperature among them) whenever no lane needed the safeguarded carbonate restart;
hx_run_kernel<looped biomes, heat flux> lost half a double in the year every lane's CH4 equals
its preindustrial value.

    python tools/check_isa.py file.s
What is reported: a register save (v_accvgpr_write_b32 aN, vM / a scratch store of vM) that sits
between the label of an s_cbranch_execz target and that block's exec restore, and whose source vM
was NOT written inside the branch that ends there, unless the same AGPR was also assigned a few
hundred lines earlier (the other arm of an if / else that merges two values into it).  (A value
the branch computed may be copied under the branch's mask: the other lanes got theirs before
the branch.  A value from outside the branch may not.)  Exit status 1 if there is any.

Second check (scan_mfma; AGPR accumulators in the one-wavefront kernels, VGPR ones in the
two-wavefront flavour): the fp64 matrix-pipe pass keeps its accumulators in AGPRs across the
loop by an `asm` statement that hides the v_mfma results from the compiler's hazard recogniser
and carries the wait states itself (hx_kernels.hip, doeclim_pass_mfma).  That is only correct
while the wait really sits between every v_mfma_f64_16x16x4_f64 and the first instruction that
reads its destination: here every read of an AGPR range (SrcC of a later v_mfma, v_accvgpr_read,
a store straight from AGPRs) is required to be at least MFMA_WAIT wait states behind the last
v_mfma that wrote that range (an `s_nop N` counts N + 1, a v_mfma its 8 passes, anything else
1; a back-to-back
accumulate into the SAME range by the next-but-seven v_mfma, the loop's own rhythm, is counted
the same way), with loop bodies that hold v_mfma's walked twice so that the back edge is
covered."""
import re
import sys

DEST = re.compile(r"^\s*(?:v_|ds_read|ds_bpermute|global_load|buffer_load|scratch_load|flat_load)\S*\s+(v\[(\d+):(\d+)\]|v(\d+))")
SAVE = re.compile(r"^\s*v_accvgpr_write_b32\s+a(\d+),\s*v(\d+)\b")
AWR = re.compile(r"^\s*v_accvgpr_(?:write|mov)_b32\s+a(\d+),")
LOOKBACK = 400
SCR = re.compile(r"^\s*(?:scratch_store|buffer_store)\S*\s+(?:off,\s*)?v(\d+)\b")


def regs(m):
    if m.group(2) is not None:
        return range(int(m.group(2)), int(m.group(3)) + 1)
    return [int(m.group(4))]


def scan(path):
    lines = open(path, errors="replace").read().split("\n")
    execz_targets = set()
    for l in lines:
        m = re.match(r"\s+s_cbranch_execz\s+(\.LBB\d+_\d+)", l)
        if m:
            execz_targets.add(m.group(1))
    found = []
    kernel, block = None, None
    stack = [set()]        # VGPRs written since each enclosing saveexec
    pend = []              # saves seen since the label of the current block
    widened = None         # inside the full-mask window of an if / else join: the SGPR pair its
                           # s_or_saveexec saved (see below), else None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|hx_\w+):", l)
        if m:
            kernel, block, stack, pend = m.group(1), None, [set()], []
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            block, pend, widened = m.group(1), [], None
            continue
        s = l.strip()
        if not s or s[0] in ";.":
            continue
        op = s.split()[0]
        if "saveexec" in op:
            if op.startswith("s_or_saveexec") and block in execz_targets:
                # an else part's head at a join label: a save AHEAD of it ran under the then-lanes'
                # mask only (none, if the label was reached by the execz skip)
                for (ln, src, text, dst) in pend:
                    if src not in stack[-1]:
                        found.append((kernel, block, ln, text))
            stack.append(set())
            pend = []      # what follows belongs to the region opened here
            # (s_or_saveexec at the head of an else part: EXEC = then-lanes | else-lanes, the whole
            #  region's mask, until the s_xor that narrows it to the else lanes)
            # (the window closes only on the s_xor of THE SAME pair -- an unrelated
            #  `s_xor_b64 exec, exec, sN` does not show that every lane of the region was active)
            mm = re.match(r"s_or_saveexec_b64\s+(s\[\d+:\d+\])\s*,", s)
            widened = mm.group(1) if mm else None
            continue
        if widened and s.replace(" ", "") == "s_xor_b64exec,exec," + widened:
            # if / else: between `s_or_saveexec_b64 sX, sX` at the join label and this `s_xor_b64 exec,
            # exec, sX` every lane of the region is active -- a save there stores the value for all
            # of them, whichever way the block was reached (round 5: hx_run_kernel<looped, HF, KERPM,
            # NBP>, two spills of values from outside the NBP-constraint block, correct code)
            pend = []
            widened = None
            continue
        if s.replace(" ", "").startswith("s_or_b64exec,exec,"):
            inner = stack.pop() if len(stack) > 1 else set()
            if block in execz_targets:
                for (ln, src, text, dst) in pend:
                    if src in inner:
                        continue
                    if dst is not None and any(AWR.match(x) and int(AWR.match(x).group(1)) == dst
                                               for x in lines[max(0, ln - 1 - LOOKBACK):ln - 1]):
                        continue
                    found.append((kernel, block, ln, text))
            stack[-1] |= inner
            pend = []
            continue
        if op.startswith(("s_cbranch", "s_branch")):
            pend = []
            continue
        m = SAVE.match(l)
        if m:
            pend.append((i + 1, int(m.group(2)), s, int(m.group(1))))
            continue
        m = SCR.match(l)
        if m:
            pend.append((i + 1, int(m.group(1)), s, None))
            continue
        m = DEST.match(l)
        if m and "cmp" not in op:
            stack[-1].update(regs(m))
    return found


MFMA = re.compile(r"^\s*v_mfma_f64_16x16x4_f64\s+([av])\[(\d+):(\d+)\],\s*([^,]+),\s*([^,]+),\s*(\S+)")
REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")
MFMA_WAIT = 18   # 8 passes of a 16x16x4 fp64 MFMA: its result may be read 18 wait states later
MFMA_PASSES = 8  # issue slots a v_mfma_f64_16x16x4_f64 itself occupies
STORES = ("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store")


def _regs(text):
    """[(file, index)] of every a / v register named in an operand string."""
    out = []
    for r in REG.finditer(text):
        if r.group(1) is not None:
            out.extend((r.group(1), k) for k in range(int(r.group(2)), int(r.group(3)) + 1))
        else:
            out.append((r.group(4), int(r.group(5))))
    return out


def scan_mfma(path):
    """-> [(kernel, line number, text, wait states seen)]: reads of a v_mfma result -- in the AGPRs
    of the one-wavefront kernels or the VGPRs of the two-wavefront flavour, as SrcC of a later
    v_mfma, by v_accvgpr_read, a store or any vector instruction -- too close behind the v_mfma that
    produced it."""
    lines = open(path, errors="replace").read().split("\n")
    found = []
    kernels, cur = [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|hx_\w+):", l)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
            continue
        if cur is None:
            continue
        s = l.split(";")[0].strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or (s[0] == "." and not s.startswith(".LBB")):
            continue
        cur[1].append((i + 1, s))
    for name, ins in kernels:
        if not any(t.startswith("v_mfma_f64") for _, t in ins):
            continue
        # walk loop bodies with MFMAs twice (the back edge)
        labels = {t.split(":")[0]: k for k, (_, t) in enumerate(ins) if t.startswith(".LBB")}
        stream = []
        for k, (ln, t) in enumerate(ins):
            stream.append((ln, t))
            op = t.split()[0]
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = t.split()[-1]
                if tgt in labels and labels[tgt] < k and any(x.startswith("v_mfma_f64") for _, x in ins[labels[tgt]:k]):
                    stream.extend(ins[labels[tgt]:k + 1])
        clock = 0
        written = {}   # (file, index) -> clock of the v_mfma that last wrote it
        for ln, t in stream:
            if t.startswith(".LBB"):
                continue
            op = t.split()[0]
            m = MFMA.match(t)
            body = t[len(op):]
            reads, writes = [], []
            if m:
                reads = _regs(m.group(4)) + _regs(m.group(5)) + _regs(m.group(6))
            elif op.startswith(STORES):
                reads = _regs(body)
            elif op.startswith(("v_", "global_load", "ds_read", "scratch_load", "buffer_load", "flat_load")):
                parts = body.split(",", 1)
                writes = _regs(parts[0])
                reads = _regs(parts[1]) if len(parts) > 1 else []
                if op.startswith(("v_cmp", "v_cmpx")):   # (compares write vcc / an SGPR pair)
                    reads, writes = _regs(body), []
            for a in reads:
                if a in written and clock - written[a] < MFMA_WAIT:
                    found.append((name, ln, t, clock - written[a]))
                    break
            if m:
                for k in range(int(m.group(2)), int(m.group(3)) + 1):
                    written[(m.group(1), k)] = clock
            else:
                for a in writes:
                    written.pop(a, None)
            n = re.match(r"s_nop\s+(\d+)", t)
            clock += int(n.group(1)) + 1 if n else (MFMA_PASSES if m else 1)
    return found


def main():
    found = scan(sys.argv[1])
    for k, b, ln, text in found:
        print("%s %s line %d: %s  <- saved under the branch's execution mask, not written in the branch" % (k, b, ln, text))
    hazards = scan_mfma(sys.argv[1])
    for k, ln, text, seen in hazards:
        print("%s line %d: %s  <- reads a v_mfma result after %d wait state(s), needs %d" % (k, ln, text, seen, MFMA_WAIT))
    print("%d suspicious register save(s)" % len(found))
    if hazards:
        print("%d v_mfma result(s) read too early" % len(hazards))
    sys.exit(1 if (found or hazards) else 0)


if __name__ == "__main__":
    main()
