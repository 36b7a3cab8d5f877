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
the branch.  A value from outside the branch may not.)  Exit status 1 if there is any."""
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
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|hx_\w+):", l)
        if m:
            kernel, block, stack, pend = m.group(1), None, [set()], []
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            block, pend = m.group(1), []
            continue
        s = l.strip()
        if not s or s[0] in ";.":
            continue
        op = s.split()[0]
        if "saveexec" in op:
            stack.append(set())
            pend = []      # what follows belongs to the region opened here
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


def main():
    found = scan(sys.argv[1])
    for k, b, ln, text in found:
        print("%s %s line %d: %s  <- saved under the branch's execution mask, not written in the branch" % (k, b, ln, text))
    print("%d suspicious register save(s)" % len(found))
    sys.exit(1 if found else 0)


if __name__ == "__main__":
    main()
