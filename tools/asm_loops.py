#!/usr/bin/env python3
"""tools/asm_loops.py dev.al.s FUNCTION_SUBSTRING: the loops of one function of an assembled kernel file (hipcc_aligned.sh
with ALIGN8_KEEP=1 keeps dev.al.s): for every backward branch, the instruction mix of the blocks between its target and
itself -- how the 64-step chunk loops and the code between them are really made up (VALU by class, LDS, waits, nops,
branches).  Used to attribute the chunk-boundary and front-end cost of the strip kernel (docs/NOTEBOOK_r1-r3.md 4.3)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith(("v_perm", "v_alignbit")):
        return "v_perm/align"
    if "dpp" in op:
        return "dpp"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main(path, func):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and func in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    labels, instrs = {}, []          # label -> index into instrs
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        dpp = " dpp" if ("row_" in s or "wave_" in s or "quad_perm" in s) else ""
        instrs.append((op + dpp, s))
    print("%s: %d instructions, %d labels" % (func, len(instrs), len(labels)))
    loops = []
    for k, (op, s) in enumerate(instrs):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                loops.append((labels[tgt], k, tgt))
    for a, b, tgt in sorted(loops, key=lambda x: x[0] - x[1])[:40]:
        mix = collections.Counter(classify(op) for op, _ in instrs[a:b + 1])
        inner = [l for l in loops if a < l[0] and l[1] < b]
        print("loop %-12s instr %6d..%6d (%5d)%s  %s" % (tgt, a, b, b - a + 1, "  [contains %d loops]" % len(inner) if inner else "",
                                                          " ".join("%s=%d" % kv for kv in sorted(mix.items(), key=lambda kv: -kv[1]))))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
