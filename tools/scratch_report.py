#!/usr/bin/env python3
"""Where does the scratch (private segment) of the strip kernels go?  Reads the assembly hipcc_aligned.sh assembled
(ALIGN8_KEEP=1 keeps it: <obj>.al/dev.al.s) and prints, per strip function, the scratch loads/stores inside every loop
with more than 300 instructions -- the unrolled 64-step chunk loops are the hot ones.

    cd masa-cudalign_amd/csrc && ALIGN8_KEEP=1 ./hipcc_aligned.sh sw_kernel_pk16_b.hip /tmp/k/b.o <flags of the Makefile>
    python tools/scratch_report.py /tmp/k/b.al/dev.al.s
"""
import re
import sys


def report(path):
    lines = open(path).read().split("\n")
    meta = {}
    name = None
    for l in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            name = m.group(1)
        for key in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "agpr_count"):
            m = re.match(r"\s+-?\s*\.%s:\s+(\d+)" % key, l)
            if m:
                meta.setdefault("pending", {})[key] = int(m.group(1))
        if l.strip() == "" and name and "pending" in meta:
            pass
    # kernel descriptors: simple second pass
    print("== %s" % path)
    cur = {}
    for l in lines:
        m = re.match(r"\s+-?\s*\.(agpr_count|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count|name):\s+(\S+)", l)
        if m:
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "vgpr_spill_count" and "name" in cur:
                print("kernel %s: private_segment %s B/lane, vgpr %s (+agpr %s), vgpr spills %s, sgpr spills %s"
                      % (cur["name"], cur.get("private_segment_fixed_size"), cur.get("vgpr_count"), cur.get("agpr_count"),
                         cur.get("vgpr_spill_count"), cur.get("sgpr_spill_count")))
                cur = {}
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN7mi355sw\w*process_strip\w*:", l)]
    for st in starts:
        end = [i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
        fname = lines[st].split(":")[0]
        blocks, cur = [], None
        for l in lines[st:end]:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = [m.group(1), 0, 0, []]
                blocks.append(cur)
                continue
            t = l.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if cur is None:
                cur = ["entry", 0, 0, []]
                blocks.append(cur)
            cur[1] += 1
            if "scratch_" in t:
                cur[2] += 1
            mb = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
            if mb:
                cur[3].append(mb.group(1) or mb.group(2))
        idx = {b[0]: i for i, b in enumerate(blocks)}
        loops = set()
        for i, b in enumerate(blocks):
            for t in b[3]:
                if t in idx and idx[t] <= i:
                    loops.add((idx[t], i))
        total = sum(b[2] for b in blocks)
        # callee-saved registers of the noinline strip function: stored at its entry, reloaded at its exit -- once per strip
        big = [(h, t) for (h, t) in loops if sum(blocks[k][1] for k in range(h, t + 1)) > 1500]
        first_loop = min(h for h, _ in big) if big else 0
        last_loop = max(t for _, t in big) if big else len(blocks) - 1
        pro = sum(b[2] for b in blocks[:first_loop])
        epi = sum(b[2] for b in blocks[last_loop + 1:])
        print("function %s: %d instructions, %d scratch loads/stores in total: %d before the first chunk loop (callee-saved "
              "registers stored, once per strip), %d after the last one (reloaded, once per strip), %d in between"
              % (fname, sum(b[1] for b in blocks), total, pro, epi, total - pro - epi))
        for (h, t) in sorted(loops):
            ins = sum(blocks[k][1] for k in range(h, t + 1))
            sc = sum(blocks[k][2] for k in range(h, t + 1))
            if ins > 300:
                inner = [x for x in loops if x != (h, t) and x[0] >= h and x[1] <= t and sum(blocks[k][1] for k in range(x[0], x[1] + 1)) > 300]
                kind = "chunk loop (outer)" if inner else ("64-step chunk body" if ins > 1500 else "other")
                print("   loop %-10s..%-10s %6d instructions %4d scratch ops   %s" % (blocks[h][0], blocks[t][0], ins, sc, kind))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        report(p)
