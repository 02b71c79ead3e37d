#!/usr/bin/env python3
"""Keep every 8-byte gfx950 instruction 8-byte aligned.

Measured on MI355X (tools/micro_fetch.hip, docs/NOTEBOOK_r1-r3.md section 5): with one wavefront per SIMD an 8-byte
instruction that starts at an address = 4 (mod 8) costs about one extra cycle to fetch/decode.  The packed
strip kernel is ~90 % 8-byte encodings (VOP3P), so whether its hot loop runs at 1.69 s or 1.91 s on the 3M x 3M
case depended on the parity of the 4-byte instructions in front of it, i.e. on unrelated edits.  This pass takes
that freedom away: it rewrites the compiler's assembly so that a 4-byte instruction is always followed by
another 4-byte instruction (natural pair), widened to its 8-byte VOP3 form (same operation, same cost), or
followed by an `s_nop 0`.

usage: align8.py in.s in.objdump out.s
  in.objdump: `llvm-objdump -d` of the assembled in.s (the instruction sizes come from its encodings; the
  mnemonics are cross-checked against in.s).
"""
import re
import sys

# VOP1/VOP2 e32 encodings with an operand-compatible VOP3 (e64) form and no implicit VCC use
WIDEN = {
    "v_mov_b32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_add_u32_e32", "v_sub_u32_e32",
    "v_subrev_u32_e32", "v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_max_i32_e32",
    "v_min_i32_e32", "v_max_u32_e32", "v_min_u32_e32", "v_not_b32_e32", "v_mul_u32_u24_e32", "v_mul_i32_i24_e32",
}
# a 4-byte instruction is left alone (the following 8-byte instructions then sit at +4 until the next 4-byte
# instruction) unless at least this many instructions would run misaligned
PAD_MIN_RUN = int(__import__("os").environ.get("ALIGN8_MIN_RUN", "6"))


def widen_vcc(line, mn):
    """VOP2/VOPC e32 forms with an implicit VCC operand -> e64 with the operand spelled out."""
    body = line.split(";", 1)[0].rstrip()
    if mn == "v_cndmask_b32_e32":
        return body.replace(mn, "v_cndmask_b32_e64", 1)     # the e32 syntax already names vcc
    m = re.match(r"^(\s*)(v_cmp_\w+)_e32\s+vcc,\s*(.*)$", body)
    if m:
        return "%s%s_e64 vcc, %s" % (m.group(1), m.group(2), m.group(3))
    return None


# instructions that the next ones address relative to themselves: nothing may be inserted behind them
PAD_BEFORE = {"s_getpc_b64"}


def is_instruction(line):
    s = line.split(";", 1)[0].split("//", 1)[0].strip()
    if not s or s.startswith((".", "#")):
        return None
    if re.match(r"^[A-Za-z_.$][\w.$@]*:\s*$", s):
        return None
    return s


def main():
    src, sizes_fn, dst = sys.argv[1:4]
    # sizes per function symbol, in program order inside the symbol (sections may be emitted in another order)
    sizes = {}
    sym = None
    for ln in open(sizes_fn):
        m = re.match(r"^[0-9a-fA-F]+ <(\S+)>:", ln)
        if m:
            sym = m.group(1)
            sizes.setdefault(sym, [])
            continue
        m = re.match(r"^\s+(\S+)\s.*//\s*[0-9A-Fa-f]+:\s*((?:[0-9A-Fa-f]{8}\s*)+)(?:<.*>)?\s*$", ln)
        if m and sym is not None:
            sizes[sym].append((m.group(1), 4 * len(m.group(2).split())))
    lines = open(src).read().split("\n")
    # pass 1: attach a size to every instruction line
    insn = []          # (line index, mnemonic, size)
    sym = None
    pos = {}
    in_meta = False
    for idx, ln in enumerate(lines):
        if ln.strip().startswith(".amdgpu_metadata"):
            in_meta = True
        elif ln.strip().startswith(".end_amdgpu_metadata"):
            in_meta = False
        if in_meta:
            continue
        m = re.match(r"^([A-Za-z_$][\w.$@]*):", ln)
        if m and m.group(1) in sizes:
            sym = m.group(1)
            pos.setdefault(sym, 0)
            continue
        s = is_instruction(ln)
        if s is None:
            continue
        mn = s.split()[0]
        if sym is None or pos[sym] >= len(sizes[sym]):
            sys.exit("align8: no encoding for line %d of %s: %s" % (idx + 1, src, s))
        omn, sz = sizes[sym][pos[sym]]
        pos[sym] += 1
        # objdump prints the same mnemonic (VOP suffixes included); tolerate suffix differences only
        if omn != mn and omn.split("_e32")[0].split("_e64")[0] != mn.split("_e32")[0].split("_e64")[0]:
            sys.exit("align8: %s instruction %d differs: %s has '%s', object has '%s'" % (sym, pos[sym], src, mn, omn))
        insn.append((idx, mn, sz))
    for k, v in sizes.items():
        # alignment / end-of-code fill behind a function shows up as trailing s_nop in the object only
        if any(mn != "s_nop" for mn, _ in v[pos.get(k, 0):]):
            sys.exit("align8: %s: %d instructions in %s, %d in %s" % (k, pos.get(k, 0), src, len(v), sizes_fn))
    size_at = {idx: (mn, sz) for idx, mn, sz in insn}
    order = [idx for idx, _, _ in insn]
    next_insn = {a: b for a, b in zip(order, order[1:])}

    out = []
    parity = 0              # 0: next instruction starts 8-byte aligned, 1: at +4
    widened = padded = misaligned = 0
    for idx, ln in enumerate(lines):
        s = ln.strip()
        if idx not in size_at:
            # a function or a section starts: make the parity known
            if re.match(r"^\.(text|section|p2align|balign|align)\b", s):
                out.append(ln)
                if s.startswith((".text", ".section")):
                    out.append("\t.p2align\t3")
                    parity = 0
                elif s.startswith(".p2align"):
                    m = re.match(r"^\.p2align\s+(\d+)", s)
                    if m and int(m.group(1)) >= 3:
                        parity = 0
                    else:
                        out.append("\t.p2align\t3")
                        parity = 0
                else:
                    out.append("\t.p2align\t3")
                    parity = 0
                continue
            out.append(ln)
            continue
        mn, sz = size_at[idx]
        if sz % 8 == 0:
            if parity:
                misaligned += 1
            out.append(ln)
            continue
        if sz % 8 != 4:
            sys.exit("align8: unexpected instruction size %d (line %d)" % (sz, idx + 1))
        if parity == 1:
            out.append(ln)      # restores the alignment
            parity = 0
            continue
        # 4-byte instruction at an aligned address.  Widening it is free; otherwise count the 8-byte instructions
        # that would run misaligned until the next 4-byte instruction restores the parity, and pad only when
        # that costs more than the s_nop (measured: s_nop 1.4 ns, misaligned instruction 0.1-0.4 ns).
        if mn in WIDEN:
            out.append(ln.replace(mn, mn[:-4] + "_e64", 1))
            widened += 1
            continue
        w = widen_vcc(ln, mn)
        if w is not None:
            out.append(w)
            widened += 1
            continue
        run = 0
        q = idx
        reset = False
        while True:
            nq = next_insn.get(q)
            if nq is None:
                reset = True
                break
            if any(re.match(r"^\.(text|section|p2align|balign|align)\b", lines[t].strip()) for t in range(q + 1, nq)):
                reset = True
                break
            if size_at[nq][1] % 8 == 4:
                break
            run += 1
            q = nq
        if reset or run >= PAD_MIN_RUN:
            if mn in PAD_BEFORE:
                out.append("\ts_nop 0")
                out.append(ln)
            else:
                out.append(ln)
                out.append("\ts_nop 0")
            padded += 1
            continue
        out.append(ln)
        parity = 1
    open(dst, "w").write("\n".join(out))
    sys.stderr.write("align8: %s: %d instructions, %d widened to e64, %d padded with s_nop, %d left at +4\n" % (
        src, len(insn), widened, padded, misaligned))


if __name__ == "__main__":
    main()
