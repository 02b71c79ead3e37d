#!/usr/bin/env python3
"""strip_pk_nops.py in.s out.s -- removes the `s_nop 0` hipcc puts between a packed (VOP3P) instruction and a packed
instruction that reads its result.

The compiler's hazard recognizer treats a VOP3P result as not forwardable to the next VALU instruction and spends one
wait state on every such pair; its scheduler does not know, so depending on the instantiation the hot loop of the
packed strip kernel carries between 0.6 and 6.5 of these per step (sw_strip_kernel_pk16<11,...>: 103 per 16 steps --
4 % of the loop, each s_nop is an issue slot of the lone wavefront).  On gfx950 the pair is interlocked in hardware:
tools/micro_hazard.hip runs dependent chains of exactly these instructions (v_pk_add_u16, v_pk_max_i16, v_pk_add_i16
clamp with op_sel_hi, v_pk_sub_i16 clamp) without any wait state and checks 1.3 * 10^9 results against the arithmetic,
none differ; and every kernel built this way goes through the bit-exact parity suite.

Only `s_nop 0` lines whose neighbours are BOTH plain v_pk_* instructions are removed: nops in front of DPP moves,
v_readlane, memory instructions or anything else (the documented software hazards) stay where the compiler put them."""
import re
import sys


def mnemonic(line):
    s = line.split(";", 1)[0].split("//", 1)[0].strip()
    if not s or s.startswith((".", "#")) or re.match(r"^[A-Za-z_.$][\w.$@]*:\s*$", s):
        return None
    return s


def plain_pk(s):
    return s is not None and s.startswith("v_pk_") and "dpp" not in s and "sdwa" not in s and "row_" not in s and "wave_" not in s


def main():
    lines = open(sys.argv[1]).read().split("\n")
    ins = [(i, mnemonic(l)) for i, l in enumerate(lines)]
    ins = [(i, s) for i, s in ins if s is not None]
    drop = set()
    for k in range(1, len(ins) - 1):
        i, s = ins[k]
        if s.split() == ["s_nop", "0"] and plain_pk(ins[k - 1][1]) and plain_pk(ins[k + 1][1]):
            # nothing but comments may separate the three (a label in between is a branch target: keep the nop)
            if all(mnemonic(lines[j]) is not None or not lines[j].strip() or lines[j].strip().startswith(";")
                   for j in range(ins[k - 1][0], ins[k + 1][0] + 1)):
                drop.add(i)
    open(sys.argv[2], "w").write("\n".join(l for i, l in enumerate(lines) if i not in drop))
    sys.stderr.write("strip_pk_nops: %s: %d of the compiler's s_nop 0 between packed instructions removed\n" % (sys.argv[1], len(drop)))


if __name__ == "__main__":
    main()
