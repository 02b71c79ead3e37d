#!/usr/bin/env python3
"""strip_pk_nops.py in.s out.s -- removes the `s_nop 0` hipcc puts behind a packed (VOP3P) instruction whose result the
next vector instruction reads.

WHICH RULE PUTS IT THERE (round 4: named, and probed with the compiler itself -- tools/hazard_probe.sh, output in
profiles/r04_pk_nop_hazard_probe.txt).  LLVM's GCNHazardRecognizer::checkVALUHazards, for subtargets with
hasDstSelForwardingHazard() (GFX940 and later), asks getDstSelForwardingOperand() whether the producer writes only PART
of its destination register -- an SDWA dst_sel other than DWORD, or a VOP3 16-bit instruction with op_sel[3] set ("result
into the high half, low half kept": `src0_modifiers & SISrcMods::DST_OP_SEL`) -- and, if a VALU instruction reads that
register in the next slot, spends one wait state ("Shift16Def").  That is the partial-register-write forwarding hazard of
the CDNA3/CDNA4 instruction-set manuals' "manually inserted wait states" table (a VALU write with dst_sel / op_sel that
keeps part of the old VGPR, followed by a VALU read of it: 1 wait state).
A VOP3P instruction has no dst_sel: v_pk_max_i16 / v_pk_add_i16 / v_pk_sub_i16 write all 32 bits.  But its operand list
stores op_sel_hi[0] -- which half of SOURCE 0 feeds the HIGH lane of the operation -- in the very bit of src0_modifiers
that DST_OP_SEL names (SISrcMods::OP_SEL_1 == DST_OP_SEL == 8), and every ordinary packed instruction has op_sel_hi =
[1,1].  The probe runs the hazard pass alone on one MIR line: `V_PK_MAX_I16 8, ...` followed by a reader gets the
`s_nop 0`; the same instruction with src0_modifiers 0 or 4 (op_sel_hi[0] = 0: a DIFFERENT source selection, same full
write) gets none, 12 gets it again -- whatever the consumer is (v_xor_b32 as well as v_pk_add_i16).  The nop follows an
aliased modifier bit, not a property of the result: a false positive of the recognizer for this instruction class.

WHAT THE HARDWARE DOES.  tools/micro_hazard.hip runs dependent chains of exactly these instructions (v_pk_add_u16,
v_pk_max_i16, v_pk_add_i16 clamp with op_sel_hi, v_pk_sub_i16 clamp) without any wait state and checks 1.3 * 10^9 results
against the arithmetic, none differ; and tests/test_gpu_nops.py runs the library built WITH the compiler's nops
(`make keepnops`, libmi355sw_keepnops.so) next to this one on C2 at full size and on seeded cases of every kernel
family and compares every output.  The scheduler does not know about the wait state, so depending on the instantiation
the hot loop of the packed strip kernel carries between 0.6 and 6.5 of them per step (sw_strip_kernel_pk16<11,...>: 103
per 16 steps -- 4 % of the loop, each s_nop is an issue slot of the lone wavefront).

WHAT IS REMOVED: an `s_nop 0` whose predecessor is a plain v_pk_* instruction (no DPP / SDWA form) writing ONE VGPR and
whose successor is a plain v_pk_* instruction that reads that VGPR -- the pair the rule above is about, restricted to the
consumers the micro-benchmark ran -- with nothing but comments between the three.  Anything else stays: nops in front of DPP moves, v_readlane, memory instructions, nops
whose successor does not read the predecessor's result (some other hazard put them there), nops behind labels.  Sites
that look like the pattern but fail the register check are counted and named in the log."""
import re
import sys


def mnemonic(line):
    s = line.split(";", 1)[0].split("//", 1)[0].strip()
    if not s or s.startswith((".", "#")) or re.match(r"^[A-Za-z_.$][\w.$@]*:\s*$", s):
        return None
    return s


def plain_pk(s):
    return s is not None and s.startswith("v_pk_") and "dpp" not in s and "sdwa" not in s and "row_" not in s and "wave_" not in s


def operands(s):
    parts = s.split(None, 1)
    if len(parts) < 2:
        return []
    return [o.strip() for o in parts[1].split(",")]


def vgprs(op):
    """set of VGPR numbers an operand names: v7, v[4:5], -v3, |v3|"""
    m = re.fullmatch(r"[-|]*v(\d+)\|?", op.split()[0]) if op else None
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"[-|]*v\[(\d+):(\d+)\]\|?", op.split()[0]) if op else None
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def reads_result(prev, nxt):
    """does vector instruction `nxt` read the single VGPR that `prev` writes?"""
    po, no = operands(prev), operands(nxt)
    if not po or not no or not nxt.startswith("v_"):
        return False
    dst = vgprs(po[0])
    if len(dst) != 1:
        return False
    srcs = set()
    # every operand but the first is a source; v_cmp*/v_cmpx* (results in VCC / SGPRs / EXEC) read all of theirs, and
    # instructions that accumulate into their destination (v_mac, v_fmac, v_dot*, v_pk_fma with tied dst, DPP `old`) read it too
    for o in (no if nxt.startswith("v_cmp") else no[1:]):
        srcs |= vgprs(o)
    if re.match(r"v_(mac|fmac|dot|mov_b32_dpp|readlane|writelane)", nxt):
        srcs |= vgprs(no[0])
    return bool(dst & srcs)


def main():
    lines = open(sys.argv[1]).read().split("\n")
    ins = [(i, mnemonic(l)) for i, l in enumerate(lines)]
    ins = [(i, s) for i, s in ins if s is not None]
    drop = set()
    odd = []
    for k in range(1, len(ins) - 1):
        i, s = ins[k]
        if s.split() != ["s_nop", "0"] or not plain_pk(ins[k - 1][1]):
            continue
        # nothing but comments may separate the three (a label in between is a branch target: keep the nop)
        if not all(mnemonic(lines[j]) is not None or not lines[j].strip() or lines[j].strip().startswith(";")
                   for j in range(ins[k - 1][0], ins[k + 1][0] + 1)):
            continue
        # (the successor must be a packed instruction as well: that is the pair tools/micro_hazard.hip ran 1.3 * 10^9 times;
        #  a nop in front of any other reader -- v_perm_b32, v_alignbit_b32 -- stays, false positive or not)
        if plain_pk(ins[k + 1][1]) and reads_result(ins[k - 1][1], ins[k + 1][1]):
            drop.add(i)
        else:
            odd.append((i + 1, ins[k - 1][1], ins[k + 1][1]))
    open(sys.argv[2], "w").write("\n".join(l for i, l in enumerate(lines) if i not in drop))
    sys.stderr.write("strip_pk_nops: %s: %d of the compiler's s_nop 0 behind packed instructions removed, %d kept (their successor "
                     "is not a packed instruction reading the result)\n" % (sys.argv[1], len(drop), len(odd)))
    if "-v" in sys.argv[3:]:
        for ln, a, b in odd:
            sys.stderr.write("strip_pk_nops:   kept line %d: %s | s_nop 0 | %s\n" % (ln, a, b))


if __name__ == "__main__":
    main()
