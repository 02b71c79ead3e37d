#!/bin/bash
# Which rule of hipcc's hazard recognizer puts an `s_nop 0` behind a packed instruction?  Runs the post-RA hazard pass
# of the ROCm LLVM on ONE producer line with the four values of the op_sel / op_sel_hi bits of source 0, in front of a
# non-packed reader.  Output: profiles/r04_pk_nop_hazard_probe.txt (see csrc/strip_pk_nops.py).  No GPU needed.
LLVM=${ROCM_LLVM:-/opt/rocm/lib/llvm/bin}
tmp=$(mktemp -d)
cat > $tmp/t.ll <<'LL'
target triple = "amdgcn-amd-amdhsa"
declare <2 x i16> @llvm.sadd.sat.v2i16(<2 x i16>, <2 x i16>)
declare <2 x i16> @llvm.smax.v2i16(<2 x i16>, <2 x i16>)
define amdgpu_kernel void @pk_to_pk(ptr addrspace(1) %p, <2 x i16> %a, <2 x i16> %b) {
  %x = call <2 x i16> @llvm.smax.v2i16(<2 x i16> %a, <2 x i16> %b)
  %z = call <2 x i16> @llvm.sadd.sat.v2i16(<2 x i16> %x, <2 x i16> %b)
  store <2 x i16> %z, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @pk_to_u32(ptr addrspace(1) %p, <2 x i16> %a, <2 x i16> %b, i32 %c) {
  %x = call <2 x i16> @llvm.smax.v2i16(<2 x i16> %a, <2 x i16> %b)
  %xi = bitcast <2 x i16> %x to i32
  %y = xor i32 %xi, %c
  store i32 %y, ptr addrspace(1) %p
  ret void
}
LL
echo "# $($LLVM/llc --version | grep -i 'LLVM version')"
echo "# hipcc's own output for a packed producer and (a) a packed, (b) a plain 32-bit reader of its result:"
$LLVM/llc -mcpu=gfx950 -O3 $tmp/t.ll -o - | grep -v '^\s*;\|^\s*\.\|^$' | grep -A12 '^pk_to_pk:\|^pk_to_u32:' | grep 'pk_to\|v_pk\|s_nop\|v_xor'
$LLVM/llc -mcpu=gfx950 -O3 $tmp/t.ll -stop-before=post-RA-hazard-rec -o $tmp/t.mir
echo "# MIR of the producer before the hazard pass (operands: src0_modifiers, src0, src1_modifiers, src1, clamp, op_sel, op_sel_hi, neg_lo, neg_hi):"
grep -n 'V_PK_MAX_I16' $tmp/t.mir | tail -1
for mod in 8 0 4 12; do
    sed "s/V_PK_MAX_I16 8, killed \$sgpr2, 8, killed \$vgpr1/V_PK_MAX_I16 $mod, killed \$sgpr2, 8, killed \$vgpr1/" $tmp/t.mir > $tmp/t_$mod.mir
    echo "# post-RA-hazard-rec alone, producer's src0_modifiers = $mod (bit 8 = SISrcMods::OP_SEL_1 = op_sel_hi[0] of a VOP3P instruction = SISrcMods::DST_OP_SEL of a VOP3 one):"
    $LLVM/llc -mcpu=gfx950 -start-before=post-RA-hazard-rec $tmp/t_$mod.mir -o - 2>&1 | grep -v '^\s*;\|^\s*\.\|^$' | grep -A9 '^pk_to_u32:' | grep 'v_pk\|s_nop\|v_xor'
done
rm -rf $tmp
