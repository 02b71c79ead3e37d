#!/bin/bash
# hipcc_aligned.sh SRC.hip OUT.o [compiler flags...]
# Compiles one HIP translation unit for gfx950 like `hipcc -c`, with one extra step between the compiler and the
# assembler: align8.py keeps every 8-byte instruction 8-byte aligned (see its header for the measurement).
#   device:  hipcc -S  ->  strip_pk_nops.py  ->  assemble (sizes)  ->  align8.py  ->  assemble  ->  lld  ->  clang-offload-bundler
#   host:    hipcc --cuda-host-only with the bundle embedded (-fcuda-include-gpubinary)
set -e
src=$1; out=$2; shift 2
here=$(cd "$(dirname "$0")" && pwd)
LLVM=${ROCM_LLVM:-/opt/rocm/lib/llvm/bin}
ARCH=${ARCH:-gfx950}
tmp=$(dirname "$out")/$(basename "$out" .o).al
mkdir -p "$tmp"
# 15-bit branch range at compile time: the padding added afterwards can never push a short branch out of its
# real 16-bit range
hipcc --offload-arch=$ARCH "$@" -mllvm -amdgpu-s-branch-bits=15 --cuda-device-only -S "$src" -o "$tmp/dev.raw.s"
# wait states the hardware does not need (see strip_pk_nops.py); ALIGN8_KEEP_PK_NOPS=1 keeps the compiler's
if [ -z "$ALIGN8_KEEP_PK_NOPS" ]; then python3 "$here/strip_pk_nops.py" "$tmp/dev.raw.s" "$tmp/dev.s"; else cp "$tmp/dev.raw.s" "$tmp/dev.s"; fi
$LLVM/clang -target amdgcn-amd-amdhsa -mcpu=$ARCH -c -x assembler "$tmp/dev.s" -o "$tmp/dev.o"
$LLVM/llvm-objdump -d "$tmp/dev.o" > "$tmp/dev.objdump"
python3 "$here/align8.py" "$tmp/dev.s" "$tmp/dev.objdump" "$tmp/dev.al.s"
$LLVM/clang -target amdgcn-amd-amdhsa -mcpu=$ARCH -c -x assembler "$tmp/dev.al.s" -o "$tmp/dev.al.o"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$tmp/dev.co" "$tmp/dev.al.o"
# what the assembled kernels really occupy (.vgpr_count / .agpr_count / .private_segment_fixed_size per kernel): kept next
# to the object, tools/kernel_resources.py turns the notes into the table under profiles/
$LLVM/llvm-readelf --notes "$tmp/dev.co" > "$(dirname "$out")/$(basename "$out" .o).notes.txt"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$tmp/dev.co" -output="$tmp/dev.hipfb"
hipcc --offload-arch=$ARCH "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$tmp/dev.hipfb" -c "$src" -o "$out"
# ALIGN8_KEEP=1 keeps the intermediate files (dev.al.s is the assembly that was actually assembled)
if [ -z "$ALIGN8_KEEP" ]; then rm -rf "$tmp"; fi
