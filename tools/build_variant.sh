#!/bin/bash
# tools/build_variant.sh NAME [-Dflags...]: rebuild only the R'=12 unpruned packed kernel (sw_kernel_pk16_b.hip, the
# C2 kernel) with extra flags and link it with the current objects into tools/_var_NAME.so (MI355SW_LIB=... A/B runs)
set -e
name=$1; shift
cd "$(dirname "$0")/../masa-cudalign_amd/csrc"
mkdir -p _var
./hipcc_aligned.sh sw_kernel_pk16_b.hip _var/b_$name.o -O3 -std=c++17 -fPIC -w -mllvm -amdgpu-sched-strategy=max-ilp "$@"
objs=$(ls _obj/*.o | grep -v sw_kernel_pk16_b.o)
hipcc --offload-arch=gfx950 -shared $objs _var/b_$name.o -o ../../tools/_var_$name.so
echo built tools/_var_$name.so
