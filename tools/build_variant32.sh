#!/bin/bash
# tools/build_variant32.sh NAME [-Dflags...]: rebuild only the int32 kernels (sw_kernel.hip) with extra flags and link
# them with the current objects into tools/_var_NAME.so (MI355SW_LIB=... A/B runs)
set -e
name=$1; shift
cd "$(dirname "$0")/../masa-cudalign_amd/csrc"
mkdir -p _var
./hipcc_aligned.sh sw_kernel.hip _var/k32_$name.o -O3 -std=c++17 -fPIC -w -mllvm -amdgpu-sched-strategy=max-ilp "$@"
objs=$(ls _obj/*.o | grep -v "_obj/sw_kernel.o")
hipcc --offload-arch=gfx950 -shared $objs _var/k32_$name.o -o ../../tools/_var_$name.so
echo built tools/_var_$name.so
