#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.
# Compiles the reference's own MASA-Core sources WHERE THEY LIE under
# /root/reference (nothing is copied into the repo) and links them with
# oracle/ref_driver.cpp into oracle/_ref/ref_driver.
#
# Excluded (need autoconf/awk generated headers the image cannot produce):
#   libmasa/libmasa.cpp, libmasa/aligners/AbstractBlockAligner.cpp,
#   libmasa/aligners/AbstractDiagonalAligner.cpp, common/configs/Configs.cpp
# The reference's own build system (autotools) is NOT run.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${REFERENCE_ROOT:-/root/reference}"
M="$REF/masa-cudalign-4.0.2.1028/libs/masa-core/src"
if [ ! -d "$M" ]; then
    echo "build_ref.sh: $M not present; skipping (GPU box uses the prebuilt oracle/_ref)" >&2
    exit 0
fi
OUT="$HERE/_ref"
OBJ="$OUT/obj"
mkdir -p "$OBJ"
CXX="${CXX:-g++}"
FLAGS="-std=gnu++98 -fpermissive -w -O2 -I$M"

list=$(find "$M" -name '*.cpp' | grep -v \
  -e 'libmasa/libmasa.cpp' -e 'AbstractBlockAligner.cpp' \
  -e 'AbstractDiagonalAligner.cpp' -e 'configs/Configs.cpp' | sort)

build_one() {
    src="$1"
    o="$OBJ/$(echo "$src" | md5sum | cut -c1-16).o"
    if [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then
        $CXX $FLAGS -c "$src" -o "$o"
    fi
}
export -f build_one
export CXX FLAGS OBJ
echo "$list" | xargs -P "${JOBS:-8}" -I{} bash -c 'build_one {}'

$CXX $FLAGS "$HERE/ref_driver.cpp" "$OBJ"/*.o -lpthread -o "$OUT/ref_driver"
echo "built $OUT/ref_driver"

# End-to-end drop-in binary: the SAME MASA-Core objects driven by the product's IAligner adapter
# (masa-cudalign_amd/host/Mi355Aligner.cpp -> C ABI -> HIP engine).  Needs libmi355sw.so.
REPO="$(dirname "$HERE")"
LIB="$REPO/masa-cudalign_amd/libmi355sw.so"
if [ -f "$LIB" ]; then
    $CXX $FLAGS -DUSE_MI355_ALIGNER -I"$REPO/include" -I"$REPO/masa-cudalign_amd/host" \
        "$HERE/ref_driver.cpp" "$REPO/masa-cudalign_amd/host/Mi355Aligner.cpp" "$REPO/masa-cudalign_amd/host/Mi355AlignerParameters.cpp" "$OBJ"/*.o \
        -L"$REPO/masa-cudalign_amd" -lmi355sw -Wl,-rpath,'$ORIGIN/../../masa-cudalign_amd' -lpthread \
        -o "$OUT/masa_mi355"
    echo "built $OUT/masa_mi355"
fi
