#!/bin/bash
# build a development variant of the C-ABI library with extra -D flags:
#   tools/build_variant.sh NAME [SRC=gemm] -DFOO -DBAR  -> sos-wsod_amd/libsoswsod_hip_NAME.so
# (only SRC.hip is recompiled, the other objects come from the last regular build; select with
#  SW_LIB_PATH=$PWD/sos-wsod_amd/libsoswsod_hip_NAME.so)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; C="$ROOT/sos-wsod_amd/csrc"; NAME="$1"; shift
SRC=gemm
case "$1" in SRC=*) SRC="${1#SRC=}"; shift;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I"$ROOT/include" -I"$C" -Wno-unused-result "$@" -c "$C/$SRC.hip" -o "$C/_obj/${SRC}_$NAME.o"
OBJS=""
for f in gemm conv_direct conv_wgrad_direct conv_winograd roipool elementwise heads detector proposals; do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS $C/_obj/${SRC}_$NAME.o"; else OBJS="$OBJS $C/_obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/sos-wsod_amd/libsoswsod_hip_$NAME.so" $OBJS
echo "built $NAME"
