#!/bin/bash
# A/B on ONE box (devices differ by up to 12 % in MFMA-bound kernels): build the library from a git revision's sources (default HEAD)
# into sos-wsod_amd/libsoswsod_hip_base.so; select it with SW_LIB_PATH=$PWD/sos-wsod_amd/libsoswsod_hip_base.so.
#   tools/build_base.sh [REV] [SRC ...]     (SRC: only these sources come from REV, the others from the last regular build)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; C="$ROOT/sos-wsod_amd/csrc"; REV="${1:-HEAD}"; shift || true
SRCS="${@:-gemm conv_direct conv_wgrad_direct conv_winograd roipool elementwise heads detector proposals}"
T="$(mktemp -d)"; mkdir -p "$T/inc"
git -C "$ROOT" show "$REV:sos-wsod_amd/csrc/common.h" > "$T/common.h"
git -C "$ROOT" show "$REV:include/soswsod_hip.h" > "$T/inc/soswsod_hip.h"
OBJS=""
for f in gemm conv_direct conv_wgrad_direct conv_winograd roipool elementwise heads detector proposals; do
  if echo " $SRCS " | grep -q " $f " && git -C "$ROOT" cat-file -e "$REV:sos-wsod_amd/csrc/$f.hip" 2>/dev/null; then
    git -C "$ROOT" show "$REV:sos-wsod_amd/csrc/$f.hip" > "$T/$f.hip"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I"$T/inc" -I"$T" -Wno-unused-result -c "$T/$f.hip" -o "$C/_obj/${f}_base.o" &
    OBJS="$OBJS $C/_obj/${f}_base.o"
  else OBJS="$OBJS $C/_obj/$f.o"; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/sos-wsod_amd/libsoswsod_hip_base.so" $OBJS
rm -rf "$T"; echo "built base from $REV: $SRCS"
