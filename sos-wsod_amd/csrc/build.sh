#!/bin/bash
# Build the C-ABI HIP library for gfx950 in-tree:  sos-wsod_amd/libsoswsod_hip.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libsoswsod_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -I$HERE -Wno-unused-result"
mkdir -p "$HERE/_obj"
pids=()
for f in gemm conv_direct roipool elementwise heads; do
  if [ ! -f "$HERE/_obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/_obj/$f.o" ] || [ "$HERE/common.h" -nt "$HERE/_obj/$f.o" ] || [ "$ROOT/include/soswsod_hip.h" -nt "$HERE/_obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/_obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE/_obj/gemm.o" "$HERE/_obj/conv_direct.o" "$HERE/_obj/roipool.o" "$HERE/_obj/elementwise.o" "$HERE/_obj/heads.o"
echo "built $OUT"
