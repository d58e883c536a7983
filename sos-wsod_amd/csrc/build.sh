#!/bin/bash
# Build the C-ABI HIP library for gfx950 in-tree:  sos-wsod_amd/libsoswsod_hip.so
# An object is rebuilt when the HASH of what it is made from (its source, the shared headers, the flags, the compiler version)
# differs from the stamp beside it — not by mtime: the tree (objects included) travels between machines as a snapshot.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libsoswsod_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -I$HERE -Wno-unused-result"
mkdir -p "$HERE/_obj"
CCV="$($HIPCC --version 2>/dev/null | head -n 2 | tr '\n' ' ')"
pids=()
rebuilt=0
for f in gemm conv_direct conv_wgrad_direct conv_winograd roipool elementwise heads detector proposals; do
  key="$( (echo "$FLAGS $CCV"; cat "$HERE/$f.hip" "$HERE/common.h" "$ROOT/include/soswsod_hip.h") | sha256sum | cut -d' ' -f1)"
  if [ ! -f "$HERE/_obj/$f.o" ] || [ "$(cat "$HERE/_obj/$f.key" 2>/dev/null)" != "$key" ]; then
    ( $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/_obj/$f.o" && echo "$key" > "$HERE/_obj/$f.key" ) &
    pids+=($!)
    rebuilt=1
  fi
done
for p in "${pids[@]}"; do wait $p; done
if [ "$rebuilt" = 1 ] || [ ! -f "$OUT" ]; then
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE/_obj/gemm.o" "$HERE/_obj/conv_direct.o" "$HERE/_obj/conv_wgrad_direct.o" "$HERE/_obj/conv_winograd.o" "$HERE/_obj/roipool.o" "$HERE/_obj/elementwise.o" "$HERE/_obj/heads.o" "$HERE/_obj/detector.o" "$HERE/_obj/proposals.o"
fi
echo "built $OUT"
