#!/bin/bash
# build a development variant of the C-ABI library with extra -D flags:  tools/build_variant.sh NAME -DFOO -DBAR  -> sos-wsod_amd/libsoswsod_hip_NAME.so
# (only gemm.hip is recompiled; select with SW_LIB_PATH=$PWD/sos-wsod_amd/libsoswsod_hip_NAME.so)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; C="$ROOT/sos-wsod_amd/csrc"; NAME="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I"$ROOT/include" -I"$C" -Wno-unused-result "$@" -c "$C/gemm.hip" -o "$C/_obj/gemm_$NAME.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/sos-wsod_amd/libsoswsod_hip_$NAME.so" "$C/_obj/gemm_$NAME.o" "$C/_obj/conv_direct.o" "$C/_obj/roipool.o" "$C/_obj/elementwise.o" "$C/_obj/heads.o"
echo "built $NAME"
