# same-box A/B: the GEMM kernels built with v_mfma_f32_32x32x16_bf16 (-DSW_MFMA32) against the default 16x16x32
for rep in 1 2; do
TAG=16x16x32 python tools/fc6_three.py 2>&1 | grep -v amdgpu
TAG=32x32x16 SW_LIB_PATH=$PWD/sos-wsod_amd/libsoswsod_hip_mfma32.so python tools/fc6_three.py 2>&1 | grep -v amdgpu
done
SW_LIB_PATH=$PWD/sos-wsod_amd/libsoswsod_hip_mfma32.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm" 2>&1 | tail -2
for rep in 1 2; do
USE_GRAPH=1 python tools/ddp_world1_step.py 60 0 2>&1 | grep "ms/step" | sed 's/^/16x16x32 step:  /'
SW_LIB_PATH=$PWD/sos-wsod_amd/libsoswsod_hip_mfma32.so USE_GRAPH=1 python tools/ddp_world1_step.py 60 0 2>&1 | grep "ms/step" | sed 's/^/32x32x16 step:  /'
done
