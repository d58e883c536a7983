// LDS read-throughput micro-benchmark for gfx950: ds_read_b128 vs ds_read_b64 vs ds_read_b64_tr_b16.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_bench.hip -o tools/micro/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(int iters, int* out, int stride) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((int*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int acc = 0;
    unsigned base = (wave * 4096) & 0xFFFF;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned off = base + ((u * 1024 + it * 64) & 0x3FFF);
            if (MODE == 0) {
                i32x4 v = *(__attribute__((address_space(3))) i32x4*)(size_t)(off + lane * 16 * stride % 16384);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else if (MODE == 1) {
                i32x2 v = *(__attribute__((address_space(3))) i32x2*)(size_t)(off + lane * 8 * stride % 16384);
                acc += v.x ^ v.y;
            } else {
                // tr read: 16 lanes x 8 B per row group, as the GEMM uses it: lane -> row (lane&15... ) contiguous 8B
                unsigned a = off + ((lane & 15) * 8 + (lane >> 4) * 512) * stride % 16384;
                s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)a);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345) out[0] = acc;
}
template <int MODE, int WAVES>
void run(const char* name, int bytes_per_lane) {
    int* out; hipMalloc(&out, 4);
    const int iters = 4096, grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 65536, 0, iters, out, 1);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr = (double)iters * 8 * WAVES;                // per CU (1 WG / CU)
    double bytes = instr * 64 * bytes_per_lane;
    printf("%-22s waves=%2d  %.3f ms  %.1f ns/instr/CU  %.1f B/ns/CU (at 2.4 GHz: %.1f B/clk)\n", name, WAVES, ms,
           ms * 1e6 / instr, bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4);
}
int main() {
    run<0, 16>("ds_read_b128", 16); run<1, 16>("ds_read_b64", 8); run<2, 16>("ds_read_b64_tr_b16", 8);
    run<0, 8>("ds_read_b128", 16); run<1, 8>("ds_read_b64", 8); run<2, 8>("ds_read_b64_tr_b16", 8);
    run<0, 4>("ds_read_b128", 16); run<1, 4>("ds_read_b64", 8); run<2, 4>("ds_read_b64_tr_b16", 8);
    return 0;
}
