// MFMA sustained-throughput micro-benchmark for gfx950 (bf16): v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16, registers
// only (no LDS, no memory), every CU busy for ~milliseconds — what the matrix pipe sustains under the power cap with each shape, and
// with data that toggles (random operands) vs data that does not (zeros).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bench.hip -o tools/micro/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(int iters, const unsigned* __restrict__ seed, float* out, unsigned long long* clk) {
  extern __shared__ char lds_pin[];                        // 96 KiB per workgroup: exactly ONE workgroup per CU
  const int lane = threadIdx.x & 63;
  if (iters < 0) lds_pin[threadIdx.x] = 1;
  const long long c0 = clock64(), w0 = wall_clock64();
  unsigned s0 = seed[(blockIdx.x * blockDim.x + threadIdx.x) & 4095], s1 = s0 * 2654435761u + 12345u;
  union { unsigned u[4]; bf16x8 v; } A[4], B[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 4; ++e) {
      s0 = s0 * 1664525u + 1013904223u; s1 = s1 * 22695477u + 1u;
      // bf16 values around 1 (exponent fixed, mantissa random): no denormals / infs, mantissa bits toggle
      A[i].u[e] = seed[4096] ? (0x3F803F80u | (s0 & 0x007F007Fu)) : 0u;
      B[i].u[e] = seed[4096] ? (0x3F803F80u | (s1 & 0x007F007Fu)) : 0u;
    }
  if (SHAPE == 0) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i].v, B[j].v, acc[i * 4 + j], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
    if (r == 123.456f) out[0] = r + lane;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i + 2 * rep].v, B[j + 2 * rep].v, acc[i * 2 + j], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][15];
    if (r == 123.456f) out[0] = r + lane;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
  }
}

template <int SHAPE, int WAVES>
void run(const char* name, bool toggling, int grid) {
  unsigned* seed; float* out; unsigned long long* clk;
  hipMalloc(&seed, 4097 * 4); hipMalloc(&out, 4); hipMalloc(&clk, 16);
  unsigned h[4097];
  for (int i = 0; i < 4096; ++i) h[i] = (unsigned)rand();
  h[4096] = toggling ? 1u : 0u;
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 40000;                                  // per iteration and wave: 16 x 16384 (shape 0) = 8 x 32768 (shape 1) = 262144 FLOP x 2
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)k<SHAPE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL((k<SHAPE, WAVES>), dim3(grid), dim3(WAVES * 64), 96 * 1024, 0, 2000, seed, out, clk);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<SHAPE, WAVES>), dim3(grid), dim3(WAVES * 64), 96 * 1024, 0, iters, seed, out, clk);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flop = (double)grid * WAVES * iters * 16.0 * 16384.0;      // 16 MFMAs of 16x16x32 (= 8 of 32x32x16) per iteration
  unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)hc[0] / ((double)hc[1] / 0.1);            // shader cycles per ns (wall_clock64 ticks at 100 MHz)
  const double mfma_cycles = (SHAPE == 0 ? 16.0 : 8.0) * (SHAPE == 0 ? 16.0 : 32.0);   // pipe cycles of one iteration's MFMAs (4 / 8 passes)
  printf("%-46s %s operands, %d waves/CU: %7.2f ms %7.1f TFLOP/s (%.3f of 2500) clock %.2f GHz, %5.1f cycles per iteration and wave (MFMA pipe: %.0f)\n",
         name, toggling ? "random" : "zero  ", WAVES * grid / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0, ghz, (double)hc[0] / iters, mfma_cycles);
  hipFree(clk);
  hipFree(seed); hipFree(out);
}

int main() {
  for (int toggling = 1; toggling >= 0; --toggling) {
    run<0, 4>("v_mfma_f32_16x16x32_bf16 x16 independent", toggling, 256);
    run<1, 4>("v_mfma_f32_32x32x16_bf16 x4 accumulators (x2)", toggling, 256);
    run<0, 8>("v_mfma_f32_16x16x32_bf16 x16 independent", toggling, 256);
    run<1, 8>("v_mfma_f32_32x32x16_bf16 x4 accumulators (x2)", toggling, 256);
    run<0, 16>("v_mfma_f32_16x16x32_bf16 x16 independent", toggling, 256);
  }
  return 0;
}
