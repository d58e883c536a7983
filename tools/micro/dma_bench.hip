// L2 -> LDS (LDS-DMA buffer loads) bandwidth per CU by access pattern, gfx950.
// Each workgroup (1024 threads, 16 waves) streams "tiles" of 64 KiB into two LDS buffers, nothing else:
//   mode 0: tile = 512 rows x 128 B, rows at a 1 KiB pitch        (NHWC activation slice / K-contiguous GEMM operand)
//   mode 1: tile = 512 rows x 128 B, rows at a 50304 B pitch      (fc6 operand rows)
//   mode 2: tile = 64 KiB contiguous                               (tile-packed operand)
//   mode 3: tile = 128 rows x 512 B, rows at a 50304 B pitch      (K-strided operand)
// The footprint per workgroup is small and shared by all workgroups -> L2 hits.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_bench.hip -o tools/micro/dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lvoid;
template <int MODE>
__global__ __launch_bounds__(1024) void k(const char* buf, unsigned bytes, int iters, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 64 KiB
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)bytes, 0x00020000);
  unsigned voff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int q = wave + 16 * s;               // instruction 0..63 of the tile (1 KiB each)
    const int idx = q * 64 + lane;             // 16-byte chunk index in the tile
    if (MODE == 0) voff[s] = (idx >> 3) * 1024 + (idx & 7) * 16;
    else if (MODE == 1) voff[s] = (idx >> 3) * 50304 + (idx & 7) * 16;
    else if (MODE == 2) voff[s] = idx * 16;
    else voff[s] = (idx >> 5) * 50304 + (idx & 31) * 16;
  }
  const unsigned step = MODE == 2 ? 65536u : (MODE == 3 ? 512u : 128u);    // next K-tile
  for (int it = 0; it < iters; ++it) {
    char* dst = smem + (it & 1) * 65536;
    const unsigned soff = (unsigned)(it & 7) * step + (blockIdx.x & 3) * 4096u * (MODE == 2 ? 16 : 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lvoid)(dst + (wave + 16 * s) * 1024), 16, (int)voff[s], (int)soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // one tile in flight behind the current one
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[tid] == 123 && iters < 0) out[0] = 1;
}
template <int MODE> void run(const char* name, const char* buf, unsigned bytes) {
  int* out; hipMalloc(&out, 4);
  const int iters = 2000, grid = 256;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 131072, 0, buf, bytes, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  const double gb = (double)grid * iters * 65536 / 1e9;
  printf("%-44s %.3f ms  %.1f GB/s per CU  %.2f TB/s\n", name, ms, gb / 256 / (ms * 1e-3), gb / (ms * 1e-3) / 1e3);
}
int main() {
  const size_t bytes = 64u << 20;
  char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
  run<0>("512 rows x 128 B, 1 KiB pitch", buf, (unsigned)bytes);
  run<1>("512 rows x 128 B, 50304 B pitch", buf, (unsigned)bytes);
  run<2>("64 KiB contiguous", buf, (unsigned)bytes);
  run<3>("128 rows x 512 B, 50304 B pitch", buf, (unsigned)bytes);
  return 0;
}
