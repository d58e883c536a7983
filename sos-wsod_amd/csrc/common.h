// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the OICR+ hot path.
// wave = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SW_F32 0
#define SW_BF16 1

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = SW_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
  __device__ static __forceinline__ void store_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }     // written once, read by a later kernel
};
template <> struct Elem<unsigned short> {   // bf16 carried as raw 16-bit words
  static constexpr int kDtype = SW_BF16;
  __device__ static __forceinline__ float load(const unsigned short* p) { return bf16_bits_to_f32(*p); }
  __device__ static __forceinline__ void store(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }
  __device__ static __forceinline__ void store_nt(unsigned short* p, float v) { __builtin_nontemporal_store(f32_to_bf16_bits(v), p); }
};

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// max of |x| carried as IEEE bits: for non-negative floats the unsigned order is the numeric order, Inf sits above every finite
// value and NaN above Inf — a NaN / Inf anywhere survives the reduction (fmaxf drops NaN)
__device__ __forceinline__ unsigned int absbits(float x) { return __float_as_uint(x) & 0x7FFFFFFFu; }
__device__ __forceinline__ unsigned int wave_reduce_max_u32(unsigned int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned int)__shfl_xor((int)v, o, 64));
  return v;
}

// block-wide reductions for blockDim.x a multiple of 64 and <= 1024; red must hold >= 16 floats.
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = (lane < nw) ? red[lane] : 0.f;
  r = wave_reduce_sum(r);
  return r;
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
  v = wave_reduce_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = (lane < nw) ? red[lane] : -3.402823466e38f;
  r = wave_reduce_max(r);
  return r;
}

// LDS-DMA (buffer_load_dwordx4 ... lds: 16 bytes per lane from a buffer resource straight into LDS at m0 + 16 * lane) issued through
// inline assembly.  Through the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc's waitcnt pass knows an LDS write is in flight, and
// in front of the next ds_read_b64_tr_b16 — the transposed-read intrinsic carries no alias information — it puts s_waitcnt vmcnt(0): a
// kernel that prefetches K-tiles and reads fragments transposed then waits for the DMA it has just issued (found in round 6: the direct
// conv weight gradient spent 52 % of its wave cycles parked there; the fc6 weight-gradient GEMM, K-strided B operand, had the same wait in
// phases 0 and 1 of every K-tile).  The kernels order DMA -> read themselves (counted vmcnt + barrier), so nothing is lost.
// (m0: written here, no other user in these kernels — no builtin LDS-DMA left beside it, no GWS, no movrel.)
typedef __attribute__((ext_vector_type(4))) int sw_i32x4;
__device__ __forceinline__ sw_i32x4 sw_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  return sw_i32x4{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void sw_dma16(const sw_i32x4 rsrc, const char* lds_dst, unsigned voff, unsigned soff) {
  const unsigned m = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds_dst;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// hipGetLastError() is a per-thread sticky value: a recoverable error of an unrelated earlier HIP call (e.g. torch probing a
// device) would otherwise be reported by the first SW_CHECK_LAUNCH of this library.  Every entry point starts clean.
#define SW_ENTER() (void)hipGetLastError()
#define SW_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)
