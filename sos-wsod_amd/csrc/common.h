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

// hipGetLastError() is a per-thread sticky value: a recoverable error of an unrelated earlier HIP call (e.g. torch probing a
// device) would otherwise be reported by the first SW_CHECK_LAUNCH of this library.  Every entry point starts clean.
#define SW_ENTER() (void)hipGetLastError()
#define SW_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)
