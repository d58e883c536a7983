// HBM-bound helpers of the OICR+ path on gfx950: image normalisation, 2x2 max pooling (fwd/bwd),
// weight staging (OIHW f32 -> kernel layouts), column sums (bias gradients), dropout keep-masks,
// SGD-momentum update, loss assembly.  Grid-stride loops capped at 256 CUs x 8 workgroups.
#include "common.h"
#include "soswsod_hip.h"

namespace {

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// ---------------------------------------------------------------- preprocess (rcnn_multi.py:256-269)
template <typename T>
__global__ void preprocess_kernel(int H, int W, int cpad, const uint8_t* __restrict__ img, float m0, float m1, float m2,
                                  float s0, float s1, float s2, T* __restrict__ out) {
  const long npix = (long)H * W;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const float v0 = __fdiv_rn((float)img[p] - m0, s0);
    const float v1 = __fdiv_rn((float)img[npix + p] - m1, s1);
    const float v2 = __fdiv_rn((float)img[2 * npix + p] - m2, s2);
    T* o = out + p * cpad;
    Elem<T>::store(o + 0, v0); Elem<T>::store(o + 1, v1); Elem<T>::store(o + 2, v2);
    for (int c = 3; c < cpad; ++c) Elem<T>::store(o + c, 0.f);
  }
}

// several images of one size in one launch (a view batch = a view and its flipped copy, x B images): blockIdx.y = image
constexpr int PRE_MAX = 16;
struct PreArgs { const uint8_t* img[PRE_MAX]; };
template <typename T>
__global__ void preprocess_multi_kernel(int H, int W, int cpad, PreArgs a, float m0, float m1, float m2, float s0, float s1, float s2,
                                        T* __restrict__ out) {
  const long npix = (long)H * W;
  const uint8_t* __restrict__ img = a.img[blockIdx.y];
  T* o_img = out + (long)blockIdx.y * npix * cpad;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const float v0 = __fdiv_rn((float)img[p] - m0, s0);
    const float v1 = __fdiv_rn((float)img[npix + p] - m1, s1);
    const float v2 = __fdiv_rn((float)img[2 * npix + p] - m2, s2);
    T* o = o_img + p * cpad;
    Elem<T>::store(o + 0, v0); Elem<T>::store(o + 1, v1); Elem<T>::store(o + 2, v2);
    for (int c = 3; c < cpad; ++c) Elem<T>::store(o + c, 0.f);
  }
}

// ---------------------------------------------------------------- maxpool 2x2 (vgg.py:99-100)
template <typename T>
__global__ void maxpool_fwd_kernel(int nimg, int H, int W, int C, int stride, int OH, int OW, const T* __restrict__ in,
                                   T* __restrict__ out) {
  const long total = (long)nimg * OH * OW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long t = i / C;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH); const int n = (int)(t / OH);
    const T* b = in + (((long)n * H + oy * stride) * W + ox * stride) * C + c;
    float m = Elem<T>::load(b);
    m = fmaxf(m, Elem<T>::load(b + C));
    m = fmaxf(m, Elem<T>::load(b + (long)W * C));
    m = fmaxf(m, Elem<T>::load(b + (long)W * C + C));
    Elem<T>::store(out + i, m);
  }
}

// gather form: every input element collects from the (<= 4) windows that contain it and whose first
// maximum (scan order (0,0),(0,1),(1,0),(1,1), strict >) is this element — torch MaxPool2d backward.
template <typename T>
__global__ void maxpool_bwd_kernel(int nimg, int H, int W, int C, int stride, int OH, int OW, const T* __restrict__ in,
                                   const T* __restrict__ dout, T* __restrict__ din, int relu_mask) {
  const long total = (long)nimg * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const int n = (int)(t / H);
    const float self = Elem<T>::load(in + i);
    float g = 0.f;
    if (!(relu_mask && !(self > 0.f))) {
      const int oy_lo = stride == 2 ? (y >> 1) : max(y - 1, 0);
      const int oy_hi = stride == 2 ? (y >> 1) : y;
      const int ox_lo = stride == 2 ? (x >> 1) : max(x - 1, 0);
      const int ox_hi = stride == 2 ? (x >> 1) : x;
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        if (oy >= OH) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
          if (ox >= OW) continue;
          const int y0 = oy * stride, x0 = ox * stride;
          const T* b = in + (((long)n * H + y0) * W + x0) * C + c;
          float m = Elem<T>::load(b); int am = 0;
          float v = Elem<T>::load(b + C); if (v > m) { m = v; am = 1; }
          v = Elem<T>::load(b + (long)W * C); if (v > m) { m = v; am = 2; }
          v = Elem<T>::load(b + (long)W * C + C); if (v > m) { m = v; am = 3; }
          if (am == (y - y0) * 2 + (x - x0)) g += Elem<T>::load(dout + (((long)n * OH + oy) * OW + ox) * C + c);
        }
      }
    }
    Elem<T>::store(din + i, g);
  }
}

// 16-byte channel vectors (8 bf16 / 4 f32 per thread): the pools are pure streaming, the scalar forms above issue a
// 2-byte access and a 64-bit division per element (65 us for the 16 MB pool3 backward; HBM time is ~5 us)
template <typename T> struct Vec16 { static constexpr int N = 16 / (int)sizeof(T); };
template <typename T>
__device__ __forceinline__ void vload(const T* p, float* v) {
  const u32x4 w = *(const u32x4*)p;
  if (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(w[i]);
  }
}
template <typename T>
__device__ __forceinline__ void vstore(T* p, const float* v) {
  u32x4 w;
  if (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16_bits(v[2 * i]) | ((unsigned)f32_to_bf16_bits(v[2 * i + 1]) << 16);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = __float_as_uint(v[i]);
  }
  *(u32x4*)p = w;
}

template <typename T>
__global__ void maxpool_fwd_vec_kernel(int nimg, int H, int W, int C, int stride, int OH, int OW, const T* __restrict__ in,
                                       T* __restrict__ out) {
  constexpr int N = Vec16<T>::N;
  const int CV = C / N;
  const long total = (long)nimg * OH * OW * CV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV); long t = i / CV;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH); const int n = (int)(t / OH);
    const T* b = in + (((long)n * H + oy * stride) * W + ox * stride) * C + cv * N;
    float a[N], v[N];
    vload(b, a);
    vload(b + C, v);
#pragma unroll
    for (int q = 0; q < N; ++q) a[q] = fmaxf(a[q], v[q]);
    vload(b + (long)W * C, v);
#pragma unroll
    for (int q = 0; q < N; ++q) a[q] = fmaxf(a[q], v[q]);
    vload(b + (long)W * C + C, v);
#pragma unroll
    for (int q = 0; q < N; ++q) a[q] = fmaxf(a[q], v[q]);
    vstore(out + ((((long)n * OH + oy) * OW + ox) * C + cv * N), a);
  }
}

// stride 2: windows do not overlap, so a thread owns one 2x2 window (x 16 bytes of channels): 4 + 1 loads, 4 stores.  (The
// gather form below recomputes every window from each of its four pixels: 6 loads per pixel; 216 vs ~100 us on a
// 2 x 300 x 400 x 256 map.)  The odd last row / column of the input belongs to no window and gets zeros.
template <typename T>
__global__ void maxpool_bwd_s2_kernel(int nimg, int H, int W, int C, int OH, int OW, const T* __restrict__ in,
                                      const T* __restrict__ dout, T* __restrict__ din, int relu_mask) {
  constexpr int N = Vec16<T>::N;
  const int CV = C / N, WH = (H + 1) >> 1, WW = (W + 1) >> 1;
  const long total = (long)nimg * WH * WW * CV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV); long t = i / CV;
    const int ox = (int)(t % WW); t /= WW;
    const int oy = (int)(t % WH); const int n = (int)(t / WH);
    const int y0 = 2 * oy, x0 = 2 * ox;
    const long base = (((long)n * H + y0) * W + x0) * C + cv * N;
    const long offs[4] = {0, (long)C, (long)W * C, (long)W * C + C};
    float g[4][N];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < N; ++q) g[k][q] = 0.f;
    if (oy < OH && ox < OW) {
      float v[4][N], d[N];
#pragma unroll
      for (int k = 0; k < 4; ++k) vload(in + base + offs[k], v[k]);
      vload(dout + (((long)n * OH + oy) * OW + ox) * C + cv * N, d);
#pragma unroll
      for (int q = 0; q < N; ++q) {
        float m = v[0][q]; int am = 0;
        if (v[1][q] > m) { m = v[1][q]; am = 1; }
        if (v[2][q] > m) { m = v[2][q]; am = 2; }
        if (v[3][q] > m) { m = v[3][q]; am = 3; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float gk = am == k ? 0.f + d[q] : 0.f;
          if (relu_mask && !(v[k][q] > 0.f)) gk = 0.f;
          g[k][q] = gk;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (y0 + (k >> 1) < H && x0 + (k & 1) < W) vstore(din + base + offs[k], g[k]);
  }
}

template <typename T>
__global__ void maxpool_bwd_vec_kernel(int nimg, int H, int W, int C, int stride, int OH, int OW, const T* __restrict__ in,
                                       const T* __restrict__ dout, T* __restrict__ din, int relu_mask) {
  constexpr int N = Vec16<T>::N;
  const int CV = C / N;
  const long total = (long)nimg * H * W * CV;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV); long t = i / CV;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const int n = (int)(t / H);
    const long off = (((long)n * H + y) * W + x) * C + cv * N;
    float self[N], g[N];
    vload(in + off, self);
#pragma unroll
    for (int q = 0; q < N; ++q) g[q] = 0.f;
    const int oy_lo = stride == 2 ? (y >> 1) : max(y - 1, 0);
    const int oy_hi = stride == 2 ? (y >> 1) : y;
    const int ox_lo = stride == 2 ? (x >> 1) : max(x - 1, 0);
    const int ox_hi = stride == 2 ? (x >> 1) : x;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      if (oy >= OH) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        if (ox >= OW) continue;
        const int y0 = oy * stride, x0 = ox * stride;
        const T* b = in + (((long)n * H + y0) * W + x0) * C + cv * N;
        float m[N], v[N], d[N]; int am[N];
        vload(b, m);
#pragma unroll
        for (int q = 0; q < N; ++q) am[q] = 0;
        vload(b + C, v);
#pragma unroll
        for (int q = 0; q < N; ++q) if (v[q] > m[q]) { m[q] = v[q]; am[q] = 1; }
        vload(b + (long)W * C, v);
#pragma unroll
        for (int q = 0; q < N; ++q) if (v[q] > m[q]) { m[q] = v[q]; am[q] = 2; }
        vload(b + (long)W * C + C, v);
#pragma unroll
        for (int q = 0; q < N; ++q) if (v[q] > m[q]) { m[q] = v[q]; am[q] = 3; }
        vload(dout + (((long)n * OH + oy) * OW + ox) * C + cv * N, d);
        const int me = (y - y0) * 2 + (x - x0);
#pragma unroll
        for (int q = 0; q < N; ++q) if (am[q] == me) g[q] += d[q];
      }
    }
    if (relu_mask) {
#pragma unroll
      for (int q = 0; q < N; ++q) if (!(self[q] > 0.f)) g[q] = 0.f;
    }
    vstore(din + off, g);
  }
}

// ---------------------------------------------------------------- conv weight staging
// mode 0: wk[co][tap][ci_pad] = w[co][ci][tap]            (forward)
// mode 1: wk[ci][8-tap][co]   = w[co][ci][tap]            (data gradient: flipped taps, swapped in/out)
template <typename T>
__global__ void conv_weight_prep_kernel(int mode, int Cout, int Cin, int cin_pad, const float* __restrict__ w,
                                        T* __restrict__ wk) {
  const long total = mode == 0 ? (long)Cout * 9 * cin_pad : (long)Cin * 9 * Cout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (mode == 0) {
      const int ci = (int)(i % cin_pad); long t = i / cin_pad;
      const int tap = (int)(t % 9); const int co = (int)(t / 9);
      if (ci < Cin) v = w[((long)co * Cin + ci) * 9 + tap];
    } else {
      const int co = (int)(i % Cout); long t = i / Cout;
      const int tapf = (int)(t % 9); const int ci = (int)(t / 9);
      v = w[((long)co * Cin + ci) * 9 + (8 - tapf)];
    }
    Elem<T>::store(wk + i, v);
  }
}

template <typename T>
__global__ void convert_2d_kernel(int rows, int cols, const float* __restrict__ src, long ld_src, T* __restrict__ dst,
                                  long ld_dst) {
  const long total = (long)rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols; const int c = (int)(i - r * cols);
    Elem<T>::store(dst + r * ld_dst + c, src[r * ld_src + c]);
  }
}

// 4 columns per thread (16-byte loads): the staging copy of fc6's 411 MB weight is pure HBM streaming
template <typename T>
__global__ void convert_2d_vec4_kernel(int rows, int cols4, const float* __restrict__ src, long ld_src, T* __restrict__ dst,
                                       long ld_dst) {
  const long total = (long)rows * cols4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols4; const int c = (int)(i - r * cols4) * 4;
    const float4 v = *(const float4*)(src + r * ld_src + c);
    T* d = dst + r * ld_dst + c;
    if (sizeof(T) == 2) {
      const unsigned lo = (unsigned)f32_to_bf16_bits(v.x) | ((unsigned)f32_to_bf16_bits(v.y) << 16);
      const unsigned hi = (unsigned)f32_to_bf16_bits(v.z) | ((unsigned)f32_to_bf16_bits(v.w) << 16);
      *(uint2*)d = make_uint2(lo, hi);
    } else {
      *(float4*)d = v;
    }
  }
}

template <typename T>
__global__ void to_f32_kernel(long n, const T* __restrict__ src, float* __restrict__ dst) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = Elem<T>::load(src + i);
}

// ---------------------------------------------------------------- column sums: out[n] = sum_m X[m][n]
// grid = (column slabs of 64, row chunks); each workgroup folds its rows (4 row-lanes x 64 columns, coalesced
// 128/256-byte row segments) and adds ONE f32 atomic per column into the zero-filled output.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, int rows_per_chunk, const T* __restrict__ X, long ld,
                                                     float* __restrict__ out) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float s = 0.f;
  if (col < N)
    for (int m = m0 + rl; m < m1; m += 4) s += Elem<T>::load(X + (long)m * ld + col);
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < N)
    atomicAdd(out + col, (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// Workspace form (the one the training step uses): no zero fill, no atomics, deterministic.  Two launches:
//  1. grid = (column slabs of 32 x 16 bytes, row chunks).  A thread streams 16-byte row pieces (8 bf16 / 4 f32 columns) of
//     every 8th row of its chunk; the workgroup folds its 8 row-lanes through LDS and stores one partial row of the slab
//     into workspace[chunk][N];
//  2. colsum_fold_kernel adds the partial rows in fixed order.
// (A single launch whose last workgroup folds — ticket counter + __threadfence — was measured 20 us SLOWER per call: an
// agent-scope release on this part writes the XCD's whole L2 back, once per workgroup.)
template <typename T>
__device__ __forceinline__ void colsum_ws_body(int M, int N, int rows_per_chunk, const T* __restrict__ X, long ld,
                                               float* __restrict__ ws, const int slab, const int chunk) {
  constexpr int VEC = 16 / (int)sizeof(T), SLAB = 32 * VEC;
  __shared__ float part[8][SLAB + 4];
  const int tid = threadIdx.x, l32 = tid & 31, rsub = tid >> 5;
  const int c0 = slab * SLAB + l32 * VEC;
  const int m0 = chunk * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  if (c0 < N) {                                              // N % VEC == 0 (host): a 16-byte piece is all in or all out
    const T* col = X + c0;
#pragma unroll 4
    for (int m = m0 + rsub; m < m1; m += 8) {
      const u32x4 q = *(const u32x4*)(col + (long)m * ld);
      if (sizeof(T) == 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc[2 * v] += __uint_as_float(q[v] << 16);
          acc[2 * v + 1] += __uint_as_float(q[v] & 0xFFFF0000u);
        }
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] += __uint_as_float(q[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) part[rsub][l32 * VEC + v] = acc[v];
  __syncthreads();
  for (int c = tid; c < SLAB; c += 256) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += part[r][c];
    if (slab * SLAB + c < N) ws[(long)chunk * N + slab * SLAB + c] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_ws_kernel(int M, int N, int rows_per_chunk, const T* __restrict__ X, long ld,
                                                        float* __restrict__ ws) {
  colsum_ws_body<T>(M, N, rows_per_chunk, X, ld, ws, (int)blockIdx.x, (int)blockIdx.y);
}

// several matrices in ONE launch (the bias gradients of every conv layer x view batch of a backward pass): workgroup -> (problem,
// slab, chunk) through a prefix table in the kernel arguments
constexpr int CPART_MAX = 40;
struct ColsumPartMulti {
  int n;
  int M[CPART_MAX], N[CPART_MAX], rpc[CPART_MAX], slabs[CPART_MAX], first_wg[CPART_MAX + 1];
  long ld[CPART_MAX];
  const void* X[CPART_MAX];
  float* ws[CPART_MAX];
};
template <typename T>
__global__ __launch_bounds__(256) void colsum_ws_multi_kernel(ColsumPartMulti f) {
  int i = 0;
  while (i + 1 < f.n && (int)blockIdx.x >= f.first_wg[i + 1]) ++i;
  const int w = (int)blockIdx.x - f.first_wg[i];
  colsum_ws_body<T>(f.M[i], f.N[i], f.rpc[i], (const T*)f.X[i], f.ld[i], f.ws[i], w % f.slabs[i], w / f.slabs[i]);
}

// out[n] = sum_c ws[c][n]: a workgroup owns 16 columns x 16 chunk-lanes (the <= 128 partial rows are 8 loads deep per
// thread; 64 columns x 4 lanes was 32 deep on 4-8 workgroups: 11 us); chunks in fixed order per lane, lanes folded in fixed
// order through LDS
__device__ __forceinline__ void colsum_fold_body(int N, int nchunk, const float* __restrict__ ws, float* __restrict__ out, int wg,
                                                 const int accumulate = 0) {
  __shared__ float fold[16][17];
  const int tid = threadIdx.x, cl = tid >> 4, c16 = tid & 15, col = wg * 16 + c16;
  float f = 0.f;
  if (col < N)
    for (int c = cl; c < nchunk; c += 16) f += ws[(long)c * N + col];
  fold[cl][c16] = f;
  __syncthreads();
  if (cl == 0 && col < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += fold[r][c16];
    out[col] = accumulate ? out[col] + s : s;
  }
}

__global__ __launch_bounds__(256) void colsum_fold_kernel(int N, int nchunk, const float* __restrict__ ws, float* __restrict__ out, int accumulate) {
  colsum_fold_body(N, nchunk, ws, out, blockIdx.x, accumulate);
}

constexpr int CFOLD_MAX = 32;
struct ColsumFoldMulti {
  int n;
  int first_wg[CFOLD_MAX + 1];
  int N[CFOLD_MAX], rows[CFOLD_MAX];
  const float* ws[CFOLD_MAX];
  float* out[CFOLD_MAX];
};
__global__ __launch_bounds__(256) void colsum_fold_multi_kernel(ColsumFoldMulti f) {
  int i = 0;
  while (i + 1 < f.n && (int)blockIdx.x >= f.first_wg[i + 1]) ++i;
  colsum_fold_body(f.N[i], f.rows[i], f.ws[i], f.out[i], blockIdx.x - f.first_wg[i]);
}

// ---------------------------------------------------------------- dropout keep mask
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void dropout_mask_kernel(uint8_t* __restrict__ keep, long n, uint64_t seed, uint64_t offset, float p) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const uint64_t z = splitmix64(splitmix64(seed) + offset + (uint64_t)i);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    keep[i] = u >= p ? 1 : 0;
  }
}

// ---------------------------------------------------------------- SGD momentum (torch.optim.SGD semantics)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr,
                           float mom, float wd, int first, float gscale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float w = p[i];
    float d = g[i] * gscale + wd * w;
    const float b = first ? d : mom * buf[i] + d;
    buf[i] = b;
    p[i] = w - lr * b;
  }
}

// Multi-tensor SGD + weight staging.  blockIdx.x -> (tensor, 4096-element chunk) through a prefix table that travels in
// the kernel arguments with the tensor descriptors.
constexpr int SGD_CHUNK = 4096;
struct SgdBatch {
  sw_sgd_tensor t[SW_SGD_MAX_TENSORS];
  int block_start[SW_SGD_MAX_TENSORS + 1];
  int n;
};

template <typename T>
__device__ __forceinline__ void sgd_stage(const sw_sgd_tensor& d, long i, float w) {
  if (d.stage_kind == 1) {
    const long r = i / d.d0; const int c = (int)(i - r * d.d0);
    Elem<T>::store((T*)d.stage0 + r * d.ld0 + c, w);
  } else {
    const int tap = (int)(i % 9); const long t = i / 9;
    const int ci = (int)(t % d.d1); const int co = (int)(t / d.d1);
    if (d.stage0) Elem<T>::store((T*)d.stage0 + ((long)co * 9 + tap) * d.d2 + ci, w);
    if (d.stage1) Elem<T>::store((T*)d.stage1 + ((long)ci * 9 + (8 - tap)) * d.d0 + co, w);
  }
}

// conv weights whose Cout and Cin are multiples of 32: one block = a 32 (co) x 32 (ci) x 9 (tap) tile.  The OIHW master is
// read / written in 1152-byte runs, the updated values pass through an LDS tile, and both compute layouts leave as 64-byte
// runs ([co][tap][ci]: 32 ci, [ci][8-tap][co]: 32 co).  The element-wise form scattered 2-byte stores 2 x Cin / Cout apart
// and spent 0.2 ms per step on 14.7 M conv weights.
__device__ __forceinline__ bool sgd_conv_tileable(const sw_sgd_tensor& d) {
  return d.stage_kind == 2 && (d.d0 % 32) == 0 && (d.d1 % 32) == 0;
}

template <typename T>
__device__ __forceinline__ void sgd_conv_tile(const sw_sgd_tensor& d, int blk, float mom, float gscale, float (*tile)[289]) {
  const int cib = d.d1 / 32;
  const int co0 = (blk / cib) * 32, ci0 = (blk % cib) * 32;
  const int Cin = d.d1, Cout = d.d0;
  const float lr = d.hyper_dev ? d.hyper_dev[0] : d.lr, wd = d.hyper_dev ? d.hyper_dev[1] : d.weight_decay;
  for (int idx = threadIdx.x; idx < 32 * 288; idx += 256) {
    const int co_l = idx / 288, rem = idx - co_l * 288;                 // rem = ci_l * 9 + tap
    const long o = ((long)(co0 + co_l) * Cin + ci0) * 9 + rem;
    const float w0 = d.param[o];
    const float dd = d.grad[o] * gscale + wd * w0;
    const float m = d.first_step ? dd : mom * d.momentum_buf[o] + dd;
    const float w1 = w0 - lr * m;
    d.momentum_buf[o] = m; d.param[o] = w1;
    tile[co_l][rem] = w1;
  }
  __syncthreads();
  if (d.stage0) {
    T* s0 = (T*)d.stage0;
    for (int idx = threadIdx.x; idx < 32 * 288; idx += 256) {
      const int co_l = idx / 288, r2 = idx - co_l * 288;
      const int tap = r2 >> 5, ci_l = r2 & 31;
      Elem<T>::store(s0 + ((long)(co0 + co_l) * 9 + tap) * d.d2 + ci0 + ci_l, tile[co_l][ci_l * 9 + tap]);
    }
  }
  if (d.stage1) {
    T* s1 = (T*)d.stage1;
    for (int idx = threadIdx.x; idx < 32 * 288; idx += 256) {
      const int ci_l = idx / 288, r3 = idx - ci_l * 288;
      const int tapf = r3 >> 5, co_l = r3 & 31;
      Elem<T>::store(s1 + ((long)(ci0 + ci_l) * 9 + tapf) * Cout + co0 + co_l, tile[co_l][ci_l * 9 + (8 - tapf)]);
    }
  }
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdBatch b, float mom, float gscale) {
  __shared__ float conv_tile[32][289];
  int ti = 0;
  while (ti + 1 < b.n && (int)blockIdx.x >= b.block_start[ti + 1]) ++ti;
  const sw_sgd_tensor& d = b.t[ti];
  if (sgd_conv_tileable(d)) {
    if (d.stage_dtype == SW_BF16) sgd_conv_tile<unsigned short>(d, (int)blockIdx.x - b.block_start[ti], mom, gscale, conv_tile);
    else sgd_conv_tile<float>(d, (int)blockIdx.x - b.block_start[ti], mom, gscale, conv_tile);
    return;
  }
  // learning rate / weight decay: kernel arguments, or (hyper_dev) two floats in device memory — a captured hipGraph of the
  // step then stays valid when the schedule changes them
  const float lr = d.hyper_dev ? d.hyper_dev[0] : d.lr, wd = d.hyper_dev ? d.hyper_dev[1] : d.weight_decay;
  const long base = (long)((int)blockIdx.x - b.block_start[ti]) * SGD_CHUNK;
  const long end = min(d.n, base + SGD_CHUNK);
  const bool vec = ((((uintptr_t)d.param) | ((uintptr_t)d.grad) | ((uintptr_t)d.momentum_buf)) & 15) == 0;
  if (vec) {
    for (long i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
      if (i + 4 <= end) {
        const float4 w4 = *(const float4*)(d.param + i);
        const float4 g4 = *(const float4*)(d.grad + i);
        float4 m4 = d.first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(d.momentum_buf + i);
        float w[4] = {w4.x, w4.y, w4.z, w4.w}; const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        float m[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dd = g[e] * gscale + wd * w[e];
          m[e] = d.first_step ? dd : mom * m[e] + dd;
          w[e] = w[e] - lr * m[e];
        }
        *(float4*)(d.momentum_buf + i) = make_float4(m[0], m[1], m[2], m[3]);
        *(float4*)(d.param + i) = make_float4(w[0], w[1], w[2], w[3]);
        if (d.stage_kind == 1 && (d.d0 & 3) == 0 && d.stage_dtype == SW_BF16) {      // 4 columns of one row: one 8-byte store
          const long r = i / d.d0; const int c = (int)(i - r * d.d0);
          const unsigned lo = (unsigned)f32_to_bf16_bits(w[0]) | ((unsigned)f32_to_bf16_bits(w[1]) << 16);
          const unsigned hi = (unsigned)f32_to_bf16_bits(w[2]) | ((unsigned)f32_to_bf16_bits(w[3]) << 16);
          unsigned short* o = (unsigned short*)d.stage0 + r * d.ld0 + c;
          if ((((uintptr_t)o) & 7) == 0) *(uint2*)o = make_uint2(lo, hi);
          else { o[0] = (unsigned short)lo; o[1] = (unsigned short)(lo >> 16); o[2] = (unsigned short)hi; o[3] = (unsigned short)(hi >> 16); }
        } else if (d.stage_kind) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (d.stage_dtype == SW_BF16) sgd_stage<unsigned short>(d, i + e, w[e]); else sgd_stage<float>(d, i + e, w[e]);
          }
        }
      } else {
        for (long j = i; j < end; ++j) {
          const float w0 = d.param[j];
          const float dd = d.grad[j] * gscale + wd * w0;
          const float m = d.first_step ? dd : mom * d.momentum_buf[j] + dd;
          const float w1 = w0 - lr * m;
          d.momentum_buf[j] = m; d.param[j] = w1;
          if (d.stage_kind) { if (d.stage_dtype == SW_BF16) sgd_stage<unsigned short>(d, j, w1); else sgd_stage<float>(d, j, w1); }
        }
      }
    }
  } else {
    for (long j = base + threadIdx.x; j < end; j += 256) {
      const float w0 = d.param[j];
      const float dd = d.grad[j] * gscale + wd * w0;
      const float m = d.first_step ? dd : mom * d.momentum_buf[j] + dd;
      const float w1 = w0 - lr * m;
      d.momentum_buf[j] = m; d.param[j] = w1;
      if (d.stage_kind) { if (d.stage_dtype == SW_BF16) sgd_stage<unsigned short>(d, j, w1); else sgd_stage<float>(d, j, w1); }
    }
  }
}

// 64 x 64 tiles of a 2D parameter: SGD update (or plain conversion when grad == nullptr), row-major compute-dtype copy and
// the transposed copy through an LDS tile, every global access a >= 128-byte run.
template <typename T>
__global__ __launch_bounds__(256) void sgd_tile_t_kernel(int rows, int cols, float* __restrict__ param,
                                                         const float* __restrict__ grad, float* __restrict__ buf, long ld_src,
                                                         float lr, float wd, int first, float mom, float gscale,
                                                         T* __restrict__ st0, long ld0, T* __restrict__ st1, long ld1,
                                                         const float* __restrict__ hyper) {
  __shared__ float tile[64][65];
  if (hyper) { lr = hyper[0]; wd = hyper[1]; }
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = tr + it * 16;
    const long o = (long)(r0 + r) * ld_src + c0 + tc;
    float4 w4 = *(const float4*)(param + o);
    float w[4] = {w4.x, w4.y, w4.z, w4.w};
    if (grad) {
      const float4 g4 = *(const float4*)(grad + o);
      const float4 m4 = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(buf + o);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
      float m[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dd = g[e] * gscale + wd * w[e];
        m[e] = first ? dd : mom * m[e] + dd;
        w[e] = w[e] - lr * m[e];
      }
      *(float4*)(buf + o) = make_float4(m[0], m[1], m[2], m[3]);
      *(float4*)(param + o) = make_float4(w[0], w[1], w[2], w[3]);
    }
    if (st0) {
      T* d = st0 + (long)(r0 + r) * ld0 + c0 + tc;
      if (sizeof(T) == 2 && (((uintptr_t)d) & 7) == 0) {
        *(uint2*)d = make_uint2((unsigned)f32_to_bf16_bits(w[0]) | ((unsigned)f32_to_bf16_bits(w[1]) << 16),
                                (unsigned)f32_to_bf16_bits(w[2]) | ((unsigned)f32_to_bf16_bits(w[3]) << 16));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) Elem<T>::store(d + e, w[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[r][tc + e] = w[e];
  }
  __syncthreads();
  // transposed write: output row c (= source column), 64 source rows = one 128-byte (bf16) / 256-byte (f32) run
  const int oc = threadIdx.x >> 2, q = (threadIdx.x & 3) * 16;
  T* d = st1 + (long)(c0 + oc) * ld1 + r0 + q;
  if (sizeof(T) == 2 && (((uintptr_t)d) & 15) == 0) {
    u32x4 o[2];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e >> 2][e & 3] = (unsigned)f32_to_bf16_bits(tile[q + 2 * e][oc]) | ((unsigned)f32_to_bf16_bits(tile[q + 2 * e + 1][oc]) << 16);
    *(u32x4*)d = o[0];
    *(u32x4*)(d + 8) = o[1];
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) Elem<T>::store(d + e, tile[q + e][oc]);
  }
}

// f32 NCHW -> dtype NHWC with channel padding (generic backbone entry; the fused path is sw_preprocess)
template <typename T>
__global__ void nchw_to_nhwc_kernel(int N, int C, int H, int W, int cpad, const float* __restrict__ in, T* __restrict__ out) {
  const long total = (long)N * H * W * cpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad); long t = i / cpad;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const int n = (int)(t / H);
    Elem<T>::store(out + i, c < C ? in[(((long)n * C + c) * H + y) * W + x] : 0.f);
  }
}

// ReLU backward: out = ref > 0 ? g : 0 (out may be g)
template <typename T>
__global__ void relu_bwd_kernel(long n, const T* __restrict__ ref, const T* g, T* out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = Elem<T>::load(ref + i) > 0.f ? g[i] : (T)0;
}

// 16-byte pieces (n a multiple of 8 bf16 / 4 f32 elements, aligned tensors)
template <typename T>
__global__ void relu_bwd_vec_kernel(long nv, const T* __restrict__ ref, const T* g, T* out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const u32x4 r = ((const u32x4*)ref)[i];
    u32x4 v = ((const u32x4*)g)[i];
    if (sizeof(T) == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool lo = __uint_as_float(r[k] << 16) > 0.f, hi = __uint_as_float(r[k] & 0xFFFF0000u) > 0.f;
        v[k] = (lo ? (v[k] & 0x0000FFFFu) : 0u) | (hi ? (v[k] & 0xFFFF0000u) : 0u);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(r[k]) > 0.f ? v[k] : 0u;
    }
    ((u32x4*)out)[i] = v;
  }
}

// out[m][n] = in[m][n] * colscale[n]  (f32 -> dtype): applies the per-loss cotangents to the unit logit gradients
template <typename T>
__global__ void scale_cols_kernel(int M, int N, const float* __restrict__ in, long ld_in, const float* __restrict__ cs,
                                  T* __restrict__ out, long ld_out) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / N; const int n = (int)(i - m * N);
    Elem<T>::store(out + m * ld_out + n, in[m * ld_in + n] * cs[n]);
  }
}

// scale_cols with the per-column scale formed in place from the cotangents of the loss vector and of the total:
//   cs[n] = ((g_losses ? g_losses[col_to_loss[n]] : 0) + (g_total ? g_total[0] : 0)) * mul   for n < n_valid, 0 beyond
// (the padding columns of the packed logits are never read: `in` may hold anything there)
template <typename T>
__global__ void scale_cols_loss_kernel(int M, int N, int n_valid, const float* __restrict__ in, long ld_in,
                                       const float* __restrict__ g_losses, const float* __restrict__ g_total,
                                       const int* __restrict__ col_to_loss, float mul, T* __restrict__ out, long ld_out) {
  const long total = (long)M * N;
  const float gt = g_total ? g_total[0] : 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / N; const int n = (int)(i - m * N);
    float v = 0.f;
    if (n < n_valid) v = in[m * ld_in + n] * (((g_losses ? g_losses[col_to_loss[n]] : 0.f) + gt) * mul);
    Elem<T>::store(out + m * ld_out + n, v);
  }
}

// the four views' proposal boxes / objectness of one image -> the packed tensors the heads work on (one launch instead of
// five torch.cat / stack): boxes [4][R][4], obj [4][R], rois [2][2R][5] (scale s: rows [0,R) = view 2s with batch index 0,
// rows [R,2R) = view 2s+1 with batch index 1; poolers.py:81-108 convert_boxes_to_pooler_format)
struct PackViewsArgs { const float* box[4]; const float* obj[4]; };
__global__ void pack_views_kernel(int R, PackViewsArgs a, float* __restrict__ boxes, float* __restrict__ obj,
                                  float* __restrict__ rois) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * R) return;
  const int v = i / R, r = i - v * R;
  const float* b = a.box[v] + (long)r * 4;
  const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
  float* o = boxes + (long)i * 4;
  o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
  obj[i] = a.obj[v][r];
  float* q = rois + (long)i * 5;                     // (scale v/2, row (v&1)*R + r) == flat row i
  q[0] = (float)(v & 1); q[1] = x1; q[2] = y1; q[3] = x2; q[4] = y2;
}

// One pass of Pillow's 8-bit separable resize (libImaging/Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc):
//   out[o] = clip8((2^21 + sum_t in[first[o] + t] * kk[o][t]) >> 22)   along x (horizontal = 1) or y, per band.
// `bounds` (first input index, tap count) and the 22-bit fixed-point coefficients `kk` come from the host, computed in double
// precision exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do (sos_wsod_amd.resize).  out_flip (optional) also
// receives the row mirrored in x: the h-flipped view of the multi-view mapper in the same launch.
__global__ void resize_pass_u8_kernel(int C, int H, int W, long in_ld, long in_plane, int OH, int OW, int horizontal,
                                      const uint8_t* __restrict__ in, const int* __restrict__ bounds, const int* __restrict__ kk,
                                      int ksize, uint8_t* __restrict__ out, uint8_t* __restrict__ out_flip) {
  const long total = (long)C * OH * OW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % OW); const long t = i / OW;
    const int y = (int)(t % OH), c = (int)(t / OH);
    const int o = horizontal ? x : y;
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (long)o * ksize;
    int ss = 1 << 21;
    if (horizontal) {
      const uint8_t* p = in + (long)c * in_plane + (long)y * in_ld + first;
      for (int j = 0; j < n; ++j) ss += (int)p[j] * k[j];
    } else {
      const uint8_t* p = in + (long)c * in_plane + (long)first * in_ld + x;
      for (int j = 0; j < n; ++j) ss += (int)p[(long)j * in_ld] * k[j];
    }
    int v = ss >> 22;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    out[i] = (uint8_t)v;
    if (out_flip) out_flip[((long)c * OH + y) * OW + (OW - 1 - x)] = (uint8_t)v;
  }
}

// RandomBrightness then RandomSaturation of the training mapper (augmentation_impl.py:403-455), each a fvcore BlendTransform on
// a uint8 HWC image: `np.clip(src_weight * src_image + dst_weight * img.astype(float32), 0, 255).astype(uint8)`.
//   brightness: src_image = 0 -> float32 product w * px, clipped, truncated.
//   saturation: src_image = img.dot([0.299, 0.587, 0.114]) of the brightened uint8 image in DOUBLE (channel 0 takes 0.299 whatever
//   the channel order is), src_weight = 1 - w a python double, dst_weight * img a float32 product -> the sum is a double.
// Planar (3, H, W) in / out; out_flip (optional) receives the rows mirrored in x (the mapper's HFlipTransform comes after the blends).
__global__ void color_jitter_u8_kernel(int H, int W, int mode, const uint8_t* __restrict__ in, float w_bright, double src_w_sat,
                                       float w_sat, uint8_t* __restrict__ out, uint8_t* __restrict__ out_flip) {
  const long plane = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
    uint8_t b[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      b[c] = in[c * plane + i];
      if (mode & 1) {
        float f = __fmul_rn(w_bright, (float)b[c]);
        f = fminf(fmaxf(f, 0.f), 255.f);
        b[c] = (uint8_t)(int)f;
      }
    }
    if (mode & 2) {
      const double g = __dadd_rn(__dadd_rn(__dmul_rn((double)b[0], 0.299), __dmul_rn((double)b[1], 0.587)), __dmul_rn((double)b[2], 0.114));
      const double sg = __dmul_rn(src_w_sat, g);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double v = __dadd_rn(sg, (double)__fmul_rn(w_sat, (float)b[c]));
        v = fmin(fmax(v, 0.0), 255.0);
        b[c] = (uint8_t)(int)v;
      }
    }
    const int x = (int)(i % W); const long row = i - x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      out[c * plane + i] = b[c];
      if (out_flip) out_flip[c * plane + row + (W - 1 - x)] = b[c];
    }
  }
}

__global__ void mean_views_kernel(int V, long n, const float* __restrict__ in, float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = in[i];
    for (int v = 1; v < V; ++v) s += in[(long)v * n + i];      // ((s0+s1)+s2)+s3, as the reference adds them
    out[i] = __fdiv_rn(s, (float)V);
  }
}

// out[i] = mean over images b of (mean over views v of lv[b][i][v]); out[nl] = sum_i out[i] (fixed order)
__global__ void loss_finalize_kernel(int nl, int V, int B, const float* __restrict__ lv, float* __restrict__ out,
                                     float* __restrict__ total) {
  __shared__ float s_l[64];
  const int i = threadIdx.x;
  if (i < nl) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* p = lv + ((long)b * nl + i) * V;
      float s = p[0];
      for (int v = 1; v < V; ++v) s += p[v];
      acc += __fdiv_rn(s, (float)V);
    }
    s_l[i] = out[i] = B == 1 ? acc : __fdiv_rn(acc, (float)B);
  }
  __syncthreads();
  if (i == 0 && total) {
    float t = s_l[0];
    for (int j = 1; j < nl; ++j) t += s_l[j];
    total[0] = t;
    total[1] = (t - t == 0.f) ? 1.f : 0.f;          // finite flag (NaN / inf of any loss reaches the sum)
  }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
  if ((dtype) == SW_BF16) { CALL_BF16; } else if ((dtype) == SW_F32) { CALL_F32; } else return -1;

extern "C" int sw_preprocess(int dtype, int H, int W, int cpad, const uint8_t* img, const float* mean3,
                             const float* std3, void* out, hipStream_t stream) {
  SW_ENTER();
  // mean/std are HOST floats here? No: device pointers are the convention, but these 6 scalars are
  // configuration constants; they are passed by value through the launch instead of dereferenced on device.
  // => mean3/std3 are HOST pointers (documented exception).
  const long n = (long)H * W;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(preprocess_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, H, W, cpad, img,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (unsigned short*)out),
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, H, W, cpad, img,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_preprocess_multi(int dtype, int n, int H, int W, int cpad, const uint8_t* const* imgs_chw, const float* mean3,
                                   const float* std3, void* out_nhwc, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  if (cpad < 3) return -1;
  const long npix = (long)H * W;
  const long es = dtype == SW_BF16 ? 2 : 4;
  for (int base = 0; base < n; base += PRE_MAX) {
    PreArgs a = {};
    const int m = n - base < PRE_MAX ? n - base : PRE_MAX;
    for (int i = 0; i < m; ++i) a.img[i] = imgs_chw[base + i];
    void* out = (char*)out_nhwc + (long)base * npix * cpad * es;
    long bx = grid_for(npix); if (bx > 512) bx = 512;
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(preprocess_multi_kernel<unsigned short>, dim3((unsigned)bx, (unsigned)m), dim3(256), 0, stream, H, W, cpad, a,
                         mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (unsigned short*)out),
      hipLaunchKernelGGL(preprocess_multi_kernel<float>, dim3((unsigned)bx, (unsigned)m), dim3(256), 0, stream, H, W, cpad, a,
                         mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (float*)out));
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int sw_maxpool2x2_fwd(int dtype, int nimg, int H, int W, int C, int stride, const void* in, void* out,
                                 hipStream_t stream) {
  SW_ENTER();
  const int OH = (H - 2) / stride + 1, OW = (W - 2) / stride + 1;
  const long n = (long)nimg * OH * OW * C;
  const int vn = dtype == SW_BF16 ? 8 : 4;
  if ((C % vn) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(maxpool_fwd_vec_kernel<unsigned short>, dim3(grid_for(n / vn)), dim3(256), 0, stream, nimg, H, W, C,
                         stride, OH, OW, (const unsigned short*)in, (unsigned short*)out),
      hipLaunchKernelGGL(maxpool_fwd_vec_kernel<float>, dim3(grid_for(n / vn)), dim3(256), 0, stream, nimg, H, W, C, stride,
                         OH, OW, (const float*)in, (float*)out));
    SW_CHECK_LAUNCH();
    return 0;
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(maxpool_fwd_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, nimg, H, W, C, stride,
                       OH, OW, (const unsigned short*)in, (unsigned short*)out),
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, nimg, H, W, C, stride, OH, OW,
                       (const float*)in, (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_maxpool2x2_bwd(int dtype, int nimg, int H, int W, int C, int stride, const void* in,
                                 const void* dout, void* din, int relu_mask, hipStream_t stream) {
  SW_ENTER();
  const int OH = (H - 2) / stride + 1, OW = (W - 2) / stride + 1;
  const long n = (long)nimg * H * W * C;
  const int vn = dtype == SW_BF16 ? 8 : 4;
  if (stride == 2 && (C % vn) == 0 && (((uintptr_t)in | (uintptr_t)dout | (uintptr_t)din) & 15) == 0) {
    const long nw = (long)nimg * ((H + 1) / 2) * ((W + 1) / 2) * (C / vn);
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(maxpool_bwd_s2_kernel<unsigned short>, dim3(grid_for(nw)), dim3(256), 0, stream, nimg, H, W, C, OH, OW,
                         (const unsigned short*)in, (const unsigned short*)dout, (unsigned short*)din, relu_mask),
      hipLaunchKernelGGL(maxpool_bwd_s2_kernel<float>, dim3(grid_for(nw)), dim3(256), 0, stream, nimg, H, W, C, OH, OW,
                         (const float*)in, (const float*)dout, (float*)din, relu_mask));
    SW_CHECK_LAUNCH();
    return 0;
  }
  if ((C % vn) == 0 && (((uintptr_t)in | (uintptr_t)dout | (uintptr_t)din) & 15) == 0) {
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(maxpool_bwd_vec_kernel<unsigned short>, dim3(grid_for(n / vn)), dim3(256), 0, stream, nimg, H, W, C,
                         stride, OH, OW, (const unsigned short*)in, (const unsigned short*)dout, (unsigned short*)din,
                         relu_mask),
      hipLaunchKernelGGL(maxpool_bwd_vec_kernel<float>, dim3(grid_for(n / vn)), dim3(256), 0, stream, nimg, H, W, C, stride,
                         OH, OW, (const float*)in, (const float*)dout, (float*)din, relu_mask));
    SW_CHECK_LAUNCH();
    return 0;
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(maxpool_bwd_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, nimg, H, W, C, stride,
                       OH, OW, (const unsigned short*)in, (const unsigned short*)dout, (unsigned short*)din, relu_mask),
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, nimg, H, W, C, stride, OH, OW,
                       (const float*)in, (const float*)dout, (float*)din, relu_mask));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_conv_weight_prep(int dtype, int mode, int Cout, int Cin, int cin_pad, const float* w, void* wk,
                                   hipStream_t stream) {
  SW_ENTER();
  if (mode != 0 && mode != 1) return -3;
  if (mode == 1 && cin_pad != Cin) return -3;
  const long n = mode == 0 ? (long)Cout * 9 * cin_pad : (long)Cin * 9 * Cout;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(conv_weight_prep_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, mode, Cout, Cin,
                       cin_pad, w, (unsigned short*)wk),
    hipLaunchKernelGGL(conv_weight_prep_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, mode, Cout, Cin, cin_pad,
                       w, (float*)wk));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_convert_2d(int dtype, int rows, int cols, const float* src, long ld_src, void* dst, long ld_dst,
                             hipStream_t stream) {
  SW_ENTER();
  const long n = (long)rows * cols;
  if (n <= 0) return 0;
  const long esz = dtype == SW_BF16 ? 2 : 4;
  if ((cols % 4) == 0 && (ld_src % 4) == 0 && (ld_dst % 4) == 0 && (((uintptr_t)src) & 15) == 0 &&
      (((uintptr_t)dst) & (4 * esz - 1)) == 0) {
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(convert_2d_vec4_kernel<unsigned short>, dim3(grid_for(n / 4)), dim3(256), 0, stream, rows, cols / 4,
                         src, ld_src, (unsigned short*)dst, ld_dst),
      hipLaunchKernelGGL(convert_2d_vec4_kernel<float>, dim3(grid_for(n / 4)), dim3(256), 0, stream, rows, cols / 4, src,
                         ld_src, (float*)dst, ld_dst));
    SW_CHECK_LAUNCH();
    return 0;
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(convert_2d_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, rows, cols, src, ld_src,
                       (unsigned short*)dst, ld_dst),
    hipLaunchKernelGGL(convert_2d_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, rows, cols, src, ld_src,
                       (float*)dst, ld_dst));
  SW_CHECK_LAUNCH();
  return 0;
}

// f32 -> three bf16 pieces a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (both differences exact in f32), written as
// the operand of the six-product GEMM of sw_split_bf16x3: `side` 0 lays the pieces out as [a1 | a1 | a2 | a1 | a2 | a3], side 1 as
// [b1 | b2 | b1 | b3 | b2 | b1], block p of the K-concatenated operand.  along_rows = 0: K runs along the columns (block p = columns
// [p * cols, (p + 1) * cols) of every row); 1: K runs along the rows (block p = rows [p * rows, (p + 1) * rows)).
__global__ __launch_bounds__(256) void split_bf16x3_kernel(int rows, int cols4, const float* __restrict__ src, long ld_src,
                                                           unsigned short* __restrict__ dst, long ld_dst, int side, int along_rows) {
  const long n = (long)rows * cols4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols4), c = (int)(i - (long)r * cols4) * 4;
    const f32x4 a = *(const f32x4*)(src + (long)r * ld_src + c);
    u32x2 pc[3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned short q[3][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float v = a[2 * h + e];
        const unsigned short b1 = f32_to_bf16_bits(v);
        const float r1 = __fsub_rn(v, bf16_bits_to_f32(b1));
        const unsigned short b2 = f32_to_bf16_bits(r1);
        const float r2 = __fsub_rn(r1, bf16_bits_to_f32(b2));
        q[0][e] = b1; q[1][e] = b2; q[2][e] = f32_to_bf16_bits(r2);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) pc[k][h] = (unsigned)q[k][0] | ((unsigned)q[k][1] << 16);
    }
    const int patA[6] = {0, 0, 1, 0, 1, 2}, patB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int k = side == 0 ? patA[p] : patB[p];
      const long o = along_rows ? ((long)p * rows + r) * ld_dst + c : (long)r * ld_dst + (long)p * cols4 * 4 + c;
      *(u32x2*)(dst + o) = pc[k];
    }
  }
}

extern "C" int sw_split_bf16x3(int rows, int cols, const float* src, long ld_src, void* dst, long ld_dst, int side, int along_rows,
                               hipStream_t stream) {
  SW_ENTER();
  if (rows <= 0 || cols <= 0) return 0;
  if ((cols % 4) || (ld_src % 4) || (ld_dst % 4) || (((uintptr_t)src) & 15) || (((uintptr_t)dst) & 7) || (side != 0 && side != 1)) return -5;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(grid_for((long)rows * (cols / 4))), dim3(256), 0, stream, rows, cols / 4, src, ld_src,
                     (unsigned short*)dst, ld_dst, side, along_rows ? 1 : 0);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_to_f32(int dtype, long n, const void* src, float* dst, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(to_f32_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, n, (const unsigned short*)src, dst),
    hipLaunchKernelGGL(to_f32_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, n, (const float*)src, dst));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_nchw_to_nhwc(int dtype, int N, int C, int H, int W, int cpad, const float* in, void* out,
                               hipStream_t stream) {
  SW_ENTER();
  const long n = (long)N * H * W * cpad;
  if (n <= 0) return 0;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, N, C, H, W, cpad, in, (unsigned short*)out),
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, N, C, H, W, cpad, in, (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_relu_bwd_out(int dtype, long n, const void* ref, const void* grad, void* out, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  {
    const long V = dtype == SW_BF16 ? 8 : 4;
    if ((n % V) == 0 && ((((uintptr_t)ref | (uintptr_t)grad | (uintptr_t)out) & 15) == 0)) {
      const long nv = n / V;
      DISPATCH_T(dtype,
        hipLaunchKernelGGL(relu_bwd_vec_kernel<unsigned short>, dim3(grid_for(nv)), dim3(256), 0, stream, nv, (const unsigned short*)ref, (const unsigned short*)grad, (unsigned short*)out),
        hipLaunchKernelGGL(relu_bwd_vec_kernel<float>, dim3(grid_for(nv)), dim3(256), 0, stream, nv, (const float*)ref, (const float*)grad, (float*)out));
      SW_CHECK_LAUNCH();
      return 0;
    }
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(relu_bwd_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, n, (const unsigned short*)ref, (const unsigned short*)grad, (unsigned short*)out),
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, n, (const float*)ref, (const float*)grad, (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_relu_bwd(int dtype, long n, const void* ref, void* grad, hipStream_t stream) {
  return sw_relu_bwd_out(dtype, n, ref, grad, grad, stream);
}

extern "C" int sw_scale_cols(int dtype, int M, int N, const float* in, long ld_in, const float* colscale, void* out,
                             long ld_out, hipStream_t stream) {
  SW_ENTER();
  const long n = (long)M * N;
  if (n <= 0) return 0;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(scale_cols_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, M, N, in, ld_in, colscale, (unsigned short*)out, ld_out),
    hipLaunchKernelGGL(scale_cols_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, M, N, in, ld_in, colscale, (float*)out, ld_out));
  SW_CHECK_LAUNCH();
  return 0;
}

namespace {
// (slabs, chunks) of the workspace form: ~512 workgroups, >= 32 rows per chunk and <= 128 partial rows
void colsum_ws_geometry(int dtype, int M, int N, int* slabs, int* chunks, int* rpc) {
  const int slab_cols = dtype == SW_BF16 ? 256 : 128;
  *slabs = (N + slab_cols - 1) / slab_cols;
  int c = 512 / *slabs;
  c = c < 1 ? 1 : (c > 128 ? 128 : c);
  int r = (M + c - 1) / c;
  if (r < 32) r = 32;
  *rpc = r;
  *chunks = (M + r - 1) / r;
}
}  // namespace

extern "C" long sw_colsum_workspace_floats(int dtype, int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  int slabs, chunks, rpc;
  colsum_ws_geometry(dtype, M, N, &slabs, &chunks, &rpc);
  return (long)chunks * N;
}

// the two halves of the workspace form on their own: several matrices (the views of a backbone pass running on different
// streams) write their partial rows into ONE workspace back to back, a single fold then sums all of them in fixed order
extern "C" int sw_colsum_partial(int dtype, int M, int N, const void* X, long ld, float* workspace, hipStream_t stream) {
  SW_ENTER();
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  const int vec = dtype == SW_BF16 ? 8 : 4;
  if (!workspace || M <= 0 || N <= 0 || (N % vec) || (ld % vec) || (((uintptr_t)X) & 15)) return -5;
  int slabs, chunks, rpc;
  colsum_ws_geometry(dtype, M, N, &slabs, &chunks, &rpc);
  dim3 grid(slabs, chunks);
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(colsum_ws_kernel<unsigned short>, grid, dim3(256), 0, stream, M, N, rpc, (const unsigned short*)X, ld,
                       workspace),
    hipLaunchKernelGGL(colsum_ws_kernel<float>, grid, dim3(256), 0, stream, M, N, rpc, (const float*)X, ld, workspace));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_colsum_partial_multi(int dtype, int n, const sw_colsum_part_desc* parts, hipStream_t stream) {
  SW_ENTER();
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  const int vec = dtype == SW_BF16 ? 8 : 4;
  for (int base = 0; base < n; base += CPART_MAX) {
    ColsumPartMulti f = {};
    f.n = n - base < CPART_MAX ? n - base : CPART_MAX;
    long wgs = 0;
    for (int i = 0; i < f.n; ++i) {
      const sw_colsum_part_desc& q = parts[base + i];
      if (!q.workspace || q.M <= 0 || q.N <= 0 || (q.N % vec) || (q.ld % vec) || (((uintptr_t)q.X) & 15)) return -5;
      int slabs, chunks, rpc;
      colsum_ws_geometry(dtype, q.M, q.N, &slabs, &chunks, &rpc);
      f.M[i] = q.M; f.N[i] = q.N; f.rpc[i] = rpc; f.slabs[i] = slabs; f.ld[i] = q.ld; f.X[i] = q.X; f.ws[i] = q.workspace;
      f.first_wg[i] = (int)wgs;
      wgs += (long)slabs * chunks;
      if (wgs > 2147483647L) return -6;
    }
    f.first_wg[f.n] = (int)wgs;
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(colsum_ws_multi_kernel<unsigned short>, dim3((unsigned)wgs), dim3(256), 0, stream, f),
      hipLaunchKernelGGL(colsum_ws_multi_kernel<float>, dim3((unsigned)wgs), dim3(256), 0, stream, f));
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int sw_colsum_fold(int N, int n_partial_rows, const float* workspace, float* out, hipStream_t stream) {
  SW_ENTER();
  if (N <= 0 || n_partial_rows < 1) return -5;
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, N, n_partial_rows, workspace, out, 0);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_colsum_fold_multi(int n, const sw_colsum_fold_desc* folds, hipStream_t stream) {
  SW_ENTER();
  for (int base = 0; base < n; base += CFOLD_MAX) {
    ColsumFoldMulti f = {};
    f.n = n - base < CFOLD_MAX ? n - base : CFOLD_MAX;
    int wgs = 0;
    for (int i = 0; i < f.n; ++i) {
      const sw_colsum_fold_desc& q = folds[base + i];
      if (q.N <= 0 || q.n_partial_rows < 1) return -5;
      f.N[i] = q.N; f.rows[i] = q.n_partial_rows; f.ws[i] = q.workspace; f.out[i] = q.out;
      f.first_wg[i] = wgs;
      wgs += (q.N + 15) / 16;
    }
    f.first_wg[f.n] = wgs;
    hipLaunchKernelGGL(colsum_fold_multi_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, f);
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int sw_colsum(int dtype, int M, int N, const void* X, long ld, float* out, float* workspace, hipStream_t stream) {
  return sw_colsum_acc(dtype, M, N, X, ld, out, workspace, 0, stream);
}

extern "C" int sw_colsum_acc(int dtype, int M, int N, const void* X, long ld, float* out, float* workspace, int accumulate,
                             hipStream_t stream) {
  SW_ENTER();
  if (N <= 0) return 0;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  const int vec = dtype == SW_BF16 ? 8 : 4;
  const bool ws_form = workspace && M > 0 && (N % vec) == 0 && (ld % vec) == 0 && (((uintptr_t)X) & 15) == 0;
  if (ws_form) {
    const int rc = sw_colsum_partial(dtype, M, N, X, ld, workspace, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, N, (int)(sw_colsum_workspace_floats(dtype, M, N) / N),
                       workspace, out, accumulate);
    SW_CHECK_LAUNCH();
    return 0;
  }
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
  }
  if (M <= 0) return 0;
  const int slabs = (N + 63) / 64;
  int chunks = (2048 + slabs - 1) / slabs;                 // aim at ~2048 workgroups (256 CUs x 8)
  int rpc = (M + chunks - 1) / chunks;
  if (rpc < 32) rpc = 32;
  chunks = (M + rpc - 1) / rpc;
  dim3 grid(slabs, chunks);
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(colsum_kernel<unsigned short>, grid, dim3(256), 0, stream, M, N, rpc, (const unsigned short*)X, ld, out),
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, M, N, rpc, (const float*)X, ld, out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_fill_zero(void* p, long bytes, hipStream_t stream) {
  SW_ENTER();
  if (bytes <= 0) return 0;
  return (int)hipMemsetAsync(p, 0, (size_t)bytes, stream);
}

extern "C" int sw_dropout_mask(uint8_t* keep, long n, uint64_t seed, uint64_t offset, float p, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, stream, keep, n, seed, offset, p);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_sgd_momentum_step(float* param, const float* grad, float* buf, long n, float lr, float momentum,
                                    float weight_decay, int first_step, float grad_scale, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, param, grad, buf, n, lr, momentum, weight_decay,
                     first_step, grad_scale);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_convert_2d_t(int dtype, int rows, int cols, const float* src, long ld_src, void* dst, long ld_dst,
                               hipStream_t stream) {
  SW_ENTER();
  if (rows <= 0 || cols <= 0) return 0;
  if ((rows % 64) || (cols % 64) || (ld_src % 4) || (((uintptr_t)src) & 15)) return -5;
  dim3 grid(cols / 64, rows / 64);
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(sgd_tile_t_kernel<unsigned short>, grid, dim3(256), 0, stream, rows, cols, (float*)src, (const float*)nullptr,
                       (float*)nullptr, ld_src, 0.f, 0.f, 0, 0.f, 0.f, (unsigned short*)nullptr, 0L, (unsigned short*)dst, ld_dst, (const float*)nullptr),
    hipLaunchKernelGGL(sgd_tile_t_kernel<float>, grid, dim3(256), 0, stream, rows, cols, (float*)src, (const float*)nullptr,
                       (float*)nullptr, ld_src, 0.f, 0.f, 0, 0.f, 0.f, (float*)nullptr, 0L, (float*)dst, ld_dst, (const float*)nullptr));
  SW_CHECK_LAUNCH();
  return 0;
}

// the tiled update kernel on a column block of a 2D parameter (gemm.hip: the peeled tail columns of a GEMM with a fused SGD epilogue)
int sw_sgd_tile_t_block(int rows, int cols, float* param, const float* grad, float* buf, long ld_src, float lr, float wd, int first,
                        float mom, float gscale, unsigned short* st0, long ld0, unsigned short* st1, long ld1, const float* hyper,
                        hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if ((rows % 64) || (cols % 64) || !st1) return -5;
  hipLaunchKernelGGL(sgd_tile_t_kernel<unsigned short>, dim3(cols / 64, rows / 64), dim3(256), 0, stream, rows, cols, param, grad, buf,
                     ld_src, lr, wd, first, mom, gscale, st0, ld0, st1, ld1, hyper);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_sgd_multi(int n_tensors, const sw_sgd_tensor* tensors, float momentum, float grad_scale,
                            hipStream_t stream) {
  SW_ENTER();
  for (int t0 = 0; t0 < n_tensors; t0 += SW_SGD_MAX_TENSORS) {
    SgdBatch b;
    b.n = 0;
    int blocks = 0;
    for (int i = t0; i < n_tensors && i < t0 + SW_SGD_MAX_TENSORS; ++i) {
      const sw_sgd_tensor& d = tensors[i];
      if (d.n <= 0) continue;
      if (d.stage_kind < 0 || d.stage_kind > 3) return -1;
      if (d.stage_kind == 3) {                 // own tiled launch (row-major + transposed copies)
        const long rows = d.n / (d.d0 > 0 ? d.d0 : 1);
        if (d.d0 <= 0 || (d.d0 % 64) || (rows % 64) || rows * d.d0 != d.n || d.stage1 == nullptr) return -5;
        if (d.stage_dtype != SW_BF16 && d.stage_dtype != SW_F32) return -1;
        dim3 grid(d.d0 / 64, (unsigned)(rows / 64));
        if (d.stage_dtype == SW_BF16)
          hipLaunchKernelGGL(sgd_tile_t_kernel<unsigned short>, grid, dim3(256), 0, stream, (int)rows, d.d0, d.param, d.grad,
                             d.momentum_buf, (long)d.d0, d.lr, d.weight_decay, d.first_step, momentum, grad_scale,
                             (unsigned short*)d.stage0, d.ld0, (unsigned short*)d.stage1, d.ld1, d.hyper_dev);
        else
          hipLaunchKernelGGL(sgd_tile_t_kernel<float>, grid, dim3(256), 0, stream, (int)rows, d.d0, d.param, d.grad,
                             d.momentum_buf, (long)d.d0, d.lr, d.weight_decay, d.first_step, momentum, grad_scale,
                             (float*)d.stage0, d.ld0, (float*)d.stage1, d.ld1, d.hyper_dev);
        SW_CHECK_LAUNCH();
        continue;
      }
      if (d.stage_kind && d.stage_dtype != SW_BF16 && d.stage_dtype != SW_F32) return -1;
      if (d.stage_kind == 1 && (d.d0 <= 0 || d.stage0 == nullptr)) return -5;
      if (d.stage_kind == 2 && (long)d.d0 * d.d1 * 9 != d.n) return -5;
      const bool conv_tiles = d.stage_kind == 2 && (d.d0 % 32) == 0 && (d.d1 % 32) == 0;
      const long nb = conv_tiles ? (long)(d.d0 / 32) * (d.d1 / 32) : (d.n + SGD_CHUNK - 1) / SGD_CHUNK;
      if (nb > (1L << 30)) return -6;
      b.t[b.n] = d;
      b.block_start[b.n] = blocks;
      blocks += (int)nb;
      ++b.n;
    }
    b.block_start[b.n] = blocks;
    if (b.n == 0) continue;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(blocks), dim3(256), 0, stream, b, momentum, grad_scale);
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int sw_mean_views(int V, long n, const float* in, float* out, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  hipLaunchKernelGGL(mean_views_kernel, dim3(grid_for(n)), dim3(256), 0, stream, V, n, in, out);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_loss_finalize(int n_losses, int V, int n_images, const float* loss_view, float* out, float* total,
                                hipStream_t stream) {
  SW_ENTER();
  if (n_losses > 64 || n_losses < 1 || V < 1 || n_images < 1) return -6;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, n_losses, V, n_images, loss_view, out, total);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_scale_cols_loss(int dtype, int M, int N, int n_valid, const float* in, long ld_in, const float* g_losses,
                                  const float* g_total, const int32_t* col_to_loss, float mul, void* out, long ld_out,
                                  hipStream_t stream) {
  SW_ENTER();
  const long n = (long)M * N;
  if (n <= 0) return 0;
  if (!col_to_loss && g_losses) return -5;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(scale_cols_loss_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, stream, M, N, n_valid, in, ld_in,
                       g_losses, g_total, col_to_loss, mul, (unsigned short*)out, ld_out),
    hipLaunchKernelGGL(scale_cols_loss_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, M, N, n_valid, in, ld_in,
                       g_losses, g_total, col_to_loss, mul, (float*)out, ld_out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_resize_pass_u8(int C, int H, int W, long in_ld, long in_plane, int out_size, int horizontal, const uint8_t* in,
                                 const int32_t* bounds, const int32_t* kk, int ksize, uint8_t* out, uint8_t* out_flip,
                                 hipStream_t stream) {
  SW_ENTER();
  if (C < 1 || H < 1 || W < 1 || out_size < 1 || ksize < 1 || in_ld < W || in_plane < (long)(H - 1) * in_ld + W) return -5;
  const int OH = horizontal ? H : out_size, OW = horizontal ? out_size : W;
  const long n = (long)C * OH * OW;
  hipLaunchKernelGGL(resize_pass_u8_kernel, dim3(grid_for(n)), dim3(256), 0, stream, C, H, W, in_ld, in_plane, OH, OW, horizontal, in,
                     bounds, kk, ksize, out, out_flip);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_color_jitter_u8(int H, int W, int mode, const uint8_t* in, float w_bright, double src_w_sat, float w_sat,
                                  uint8_t* out, uint8_t* out_flip, hipStream_t stream) {
  SW_ENTER();
  if (H < 1 || W < 1 || (mode & ~3)) return -5;
  hipLaunchKernelGGL(color_jitter_u8_kernel, dim3(grid_for((long)H * W)), dim3(256), 0, stream, H, W, mode, in, w_bright, src_w_sat,
                     w_sat, out, out_flip);
  SW_CHECK_LAUNCH();
  return 0;
}

namespace {
// dst[c][r] = src[r][c] through a 64 x 64 LDS tile: both sides move 16-byte row pieces (>= 128-byte runs per row)
template <typename T>
__global__ __launch_bounds__(256) void transpose_2d_kernel(int rows, int cols, const T* __restrict__ src, long ld_src,
                                                           T* __restrict__ dst, long ld_dst) {
  constexpr int V = 16 / (int)sizeof(T);                 // elements per 16-byte piece
  __shared__ T tile[64][64 + V];                         // [source column][source row], padded rows
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * (64 / V); i += 256) {
    const int r = i / (64 / V), cv = (i % (64 / V)) * V;
    T v[V];
    if (r0 + r < rows && c0 + cv + V <= cols) *(u32x4*)v = *(const u32x4*)(src + (long)(r0 + r) * ld_src + c0 + cv);
    else
      for (int t = 0; t < V; ++t) v[t] = (r0 + r < rows && c0 + cv + t < cols) ? src[(long)(r0 + r) * ld_src + c0 + cv + t] : (T)0;
#pragma unroll
    for (int t = 0; t < V; ++t) tile[cv + t][r] = v[t];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * (64 / V); i += 256) {
    const int c = i / (64 / V), rv = (i % (64 / V)) * V;
    if (c0 + c >= cols) continue;
    T* d = dst + (long)(c0 + c) * ld_dst + r0 + rv;
    if (r0 + rv + V <= rows) *(u32x4*)d = *(const u32x4*)&tile[c][rv];
    else
      for (int t = 0; t < V; ++t) if (r0 + rv + t < rows) d[t] = tile[c][rv + t];
  }
}
}  // namespace

extern "C" int sw_transpose_2d(int dtype, int rows, int cols, const void* src, long ld_src, void* dst, long ld_dst,
                               hipStream_t stream) {
  SW_ENTER();
  if (rows <= 0 || cols <= 0) return 0;
  const int vec = dtype == SW_BF16 ? 8 : 4;
  if ((ld_src % vec) || (ld_dst % vec) || (((uintptr_t)src) & 15) || (((uintptr_t)dst) & 15)) return -5;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(transpose_2d_kernel<unsigned short>, grid, dim3(256), 0, stream, rows, cols, (const unsigned short*)src, ld_src,
                       (unsigned short*)dst, ld_dst),
    hipLaunchKernelGGL(transpose_2d_kernel<float>, grid, dim3(256), 0, stream, rows, cols, (const float*)src, ld_src, (float*)dst, ld_dst));
  SW_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- Stage-3 (Unbiased Teacher) step pieces
namespace {
constexpr int EMA_MAX = 48;
struct EmaBatch {
  float* teacher[EMA_MAX];
  const float* student[EMA_MAX];
  long n[EMA_MAX];
  int block_start[EMA_MAX + 1];
  int count;
};
// teacher = student * (1 - keep) + teacher * keep  (ubteacher/engine/trainer.py:588-604 _update_teacher_model), the products and
// the sum rounded as torch rounds them (two multiplies, one add: no fused multiply-add)
__global__ __launch_bounds__(256) void ema_multi_kernel(EmaBatch b, float keep, float one_minus_keep) {
  int t = 0;
  while (t + 1 < b.count && (int)blockIdx.x >= b.block_start[t + 1]) ++t;
  const long base = (long)(blockIdx.x - b.block_start[t]) * 4096;
  float* te = b.teacher[t];
  const float* st = b.student[t];
  const long n = b.n[t];
  for (long i = base + threadIdx.x; i < base + 4096 && i < n; i += 256)
    te[i] = __fadd_rn(__fmul_rn(st[i], one_minus_keep), __fmul_rn(te[i], keep));
}

// keep[i] = score[i] > thres (and class[i] in the image's label set when given); stable compaction in input order by one
// workgroup (the teacher emits <= a few hundred detections per image)  (trainer.py:361-400 threshold_bbox)
__global__ __launch_bounds__(1024) void threshold_select_kernel(int n, const float* __restrict__ scores, const int* __restrict__ classes,
                                                                const float* __restrict__ boxes, float thres,
                                                                const int* __restrict__ allowed, int n_allowed,
                                                                int* __restrict__ out_count, float* __restrict__ out_boxes,
                                                                int* __restrict__ out_classes, float* __restrict__ out_scores,
                                                                int* __restrict__ out_index) {
  __shared__ int s_scan[1024];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    bool ok = i < n && scores[i] > thres;
    if (ok && allowed) {
      bool in = false;
      for (int a = 0; a < n_allowed; ++a) in = in || (allowed[a] == classes[i]);
      ok = in;
    }
    s_scan[threadIdx.x] = ok ? 1 : 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int add = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += add;
      __syncthreads();
    }
    if (ok) {
      const int o = s_base + s_scan[threadIdx.x] - 1;
      out_boxes[4 * o + 0] = boxes[4 * i + 0]; out_boxes[4 * o + 1] = boxes[4 * i + 1];
      out_boxes[4 * o + 2] = boxes[4 * i + 2]; out_boxes[4 * o + 3] = boxes[4 * i + 3];
      if (out_classes && classes) out_classes[o] = classes[i];
      out_scores[o] = scores[i];
      if (out_index) out_index[o] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_scan[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_count[0] = s_base;
}
}  // namespace

extern "C" int sw_ema_multi(int n_tensors, float* const* teacher, const float* const* student, const long* numel, double keep_rate,
                            hipStream_t stream) {
  SW_ENTER();
  for (int t0 = 0; t0 < n_tensors; t0 += EMA_MAX) {
    EmaBatch b = {};
    int blocks = 0;
    for (int i = t0; i < n_tensors && i < t0 + EMA_MAX; ++i) {
      if (numel[i] <= 0) continue;
      b.teacher[b.count] = teacher[i]; b.student[b.count] = student[i]; b.n[b.count] = numel[i];
      b.block_start[b.count] = blocks;
      blocks += (int)((numel[i] + 4095) / 4096);
      ++b.count;
    }
    b.block_start[b.count] = blocks;
    if (blocks == 0) continue;
    // the two python-float scalars of `student * (1 - keep) + teacher * keep`, each rounded to f32 as torch rounds a scalar operand
    hipLaunchKernelGGL(ema_multi_kernel, dim3(blocks), dim3(256), 0, stream, b, (float)keep_rate, (float)(1.0 - keep_rate));
    SW_CHECK_LAUNCH();
  }
  return 0;
}

// ---- weighted sum of scalar losses (and its backward): one launch each instead of n multiplications + n additions
namespace {
constexpr int WS_MAX = 32;
struct ScalarList { const float* p[WS_MAX]; float w[WS_MAX]; int n; };
__global__ void weighted_sum_kernel(ScalarList a, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < a.n; ++i) {                                   // ((0 + w0 v0) + w1 v1) + ...: Python's sum() over the weighted dict
    const float v = __fmul_rn(*a.p[i], a.w[i]);
    out[i] = v;
    s = i == 0 ? v : __fadd_rn(s, v);
  }
  out[a.n] = s;
}
__global__ void scale_scalars_kernel(ScalarList a, const float* __restrict__ g, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (blockIdx.x == 0 && i < a.n) out[i] = __fmul_rn(g[0], a.w[i]);
}
}  // namespace

extern "C" int sw_weighted_sum(int n, const float* const* values, const float* weights, float* out, hipStream_t stream) {
  SW_ENTER();
  if (n < 1 || n > WS_MAX) return -5;
  ScalarList a = {};
  a.n = n;
  for (int i = 0; i < n; ++i) { a.p[i] = values[i]; a.w[i] = weights[i]; }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, stream, a, out);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_scale_scalars(int n, const float* g, const float* weights, float* out, hipStream_t stream) {
  SW_ENTER();
  if (n < 1 || n > WS_MAX) return -5;
  ScalarList a = {};
  a.n = n;
  for (int i = 0; i < n; ++i) a.w[i] = weights[i];
  hipLaunchKernelGGL(scale_scalars_kernel, dim3(1), dim3(64), 0, stream, a, g, out);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_threshold_select(int n, const float* scores, const int32_t* classes, const float* boxes, float thres,
                                   const int32_t* allowed_classes, int n_allowed, int32_t* out_count, float* out_boxes,
                                   int32_t* out_classes, float* out_scores, int32_t* out_index, hipStream_t stream) {
  SW_ENTER();
  if (n < 0) return -5;
  hipLaunchKernelGGL(threshold_select_kernel, dim3(1), dim3(1024), 0, stream, n, scores, classes, boxes, thres, allowed_classes,
                     n_allowed, out_count, out_boxes, out_classes, out_scores, out_index);
  SW_CHECK_LAUNCH();
  return 0;
}

namespace { __global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) { *c += inc; } }

extern "C" int sw_counter_add(uint64_t* counter, uint64_t increment, hipStream_t stream) {
  SW_ENTER();
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, stream, (unsigned long long*)counter, (unsigned long long)increment);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_pack_views(int R, const float* const* box_ptrs4, const float* const* obj_ptrs4, float* boxes, float* obj,
                             float* rois, hipStream_t stream) {
  SW_ENTER();
  if (R <= 0) return 0;
  PackViewsArgs a;
  for (int v = 0; v < 4; ++v) { a.box[v] = box_ptrs4[v]; a.obj[v] = obj_ptrs4[v]; }
  hipLaunchKernelGGL(pack_views_kernel, dim3((4 * R + 255) / 256), dim3(256), 0, stream, R, a, boxes, obj, rois);
  SW_CHECK_LAUNCH();
  return 0;
}

// ---- several device-to-device copies in one launch (the step's input staging: 4 images + 4 x (boxes, objectness) into the
// captured graph's static buffers was 12 copy-engine launches, ~75 us of stream time for 3 MB)
namespace {
constexpr int COPY_MAX = 16;
struct CopyArgs { const char* src[COPY_MAX]; char* dst[COPY_MAX]; long bytes[COPY_MAX]; };
__global__ __launch_bounds__(256) void copy_multi_kernel(CopyArgs a) {
  const int t = blockIdx.y;
  const char* __restrict__ src = a.src[t]; char* __restrict__ dst = a.dst[t];
  const long n = a.bytes[t];
  const long stride = (long)gridDim.x * blockDim.x, first = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
    const long nv = n >> 4;
    for (long i = first; i < nv; i += stride) ((u32x4*)dst)[i] = ((const u32x4*)src)[i];
    for (long i = (nv << 4) + first; i < n; i += stride) dst[i] = src[i];
  } else {
    for (long i = first; i < n; i += stride) dst[i] = src[i];
  }
}
}  // namespace

extern "C" int sw_copy_multi(int n, const sw_copy_desc* copies, hipStream_t stream) {
  SW_ENTER();
  for (int base = 0; base < n; base += COPY_MAX) {
    CopyArgs a = {};
    const int m = n - base < COPY_MAX ? n - base : COPY_MAX;
    long longest = 0;
    for (int i = 0; i < m; ++i) {
      a.src[i] = (const char*)copies[base + i].src; a.dst[i] = (char*)copies[base + i].dst; a.bytes[i] = copies[base + i].bytes;
      if (a.bytes[i] < 0 || (a.bytes[i] > 0 && (!a.src[i] || !a.dst[i]))) return -1;
      longest = a.bytes[i] > longest ? a.bytes[i] : longest;
    }
    if (longest == 0) continue;
    long blocks = (longest / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
    hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, stream, a);
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" const char* sw_version(void) { return "soswsod-hip 0.1 (gfx950)"; }
