// Kernels of the Stage-3 detector (the ResNet-50-FPN Faster R-CNN of the Unbiased-Teacher step) that the
// Stage-1 path did not need.  Dense work stays on the MFMA kernels of gemm.hip / conv_direct.hip (1x1 convolutions = sw_gemm on
// NHWC pixels, 3x3 = sw_conv3x3_igemm, fc = sw_gemm); here: the frozen stem (7x7 stride-2 convolution with the FrozenBN affine +
// ReLU, 3x3 stride-2 max pooling), the stride-2 pixel subsampling in front of a 1x1 stride-2 convolution and its scatter
// backward, residual add + ReLU, FPN's nearest 2x upsample + add and its backward, ROIAlign (aligned, adaptive sampling grid)
// forward / backward over FPN levels, anchor-delta decoding, and the RPN's objectness / localisation losses with their unit
// gradients.  All NHWC, wave64, HBM-bound elementwise / gather work: coalesced channel-contiguous accesses, no MFMA.
#include <float.h>
#include <math.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {

inline int grid_for_n(long n, int block = 256) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 1048576 ? 1048576 : g));
}

// ------------------------------------------------------------------------------------------- input: normalise + pad
// img u8 [3][h][w] -> out [H][W][4] = (img - mean) / std inside the image, 0 in the padding and in channel 3
// (detectron2/modeling/meta_arch/rcnn.py:220-228 + structures/image_list.py:60-124)
template <typename T>
__global__ void preprocess_pad_kernel(int h, int w, int H, int W, const uint8_t* __restrict__ img, float m0, float m1, float m2,
                                      float s0, float s1, float s2, T* __restrict__ out) {
  const long n = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i - (long)y * W);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (y < h && x < w) {
      const long p = (long)y * w + x, hw = (long)h * w;
      v0 = ((float)img[p] - m0) / s0; v1 = ((float)img[hw + p] - m1) / s1; v2 = ((float)img[2 * hw + p] - m2) / s2;
    }
    T* o = out + i * 4;
    Elem<T>::store(o, v0); Elem<T>::store(o + 1, v1); Elem<T>::store(o + 2, v2); Elem<T>::store(o + 3, 0.f);
  }
}

// ------------------------------------------------------------------------------------------- stem: conv 7x7 s2 p3 + affine + ReLU
// in [N][H][W][4], w f32 [64][3][7][7] (OIHW), scale / bias f32 [64] (the FrozenBN fold), out [N][OH][OW][64], OH = (H + 6 - 7) / 2 + 1.
// Workgroup = an 8x8 tile of output pixels x 64 channels; thread = (pixel, 16-channel group): the 21x21x3 input patch and the
// 147x64 weights sit in LDS, the weight reads of a wave are broadcasts.  Frozen layer (FREEZE_AT 2): forward only.
template <typename T>
__global__ __launch_bounds__(256) void stem_conv7_kernel(int N, int H, int W, int OH, int OW, const T* __restrict__ in,
                                                         const float* __restrict__ w, const float* __restrict__ scale,
                                                         const float* __restrict__ bias, T* __restrict__ out) {
  __shared__ float s_w[147 * 64];
  __shared__ float s_in[21 * 21 * 3];
  const int tiles_x = (OW + 7) / 8, tiles_y = (OH + 7) / 8;
  const int t = blockIdx.x;
  const int n = t / (tiles_x * tiles_y), ty = (t / tiles_x) % tiles_y, tx = t % tiles_x;
  for (int i = threadIdx.x; i < 147 * 64; i += 256) {
    const int k = i >> 6, co = i & 63;                       // k = (ci * 7 + ky) * 7 + kx
    s_w[i] = w[co * 147 + k];
  }
  const int iy0 = ty * 16 - 3, ix0 = tx * 16 - 3;
  for (int i = threadIdx.x; i < 21 * 21; i += 256) {
    const int py = i / 21, px = i - py * 21, y = iy0 + py, x = ix0 + px;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const T* p = in + (((long)n * H + y) * W + x) * 4;
      v0 = Elem<T>::load(p); v1 = Elem<T>::load(p + 1); v2 = Elem<T>::load(p + 2);
    }
    s_in[i * 3] = v0; s_in[i * 3 + 1] = v1; s_in[i * 3 + 2] = v2;
  }
  __syncthreads();
  const int p = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int oy = p >> 3, ox = p & 7;
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  for (int ci = 0; ci < 3; ++ci)
    for (int ky = 0; ky < 7; ++ky)
      for (int kx = 0; kx < 7; ++kx) {
        const float v = s_in[((oy * 2 + ky) * 21 + ox * 2 + kx) * 3 + ci];
        const float* wr = s_w + ((ci * 7 + ky) * 7 + kx) * 64 + g * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
      }
  const int Y = ty * 8 + oy, X = tx * 8 + ox;
  if (Y < OH && X < OW) {
    T* o = out + (((long)n * OH + Y) * OW + X) * 64 + g * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) Elem<T>::store(o + c, fmaxf(fmaf(acc[c], scale[g * 16 + c], bias[g * 16 + c]), 0.f));
  }
}

// 3x3 max pooling, stride 2, padding 1 (resnet.py:358): out [N][OH][OW][C], OH = (H + 2 - 3) / 2 + 1
template <typename T>
__global__ void maxpool3x3s2_kernel(int N, int H, int W, int C, int OH, int OW, const T* __restrict__ in, T* __restrict__ out) {
  const long n = (long)N * OH * OW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int b = (int)(r / OH);
    float m = -FLT_MAX;
    for (int ky = 0; ky < 3; ++ky) {
      const int y = oy * 2 - 1 + ky;
      if (y < 0 || y >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int x = ox * 2 - 1 + kx;
        if (x < 0 || x >= W) continue;
        m = fmaxf(m, Elem<T>::load(in + (((long)b * H + y) * W + x) * C + c));
      }
    }
    Elem<T>::store(out + i, m);
  }
}

// 16-byte pieces (C a multiple of 8 bf16 / 4 f32 elements)
template <typename T>
__global__ void maxpool3x3s2_vec_kernel(int N, int H, int W, int C, int OH, int OW, const T* __restrict__ in, T* __restrict__ out) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cp = C / V;
  const long n = (long)N * OH * OW * cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cp) * V; long r = i / cp;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int b = (int)(r / OH);
    float m[V];
#pragma unroll
    for (int k = 0; k < V; ++k) m[k] = -FLT_MAX;
    for (int ky = 0; ky < 3; ++ky) {
      const int y = oy * 2 - 1 + ky;
      if (y < 0 || y >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int x = ox * 2 - 1 + kx;
        if (x < 0 || x >= W) continue;
        const u32x4 q = *(const u32x4*)(in + (((long)b * H + y) * W + x) * C + c);
        if (sizeof(T) == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            m[2 * k] = fmaxf(m[2 * k], __uint_as_float(q[k] << 16));
            m[2 * k + 1] = fmaxf(m[2 * k + 1], __uint_as_float(q[k] & 0xFFFF0000u));
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], __uint_as_float(q[k]));
        }
      }
    }
    u32x4 o;
    if (sizeof(T) == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (unsigned)f32_to_bf16_bits(m[2 * k]) | ((unsigned)f32_to_bf16_bits(m[2 * k + 1]) << 16);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = __float_as_uint(m[k]);
    }
    *(u32x4*)(out + ((((long)b * OH + oy) * OW + ox) * C + c)) = o;
  }
}

// ------------------------------------------------------------------------------------------- stride-2 subsample / scatter
// out[n][y][x][:] = in[n][2y][2x][:]  (the pixels a 1x1 stride-2 convolution reads; also FPN's p6 = max_pool(k 1, s 2) of p5)
template <typename T>
__global__ void subsample2_kernel(int N, int H, int W, int C, int OH, int OW, const T* __restrict__ in, T* __restrict__ out) {
  const long n = (long)N * OH * OW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const int b = (int)(r / OH);
    out[i] = in[(((long)b * H + oy * 2) * W + ox * 2) * C + c];
  }
}
// backward: out [N][H][W][C] = g at the even pixels, 0 elsewhere (fully written)
template <typename T>
__global__ void scatter2_kernel(int N, int H, int W, int C, int OH, int OW, const T* __restrict__ g, T* __restrict__ out) {
  const long n = (long)N * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H); const int b = (int)(r / H);
    T v = {};
    if (!(y & 1) && !(x & 1)) v = g[(((long)b * OH + (y >> 1)) * OW + (x >> 1)) * C + c];
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------- residual add + ReLU, FPN top-down
template <typename T>
__global__ void add_relu_kernel(long n, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int relu) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = Elem<T>::load(a + i) + Elem<T>::load(b + i);
    if (relu) v = fmaxf(v, 0.f);
    Elem<T>::store(out + i, v);
  }
}
// out [N][2h][2w][C] = lateral + nearest-neighbour 2x upsampling of top [N][h][w][C]  (fpn.py:142-144)
template <typename T>
__global__ void upsample2_add_kernel(int N, int h, int w, int C, const T* __restrict__ lateral, const T* __restrict__ top,
                                     T* __restrict__ out) {
  const long n = (long)N * 2 * h * 2 * w * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int x = (int)(r % (2 * w)); r /= (2 * w);
    const int y = (int)(r % (2 * h)); const int b = (int)(r / (2 * h));
    Elem<T>::store(out + i, Elem<T>::load(lateral + i) + Elem<T>::load(top + (((long)b * h + (y >> 1)) * w + (x >> 1)) * C + c));
  }
}
// the same with 16-byte pieces (C a multiple of 8 bf16 / 4 f32 elements, 16-byte aligned tensors): one thread = one piece of a pixel
template <typename T>
__global__ void upsample2_add_vec_kernel(int N, int h, int w, int C, const T* __restrict__ lateral, const T* __restrict__ top,
                                         T* __restrict__ out) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cp = C / V;
  const long n = (long)N * 2 * h * 2 * w * cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cp) * V; long r = i / cp;
    const int x = (int)(r % (2 * w)); r /= (2 * w);
    const int y = (int)(r % (2 * h)); const int b = (int)(r / (2 * h));
    const u32x4 a = *(const u32x4*)(lateral + i * V);
    const u32x4 t = *(const u32x4*)(top + (((long)b * h + (y >> 1)) * w + (x >> 1)) * C + c);
    u32x4 o;
    if (sizeof(T) == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(a[k] << 16) + __uint_as_float(t[k] << 16);
        const float hi = __uint_as_float(a[k] & 0xFFFF0000u) + __uint_as_float(t[k] & 0xFFFF0000u);
        o[k] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = __float_as_uint(__uint_as_float(a[k]) + __uint_as_float(t[k]));
    }
    *(u32x4*)(out + i * V) = o;
  }
}
// backward of the upsampling: out [N][h][w][C] = sum of the 2x2 block of g [N][2h][2w][C] (fixed order)
template <typename T>
__global__ void downsample2_sum_kernel(int N, int h, int w, int C, const T* __restrict__ g, T* __restrict__ out) {
  const long n = (long)N * h * w * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h); const int b = (int)(r / h);
    const T* p = g + (((long)b * 2 * h + 2 * y) * 2 * w + 2 * x) * C + c;
    const float v = (Elem<T>::load(p) + Elem<T>::load(p + C)) + (Elem<T>::load(p + (long)2 * w * C) + Elem<T>::load(p + (long)2 * w * C + C));
    Elem<T>::store(out + i, v);
  }
}

// the same with 16-byte pieces: (a + b) + (c + d) per element in f32, as above
template <typename T>
__global__ void downsample2_sum_vec_kernel(int N, int h, int w, int C, const T* __restrict__ g, T* __restrict__ out) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cp = C / V;
  const long n = (long)N * h * w * cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cp) * V; long r = i / cp;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h); const int b = (int)(r / h);
    const T* p = g + (((long)b * 2 * h + 2 * y) * 2 * w + 2 * x) * C + c;
    const u32x4 q0 = *(const u32x4*)p, q1 = *(const u32x4*)(p + C), q2 = *(const u32x4*)(p + (long)2 * w * C), q3 = *(const u32x4*)(p + (long)2 * w * C + C);
    u32x4 o;
    if (sizeof(T) == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = (__uint_as_float(q0[k] << 16) + __uint_as_float(q1[k] << 16)) + (__uint_as_float(q2[k] << 16) + __uint_as_float(q3[k] << 16));
        const float hi = (__uint_as_float(q0[k] & 0xFFFF0000u) + __uint_as_float(q1[k] & 0xFFFF0000u)) +
                         (__uint_as_float(q2[k] & 0xFFFF0000u) + __uint_as_float(q3[k] & 0xFFFF0000u));
        o[k] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = __float_as_uint((__uint_as_float(q0[k]) + __uint_as_float(q1[k])) + (__uint_as_float(q2[k]) + __uint_as_float(q3[k])));
    }
    *(u32x4*)(out + ((((long)b * h + y) * w + x) * C + c)) = o;
  }
}

// ------------------------------------------------------------------------------------------- ROIAlign (aligned = true)
// The arithmetic of uwsod/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:20-218 (= torchvision.ops.roi_align, which
// detectron2/layers/roi_align.py:56-64 calls): offset 0.5, ROI size not clamped, sampling grid ceil(roi / pooled) when
// sampling_ratio == 0, samples outside [-1, size] contribute 0, average over the grid.  Compiled with -ffp-contract=off.
struct AlignGeom { float start_h, start_w, bin_h, bin_w, count; int grid_h, grid_w, batch; };
__device__ __forceinline__ AlignGeom align_geom(const float* roi, float scale, int PH, int PW, int sampling_ratio) {
  AlignGeom g;
  g.batch = (int)roi[0];
  g.start_w = __fsub_rn(__fmul_rn(roi[1], scale), 0.5f);
  g.start_h = __fsub_rn(__fmul_rn(roi[2], scale), 0.5f);
  const float end_w = __fsub_rn(__fmul_rn(roi[3], scale), 0.5f), end_h = __fsub_rn(__fmul_rn(roi[4], scale), 0.5f);
  const float rw = __fsub_rn(end_w, g.start_w), rh = __fsub_rn(end_h, g.start_h);
  g.bin_h = __fdiv_rn(rh, (float)PH);
  g.bin_w = __fdiv_rn(rw, (float)PW);
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(__fdiv_rn(rh, (float)PH));
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(__fdiv_rn(rw, (float)PW));
  const int c = g.grid_h * g.grid_w;
  g.count = (float)(c > 1 ? c : 1);
  return g;
}
__device__ __forceinline__ bool align_corners(int H, int W, float y, float x, int* yl, int* xl, int* yh, int* xh, float* wgt) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
  const float ly = __fsub_rn(y, (float)y_low), lx = __fsub_rn(x, (float)x_low);
  const float hy = __fsub_rn(1.0f, ly), hx = __fsub_rn(1.0f, lx);
  wgt[0] = __fmul_rn(hy, hx); wgt[1] = __fmul_rn(hy, lx); wgt[2] = __fmul_rn(ly, hx); wgt[3] = __fmul_rn(ly, lx);
  *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;
  return true;
}
__device__ __forceinline__ float align_coord(float start, int p, float bin, int i, int grid) {
  // start + p * bin + (i + .5f) * bin / grid, left to right as the reference writes it
  return __fadd_rn(__fadd_rn(start, __fmul_rn((float)p, bin)), __fdiv_rn(__fmul_rn((float)i + .5f, bin), (float)grid));
}

// thread = (listed ROI, bin, channel): consecutive threads = consecutive channels -> coalesced NHWC reads.  Only the ROIs listed in
// `sel` (n_sel row indices into rois / out) are computed: the FPN pooler calls once per level with that level's ROIs.
// out [R][C][PH][PW] (the reference's flatten order: fc1.weight needs no permutation), row pitch ld.
template <typename T>
__global__ void roi_align_fwd_kernel(int H, int W, int C, int PH, int PW, float scale, int sampling_ratio,
                                     const T* __restrict__ feat, const float* __restrict__ rois, const int* __restrict__ sel,
                                     int n_sel, const int* __restrict__ n_sel_dev, T* __restrict__ out, long ld) {
  const long n = (long)(n_sel_dev ? min(n_sel_dev[0], n_sel) : n_sel) * PH * PW * C;     // device count: no host round trip for the level lists
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int pw = (int)(r % PW); r /= PW;
    const int ph = (int)(r % PH); const int s = (int)(r / PH);
    const int row = sel[s];
    const AlignGeom g = align_geom(rois + (long)row * 5, scale, PH, PW, sampling_ratio);
    const T* f = feat + (long)g.batch * H * W * C + c;
    float v = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float y = align_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float x = align_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
        int yl, xl, yh, xh; float wg[4];
        if (!align_corners(H, W, y, x, &yl, &xl, &yh, &xh, wg)) continue;
        // ((w1 v1 + w2 v2) + w3 v3) + w4 v4 added to the running sum, as ROIAlign_cpu.cpp:204-206 evaluates it
        const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(wg[0], Elem<T>::load(f + ((long)yl * W + xl) * C)),
                                                      __fmul_rn(wg[1], Elem<T>::load(f + ((long)yl * W + xh) * C))),
                                            __fmul_rn(wg[2], Elem<T>::load(f + ((long)yh * W + xl) * C))),
                                  __fmul_rn(wg[3], Elem<T>::load(f + ((long)yh * W + xh) * C)));
        v = __fadd_rn(v, t);
      }
    }
    Elem<T>::store(out + (long)row * ld + ((long)c * PH + ph) * PW + pw, __fdiv_rn(v, g.count));
  }
}
// backward: dfeat f32 [N][H][W][C] (caller zero-fills) += w * g / count by f32 atomics (ROIAlign_cpu.cpp:286-400).
// The bilinear weight of a sample factors into a row part and a column part, and validity / border clamping act per axis, so the
// contributions of a bin's grid_h x grid_w samples to pixel (Y, X) sum to (sum_iy wy_iy(Y)) * (sum_ix wx_ix(X)) * g / count: with the
// per-axis sums in registers a bin issues (rows touched) x (columns touched) <= (grid_h + 1)(grid_w + 1) atomics instead of
// 4 * grid_h * grid_w (grid 4 x 4: 25 for 64).  Sample spacing is bin / ceil(bin) <= 1 pixel, so the touched rows are consecutive.
// Grids above AXIS_MAX - 1 per axis (a ROI far too large for its level) take the sample-by-sample form.
constexpr int AXIS_MAX = 9;
struct AxisW { float w[AXIS_MAX]; int base; };
__device__ __forceinline__ AxisW align_axis_weights(float start, int p, float bin, int grid, int size) {
  AxisW a;
#pragma unroll
  for (int s = 0; s < AXIS_MAX; ++s) a.w[s] = 0.f;
  a.base = -1;
  for (int i = 0; i < grid; ++i) {
    float y = align_coord(start, p, bin, i, grid);
    if (y < -1.0f || y > (float)size) continue;
    if (y <= 0) y = 0;
    int lo = (int)y, hi;
    if (lo >= size - 1) { hi = lo = size - 1; y = (float)lo; } else hi = lo + 1;
    const float l = __fsub_rn(y, (float)lo), h = __fsub_rn(1.0f, l);
    if (a.base < 0) a.base = lo;
    const int k0 = lo - a.base, k1 = hi - a.base;
#pragma unroll
    for (int s = 0; s < AXIS_MAX; ++s) {                     // (static register indices: a dynamic one would put the array in scratch)
      if (s == k0) a.w[s] += h;
      if (s == k1) a.w[s] += l;
    }
  }
  return a;
}
template <typename T>
__global__ void roi_align_bwd_kernel(int H, int W, int C, int PH, int PW, float scale, int sampling_ratio,
                                     const T* __restrict__ gout, long ld, const float* __restrict__ rois,
                                     const int* __restrict__ sel, int n_sel, const int* __restrict__ n_sel_dev, float* __restrict__ dfeat) {
  const long n = (long)(n_sel_dev ? min(n_sel_dev[0], n_sel) : n_sel) * PH * PW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int pw = (int)(r % PW); r /= PW;
    const int ph = (int)(r % PH); const int s = (int)(r / PH);
    const int row = sel[s];
    const AlignGeom g = align_geom(rois + (long)row * 5, scale, PH, PW, sampling_ratio);
    const float go = Elem<T>::load(gout + (long)row * ld + ((long)c * PH + ph) * PW + pw);
    float* d = dfeat + (long)g.batch * H * W * C + c;
    if (g.grid_h < AXIS_MAX && g.grid_w < AXIS_MAX && g.bin_h <= (float)g.grid_h && g.bin_w <= (float)g.grid_w) {     // (sample spacing <= 1)
      const AxisW ay = align_axis_weights(g.start_h, ph, g.bin_h, g.grid_h, H);
      const AxisW ax = align_axis_weights(g.start_w, pw, g.bin_w, g.grid_w, W);
      if (ay.base < 0 || ax.base < 0) continue;
      const float gs = __fdiv_rn(go, g.count);
#pragma unroll
      for (int ry = 0; ry < AXIS_MAX; ++ry) {
        if (ay.w[ry] == 0.f) continue;
        const float gy = __fmul_rn(gs, ay.w[ry]);
        float* drow = d + (long)(ay.base + ry) * W * C;
#pragma unroll
        for (int rx = 0; rx < AXIS_MAX; ++rx)
          if (ax.w[rx] != 0.f) atomicAdd(drow + (long)(ax.base + rx) * C, __fmul_rn(gy, ax.w[rx]));
      }
      continue;
    }
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float y = align_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float x = align_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
        int yl, xl, yh, xh; float wg[4];
        if (!align_corners(H, W, y, x, &yl, &xl, &yh, &xh, wg)) continue;
        atomicAdd(d + ((long)yl * W + xl) * C, __fdiv_rn(__fmul_rn(go, wg[0]), g.count));
        atomicAdd(d + ((long)yl * W + xh) * C, __fdiv_rn(__fmul_rn(go, wg[1]), g.count));
        atomicAdd(d + ((long)yh * W + xl) * C, __fdiv_rn(__fmul_rn(go, wg[2]), g.count));
        atomicAdd(d + ((long)yh * W + xh) * C, __fdiv_rn(__fmul_rn(go, wg[3]), g.count));
      }
    }
  }
}

// Deterministic backward (round 6).  The f32 atomics above add in arrival order: two runs of one iteration differ in the last bits of
// the feature gradients, the student drifts, pseudo boxes jitter, position-keyed anchor sampling flips (tests had to allow 6e-2 on the
// pseudo RPN losses).  Here the SAME contributions are accumulated as 64-bit fixed-point integers — integer addition commutes, so the sum
// does not depend on the order: contribution v (|v| <= max|gout|: a bin hands out exactly its gradient, weights sum to count) is added as
// llrint(v * 2^40 / max|gout|); a pixel collects at most R * PH * PW * 2^40 < 2^63 for R < 170 000 ROIs; resolution 2^-40 of the largest
// gradient against f32's 2^-24.  acc [N][H][W][C] int64, zero-filled by the caller; fx_to_float_kernel turns it into the map (NaN / Inf in
// gout: absmax reports it, the whole map comes out NaN — the f32 form poisoned the pixels the ROI touched).
constexpr float FX_ONE = 1099511627776.0f;          // 2^40
template <typename T>
__global__ void roi_align_bwd_fx_kernel(int H, int W, int C, int PH, int PW, float scale, int sampling_ratio,
                                        const T* __restrict__ gout, long ld, const float* __restrict__ rois,
                                        const int* __restrict__ sel, int n_sel, const int* __restrict__ n_sel_dev,
                                        const float* __restrict__ absmax, unsigned long long* __restrict__ acc) {
  const float am = absmax[0];
  if (!(am > 0.f) || !(am <= 3.0e38f)) return;                      // zero gradient: nothing to add; NaN / Inf: the conversion reports it
  const float S = __fdiv_rn(FX_ONE, am);
  const long n = (long)(n_sel_dev ? min(n_sel_dev[0], n_sel) : n_sel) * PH * PW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long r = i / C;
    const int pw = (int)(r % PW); r /= PW;
    const int ph = (int)(r % PH); const int s = (int)(r / PH);
    const int row = sel[s];
    const AlignGeom g = align_geom(rois + (long)row * 5, scale, PH, PW, sampling_ratio);
    const float go = Elem<T>::load(gout + (long)row * ld + ((long)c * PH + ph) * PW + pw);
    unsigned long long* d = acc + (long)g.batch * H * W * C + c;
    auto add = [&](unsigned long long* p, float v) { atomicAdd(p, (unsigned long long)__float2ll_rn(__fmul_rn(v, S))); };
    if (g.grid_h < AXIS_MAX && g.grid_w < AXIS_MAX && g.bin_h <= (float)g.grid_h && g.bin_w <= (float)g.grid_w) {
      const AxisW ay = align_axis_weights(g.start_h, ph, g.bin_h, g.grid_h, H);
      const AxisW ax = align_axis_weights(g.start_w, pw, g.bin_w, g.grid_w, W);
      if (ay.base < 0 || ax.base < 0) continue;
      const float gs = __fdiv_rn(go, g.count);
#pragma unroll
      for (int ry = 0; ry < AXIS_MAX; ++ry) {
        if (ay.w[ry] == 0.f) continue;
        const float gy = __fmul_rn(gs, ay.w[ry]);
        unsigned long long* drow = d + (long)(ay.base + ry) * W * C;
#pragma unroll
        for (int rx = 0; rx < AXIS_MAX; ++rx)
          if (ax.w[rx] != 0.f) add(drow + (long)(ax.base + rx) * C, __fmul_rn(gy, ax.w[rx]));
      }
      continue;
    }
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float y = align_coord(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float x = align_coord(g.start_w, pw, g.bin_w, ix, g.grid_w);
        int yl, xl, yh, xh; float wg[4];
        if (!align_corners(H, W, y, x, &yl, &xl, &yh, &xh, wg)) continue;
        add(d + ((long)yl * W + xl) * C, __fdiv_rn(__fmul_rn(go, wg[0]), g.count));
        add(d + ((long)yl * W + xh) * C, __fdiv_rn(__fmul_rn(go, wg[1]), g.count));
        add(d + ((long)yh * W + xl) * C, __fdiv_rn(__fmul_rn(go, wg[2]), g.count));
        add(d + ((long)yh * W + xh) * C, __fdiv_rn(__fmul_rn(go, wg[3]), g.count));
      }
    }
  }
}
template <typename T>
__global__ void fx_to_float_kernel(long n, const long long* __restrict__ acc, const float* __restrict__ absmax, T* __restrict__ out) {
  const float am = absmax[0];
  const bool bad = !(am <= 3.0e38f);                                  // NaN or Inf somewhere in the gradient
  const double inv = (double)am / (double)FX_ONE;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    Elem<T>::store(out + i, bad ? __uint_as_float(0x7FC00000u) : (float)((double)acc[i] * inv));
}

// ------------------------------------------------------------------------------------------- box decoding, RPN losses
// Box2BoxTransform.apply_deltas (detectron2/modeling/box_regression.py:76-116): deltas [n][4], boxes [n][4] (row r of boxes is
// r % n_boxes: the anchors repeat over the images) -> out [n][4]
__global__ void decode_boxes_kernel(long n, long n_boxes, const float* __restrict__ deltas, long ld_d, const float* __restrict__ boxes,
                                    float wx, float wy, float ww, float wh, float clamp, float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float* b = boxes + (i % n_boxes) * 4;
    const float* d = deltas + i * ld_d;
    const float w = __fsub_rn(b[2], b[0]), h = __fsub_rn(b[3], b[1]);
    const float cx = __fadd_rn(b[0], __fmul_rn(0.5f, w)), cy = __fadd_rn(b[1], __fmul_rn(0.5f, h));
    const float dx = __fdiv_rn(d[0], wx), dy = __fdiv_rn(d[1], wy);
    const float dw = fminf(__fdiv_rn(d[2], ww), clamp), dh = fminf(__fdiv_rn(d[3], wh), clamp);
    const float px = __fadd_rn(__fmul_rn(dx, w), cx), py = __fadd_rn(__fmul_rn(dy, h), cy);
    const float pw = __fmul_rn(expf(dw), w), ph = __fmul_rn(expf(dh), h);
    float* o = out + i * 4;
    o[0] = __fsub_rn(px, __fmul_rn(0.5f, pw)); o[1] = __fsub_rn(py, __fmul_rn(0.5f, ph));
    o[2] = __fadd_rn(px, __fmul_rn(0.5f, pw)); o[3] = __fadd_rn(py, __fmul_rn(0.5f, ph));
  }
}

// RPN losses (detectron2/modeling/proposal_generator/rpn.py:362-420, box_regression.py:229-260): over all N * A anchors,
//   objectness: sum over label >= 0 of BCE-with-logits(x, label);   localisation: sum over label == 1 of |delta - get_deltas(anchor, gt)|
// both x inv_norm (1 / (batch_size_per_image * N)).  One partial pair per workgroup into `partial` (ordered fold by the second
// kernel: deterministic), and the unit gradients dlogit = (sigmoid(x) - y) inv_norm, ddelta = sign(delta - target) inv_norm
// (0 where the anchor does not take part).
__global__ __launch_bounds__(256) void rpn_loss_kernel(long n, long n_anchors, const float* __restrict__ logits,
                                                       const float* __restrict__ deltas, const signed char* __restrict__ labels,
                                                       const float* __restrict__ anchors, const float* __restrict__ gt_boxes,
                                                       float wx, float wy, float ww, float wh, float inv_norm,
                                                       float* __restrict__ partial, float* __restrict__ dlogits,
                                                       float* __restrict__ ddeltas) {
  __shared__ float red[32];
  float s_obj = 0.f, s_loc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int lab = labels[i];
    const float x = logits[i];
    float dl = 0.f;
    if (lab >= 0) {
      const float y = (float)lab;
      // max(x, 0) - x y + log(1 + exp(-|x|))   (torch's binary_cross_entropy_with_logits)
      s_obj += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
      dl = (1.f / (1.f + expf(-x)) - y) * inv_norm;
    }
    if (dlogits) dlogits[i] = dl;
    float dd[4] = {0.f, 0.f, 0.f, 0.f};
    if (lab == 1) {
      const float* a = anchors + (i % n_anchors) * 4;
      const float* g = gt_boxes + i * 4;
      const float sw = a[2] - a[0], sh = a[3] - a[1], sx = a[0] + 0.5f * sw, sy = a[1] + 0.5f * sh;
      const float tw = g[2] - g[0], th = g[3] - g[1], tx = g[0] + 0.5f * tw, ty = g[1] + 0.5f * th;
      const float t[4] = {wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * logf(tw / sw), wh * logf(th / sh)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = deltas[i * 4 + k] - t[k];
        s_loc += fabsf(e);
        dd[k] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * inv_norm;
      }
    }
    if (ddeltas) { ddeltas[i * 4] = dd[0]; ddeltas[i * 4 + 1] = dd[1]; ddeltas[i * 4 + 2] = dd[2]; ddeltas[i * 4 + 3] = dd[3]; }
  }
  s_obj = block_reduce_sum(s_obj, red);
  __syncthreads();
  s_loc = block_reduce_sum(s_loc, red);
  if (threadIdx.x == 0) { partial[blockIdx.x * 2] = s_obj; partial[blockIdx.x * 2 + 1] = s_loc; }
}
__global__ void rpn_loss_fold_kernel(int nblocks, const float* __restrict__ partial, float inv_norm, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < nblocks; ++i) { a += partial[2 * i]; b += partial[2 * i + 1]; }
    out[0] = a * inv_norm; out[1] = b * inv_norm;
  }
}
constexpr int RPN_LOSS_BLOCKS = 256;

// ---------------------------------------------------------------- every weight of a detector staged by ONE launch
// entry kinds: 0 linear (rows, cols) f32 -> compute dtype rows of pitch cols; 1 conv3x3 OIHW -> [co][tap][ci]; 2 conv3x3 OIHW ->
// [ci][8 - tap][co] (data-gradient layout); 3 f32 copy (bias pieces of packed heads).  With FrozenBN buffers the folded weight
// w * scale[co] is staged (scale = bn_weight * rsqrt(bn_var + eps), layers/batch_norm.py:52-60; rsqrt spelled 1 / sqrt with the
// _rn intrinsics: within one ulp of the reference's CPU result, tests/test_gpu_stage3.py) and scale / shift are written.
constexpr int STAGE_CHUNK = 4096;
// block geometry per kind (sw_stage_blocks must agree):
//   0 / 3: STAGE_CHUNK consecutive elements, 4 per thread and step (16-byte loads when a row is a multiple of 4 elements);
//   1: one output channel x a range of <= STAGE_CI1 input channels: the [ci][tap] run is read as it lies (contiguous), turned in LDS,
//      and written as 9 runs over ci (the first version gathered 4 bytes every 36: 9x the read transactions);
//   2: 64 output channels x 7 input channels: 63-float runs read per output channel, written as (ci, tap) rows of 64 channels.
constexpr int STAGE_CI1 = 448, STAGE_CO2 = 64, STAGE_CI2 = 7;
__host__ __device__ inline int stage_ci_chunk(int cols) { return cols <= STAGE_CI1 ? cols : 256; }
template <typename T>
__global__ __launch_bounds__(256) void stage_weights_multi_kernel(int n, const sw_stage_desc* __restrict__ descs, float eps) {
  __shared__ float s_t[STAGE_CO2 * (STAGE_CI2 * 9 + 2) > STAGE_CI1 * 9 ? STAGE_CO2 * (STAGE_CI2 * 9 + 2) : STAGE_CI1 * 9];
  int lo = 0, hi = n - 1;                                     // the last entry whose block_start <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const sw_stage_desc d = descs[lo];
  const int rows = d.rows, cols = d.cols;
  const bool bn = d.bn_weight != nullptr;
  const int blk = (int)blockIdx.x - d.block_start;
  if (bn && blk == 0 && d.scale)
    for (int r = threadIdx.x; r < rows; r += 256) {
      const float sc = __fmul_rn(d.bn_weight[r], __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(d.bn_var[r], eps))));
      d.scale[r] = sc;
      if (d.shift) d.shift[r] = __fsub_rn(d.bn_bias[r], __fmul_rn(d.bn_mean[r], sc));
    }
  auto scale_of = [&](int co) { return __fmul_rn(d.bn_weight[co], __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(d.bn_var[co], eps)))); };
  if (d.kind == 0 || d.kind == 3) {
    const long total = (long)rows * cols;
    const long base = (long)blk * STAGE_CHUNK;
    const bool vec = (cols % 4) == 0 && ((((uintptr_t)d.w) & 15) == 0) && ((((uintptr_t)d.dst) & 15) == 0);
    if (vec) {
      for (long i = base + threadIdx.x * 4; i < base + STAGE_CHUNK && i < total; i += 1024) {
        f32x4 v = *(const f32x4*)(d.w + i);
        if (bn) { const float sc = scale_of((int)(i / cols));
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = __fmul_rn(v[t], sc); }
        if (d.kind == 3) *(f32x4*)((float*)d.dst + i) = v;
        else if (sizeof(T) == 4) *(f32x4*)((float*)d.dst + i) = v;
        else {
          unsigned int* o = (unsigned int*)((unsigned short*)d.dst + i);
          o[0] = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
          o[1] = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
        }
      }
    } else {
      for (long i = base + threadIdx.x; i < base + STAGE_CHUNK && i < total; i += 256) {
        float v = d.w[i];
        if (bn) v = __fmul_rn(v, scale_of((int)(i / cols)));
        if (d.kind == 3) ((float*)d.dst)[i] = v;
        else Elem<T>::store((T*)d.dst + i, v);
      }
    }
  } else if (d.kind == 1) {                                   // OIHW [co][ci][tap] -> [co][tap][ci]
    const int cic = stage_ci_chunk(cols), nch = (cols + cic - 1) / cic;
    const int co = blk / nch, ci0 = (blk - co * nch) * cic;
    const int CI = min(cic, cols - ci0);
    const float* src = d.w + ((long)co * cols + ci0) * 9;
    for (int e = threadIdx.x; e < CI * 9; e += 256) s_t[e] = src[e];
    __syncthreads();
    const float sc = bn ? scale_of(co) : 1.f;
    T* dst = (T*)d.dst + (long)co * 9 * cols + ci0;
    for (int e = threadIdx.x; e < CI * 9; e += 256) {
      const int tap = e / CI, ci = e - tap * CI;
      const float v = s_t[ci * 9 + tap];
      Elem<T>::store(dst + (long)tap * cols + ci, bn ? __fmul_rn(v, sc) : v);
    }
  } else {                                                    // OIHW -> [ci][8 - tap][co]
    constexpr int PITCH = STAGE_CI2 * 9 + 2;                  // 65: the write phase walks co at a fixed (ci, tap)
    const int nci = (cols + STAGE_CI2 - 1) / STAGE_CI2;
    const int cob = blk / nci, ci0 = (blk - cob * nci) * STAGE_CI2, co0 = cob * STAGE_CO2;
    const int CI = min(STAGE_CI2, cols - ci0), CO = min(STAGE_CO2, rows - co0);
    const int run = CI * 9;
    for (int e = threadIdx.x; e < CO * run; e += 256) {
      const int col = e / run, k = e - col * run;
      s_t[col * PITCH + k] = d.w[((long)(co0 + col) * cols + ci0) * 9 + k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < run * CO; e += 256) {
      const int col = e % CO, k = e / CO;                     // k = ci_l * 9 + tapf
      const int cil = k / 9, tapf = k - cil * 9;
      float v = s_t[col * PITCH + cil * 9 + (8 - tapf)];
      if (bn) v = __fmul_rn(v, scale_of(co0 + col));
      Elem<T>::store((T*)d.dst + ((long)(ci0 + cil) * 9 + tapf) * rows + co0 + col, v);
    }
  }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
  if ((dtype) == SW_BF16) { CALL_BF16; } else if ((dtype) == SW_F32) { CALL_F32; } else return -1;

extern "C" int sw_preprocess_pad(int dtype, int h, int w, int H, int W, const uint8_t* img_chw, const float* mean3,
                                 const float* std3, void* out_nhwc4, hipStream_t stream) {
  SW_ENTER();
  if (h > H || w > W || H <= 0 || W <= 0) return -5;
  const long n = (long)H * W;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(preprocess_pad_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, h, w, H, W, img_chw,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (unsigned short*)out_nhwc4),
    hipLaunchKernelGGL(preprocess_pad_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, h, w, H, W, img_chw, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2], (float*)out_nhwc4));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_stem_conv7x7(int dtype, int N, int H, int W, const void* in_nhwc4, const float* w_oihw, const float* scale,
                               const float* bias, void* out_nhwc64, hipStream_t stream) {
  SW_ENTER();
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  if (N <= 0 || OH <= 0 || OW <= 0) return 0;
  const long tiles = (long)N * ((OH + 7) / 8) * ((OW + 7) / 8);
  if (tiles > 2147483647L) return -6;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(stem_conv7_kernel<unsigned short>, dim3((unsigned)tiles), dim3(256), 0, stream, N, H, W, OH, OW,
                       (const unsigned short*)in_nhwc4, w_oihw, scale, bias, (unsigned short*)out_nhwc64),
    hipLaunchKernelGGL(stem_conv7_kernel<float>, dim3((unsigned)tiles), dim3(256), 0, stream, N, H, W, OH, OW,
                       (const float*)in_nhwc4, w_oihw, scale, bias, (float*)out_nhwc64));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_maxpool3x3s2(int dtype, int N, int H, int W, int C, const void* in, void* out, hipStream_t stream) {
  SW_ENTER();
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long n = (long)N * OH * OW * C;
  if (n <= 0) return 0;
  {
    const int V = dtype == SW_BF16 ? 8 : 4;
    if ((C % V) == 0 && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0)) {
      DISPATCH_T(dtype,
        hipLaunchKernelGGL(maxpool3x3s2_vec_kernel<unsigned short>, dim3(grid_for_n(n / V)), dim3(256), 0, stream, N, H, W, C, OH, OW,
                           (const unsigned short*)in, (unsigned short*)out),
        hipLaunchKernelGGL(maxpool3x3s2_vec_kernel<float>, dim3(grid_for_n(n / V)), dim3(256), 0, stream, N, H, W, C, OH, OW,
                           (const float*)in, (float*)out));
      SW_CHECK_LAUNCH();
      return 0;
    }
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(maxpool3x3s2_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, H, W, C, OH, OW,
                       (const unsigned short*)in, (unsigned short*)out),
    hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, H, W, C, OH, OW, (const float*)in,
                       (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_subsample2x(int dtype, int N, int H, int W, int C, const void* in, void* out, hipStream_t stream) {
  SW_ENTER();
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long n = (long)N * OH * OW * C;
  if (n <= 0) return 0;
  {                                                           // a copy: 16-byte pieces of a pixel's channel run when they exist
    const long es = dtype == SW_BF16 ? 2 : 4;
    if (((long)C * es) % 16 == 0 && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) && (dtype == SW_BF16 || dtype == SW_F32)) {
      const int Cp = (int)((long)C * es / 16);
      hipLaunchKernelGGL(subsample2_kernel<u32x4>, dim3(grid_for_n((long)N * OH * OW * Cp)), dim3(256), 0, stream, N, H, W, Cp, OH, OW,
                         (const u32x4*)in, (u32x4*)out);
      SW_CHECK_LAUNCH();
      return 0;
    }
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(subsample2_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, H, W, C, OH, OW,
                       (const unsigned short*)in, (unsigned short*)out),
    hipLaunchKernelGGL(subsample2_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, H, W, C, OH, OW, (const float*)in,
                       (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_scatter2x(int dtype, int N, int H, int W, int C, const void* g, void* out, hipStream_t stream) {
  SW_ENTER();
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long n = (long)N * H * W * C;
  if (n <= 0) return 0;
  {
    const long es = dtype == SW_BF16 ? 2 : 4;
    if (((long)C * es) % 16 == 0 && ((((uintptr_t)g | (uintptr_t)out) & 15) == 0) && (dtype == SW_BF16 || dtype == SW_F32)) {
      const int Cp = (int)((long)C * es / 16);
      hipLaunchKernelGGL(scatter2_kernel<u32x4>, dim3(grid_for_n((long)N * H * W * Cp)), dim3(256), 0, stream, N, H, W, Cp, OH, OW,
                         (const u32x4*)g, (u32x4*)out);
      SW_CHECK_LAUNCH();
      return 0;
    }
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(scatter2_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, H, W, C, OH, OW,
                       (const unsigned short*)g, (unsigned short*)out),
    hipLaunchKernelGGL(scatter2_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, H, W, C, OH, OW, (const float*)g,
                       (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_add_relu(int dtype, long n, const void* a, const void* b, void* out, int relu, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(add_relu_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, n, (const unsigned short*)a,
                       (const unsigned short*)b, (unsigned short*)out, relu),
    hipLaunchKernelGGL(add_relu_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, n, (const float*)a, (const float*)b,
                       (float*)out, relu));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_upsample2x_add(int dtype, int N, int h, int w, int C, const void* lateral, const void* top, void* out,
                                 hipStream_t stream) {
  SW_ENTER();
  const long n = (long)N * 4 * h * w * C;
  if (n <= 0) return 0;
  const int V = dtype == SW_BF16 ? 8 : 4;
  if ((C % V) == 0 && ((((uintptr_t)lateral | (uintptr_t)top | (uintptr_t)out) & 15) == 0)) {
    const long nv = n / V;
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(upsample2_add_vec_kernel<unsigned short>, dim3(grid_for_n(nv)), dim3(256), 0, stream, N, h, w, C,
                         (const unsigned short*)lateral, (const unsigned short*)top, (unsigned short*)out),
      hipLaunchKernelGGL(upsample2_add_vec_kernel<float>, dim3(grid_for_n(nv)), dim3(256), 0, stream, N, h, w, C, (const float*)lateral,
                         (const float*)top, (float*)out));
    SW_CHECK_LAUNCH();
    return 0;
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(upsample2_add_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, h, w, C,
                       (const unsigned short*)lateral, (const unsigned short*)top, (unsigned short*)out),
    hipLaunchKernelGGL(upsample2_add_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, h, w, C, (const float*)lateral,
                       (const float*)top, (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_downsample2x_sum(int dtype, int N, int h, int w, int C, const void* g, void* out, hipStream_t stream) {
  SW_ENTER();
  const long n = (long)N * h * w * C;
  if (n <= 0) return 0;
  {
    const int V = dtype == SW_BF16 ? 8 : 4;
    if ((C % V) == 0 && ((((uintptr_t)g | (uintptr_t)out) & 15) == 0)) {
      DISPATCH_T(dtype,
        hipLaunchKernelGGL(downsample2_sum_vec_kernel<unsigned short>, dim3(grid_for_n(n / V)), dim3(256), 0, stream, N, h, w, C,
                           (const unsigned short*)g, (unsigned short*)out),
        hipLaunchKernelGGL(downsample2_sum_vec_kernel<float>, dim3(grid_for_n(n / V)), dim3(256), 0, stream, N, h, w, C, (const float*)g,
                           (float*)out));
      SW_CHECK_LAUNCH();
      return 0;
    }
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(downsample2_sum_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, h, w, C,
                       (const unsigned short*)g, (unsigned short*)out),
    hipLaunchKernelGGL(downsample2_sum_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, N, h, w, C, (const float*)g,
                       (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_roi_align_fwd(int dtype, int H, int W, int C, int PH, int PW, float spatial_scale, int sampling_ratio,
                                const void* feat, const float* rois, const int32_t* sel, int n_sel, const int32_t* n_sel_dev, void* out, long ld_out,
                                hipStream_t stream) {
  SW_ENTER();
  if (n_sel <= 0) return 0;
  if (ld_out < (long)C * PH * PW) return -5;
  const long n = (long)n_sel * PH * PW * C;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(roi_align_fwd_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, H, W, C, PH, PW,
                       spatial_scale, sampling_ratio, (const unsigned short*)feat, rois, sel, n_sel, n_sel_dev, (unsigned short*)out, ld_out),
    hipLaunchKernelGGL(roi_align_fwd_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, H, W, C, PH, PW, spatial_scale,
                       sampling_ratio, (const float*)feat, rois, sel, n_sel, n_sel_dev, (float*)out, ld_out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_roi_align_bwd(int dtype, int H, int W, int C, int PH, int PW, float spatial_scale, int sampling_ratio,
                                const void* gout, long ld, const float* rois, const int32_t* sel, int n_sel, const int32_t* n_sel_dev, float* dfeat_f32,
                                hipStream_t stream) {
  SW_ENTER();
  if (n_sel <= 0) return 0;
  if (ld < (long)C * PH * PW) return -5;
  const long n = (long)n_sel * PH * PW * C;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(roi_align_bwd_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, H, W, C, PH, PW,
                       spatial_scale, sampling_ratio, (const unsigned short*)gout, ld, rois, sel, n_sel, n_sel_dev, dfeat_f32),
    hipLaunchKernelGGL(roi_align_bwd_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, H, W, C, PH, PW, spatial_scale,
                       sampling_ratio, (const float*)gout, ld, rois, sel, n_sel, n_sel_dev, dfeat_f32));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_roi_align_bwd_fx(int dtype, int H, int W, int C, int PH, int PW, float spatial_scale, int sampling_ratio,
                                   const void* gout, long ld, const float* rois, const int32_t* sel, int n_sel, const int32_t* n_sel_dev,
                                   const float* gout_absmax, long long* acc_i64, hipStream_t stream) {
  SW_ENTER();
  if (n_sel <= 0) return 0;
  if (ld < (long)C * PH * PW || !gout_absmax || !acc_i64) return -5;
  const long n = (long)n_sel * PH * PW * C;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(roi_align_bwd_fx_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, H, W, C, PH, PW,
                       spatial_scale, sampling_ratio, (const unsigned short*)gout, ld, rois, sel, n_sel, n_sel_dev, gout_absmax,
                       (unsigned long long*)acc_i64),
    hipLaunchKernelGGL(roi_align_bwd_fx_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, H, W, C, PH, PW, spatial_scale,
                       sampling_ratio, (const float*)gout, ld, rois, sel, n_sel, n_sel_dev, gout_absmax, (unsigned long long*)acc_i64));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_fx_to_float(int out_dtype, long n, const long long* acc_i64, const float* absmax, void* out, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  if (!acc_i64 || !absmax || !out) return -5;
  DISPATCH_T(out_dtype,
    hipLaunchKernelGGL(fx_to_float_kernel<unsigned short>, dim3(grid_for_n(n)), dim3(256), 0, stream, n, acc_i64, absmax, (unsigned short*)out),
    hipLaunchKernelGGL(fx_to_float_kernel<float>, dim3(grid_for_n(n)), dim3(256), 0, stream, n, acc_i64, absmax, (float*)out));
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_decode_boxes(long n, long n_boxes, const float* deltas, long ld_deltas, const float* boxes,
                               const float* weights4, float scale_clamp, float* out, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 0;
  if (n_boxes <= 0 || ld_deltas < 4) return -5;
  hipLaunchKernelGGL(decode_boxes_kernel, dim3(grid_for_n(n)), dim3(256), 0, stream, n, n_boxes, deltas, ld_deltas, boxes,
                     weights4[0], weights4[1], weights4[2], weights4[3], scale_clamp, out);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" long sw_rpn_loss_workspace_floats(void) { return 2L * RPN_LOSS_BLOCKS; }

extern "C" int sw_rpn_loss(long n, long n_anchors, const float* logits, const float* deltas, const int8_t* labels,
                           const float* anchors, const float* matched_gt_boxes, const float* weights4, float inv_norm,
                           float* losses2, float* dlogits, float* ddeltas, float* workspace, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0 || n_anchors <= 0) return -5;
  int blocks = grid_for_n(n);
  if (blocks > RPN_LOSS_BLOCKS) blocks = RPN_LOSS_BLOCKS;
  hipLaunchKernelGGL(rpn_loss_kernel, dim3(blocks), dim3(256), 0, stream, n, n_anchors, logits, deltas, (const signed char*)labels,
                     anchors, matched_gt_boxes, weights4[0], weights4[1], weights4[2], weights4[3], inv_norm, workspace, dlogits,
                     ddeltas);
  hipLaunchKernelGGL(rpn_loss_fold_kernel, dim3(1), dim3(64), 0, stream, blocks, (const float*)workspace, inv_norm, losses2);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_stage_blocks(int kind, int rows, int cols) {
  if (kind == 1) { const int cic = stage_ci_chunk(cols); return rows * ((cols + cic - 1) / cic); }
  if (kind == 2) return ((rows + STAGE_CO2 - 1) / STAGE_CO2) * ((cols + STAGE_CI2 - 1) / STAGE_CI2);
  const long total = (long)rows * cols;
  return (int)((total + STAGE_CHUNK - 1) / STAGE_CHUNK);
}

extern "C" int sw_stage_weights_multi(int dtype, int n, const sw_stage_desc* descs_dev, int total_blocks, float eps,
                                      hipStream_t stream) {
  SW_ENTER();
  if (n <= 0 || total_blocks <= 0) return 0;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(stage_weights_multi_kernel<unsigned short>, dim3(total_blocks), dim3(256), 0, stream, n, descs_dev, eps),
             hipLaunchKernelGGL(stage_weights_multi_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, n, descs_dev, eps));
  SW_CHECK_LAUNCH();
  return 0;
}
