// Max ROI pooling over an NHWC feature map, gfx950.
// Arithmetic follows the reference statement in
//   uwsod/projects/WSL/wsl/layers/csrc/ROILoopPool/ROILoopPool_cpu.cpp:26-79 (fwd), :98-123 (bwd)
// (= torchvision 0.7 RoIPool): round(x*scale), +1 extent, float bin = extent/P, floor/ceil bin edges,
// clip to the map, strict '>' from -FLT_MAX (first max in row-major order), empty bin -> 0 / -1.
//
// Forward: one workgroup per (roi, 64-channel slab).  Lanes run over channels so that every pixel read
// is one coalesced 128/256-byte row segment of the NHWC map (the whole map is L2 / Infinity-Cache
// resident); the 4 waves split the PHxPW bins; results are transposed through LDS so that the
// (R, C, PH, PW) output — the layout the reference's fc6 weight expects — is written as one contiguous
// run of 64*PH*PW elements per workgroup.  HBM-bound on the output + argmax stream.
//
// Backward: one workgroup per (image, CB-channel slab) owns dfeat[:, :, slab] in LDS (H*W*CB f32),
// sweeps every ROI of that image once, accumulates with LDS atomics and writes its slab exactly once:
// no global atomics, no cross-workgroup traffic (a per-channel argmax scatters 64 lanes to 64 different
// rows, which global float atomics serve ~17x below their peak rate).
#include <float.h>
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {


struct RoiGeom {
  int batch, start_w, start_h;
  float bin_h, bin_w;
};

// Channel slab of workgroup blockIdx.x.  Workgroup ids go round-robin over the 8 XCDs (id % 8), so slab = blockIdx.x puts
// NEIGHBOURING slabs on different XCDs — and neighbouring slabs write (forward) / read (backward) neighbouring 784-byte runs of
// every ROI row: the 128-byte lines at the run boundaries were held, half written, by two private L2s.  Dealt as below the 8 slabs
// of a 64-channel group (49 whole lines per ROI row) run on ONE XCD at the same time.  Measured per 4000-ROI call, 63x63 map
// (tools/roi_bench_voc.py): bf16 forward 259 -> 246 us, backward 127 -> 115 us; fp32 forward 455 -> 397 us.
__device__ __forceinline__ int xcd_grouped_slab() {
  return ((gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
}

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int PH, int PW) {
  RoiGeom g;
  g.batch = (int)roi[0];
  g.start_w = (int)roundf(__fmul_rn(roi[1], scale));
  g.start_h = (int)roundf(__fmul_rn(roi[2], scale));
  const int end_w = (int)roundf(__fmul_rn(roi[3], scale));
  const int end_h = (int)roundf(__fmul_rn(roi[4], scale));
  const int rw = max(end_w - g.start_w + 1, 1);
  const int rh = max(end_h - g.start_h + 1, 1);
  g.bin_h = __fdiv_rn((float)rh, (float)PH);
  g.bin_w = __fdiv_rn((float)rw, (float)PW);
  return g;
}

// argmax storage: int32 (the reference's tensor) or uint16 with 0xFFFF = "no pixel" (maps of < 65535 pixels; a third less
// HBM traffic on the forward's output stream and half of the backward's index stream)
template <typename IT> struct ArgIdx;
template <> struct ArgIdx<int> {
  __device__ static __forceinline__ int enc(int i) { return i; }
  __device__ static __forceinline__ int dec(int v) { return v; }
};
template <> struct ArgIdx<unsigned short> {
  __device__ static __forceinline__ unsigned short enc(int i) { return (unsigned short)i; }     // i is a pixel index or -1 (-> 0xFFFF)
  __device__ static __forceinline__ int dec(unsigned short v) { return v == 0xFFFF ? -1 : (int)v; }
};

// Forward.  Workgroup = (roi, slab of 64*VEC channels), 4 waves split the PHxPW bins, lane = VEC adjacent channels
// (bf16: one 4-byte load carries 2 channels => 256-byte wave loads).  The scan of a bin is latency bound (every NHWC
// row segment is an L2 / Infinity-Cache hit), so a wave keeps 8 pixel loads in flight: the bin window is walked as a
// flat list (h ascending, w ascending — a wave-uniform scalar cursor), 8 elements per batch; slots past the end
// re-read the last pixel, which can never win the strict '>' => the first maximum in row-major order is kept
// exactly as in the serial reference.
template <typename T, int VEC> struct VecLoad;
template <> struct VecLoad<float, 1> {
  __device__ static __forceinline__ void load(const float* p, float* v) { v[0] = *p; }
};
template <> struct VecLoad<unsigned short, 2> {
  __device__ static __forceinline__ void load(const unsigned short* p, float* v) {
    const unsigned int u = *(const unsigned int*)p;
    v[0] = __uint_as_float(u << 16); v[1] = __uint_as_float(u & 0xFFFF0000u);
  }
};

template <typename T, int VEC, typename IT>
__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(int H, int W, int C, long ld, int PH, int PW, float scale,
                                                           const T* __restrict__ feat, const float* __restrict__ rois,
                                                           const float* __restrict__ row_scale, float row_scale_add,
                                                           T* __restrict__ out, IT* __restrict__ argmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = 64 * VEC;
  const int nb = PH * PW;
  int* s_arg = (int*)smem;                     // [CH][nb]  (row stride nb: odd for 7x7 -> conflict free)
  T* s_val = (T*)(smem + (size_t)CH * nb * 4); // [CH][nb]  stored in the output dtype: 6 B/entry -> 4 workgroups per CU
  const int r = blockIdx.x, c0 = blockIdx.y * CH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const RoiGeom g = roi_geom(rois + (long)r * 5, scale, PH, PW);
  const float mul = row_scale ? (row_scale[r] + row_scale_add) : 1.0f;
  const int c = c0 + lane * VEC;
  const bool cok = c < C;                       // C % VEC == 0 is checked by the host
  const T* fimg = feat + (long)g.batch * H * W * C + (cok ? c : 0);
  for (int b = wave; b < nb; b += 4) {
    const int ph = b / PW, pw = b - ph * PW;
    int hs = (int)floorf(__fmul_rn((float)ph, g.bin_h));
    int ws = (int)floorf(__fmul_rn((float)pw, g.bin_w));
    int he = (int)ceilf(__fmul_rn((float)(ph + 1), g.bin_h));
    int we = (int)ceilf(__fmul_rn((float)(pw + 1), g.bin_w));
    hs = min(max(hs + g.start_h, 0), H); he = min(max(he + g.start_h, 0), H);
    ws = min(max(ws + g.start_w, 0), W); we = min(max(we + g.start_w, 0), W);
    const bool empty = (he <= hs) || (we <= ws);
    float mv[VEC]; int mi[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) { mv[q] = empty ? 0.f : -FLT_MAX; mi[q] = -1; }
    const int n = empty ? 0 : (he - hs) * (we - ws);
    int hh = hs, ww = ws;                        // wave-uniform cursor over the window, row-major
    for (int e0 = 0; e0 < n; e0 += 6) {         // 6 loads in flight (typical bins hold 6..16 pixels)
      float v[6][VEC]; int idx[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        idx[u] = hh * W + ww;
        VecLoad<T, VEC>::load(fimg + (long)idx[u] * C, v[u]);
        if (e0 + u + 1 < n) { ++ww; if (ww == we) { ww = ws; ++hh; } }
      }
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int q = 0; q < VEC; ++q)
          if (v[u][q] > mv[q]) { mv[q] = v[u][q]; mi[q] = idx[u]; }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      Elem<T>::store(&s_val[(lane * VEC + q) * nb + b], __fmul_rn(mv[q], mul));
      s_arg[(lane * VEC + q) * nb + b] = mi[q];
    }
  }
  __syncthreads();
  const int nch = min(CH, C - c0);
  const int total = nch * nb;
  const long obase = (long)r * ld + (long)c0 * nb;
  if (sizeof(IT) == 4 && (total & 7) == 0 && ((obase * (long)sizeof(T)) & 15) == 0) {
    // 16-byte stores: 8 (bf16) / 4 (f32) values and 4 argmax words per lane
    constexpr int VPV = 16 / (int)sizeof(T);
    for (int i = threadIdx.x; i < total / VPV; i += blockDim.x)
      *(u32x4*)(out + obase + (long)i * VPV) = *(const u32x4*)(s_val + i * VPV);
    for (int i = threadIdx.x; i < total / 4; i += blockDim.x)
      *(u32x4*)((int*)argmax + obase + (long)i * 4) = *(const u32x4*)(s_arg + i * 4);
  } else {
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      out[obase + i] = s_val[i];
      argmax[obase + i] = ArgIdx<IT>::enc(s_arg[i]);
    }
  }
}

template <typename T, typename IT>
__global__ __launch_bounds__(1024) void roi_pool_bwd_kernel(int H, int W, int C, long ld, int nb, int CB,
                                                            const T* __restrict__ dout, const IT* __restrict__ argmax,
                                                            const float* __restrict__ rois, int R,
                                                            const float* __restrict__ row_scale, float row_scale_add,
                                                            const T* __restrict__ relu_ref, T* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc = (float*)smem;                   // [CB][H*W]
  const int img = blockIdx.y, c0 = blockIdx.x * CB;
  const int npix = H * W;
  for (int i = threadIdx.x; i < npix * CB; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int per_roi = CB * nb;                 // contiguous (c, bin) run of this slab inside one ROI
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int iters = (per_roi + 63) / 64;
  for (int r0 = wave * 4; r0 < R; r0 += nwave * 4) {
    // stage 4 ROIs x (argmax, grad) in registers first (memory-level parallelism), then scatter into LDS
    for (int j = 0; j < iters; ++j) {
      const int i = lane + 64 * j;
      int a[4]; float d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u;
        a[u] = -1; d[u] = 0.f;
        if (r < R && i < per_roi && (int)rois[(long)r * 5] == img) {
          const long base = (long)r * ld + (long)c0 * nb;
          a[u] = ArgIdx<IT>::dec(argmax[base + i]);
          d[u] = Elem<T>::load(dout + base + i);
        }
      }
      const int cc = i / nb;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (a[u] >= 0) {
          const float mul = row_scale ? (row_scale[r0 + u] + row_scale_add) : 1.0f;
          atomicAdd(&acc[cc * npix + a[u]], __fmul_rn(d[u], mul));
        }
      }
    }
  }
  __syncthreads();
  T* dimg = dfeat + (long)img * npix * C;
  const T* rimg = relu_ref ? relu_ref + (long)img * npix * C : nullptr;
  for (int i = threadIdx.x; i < npix * CB; i += blockDim.x) {
    const int p = i / CB, cc = i - p * CB;
    float v = acc[cc * npix + p];
    if (rimg && !(Elem<T>::load(rimg + (long)p * C + c0 + cc) > 0.f)) v = 0.f;
    Elem<T>::store(dimg + (long)p * C + c0 + cc, v);
  }
}

// Backward, fixed-point path.  Measured on MI355X (tools/probes/roi_bench.py): LDS float atomics (ds_add_f32) run ~3.4x
// slower than LDS integer atomics (ds_add_u32 / ds_add_u64 ~ plain ds_write rate) and set the kernel's time.  So the slab
// accumulates in 64-bit fixed point: every product grad*scale (rounded to f32 exactly as the float path does) is
// converted with 2^FRAC, FRAC chosen from max|grad|*max|scale| so that 2^40 bounds one term and PH*PW*R_image terms (a pixel
// can be the argmax of every bin of a ROI whose bins are smaller than a pixel) cannot overflow 63 bits.  Integer adds are associative => the result is
// BITWISE REPRODUCIBLE and at least as accurate as f32 accumulation.
// Workgroup = (image, slab of CB channels, CB % 4 == 0) owning H*W*CB int64 in LDS; a lane loads 4 gradients and 4
// argmax words of 4 ROIs (16-byte aligned, 8 loads in flight) before scattering.
constexpr int FX_CHUNK = 1024;          // ROIs per compaction round of the fixed-point backward

// Accumulator forms (ACCMODE):
//   0  one 64-bit word, 40 bits per term (fp32 mode): room for 2^23 terms.
//   2  a PAIR of 32-bit words (bf16 mode, the default): a pixel-channel receives at most PH*PW bins of every ROI of its image, so
//      with bits = ceil(log2(PH*PW*R_image)) a word may take 30 - bits bits per term without overflow whatever the ROI sizes
//      (13 at R_image = 2000).  13 bits relative to the GLOBAL max|dpooled|*max|objectness+1| are a dead zone of max/16384 on a
//      gradient whose per-row weights span 30 decades (ignored / background rows), so the term is split: hi = rint(t * 2^frac),
//      lo = rint((t * 2^frac - hi) * 2^bits') with |lo| <= 2^(bits'-1): two v_cvt_i32_f32 + two 32-bit LDS atomics = 2 x (30 - bits) + 1
//      bits per term (27 at R_image = 2000, 25 at 4000: 2^-26 of the largest term against bf16's 2^-8 outputs), still without the
//      emulated f32 -> i64 conversion (the kernel is VALU-issue bound: 88 M wave instructions per 4000 ROIs).
//   1  one 32-bit word (round 2's form, 30 - bits bits per term): kept behind SW_ROI_BWD_ACC32 for A/B timing only.
template <typename T, typename IT, int ACCMODE>
__global__ __launch_bounds__(1024) void roi_pool_bwd_fx_kernel(int H, int W, int C, long ld, int nb, int CB,
                                                               const T* __restrict__ dout, const IT* __restrict__ argmax,
                                                               const float* __restrict__ rois, int R,
                                                               const float* __restrict__ row_scale, float row_scale_add,
                                                               const float* __restrict__ dout_absmax,
                                                               const T* __restrict__ relu_ref, T* __restrict__ dfeat,
                                                               float spatial_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[32];
  __shared__ int s_cnt;
  // large maps: the workgroup owns only the pixel range [p0, p1) of the plane (blockIdx.z); gradients whose argmax falls
  // outside are skipped (the other ranges' workgroups stream the same ROI data again)
  const int npix_all = H * W;
  const int px_per = (npix_all + gridDim.z - 1) / gridDim.z;
  const int p0 = blockIdx.z * px_per, p1 = min(npix_all, p0 + px_per);
  typedef typename std::conditional<ACCMODE == 0, unsigned long long, unsigned int>::type ACC;
  constexpr int ACC_BYTES = ACCMODE == 1 ? 4 : 8;                // per pixel-channel
  ACC* acc = (ACC*)smem;          // [CB][H*W] two's-complement fixed point: the lanes of a wave
                                                                // hold bins of ONE channel => neighbouring pixels => distinct banks
                                                                // (pixel-major [H*W][CB] put them 64 B apart: 8-16-way conflicts)
  const int img = blockIdx.y, c0 = xcd_grouped_slab() * CB;
  const int npix = max(p1 - p0, 0);
  for (int i = threadIdx.x; i < npix * CB * (ACCMODE == 2 ? 2 : 1); i += blockDim.x) acc[i] = (ACC)0;     // mode 2: hi words, then lo words
  float smax = 0.f;                                             // max |row_scale + add| (wave/block reduce, tiny)
  if (row_scale) { for (int r = threadIdx.x; r < R; r += blockDim.x) smax = fmaxf(smax, fabsf(row_scale[r] + row_scale_add)); }
  else smax = 1.f;
  smax = block_reduce_max(smax, red);
  smax = __shfl(smax, 0, 64);
  const float bound = dout_absmax[0] * smax;
  // A diverged gradient (NaN / Inf anywhere in dout: its |max| is then NaN or Inf) must stay visible: the integer conversion
  // would turn NaN into 0 and saturate Inf, so the whole slab is written as NaN instead.
  const bool poisoned = !(bound < 3.0e38f);
  // Terms per accumulator: a pixel-channel can be the argmax of EVERY bin of a ROI whose bins are smaller than a pixel (ROIs
  // narrower than PW feature pixels: MIN_SIZE 20 at stride 8), so the bound is nb per ROI of this image — not the 4 of
  // well-formed ROIs.  32-bit accumulators: 30 - ceil(log2(nb * R_image)) bits per term (R = 2000: 13 bits, still 32x finer
  // than the bf16 result); 64-bit: 40 bits per term leave room for 2^23 terms.
  int n_img = 0;
  for (int r = threadIdx.x; r < R; r += blockDim.x) n_img += ((int)rois[(long)r * 5] == img) ? 1 : 0;
  n_img = (int)(block_reduce_sum((float)n_img, red) + 0.5f);        // exact: counts < 2^24
  n_img = __shfl(n_img, 0, 64);
  const unsigned terms = (unsigned)max(nb * max(n_img, 1), 1);
  int frac = 0;
  const int term_bits = ACCMODE == 0 ? 40 : max(30 - (32 - __clz(terms)), 1);
  const int lo_off = npix * CB;                                  // mode 2: index distance hi -> lo word
  // mode 2, lo word: |t - rint(t)| <= 1/2, so with lo_bits = term_bits + 1 a term is at most 2^term_bits and `terms` of them
  // stay below 2^30 like the hi word's
  const int lo_bits = term_bits + 1;
  if (bound > 0.f && !poisoned) frac = term_bits - (ilogbf(bound) + 1);
  __syncthreads();
  // ROIs are taken in chunks of FX_CHUNK: the workgroup first compacts (roi, scale) of the ROIs of ITS image into LDS,
  // then every wave streams 8 listed ROIs per step with all 16 loads issued before the first use — the only global
  // loads in the loop are the two data streams (a per-ROI batch-index / scale lookup in front of them made every step
  // three dependent memory latencies long and set the kernel's time).  List order is arbitrary; integer accumulation
  // does not depend on it.
  const int nvec = (CB * nb) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  int* s_r = (int*)(smem + (size_t)px_per * CB * ACC_BYTES);
  float* s_m = (float*)(s_r + FX_CHUNK);
  for (int rc = 0; rc < R; rc += FX_CHUNK) {
    __syncthreads();                            // previous chunk's list fully consumed (and acc zeroed, first time)
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (int r = rc + threadIdx.x; r < min(R, rc + FX_CHUNK); r += blockDim.x)
      if ((int)rois[(long)r * 5] == img) {
        if (gridDim.z > 1 && spatial_scale > 0.f) {
          // large maps (several pixel ranges per plane): a workgroup lists only the ROIs whose rows can reach its range — every
          // argmax of a ROI lies in rows [rs, max(re, rs)] (ROILoopPool_cpu.cpp:36-52) — instead of streaming the gradients of
          // ALL ROIs of the image once per range (99x165 map, 2 x 4000 ROIs, 4 ranges: 765 -> ~400 us)
          const int rs = (int)roundf(__fmul_rn(rois[(long)r * 5 + 2], spatial_scale));
          const int re = max((int)roundf(__fmul_rn(rois[(long)r * 5 + 4], spatial_scale)), rs);
          if (re < p0 / W || rs > (p1 - 1) / W) continue;
        }
        const int k = atomicAdd(&s_cnt, 1);
        s_r[k] = r;
        s_m[k] = row_scale ? (row_scale[r] + row_scale_add) : 1.0f;
      }
    __syncthreads();
    const int cnt = s_cnt;
    for (int l0 = wave * 8; l0 < cnt; l0 += nwave * 8) {
      for (int j = lane; j < nvec; j += 64) {
        // channel row of each of the lane's 4 elements, formed ONCE per step and outside the range test below: left inside it, the
        // compiler sank the division by nb into every element's guarded block (16 of its 27 VALU instructions)
        int crow[4];
        {
          const int q0 = (j * 4) / nb, rem0 = j * 4 - q0 * nb;
#pragma unroll
          for (int e = 0; e < 4; ++e) crow[e] = (q0 + (rem0 + e >= nb ? 1 : 0)) * npix;
        }
        u32x4 av[8]; u32x4 dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = s_r[min(l0 + u, cnt - 1)];
          const long base = (long)r * ld + (long)c0 * nb + (long)j * 4;
          // both streams are read exactly once: nontemporal loads (126 -> 115 us)
#ifndef SW_ROI_PLAIN_LOADS
          if (sizeof(IT) == 4) av[u] = __builtin_nontemporal_load((const u32x4*)(argmax + base));
          else { const u32x2 t = __builtin_nontemporal_load((const u32x2*)(argmax + base)); av[u][0] = t[0]; av[u][1] = t[1]; }
          if (sizeof(T) == 2) { const u32x2 t = __builtin_nontemporal_load((const u32x2*)(dout + base)); dv[u][0] = t[0]; dv[u][1] = t[1]; }
          else dv[u] = __builtin_nontemporal_load((const u32x4*)(dout + base));
#else
          if (sizeof(IT) == 4) av[u] = *(const u32x4*)(argmax + base);
          else { const u32x2 t = *(const u32x2*)(argmax + base); av[u][0] = t[0]; av[u][1] = t[1]; }
          if (sizeof(T) == 2) { const u32x2 t = *(const u32x2*)(dout + base); dv[u][0] = t[0]; dv[u][1] = t[1]; }
          else dv[u] = *(const u32x4*)(dout + base);
#endif
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (l0 + u >= cnt) continue;
          const float mul = s_m[l0 + u];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // "no pixel" (-1, or 0xFFFF in the 16-bit form: maps hold < 65535 pixels) and pixels of other ranges fail ONE
            // unsigned compare of the index relative to this workgroup's range
            unsigned int rel; float d;
            if (sizeof(IT) == 4) rel = av[u][e] - (unsigned)p0;
            else rel = ((av[u][e >> 1] >> (16 * (e & 1))) & 0xFFFFu) - (unsigned)p0;
            if (sizeof(T) == 2) d = __uint_as_float((e & 1) ? (dv[u][e >> 1] & 0xFFFF0000u) : (dv[u][e >> 1] << 16));
            else d = __uint_as_float(dv[u][e]);
            if (rel < (unsigned)npix) {
              if (ACCMODE == 0) {
                const long long q = __float2ll_rn(scalbnf(__fmul_rn(d, mul), frac));
                atomicAdd((unsigned long long*)&acc[crow[e] + (int)rel], (unsigned long long)q);
              } else if (ACCMODE == 1) {
                const int q = __float2int_rn(scalbnf(__fmul_rn(d, mul), frac));
                atomicAdd((unsigned int*)&acc[crow[e] + (int)rel], (unsigned int)q);
              } else {
                const float t = scalbnf(__fmul_rn(d, mul), frac);            // |t| < 2^term_bits
                const float hf = rintf(t);                                    // v_rndne_f32; t - hf is exact, |t - hf| <= 0.5
                const int qh = __float2int_rn(hf), ql = __float2int_rn(scalbnf(t - hf, lo_bits));
                atomicAdd((unsigned int*)&acc[crow[e] + (int)rel], (unsigned int)qh);
                atomicAdd((unsigned int*)&acc[lo_off + crow[e] + (int)rel], (unsigned int)ql);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  T* dimg = dfeat + ((long)img * npix_all + p0) * C;
  const T* rimg = relu_ref ? relu_ref + ((long)img * npix_all + p0) * C : nullptr;
  for (int i = threadIdx.x; i < npix * CB; i += blockDim.x) {
    const int p = i / CB, cc = i - p * CB;
    float v;
    if (ACCMODE == 0) v = scalbnf((float)(long long)acc[cc * npix + p], -frac);
    else if (ACCMODE == 1) v = scalbnf((float)(int)acc[cc * npix + p], -frac);
    else v = scalbnf((float)(((long long)(int)acc[cc * npix + p] << lo_bits) + (long long)(int)acc[lo_off + cc * npix + p]),
                     -(frac + lo_bits));
    if (rimg && !(Elem<T>::load(rimg + (long)p * C + c0 + cc) > 0.f)) v = 0.f;
    if (poisoned) v = __uint_as_float(0x7FC00000u);
    Elem<T>::store(dimg + (long)p * C + c0 + cc, v);
  }
}

// Forward, feature-stationary form (the one the hot path runs).  The gather form above re-reads every map pixel of a
// ROI once per overlapping bin from L2 — ~5 GB of 256-byte L2 reads per 8000-ROI call, latency bound at ~5 TB/s.  Here
// a workgroup owns CB channels of ONE image: it copies that H x W x CB slab of the map into LDS once (16 bytes per
// pixel for CB = 8 bf16), then streams every ROI of that image in its ROI chunk through it.  Lane = one (roi, bin):
// the bin window is walked in the reference's row-major order with one 16-byte LDS read per pixel carrying all CB
// channels; strict '>' from -FLT_MAX keeps the first maximum exactly as the serial reference.  Results go straight to
// the (R, C, PH, PW) output: for a fixed channel the PH*PW lanes of a ROI write one contiguous run.
template <int PXB> struct PixWord;
template <> struct PixWord<16> { typedef u32x4 type; };
template <> struct PixWord<8> { typedef unsigned long long type; };
template <> struct PixWord<4> { typedef unsigned int type; };

template <typename T, int CB>
__device__ __forceinline__ void pix_decode(const typename PixWord<CB * (int)sizeof(T)>::type& w, float* v) {
  const unsigned int* u = (const unsigned int*)&w;
  if (sizeof(T) == 2) {
#pragma unroll
    for (int q = 0; q < CB; q += 2) {
      v[q] = __uint_as_float(u[q >> 1] << 16);
      if (q + 1 < CB) v[q + 1] = __uint_as_float(u[q >> 1] & 0xFFFF0000u);
    }
  } else {
#pragma unroll
    for (int q = 0; q < CB; ++q) v[q] = __uint_as_float(u[q]);
  }
}

// KEY = true (bf16 maps of < 65535 pixels): the LDS copy holds order-preserving 16-bit keys instead of bf16 bits
// (positive: b ^ 0x8000, negative: ~b, +NaN -> 0 so that it never wins, as with the reference's '>'), and the running
// maximum of a (bin, channel) is ONE u32  key << 16 | (0xFFFE - pixel): v_max_u32 then implements "larger value, else
// earlier pixel" — 2 VALU instructions per channel and pixel instead of convert + compare + two selects.  The start
// value key(-inf) << 16 | 0xFFFF can only be beaten by values > -inf, i.e. exactly those that beat -FLT_MAX.
// (-0.0 shares +0.0's key, as they compare equal; a bin won by a -0.0 pixel then outputs +0.0.)
//
// BAND = true (maps whose whole plane does not fit LDS with 16-byte pixels; KEY form only): a workgroup holds `band_rows`
// consecutive map ROWS of all W columns instead of the whole plane, and the (ROI, bin row) pairs of its ROI chunk are dealt to the
// band whose first `band_S` rows hold the bin's first window row; band_rows - band_S >= ceil(H / PH) + 1 rows of overlap hold the
// rest of any window of a ROI that lies inside the image.  So large maps keep 8 channels per lane (the per-task cost outside the
// scan and the scan's loop overhead are paid per slab: 150x200 map, 4000 ROIs: 4-byte slabs 1.76 ms) at the price of loading
// each map row ~2x.  Window rows past the band (ROIs reaching far outside the image: their clipped bins can be as tall as the
// map) are read from global memory by a cold loop.  The pairs owned by a band are compacted into an LDS list first.
template <typename T, int CB>
__device__ __forceinline__ typename PixWord<CB * (int)sizeof(T)>::type pix_to_keys(typename PixWord<CB * (int)sizeof(T)>::type w) {
  unsigned int* u = (unsigned int*)&w;
#pragma unroll
  for (int i = 0; i < CB * (int)sizeof(T) / 4; ++i) {
    unsigned int k = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned int bts = (u[i] >> (16 * h)) & 0xFFFFu;
      unsigned int key = (bts & 0x8000u) ? (bts ^ 0xFFFFu) : (bts | 0x8000u);
      if (bts > 0x7F80u && bts < 0x8000u) key = 0;              // +NaN never wins
      if (bts == 0x8000u) key = 0x8000u;                        // -0.0 == +0.0 under '>': the earlier pixel wins
      k |= key << (16 * h);
    }
    u[i] = k;
  }
  return w;
}

template <typename T, int CB, int NT, typename IT, bool KEY, bool BAND = false>
__global__ __launch_bounds__(NT) void roi_pool_fwd_plane_kernel(int H, int W, int C, long ld, int PH, int PW, float scale,
                                                                const T* __restrict__ feat, const float* __restrict__ rois,
                                                                int R, int chunk, const float* __restrict__ row_scale,
                                                                float row_scale_add, T* __restrict__ out,
                                                                IT* __restrict__ argmax, int band_S, int band_rows, int n_bands, int n_zsplit) {
  static_assert(!BAND || KEY, "the band form exists for the packed-key scan only");
  constexpr int PXB = CB * (int)sizeof(T);
  constexpr int NWORD = PXB / 4;
  typedef typename PixWord<PXB>::type word_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_cnt, s_ntask;
  const int band = BAND ? (int)(blockIdx.z % n_bands) : 0, zfirst = BAND ? (int)(blockIdx.z / n_bands) : (int)blockIdx.z;
  const int y0 = BAND ? band * band_S : 0, y1 = BAND ? min(H, y0 + band_rows) : H;      // map rows [y0, y1) live in LDS
  const int lds_rows = BAND ? band_rows : H;
  word_t* plane = (word_t*)smem;                                   // [rows*W] pixels x CB channels
  int* s_list = (int*)(smem + (((size_t)lds_rows * W * PXB + 15) & ~(size_t)15));   // [chunk] ROIs of this image
  unsigned short* s_hb = (unsigned short*)(s_list + chunk);        // [chunk][PH] bin row range  start | end << 8  (H, W <= 255)
  unsigned short* s_wb = s_hb + chunk * PH;                        // [chunk][PW] bin column range
  float* s_mul = (float*)(s_wb + chunk * PW + ((chunk * (PH + PW)) & 1));   // [chunk] output scale of the ROI (4-byte aligned)
  unsigned int* s_task = (unsigned int*)(s_mul + chunk);           // BAND: [chunk * PH] owned (ROI << 8 | bin row)
  const int c0 = xcd_grouped_slab() * CB, img = blockIdx.y;
  const int tid = threadIdx.x;
  const int npx = H * W;
  const T* fimg = feat + (long)img * npx * C + c0;
  constexpr int NC1 = 7, NCLS = NC1 * NC1;
  __shared__ int s_hist[NCLS + 1];
  bool plane_loaded = false;
  // The workgroup keeps its slab (band) of the map and walks ROI chunks zfirst, zfirst + n_zsplit, ...: the slab is fetched once
  // per workgroup, not once per chunk — every pixel's 16 bytes are their own 64-byte L2 sector at a 1 KiB NHWC pixel pitch, so the
  // plane loads of one workgroup per chunk cost as much as the scan on large maps.  n_zsplit is sized by the host to fill the chip.
  for (int zchunk = zfirst; zchunk * chunk < R; zchunk += n_zsplit) {
  const int r0 = zchunk * chunk, r1 = min(R, r0 + chunk);
  __syncthreads();                                            // the previous chunk's tables are no longer read
  // ROIs of this image, grouped by bin-window size class (counting sort): a wave holds the 49 bins of one
  // ROI plus 15 of the next, and walks every lane's window to the longest one — neighbours of similar size waste less
  // The scan below runs every lane of a wave max(window rows) x max(window columns) times, so the classes are 2-D: 7 classes
  // of the window height x 7 of its width (an area class put 2x8 and 8x2 windows side by side: 64 iterations for 16 pixels).
  if (tid <= NCLS) s_hist[tid] = 0;
  if (tid == 0) { s_cnt = 0; s_ntask = 0; }
  __syncthreads();
  int my_cls = -1;                                            // chunk <= NT: at most one ROI per thread
  for (int r = r0 + tid; r < r1; r += NT)
    if ((int)rois[(long)r * 5] == img) {
      const RoiGeom g0 = roi_geom(rois + (long)r * 5, scale, PH, PW);
      auto cls1 = [](float b) { return b < 1.f ? 0 : b < 2.f ? 1 : b < 3.f ? 2 : b < 4.f ? 3 : b < 6.f ? 4 : b < 8.f ? 5 : 6; };
      my_cls = cls1(g0.bin_h) * NC1 + cls1(g0.bin_w);          // window ~ (ceil(bin_h) + 1) x (ceil(bin_w) + 1) pixels
      atomicAdd(&s_hist[my_cls + 1], 1);
      atomicAdd(&s_cnt, 1);
    }
  __syncthreads();
  if (tid == 0) { for (int c = 1; c <= NCLS; ++c) s_hist[c] += s_hist[c - 1]; }
  __syncthreads();
  if (my_cls >= 0) s_list[atomicAdd(&s_hist[my_cls], 1)] = r0 + tid;
  __syncthreads();
  const int cnt = s_cnt;
  if (cnt == 0) continue;
  if (!plane_loaded) {
    plane_loaded = true;
    for (int px = tid; px < (y1 - y0) * W; px += NT) {
      word_t w = *(const word_t*)(fimg + (long)(y0 * W + px) * C);
      if (KEY) w = pix_to_keys<T, CB>(w);
      plane[px] = w;
    }
    __syncthreads();
  }
  // bin ranges once per (ROI, bin row / column) instead of once per (ROI, bin, channel slab lane): the scan below is
  // VALU-issue bound (rocprof: 31 VALU instructions per window pixel, a third of them this per-task geometry)
  for (int i = tid; i < cnt * (PH + PW); i += NT) {
    const int li = i / (PH + PW), k = i - li * (PH + PW);
    const RoiGeom g = roi_geom(rois + (long)s_list[li] * 5, scale, PH, PW);
    if (k == 0) s_mul[li] = row_scale ? (row_scale[s_list[li]] + row_scale_add) : 1.0f;
    if (k < PH) {
      int hs = (int)floorf(__fmul_rn((float)k, g.bin_h)), he = (int)ceilf(__fmul_rn((float)(k + 1), g.bin_h));
      hs = min(max(hs + g.start_h, 0), H); he = min(max(he + g.start_h, 0), H);
      s_hb[li * PH + k] = (unsigned short)(hs | (he << 8));
    } else {
      const int pw = k - PH;
      int ws = (int)floorf(__fmul_rn((float)pw, g.bin_w)), we = (int)ceilf(__fmul_rn((float)(pw + 1), g.bin_w));
      ws = min(max(ws + g.start_w, 0), W); we = min(max(we + g.start_w, 0), W);
      s_wb[li * PW + pw] = (unsigned short)(ws | (we << 8));
    }
  }
  __syncthreads();
  const int nb = PH * PW;
  constexpr unsigned KEY_INIT = 0x007FFFFFu;                       // key(-inf) << 16 | 0xFFFF
  int ntask = cnt * PH;                                            // (ROI, bin row) pairs this workgroup works on
  if (BAND) {
    // the pairs whose first window row falls into this band's first band_S rows, compacted (order inside a wave kept: the size
    // classes stay together; across waves it is whatever order the counter was reached in — the results do not depend on it)
    const int lane = tid & 63;
    for (int e0 = 0; e0 < cnt * PH; e0 += NT) {
      const int e = e0 + tid;
      bool mine = false;
      if (e < cnt * PH) mine = min((int)(s_hb[e] & 0xFF) / band_S, n_bands - 1) == band;
      const unsigned long long m = __ballot(mine);
      int base = 0;
      if (lane == 0 && m) base = atomicAdd(&s_ntask, __popcll(m));
      base = __shfl(base, 0);
      if (mine) { const int li_ = e / PH; s_task[base + __popcll(m & ((1ull << lane) - 1))] = (unsigned)(li_ << 8) | (unsigned)(e - li_ * PH); }
    }
    __syncthreads();
    ntask = s_ntask;
  }
  const int total = ntask * PW;
  // task t = (ROI li of the list, bin row ph, bin column pw), t advancing by NT per iteration: carried as three counters
  // (two integer divisions per task were a fifth of the ~230 VALU instructions a task spends outside its window scan)
  const int dli = NT / nb, dph = (NT - dli * nb) / PW, dpw = NT - dli * nb - dph * PW;
  int li = tid / nb, ph = (tid - li * nb) / PW, pw = tid - li * nb - ph * PW;
  const int dte = NT / PW, dtw = NT - dte * PW;                    // BAND: (entry of the owned list, bin column)
  int te = tid / PW;
  if (BAND) pw = tid - te * PW;
  const unsigned int inv_base = 0xFFFEu - (unsigned)(y0 * W);      // plane index i holds map pixel y0 * W + i
  for (int t = tid; t < total; t += NT) {
    if (BAND) { const unsigned int pk = s_task[te]; li = (int)(pk >> 8); ph = (int)(pk & 0xFF); }
    const int b = ph * PW + pw;
    const int r = s_list[li];
    const int hb = s_hb[li * PH + ph], wb = s_wb[li * PW + pw];
    const int hs = hb & 0xFF, he = hb >> 8, ws = wb & 0xFF, we = wb >> 8;
    const bool empty = (he <= hs) || (we <= ws);
    float mv[CB]; int mi[CB];
    if (KEY) {
      unsigned int best[CB];
#pragma unroll
      for (int q = 0; q < CB; ++q) best[q] = KEY_INIT;
      if (!empty) {
        // two window pixels per step (the second clamped to the row's last pixel: re-reading a pixel cannot change a
        // maximum): v_max3_u32 folds both keys into the running best — 3 VALU per channel and pixel PAIR instead of 4 — and the
        // loop / address overhead (a third of the 27 VALU per visit this scan spent; rocprof: 126 M wave instructions per
        // 4000 ROIs, VALU 86 % busy) is paid once per pair; windows 3 and 4 pixels wide both take two steps
        const int bw = we - ws;
        const int he_lds = BAND ? min(he, y1) : he;
        for (int hh = hs; hh < he_lds; ++hh) {
          const int rowi = (hh - y0) * W + ws;
          for (int x = 0; x < bw; x += 2) {
            const int i0 = rowi + x, i1 = rowi + min(x + 1, bw - 1);
            const word_t w0 = plane[i0], w1 = plane[i1];
            const unsigned int* u0 = (const unsigned int*)&w0;
            const unsigned int* u1 = (const unsigned int*)&w1;
            const unsigned int inv0 = inv_base - (unsigned)i0, inv1 = inv_base - (unsigned)i1;
#pragma unroll
            for (int i = 0; i < NWORD; ++i) {
              best[2 * i] = max(max(best[2 * i], (u0[i] << 16) | inv0), (u1[i] << 16) | inv1);
              best[2 * i + 1] = max(max(best[2 * i + 1], (u0[i] & 0xFFFF0000u) | inv0), (u1[i] & 0xFFFF0000u) | inv1);
            }
          }
        }
        if (BAND) {                            // window rows below the band (a ROI reaching far outside the image): from global memory
          for (int hh = max(hs, y1); hh < he; ++hh)
            for (int x = ws; x < we; ++x) {
              const int gi = hh * W + x;
              const word_t w0 = pix_to_keys<T, CB>(*(const word_t*)(fimg + (long)gi * C));
              const unsigned int* u0 = (const unsigned int*)&w0;
              const unsigned int inv0 = 0xFFFEu - (unsigned)gi;
#pragma unroll
              for (int i = 0; i < NWORD; ++i) {
                best[2 * i] = max(best[2 * i], (u0[i] << 16) | inv0);
                best[2 * i + 1] = max(best[2 * i + 1], (u0[i] & 0xFFFF0000u) | inv0);
              }
            }
        }
      }
      // key -> bf16 bits: positive values (key bit 15 set) flip that bit back, negative ones were stored complemented; the pixel
      // index 0xFFFE - low half is -1 by itself when nothing won (low half still 0xFFFF).  best never drops below KEY_INIT, and a
      // channel still AT it (empty bin, or a window of -inf / NaN only) is the one case the reference answers differently
      // (0 resp. -FLT_MAX): one min over the channels finds it, so the common path pays no per-channel selects
      unsigned int lowest = best[0];
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const unsigned int m = (unsigned int)((int)best[q] >> 31);
        mv[q] = __uint_as_float((best[q] & 0xFFFF0000u) ^ (0x80000000u | (~m & 0x7FFF0000u)));
        mi[q] = (int)(0xFFFEu - (best[q] & 0xFFFFu));
        lowest = min(lowest, best[q]);
      }
      if (lowest == KEY_INIT) {
#pragma unroll
        for (int q = 0; q < CB; ++q)
          if (best[q] == KEY_INIT) mv[q] = empty ? 0.f : -FLT_MAX;
      }
    } else {
#pragma unroll
      for (int q = 0; q < CB; ++q) { mv[q] = empty ? 0.f : -FLT_MAX; mi[q] = -1; }
      if (!empty) {
        const int bw = we - ws;
        for (int hh = hs; hh < he; ++hh) {
          const int rowi = hh * W + ws;
          for (int x = 0; x < bw; ++x) {
            const int idx = rowi + x;
            float v[CB];
            pix_decode<T, CB>(plane[idx], v);
#pragma unroll
            for (int q = 0; q < CB; ++q)
              if (v[q] > mv[q]) { mv[q] = v[q]; mi[q] = idx; }
          }
        }
      }
    }
    // 2 x CB two-byte stores per task (a 64-lane wave writes one contiguous 98-byte run per channel and ROI): measured 45 us of the
    // 260 us a 4000-ROI call takes (stores compiled out: 215 us), the rest is the scan's VALU issue — staging the rows through LDS
    // for 16-byte stores would need 31 KiB more LDS and drop 63x63 maps to one workgroup per CU (measured with padded LDS: +10 %),
    // so the stores stay direct
    const float mul = s_mul[li];
    const long o = (long)r * ld + (long)c0 * nb + b;
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      Elem<T>::store(out + o + (long)q * nb, __fmul_rn(mv[q], mul));        // (nontemporal 2-byte stores: 250 -> 415 us, they are
      argmax[o + (long)q * nb] = ArgIdx<IT>::enc(mi[q]);                    //  not merged into full lines on the way out)
    }
    if (BAND) {
      pw += dtw; te += dte;
      if (pw >= PW) { pw -= PW; ++te; }
    } else {
      pw += dpw; ph += dph; li += dli;
      if (pw >= PW) { pw -= PW; ++ph; }
      if (ph >= PH) { ph -= PH; ++li; }
    }
  }
  }   // ROI chunks
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Forward, ROW SPARSE TABLE form (round 5; bf16 maps of < 65535 pixels, the hot path).
//
// The feature-stationary scan above visits every pixel of every bin window: ~15 VALU per visited pixel and 8 channels, so a call
// costs in proportion to the ROI AREA in map pixels (63x63 map / 4000 ROIs: 202 us = 0.25 of the output stream's HBM time;
// 99x165 / 8000 ROIs: 1.18 ms = 0.085).  Here the LDS copy of the slab holds, per pixel and channel, the FINISHED 32-bit candidate
//     key << 16 | (0xFFFE - pixel)
// (order-preserving 16-bit key of the bf16 value; larger candidate = larger value, else EARLIER pixel in row-major order: exactly
// the winner of the reference's strict '>' scan, ROILoopPool_cpu.cpp:60-72) and is then turned IN PLACE into the row sparse table of
// level L:  T_L[y][x] = max of the candidates of pixels x .. min(x + 2^L, W) - 1 of row y  (T_L = max(T_{L-1}[x], T_{L-1}[x + 2^{L-1}])).
// max is associative and the candidate carries its own pixel, so the maximum of ANY union of spans is still the reference's winner.
// A window row [ws, we) of width bw >= 2^L is the union of the two spans at ws and we - 2^L: TWO reads per window row, whatever
// its width.  Levels are per ROI: L = floor(log2(narrowest unclipped bin window)) — the unclipped widths of one ROI differ by at
// most one pixel, so 2^L <= bw <= 2^(L+1) and the two spans cover.  Windows the map's right edge clips below 2^L read ONE span
// (the table is clipped at W); anything else narrower (a ROI sticking out on the left) takes a pixel loop over global memory.
// A workgroup sorts ITS share of the image's ROIs (every n_zsplit-th block of 128 rows) by (level, window-height class), then walks the
// levels upwards, advancing its table between them, in chunks of 128 ROIs.  Per (ROI, bin, 8 channels): ~13 VALU per
// window ROW instead of ~31 per pixel PAIR; what remains is the per-task work outside the scan (16 two-byte stores, key -> value ->
// x prior -> bf16).
// CB = 8: two 16-byte planes (channels 0-3 | 4-7) = 32 B per pixel, the whole map in LDS (<= ~4600 pixels: 63x63);
// CB = 4 + BAND: one plane, row bands with a halo as in the scan form above (large maps: 99x165 -> 3 bands), rows below a band from
// global memory.
constexpr int SP_NT = 1024, SP_CH = 128, SP_NLEV = 6, SP_NHC = 7, SP_NCLS = SP_NLEV * SP_NHC;

// development instrumentation (tools/build_variant.sh phases SRC=roipool -DSW_ROI_PHASES; tools/probes/roi_phases.py): shader-clock cycles
// of workgroup thread 0 per phase, summed over the workgroups.  0 sort, 1 level-0 table, 2 level advances, 3 chunk tables + task list,
// 4 task scan, 5 wait at the barriers behind a scan (imbalance inside a chunk), 6 whole kernel, 7 workgroups
#ifdef SW_ROI_PHASES
__device__ unsigned long long g_roi_phase[8];
#define SP_T(var) const long long var = (long long)clock64()
#define SP_ADD(i, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_roi_phase[i], (unsigned long long)((b) - (a))); } while (0)
#else
#define SP_T(var) do {} while (0)
#define SP_ADD(i, a, b) do {} while (0)
#endif

__device__ __forceinline__ unsigned int key16_of(unsigned int bts) {          // pix_to_keys for one bf16 value
  unsigned int key = (bts & 0x8000u) ? (bts ^ 0xFFFFu) : (bts | 0x8000u);
  if (bts > 0x7F80u && bts < 0x8000u) key = 0;                                // +NaN never wins
  if (bts == 0x8000u) key = 0x8000u;                                          // -0.0 == +0.0 under '>'
  return key;
}

template <typename IT, int CB, bool BAND, int PFIX = 0>
__global__ __launch_bounds__(SP_NT) void roi_pool_fwd_sparse_kernel(int H, int W, int C, long ld, int PH_, int PW_, float scale,
                                                                    const unsigned short* __restrict__ feat,
                                                                    const float* __restrict__ rois, int R,
                                                                    const float* __restrict__ row_scale, float row_scale_add,
                                                                    unsigned short* __restrict__ out, IT* __restrict__ argmax,
                                                                    int band_S, int band_rows, int n_bands, int n_zsplit, int SP_CHK) {
  // SP_CHK: ROIs per chunk (round 6: a run-time value; see launch_fwd_sparse).  PFIX = 7: the pooled size as a compile-time constant (the hot path's 7 x 7): the 16 stores of a task then take immediate offsets
  // (q * 98 bytes) instead of 64-bit address arithmetic per channel, and the task counters fold
  const int PH = PFIX ? PFIX : PH_, PW = PFIX ? PFIX : PW_;
  constexpr int NPL = CB / 4;                                       // 16-byte planes
  constexpr int PXT = CB == 8 ? 5 : 10;                             // table pixels per thread (host: rows * W <= PXT * SP_NT)
  constexpr int NT = SP_NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_start[SP_NCLS + 1], s_fill[SP_NCLS];
  __shared__ int s_ntask, s_claim;
  const int band = BAND ? (int)(blockIdx.z % n_bands) : 0, zfirst = BAND ? (int)(blockIdx.z / n_bands) : (int)blockIdx.z;
  const int y0 = BAND ? band * band_S : 0, y1 = BAND ? min(H, y0 + band_rows) : H;      // map rows [y0, y1) live in LDS
  const int lds_rows = BAND ? band_rows : H;
  const int npl_px = lds_rows * W;                                  // pixels per plane (allocated)
  const int npx_t = (y1 - y0) * W;                                  // pixels held
  u32x4* tab = (u32x4*)smem;                                        // [NPL][npl_px]
  unsigned short* s_lvl = (unsigned short*)(smem + (size_t)NPL * npl_px * 16);          // [R] ROI rows of this image, by class
  const int r_even = (R + 1) & ~1;
  int* s_r = (int*)(s_lvl + r_even);                                // [CH] ROI row of the chunk's entries
  float* s_mul = (float*)(s_r + SP_CHK);                            // [CH] output scale
  unsigned short* s_hb = (unsigned short*)(s_mul + SP_CHK);         // [CH][PH] bin row range  start | end << 8
  unsigned short* s_wb = s_hb + SP_CHK * PH;                        // [CH][PW]
  unsigned int* s_task = (unsigned int*)(s_wb + SP_CHK * PW + ((SP_CHK * (PH + PW)) & 1));   // BAND: [CH * PH] owned (entry << 8 | bin row)
  const int c0 = xcd_grouped_slab() * CB, img = blockIdx.y;
  const int tid = threadIdx.x;
  const int nb = PH * PW;
  const unsigned short* fimg = feat + (long)img * H * W * C + c0;

  // ---- 1. this image's ROIs sorted by (level, height class): counting sort, two passes over the ROI geometry
  auto roi_class = [&](int r) {
    const RoiGeom g = roi_geom(rois + (long)r * 5, scale, PH, PW);
    int m = 1 << 30;                                                // narrowest UNCLIPPED window of the ROI
    for (int pw = 0; pw < PW; ++pw) {
      const int ws = (int)floorf(__fmul_rn((float)pw, g.bin_w)), we = (int)ceilf(__fmul_rn((float)(pw + 1), g.bin_w));
      m = min(m, we - ws);
    }
    m = max(m, 1);
    const int L = min(31 - __clz(m), SP_NLEV - 1);
    const float b = g.bin_h;
    const int hc = b < 1.f ? 0 : b < 2.f ? 1 : b < 3.f ? 2 : b < 4.f ? 3 : b < 6.f ? 4 : b < 8.f ? 5 : 6;
    return L * SP_NHC + hc;
  };
  // The n_zsplit workgroups of one (slab, image, band) split the ROIs by ROW BLOCK (128 consecutive rows of `rois` each, dealt
  // round-robin), not by position in the sorted list: the order inside a class is whatever the atomics give, so a list position means
  // something to the workgroup that built the list only.  BAND: only ROIs with a bin row that STARTS in this band's own rows enter
  // the list (first / last bin row start are monotone in the bin row) — every chunk's tables are then built for ROIs that have
  // work here (with all ROIs listed, a band of nine spent most of a chunk on the geometry of ROIs whose bins belong to other bands).
  auto for_my_rois = [&](auto&& fn) {
    for (int b0 = zfirst; b0 * 128 < R; b0 += 8 * n_zsplit) {
      const int r = (b0 + (tid >> 7) * n_zsplit) * 128 + (tid & 127);
      if (r < R && (int)rois[(long)r * 5] == img) {
        if (BAND) {
          const RoiGeom g = roi_geom(rois + (long)r * 5, scale, PH, PW);
          const int h_first = min(max(g.start_h, 0), H);
          const int h_last = min(max((int)floorf(__fmul_rn((float)(PH - 1), g.bin_h)) + g.start_h, 0), H);
          if (min(h_first / band_S, n_bands - 1) > band || min(h_last / band_S, n_bands - 1) < band) continue;
        }
        fn(r);
      }
    }
  };
  SP_T(t_begin);
  if (tid <= SP_NCLS) s_start[tid] = 0;
  __syncthreads();
  for_my_rois([&](int r) { atomicAdd(&s_start[roi_class(r) + 1], 1); });
  __syncthreads();
  if (tid == 0) { for (int c = 1; c <= SP_NCLS; ++c) s_start[c] += s_start[c - 1]; }
  __syncthreads();
  if (tid < SP_NCLS) s_fill[tid] = s_start[tid];
  __syncthreads();
  for_my_rois([&](int r) { s_lvl[atomicAdd(&s_fill[roi_class(r)], 1)] = (unsigned short)r; });
  const int n_rois = s_start[SP_NCLS];
  SP_T(t_sorted); SP_ADD(0, t_begin, t_sorted); SP_ADD(7, 0, 1);
  if (n_rois == 0) return;                                          // (uniform: s_start is final since the barrier above)

  // ---- 2. level 0: candidates of the slab's pixels
  int xk[PXT];                                                      // column of this thread's k-th table pixel (for the level advance)
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const int px = tid + k * NT;
    xk[k] = px < npx_t ? px - (px / W) * W : -1;
    if (px < npx_t) {
      const int gpx = y0 * W + px;
      const unsigned int inv = 0xFFFEu - (unsigned)gpx;
      unsigned int u[CB / 2];
      if (CB == 8) { const u32x4 w = *(const u32x4*)(fimg + (long)gpx * C); u[0] = w[0]; u[1] = w[1]; u[2] = w[2]; u[3] = w[3]; }
      else { const u32x2 w = *(const u32x2*)(fimg + (long)gpx * C); u[0] = w[0]; u[1] = w[1]; }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        u32x4 t;
        t[0] = (key16_of(u[2 * pl] & 0xFFFFu) << 16) | inv;     t[1] = (key16_of(u[2 * pl] >> 16) << 16) | inv;
        t[2] = (key16_of(u[2 * pl + 1] & 0xFFFFu) << 16) | inv; t[3] = (key16_of(u[2 * pl + 1] >> 16) << 16) | inv;
        tab[pl * npl_px + px] = t;
      }
    }
  }
  constexpr unsigned KEY_INIT = 0x007FFFFFu;                        // key(-inf) << 16 | 0xFFFF
  SP_T(t_tab0); SP_ADD(1, t_sorted, t_tab0);
  for (int L = 0; L < SP_NLEV; ++L) {
    const int c_lo = s_start[L * SP_NHC], c_hi = s_start[(L + 1) * SP_NHC];
    if (c_lo >= n_rois) break;                                      // no ROI at this or a higher level
    SP_T(t_lv0);
    __syncthreads();                                                // level L - 1 fully scanned (L = 0: table and list written)
    SP_T(t_lv1); SP_ADD(5, t_lv0, t_lv1);
    if (L > 0) {
      // ---- 3. T_L from T_{L-1}, in place: every thread reads its pixels' two spans, barrier, writes
      const int d = 1 << (L - 1);
      u32x4 nv[PXT][NPL];
#pragma unroll
      for (int k = 0; k < PXT; ++k) {
        const int px = tid + k * NT;
        if (px < npx_t) {
          const bool two = xk[k] + d < W;                           // the second span starts inside the row
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            const u32x4 a = tab[pl * npl_px + px];
            const u32x4 b = two ? tab[pl * npl_px + px + d] : a;
            nv[k][pl] = u32x4{max(a[0], b[0]), max(a[1], b[1]), max(a[2], b[2]), max(a[3], b[3])};
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < PXT; ++k) {
        const int px = tid + k * NT;
        if (px < npx_t) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) tab[pl * npl_px + px] = nv[k][pl];
        }
      }
      __syncthreads();
    }
    const int span = 1 << L;
    SP_T(t_lv2); SP_ADD(2, t_lv1, t_lv2);
    for (int cs = c_lo; cs < c_hi; cs += SP_CHK) {
      const int cnt = min(SP_CHK, c_hi - cs);
      SP_T(t_c0);
      __syncthreads();                                              // the previous chunk's tables are no longer read
      SP_T(t_c1); SP_ADD(5, t_c0, t_c1);
      if (tid == 0) { s_ntask = 0; s_claim = 0; }
      for (int i = tid; i < cnt * (PH + PW); i += NT) {
        const int li = i / (PH + PW), k = i - li * (PH + PW);
        const int r = s_lvl[cs + li];
        const RoiGeom g = roi_geom(rois + (long)r * 5, scale, PH, PW);
        if (k == 0) { s_r[li] = r; s_mul[li] = row_scale ? (row_scale[r] + row_scale_add) : 1.0f; }
        if (k < PH) {
          int hs = (int)floorf(__fmul_rn((float)k, g.bin_h)), he = (int)ceilf(__fmul_rn((float)(k + 1), g.bin_h));
          hs = min(max(hs + g.start_h, 0), H); he = min(max(he + g.start_h, 0), H);
          s_hb[li * PH + k] = (unsigned short)(hs | (he << 8));
        } else {
          const int pw = k - PH;
          int ws = (int)floorf(__fmul_rn((float)pw, g.bin_w)), we = (int)ceilf(__fmul_rn((float)(pw + 1), g.bin_w));
          ws = min(max(ws + g.start_w, 0), W); we = min(max(we + g.start_w, 0), W);
          s_wb[li * PW + pw] = (unsigned short)(ws | (we << 8));
        }
      }
      __syncthreads();
      int ntask = cnt * PH;                                         // (ROI, bin row) pairs this workgroup works on
      if (BAND) {
        const int lane = tid & 63;
        for (int e0 = 0; e0 < cnt * PH; e0 += NT) {
          const int e = e0 + tid;
          bool mine = false;
          if (e < cnt * PH) mine = min((int)(s_hb[e] & 0xFF) / band_S, n_bands - 1) == band;
          const unsigned long long mk = __ballot(mine);
          int base = 0;
          if (lane == 0 && mk) base = atomicAdd(&s_ntask, __popcll(mk));
          base = __shfl(base, 0);
          if (mine) { const int li_ = e / PH; s_task[base + __popcll(mk & ((1ull << lane) - 1))] = (unsigned)(li_ << 8) | (unsigned)(e - li_ * PH); }
        }
        __syncthreads();
        ntask = s_ntask;
      }
      SP_T(t_c2); SP_ADD(3, t_c1, t_c2);
      const int total = ntask * PW;
      // 64-item units CLAIMED by the waves from an LDS counter (round 6; dealt by position, a wave waited at the chunk's closing barrier
      // for 23 % of the workgroup's cycles on the 63x63 map: 98 units per 128-ROI chunk over 16 waves, and the units differ in length)
      for (;;) {
        int u = 0;
        if ((tid & 63) == 0) u = atomicAdd(&s_claim, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u * 64 >= total) break;
        const int t = min(u * 64 + (int)(tid & 63), total - 1);     // lanes past the end redo the last item (same stores, same values)
        int li, ph, pw;
        if (BAND) { const int te = t / PW; pw = t - te * PW; const unsigned int pk = s_task[te]; li = (int)(pk >> 8); ph = (int)(pk & 0xFF); }
        else { li = t / nb; const int rem = t - li * nb; ph = rem / PW; pw = rem - ph * PW; }
        const int b = ph * PW + pw;
        const int r = s_r[li];
        const int hb = s_hb[li * PH + ph], wb = s_wb[li * PW + pw];
        const int hs = hb & 0xFF, he = hb >> 8, ws = wb & 0xFF, we = wb >> 8;
        const bool empty = (he <= hs) || (we <= ws);
        unsigned int best[CB];
#pragma unroll
        for (int q = 0; q < CB; ++q) best[q] = KEY_INIT;
        if (!empty) {
          const int bw = we - ws;
          const int he_lds = BAND ? min(he, y1) : he;
          int cold_from = BAND ? max(hs, y1) : he;                  // first window row read from global memory
          if (bw >= span) {
            const int offB = bw - span;
            // two window rows per step (the second clamped to the last row: re-reading a row cannot change a maximum): all 4 * NPL
            // reads of a step are issued before the first use — one LDS round trip per row PAIR instead of per row (rocprof, one row
            // per step: the waves of this kernel were parked at s_waitcnt / barriers 53 % of their time, VALU and LDS both < 50 % busy)
            const int last = (he_lds - 1 - y0) * W + ws;
            for (int hh = hs; hh < he_lds; hh += 2) {
              const int i0 = (hh - y0) * W + ws, i1 = min(i0 + W, last);
              u32x4 a0[NPL], b0[NPL], a1[NPL], b1[NPL];
#pragma unroll
              for (int pl = 0; pl < NPL; ++pl) {
                a0[pl] = tab[pl * npl_px + i0]; b0[pl] = tab[pl * npl_px + i0 + offB];
                a1[pl] = tab[pl * npl_px + i1]; b1[pl] = tab[pl * npl_px + i1 + offB];
              }
#pragma unroll
              for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  best[4 * pl + e] = max(max(max(max(best[4 * pl + e], a0[pl][e]), b0[pl][e]), a1[pl][e]), b1[pl][e]);     // 2 x v_max3_u32
              if (2 * span < bw) {                                  // only when the level was capped (SP_NLEV): spans in between
                for (int rr = 0; rr < 2; ++rr) {
                  const int ib = rr ? i1 : i0;
                  for (int x = span; x < offB; x += span) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                      const u32x4 a = tab[pl * npl_px + ib + x];
#pragma unroll
                      for (int e = 0; e < 4; ++e) best[4 * pl + e] = max(best[4 * pl + e], a[e]);
                    }
                  }
                }
              }
            }
          } else if (we == W) {                                     // clipped by the map's right edge: the span at ws ends at W
            for (int hh = hs; hh < he_lds; ++hh) {
              const int i0 = (hh - y0) * W + ws;
#pragma unroll
              for (int pl = 0; pl < NPL; ++pl) {
                const u32x4 a = tab[pl * npl_px + i0];
#pragma unroll
                for (int e = 0; e < 4; ++e) best[4 * pl + e] = max(best[4 * pl + e], a[e]);
              }
            }
          } else {
            cold_from = hs;                                         // narrower than the ROI's span for another reason: pixel loop
          }
          for (int hh = cold_from; hh < he; ++hh)
            for (int x = ws; x < we; ++x) {
              const int gi = hh * W + x;
              const unsigned int inv0 = 0xFFFEu - (unsigned)gi;
              unsigned int u[CB / 2];
              if (CB == 8) { const u32x4 w = *(const u32x4*)(fimg + (long)gi * C); u[0] = w[0]; u[1] = w[1]; u[2] = w[2]; u[3] = w[3]; }
              else { const u32x2 w = *(const u32x2*)(fimg + (long)gi * C); u[0] = w[0]; u[1] = w[1]; }
#pragma unroll
              for (int i = 0; i < CB / 2; ++i) {
                best[2 * i] = max(best[2 * i], (key16_of(u[i] & 0xFFFFu) << 16) | inv0);
                best[2 * i + 1] = max(best[2 * i + 1], (key16_of(u[i] >> 16) << 16) | inv0);
              }
            }
        }
        // candidate -> (bf16 bits, pixel): as in the scan form above
        float mv[CB]; int mi[CB];
        unsigned int lowest = best[0];
#pragma unroll
        for (int q = 0; q < CB; ++q) {
          const unsigned int m = (unsigned int)((int)best[q] >> 31);
          mv[q] = __uint_as_float((best[q] & 0xFFFF0000u) ^ (0x80000000u | (~m & 0x7FFF0000u)));
          mi[q] = (int)(0xFFFEu - (best[q] & 0xFFFFu));
          lowest = min(lowest, best[q]);
        }
        if (lowest == KEY_INIT) {
#pragma unroll
          for (int q = 0; q < CB; ++q)
            if (best[q] == KEY_INIT) mv[q] = empty ? 0.f : -FLT_MAX;
        }
        const float mul = s_mul[li];
        const long o = (long)r * ld + (long)c0 * nb + b;
#pragma unroll
        for (int q = 0; q < CB; ++q) {
          Elem<unsigned short>::store(out + o + (long)q * nb, __fmul_rn(mv[q], mul));
          argmax[o + (long)q * nb] = ArgIdx<IT>::enc(mi[q]);
        }
      }
      SP_T(t_c3); SP_ADD(4, t_c2, t_c3);
    }   // chunks of the level
  }     // levels
  SP_T(t_end); SP_ADD(6, t_begin, t_end);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Forward, ROW SPARSE TABLE form over a PREPARED TASK LIST (round 6; sw_roi_pool_fwd_ws with a workspace).
//
// The sparse-table kernel above spends half of a workgroup's cycles outside the scan on large maps (99x165 / 8000 ROIs, phase
// clocks in profiles/r06_roi_phases.txt: ROI sort 5 %, per-chunk bin tables + owned-pair lists 29 %, barriers behind the chunk
// scans 21 %) — and all of that is the SAME work in each of the 128 channel slabs.  Here a small kernel does it once per call:
// every (ROI, bin row) pair becomes a 32-byte task record
//     { r | ph << 16,  hs | he << 8,  output scale,  - ,   wb[0] | wb[1] << 16, ... wb[6] | wb[7] << 16 }      (wb = ws | we << 8, clipped)
// filed under (image, row band of its first window row, class = level x window-height class) — a counting sort in ONE workgroup
// per image.  The pooling kernel then only builds its table and walks the levels; at each level its lanes stream the (task, bin
// column) items of that level straight from the list (records of the next item requested before the current one is scanned):
// no chunks, no per-chunk barriers, one barrier per level.  Arithmetic, table and scan are the sparse kernel's, results
// identical bit for bit.
constexpr int TK_MAXB = 16;                                         // row bands per map at most (4 bits in RoiGeo::bands)
constexpr int TK_SEG = SP_NCLS + 1;                                 // offsets per (image, band)

__device__ __forceinline__ int tk_roi_class(const RoiGeom& g, int PW) {
  int m = 1 << 30;                                                  // narrowest UNCLIPPED window of the ROI
  for (int pw = 0; pw < PW; ++pw) {
    const int ws = (int)floorf(__fmul_rn((float)pw, g.bin_w)), we = (int)ceilf(__fmul_rn((float)(pw + 1), g.bin_w));
    m = min(m, we - ws);
  }
  m = max(m, 1);
  const int L = min(31 - __clz(m), SP_NLEV - 1);
  const float b = g.bin_h;
  const int hc = b < 1.f ? 0 : b < 2.f ? 1 : b < 3.f ? 2 : b < 4.f ? 3 : b < 6.f ? 4 : b < 8.f ? 5 : 6;
  return L * SP_NHC + hc;
}

// The list is built by two small launches, 256 ROIs per workgroup, thread = ROI (one workgroup per image took 25 us per 4000 ROIs,
// all of it the ROI geometry on four SIMDs):
//   roi_pool_geo_kernel    geometry of the ROI -> geo[r] (bin row / column ranges, class, bands, output scale) and the workgroup's
//                          histogram over the cells (image, band, class) -> hist[workgroup][cell]
//   roi_pool_tasks_kernel  every workgroup sums the histograms (cell totals -> the segment offsets, written by workgroup 0; the
//                          workgroups before it -> where ITS tasks of a cell start) and files its ROIs' records; positions inside
//                          a workgroup's share of a cell come from LDS atomics — any order gives the same pooled output.
// tasks: [nimg][cap] records of 2 x u32x4; seg: [nimg][n_bands][TK_SEG] offsets into the image's records.
constexpr int TK_CELLS = 64 * SP_NCLS;                              // nimg * n_bands <= 64
constexpr int TK_PREP_NT = 256;
struct RoiGeo { unsigned int hb[4], wb[4]; float mul; unsigned int bands; int cell0; unsigned int pad; };      // 48 bytes; cell0 = -1: no image

__global__ __launch_bounds__(TK_PREP_NT) void roi_pool_geo_kernel(int nimg, int H, int W, int PH, int PW, float scale,
                                                                  const float* __restrict__ rois, int R,
                                                                  const float* __restrict__ row_scale, float row_scale_add, int band_S,
                                                                  int band_rows, int n_bands, RoiGeo* __restrict__ geo, int* __restrict__ hist) {
  __shared__ int s_cnt[TK_CELLS];
  const int tid = threadIdx.x, ncell = nimg * n_bands * SP_NCLS;
  for (int i = tid; i < ncell; i += TK_PREP_NT) s_cnt[i] = 0;
  __syncthreads();
  const int r = blockIdx.x * TK_PREP_NT + tid;
  if (r < R) {
    RoiGeo o;
    const RoiGeom g = roi_geom(rois + (long)r * 5, scale, PH, PW);
    const int img = g.batch;
    o.cell0 = -1; o.bands = 0u; o.pad = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.hb[k] = 0u; o.wb[k] = 0u; }
    o.mul = row_scale ? (row_scale[r] + row_scale_add) : 1.0f;
    if (img >= 0 && img < nimg) {
      o.cell0 = img * n_bands * SP_NCLS + tk_roi_class(g, PW);
      for (int pw = 0; pw < PW; ++pw) {
        int ws = (int)floorf(__fmul_rn((float)pw, g.bin_w)), we = (int)ceilf(__fmul_rn((float)(pw + 1), g.bin_w));
        ws = min(max(ws + g.start_w, 0), W); we = min(max(we + g.start_w, 0), W);
        o.wb[pw >> 1] |= (unsigned)(ws | (we << 8)) << (16 * (pw & 1));
      }
      int row_lo = H, row_hi = 0;                                   // map rows any non-empty bin of the ROI reads
      for (int ph = 0; ph < PH; ++ph) {
        int hs = (int)floorf(__fmul_rn((float)ph, g.bin_h)), he = (int)ceilf(__fmul_rn((float)(ph + 1), g.bin_h));
        hs = min(max(hs + g.start_h, 0), H); he = min(max(he + g.start_h, 0), H);
        o.hb[ph >> 1] |= (unsigned)(hs | (he << 8)) << (16 * (ph & 1));
        if (he > hs) { row_lo = min(row_lo, hs); row_hi = max(row_hi, he); }
      }
      // A ROI whose rows all lie inside ONE band's rows goes there whole: its 7 x 7 outputs per channel are then written by one
      // workgroup as one 98-byte run, as on a map that fits LDS.  Bin rows of the other ROIs go to the band that owns the bin's first
      // row (the 14-byte pieces of a channel's run then come from up to n_bands workgroups at different times: the partly written
      // lines leave the L2 in between — 99x165 map / 8000 ROIs: 235 us of the call, tools/roi_tasks_forms.py with ROI_MAXH).
      int whole = -1;
      for (int b = 0; b < n_bands && whole < 0; ++b) {
        const int y0 = min(b * band_S, max(0, H - band_rows));
        if (row_lo >= row_hi || (row_lo >= y0 && row_hi <= y0 + band_rows)) whole = b;
      }
      int prev = -1, run = 0;
      for (int ph = 0; ph < PH; ++ph) {
        const int hs = (int)((o.hb[ph >> 1] >> (16 * (ph & 1))) & 0xFFu);
        const int b = whole >= 0 ? whole : min(hs / band_S, n_bands - 1);                // (non-decreasing in ph)
        o.bands |= (unsigned)b << (4 * ph);
        if (b != prev) { if (run) atomicAdd(&s_cnt[o.cell0 + prev * SP_NCLS], run); prev = b; run = 0; }
        ++run;
      }
      if (run) atomicAdd(&s_cnt[o.cell0 + prev * SP_NCLS], run);
    }
    geo[r] = o;
  }
  __syncthreads();
  for (int i = tid; i < ncell; i += TK_PREP_NT) hist[(long)blockIdx.x * ncell + i] = s_cnt[i];
}

__global__ __launch_bounds__(TK_PREP_NT) void roi_pool_tasks_kernel(int nimg, int PH, int R, int n_bands, const RoiGeo* __restrict__ geo,
                                                                    const int* __restrict__ hist, u32x4* __restrict__ tasks,
                                                                    int* __restrict__ seg, long cap) {
  __shared__ int s_tot[TK_CELLS], s_fill[TK_CELLS], s_band[65];
  const int tid = threadIdx.x, ncell = nimg * n_bands * SP_NCLS, nwg = gridDim.x, me = blockIdx.x;
  for (int c = tid; c < ncell; c += TK_PREP_NT) {
    int tot = 0, before = 0;
    for (int w = 0; w < nwg; ++w) { const int v = hist[(long)w * ncell + c]; tot += v; before += w < me ? v : 0; }
    s_tot[c] = tot; s_fill[c] = before;
  }
  __syncthreads();
  const int nib = nimg * n_bands;                                   // (image, band) pairs: tasks of a pair = one run of classes
  if (tid < nib) { int t = 0; for (int c = 0; c < SP_NCLS; ++c) t += s_tot[tid * SP_NCLS + c]; s_band[tid + 1] = t; }
  __syncthreads();
  if (tid < nimg) {                                                 // exclusive offsets of the bands inside their image
    int t = 0;
    for (int b = 0; b < n_bands; ++b) { const int n = s_band[tid * n_bands + b + 1]; s_band[tid * n_bands + b + 1] = t; t += n; }
  }
  __syncthreads();
  if (tid < nib) {
    int t = s_band[tid + 1];
    int* sg = seg + (long)tid * TK_SEG;
    for (int c = 0; c < SP_NCLS; ++c) {
      const int n = s_tot[tid * SP_NCLS + c];
      s_fill[tid * SP_NCLS + c] += t;                               // cell start + the tasks earlier workgroups put there
      if (me == 0) sg[c] = t;
      t += n;
    }
    if (me == 0) sg[SP_NCLS] = t;
  }
  __syncthreads();
  const int r = me * TK_PREP_NT + tid;
  if (r >= R) return;
  const RoiGeo o = geo[r];
  if (o.cell0 < 0) return;
  const int img = o.cell0 / (n_bands * SP_NCLS);
  u32x4* my = tasks + (long)img * cap * 2;
  int prev = -1, pos = 0;
  for (int ph = 0; ph < PH; ++ph) {
    const int b = (int)((o.bands >> (4 * ph)) & 15u);
    if (b != prev) {                                                // reserve the whole run of bin rows this ROI has in band b
      int run = 1;
      for (int q = ph + 1; q < PH && (int)((o.bands >> (4 * q)) & 15u) == b; ++q) ++run;
      pos = atomicAdd(&s_fill[o.cell0 + b * SP_NCLS], run);
      prev = b;
    }
    const unsigned int hb = (o.hb[ph >> 1] >> (16 * (ph & 1))) & 0xFFFFu;
    my[2 * (long)pos] = u32x4{(unsigned)r | ((unsigned)ph << 16), hb, __float_as_uint(o.mul), 0u};
    my[2 * (long)pos + 1] = u32x4{o.wb[0], o.wb[1], o.wb[2], o.wb[3]};
    ++pos;
  }
}

template <typename IT, int CB, int NT, int PFIX = 0>
__global__ __launch_bounds__(NT) void roi_pool_fwd_tasks_kernel(int H, int W, int C, long ld, int PH_, int PW_,
                                                                const unsigned short* __restrict__ feat,
                                                                const u32x4* __restrict__ tasks, const int* __restrict__ seg, long cap,
                                                                unsigned short* __restrict__ out, IT* __restrict__ argmax,
                                                                int band_S, int band_rows, int n_bands, int n_zsplit) {
  const int PH = PFIX ? PFIX : PH_, PW = PFIX ? PFIX : PW_;
  constexpr int EW = CB == 2 ? 2 : 4;                               // channels per table entry: 8-byte entries for 2-channel slabs, else 16
  constexpr int NPL = CB / EW;                                      // planes of entries
  typedef typename std::conditional<CB == 2, u32x2, u32x4>::type TE;
  constexpr int PXT = (CB == 8 ? 5 : CB == 4 ? 10 : 20) * (1024 / NT);   // table pixels per thread (host: rows * W <= PXT * NT)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_seg[TK_SEG], s_next[SP_NLEV];                  // s_next[L]: units of level L claimed so far
  // (band = blockIdx.z % n_bands keeps the 8 slabs of a 64-channel group, which write neighbouring runs of every ROI row, in the same
  // band at the same time on one XCD; dealing the bands so that the workgroups of a round differ in length — to take the chip out of
  // step: tables, then stores, everywhere at once — was slower: 99x165 / 8000 ROIs 486 -> 520 us per 8-slab group, 613 per slab)
  const int band = (int)(blockIdx.z % n_bands), zfirst = (int)(blockIdx.z / n_bands);
  const int y0 = min(band * band_S, max(0, H - band_rows)), y1 = min(H, y0 + band_rows);        // map rows [y0, y1) live in LDS
  const int npl_px = band_rows * W;                                 // pixels per plane (allocated)
  const int npx_t = (y1 - y0) * W;                                  // pixels held
  TE* tab = (TE*)smem;                                              // [NPL][npl_px]
  const int c0 = xcd_grouped_slab() * CB, img = blockIdx.y;
  const int tid = threadIdx.x;
  const int nb = PH * PW;
  const unsigned short* fimg = feat + (long)img * H * W * C + c0;
  SP_T(t_begin);
  if (tid < TK_SEG) s_seg[tid] = seg[((long)img * n_bands + band) * TK_SEG + tid];
  if (tid < SP_NLEV) s_next[tid] = 0;
  __syncthreads();
  const int n_tasks_end = s_seg[SP_NCLS];
  if (n_tasks_end == s_seg[0]) return;                              // no bin row of this image starts in this band
  const u32x4* my = tasks + (long)img * cap * 2;

  // ---- level 0: candidates of the slab's pixels.  All of a thread's pixels are requested before the first is converted (one
  // s_waitcnt per pixel left the 10 loads of a 4-channel slab in series: every one a 64-byte sector of its own at the NHWC pixel pitch)
  int xk[PXT];
  unsigned int raw[PXT][CB / 2];
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const int px = tid + k * NT;
    xk[k] = px < npx_t ? px - (px / W) * W : -1;
#pragma unroll
    for (int j = 0; j < CB / 2; ++j) raw[k][j] = 0u;
    if (px < npx_t) {
      const unsigned short* src = fimg + (long)(y0 * W + px) * C;
      if constexpr (CB == 8) { const u32x4 w = *(const u32x4*)src; raw[k][0] = w[0]; raw[k][1] = w[1]; raw[k][2] = w[2]; raw[k][3] = w[3]; }
      else if constexpr (CB == 4) { const u32x2 w = *(const u32x2*)src; raw[k][0] = w[0]; raw[k][1] = w[1]; }
      else raw[k][0] = *(const unsigned int*)src;
    }
  }
#pragma unroll
  for (int k = 0; k < PXT; ++k) {
    const int px = tid + k * NT;
    if (px < npx_t) {
      const unsigned int inv = 0xFFFEu - (unsigned)(y0 * W + px);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        TE t;
#pragma unroll
        for (int e = 0; e < EW; ++e) {
          const unsigned int w2 = raw[k][(EW * pl + e) >> 1];
          t[e] = (key16_of((e & 1) ? (w2 >> 16) : (w2 & 0xFFFFu)) << 16) | inv;
        }
        tab[pl * npl_px + px] = t;
      }
    }
  }
  // Barriers of the level walk order LDS traffic only (table reads of level L - 1 before the advance writes, writes before the reads of
  // level L): __syncthreads() also waits for vmcnt(0), i.e. for every output store the wave still has in flight — the phase clocks showed
  // a wave parked at the level barriers for 26 % of a workgroup's cycles on the 99x165 map (most of it turned out to be the spread of
  // the units' lengths, see the claims below; the stores' share is what this form removes)
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr unsigned KEY_INIT = 0x007FFFFFu;                        // key(-inf) << 16 | 0xFFFF
  // (development instrumentation as in the sparse kernel, -DSW_ROI_PHASES: 0 -, 1 level-0 table, 2 level advances, 3 first records of
  // a level, 4 item loop, 5 barrier at a level's start, 6 whole kernel, 7 workgroups)
  SP_T(t_tab0); SP_ADD(1, t_begin, t_tab0); SP_ADD(7, 0, 1);
  for (int L = 0; L < SP_NLEV; ++L) {
    const int t_lo = s_seg[L * SP_NHC], t_hi = s_seg[(L + 1) * SP_NHC];
    if (t_lo >= n_tasks_end) break;                                 // no task at this or a higher level
    SP_T(t_lv0);
    lds_barrier();                                                // level L - 1 fully scanned (L = 0: table written)
    SP_T(t_lv1); SP_ADD(L == 0 ? 0 : 5, t_lv0, t_lv1);
    if (L > 0) {
      const int d = 1 << (L - 1);
      TE nv[PXT][NPL];
#pragma unroll
      for (int k = 0; k < PXT; ++k) {
        const int px = tid + k * NT;
        if (px < npx_t) {
          const bool two = xk[k] + d < W;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            const TE a = tab[pl * npl_px + px];
            const TE b = two ? tab[pl * npl_px + px + d] : a;
#pragma unroll
            for (int e = 0; e < EW; ++e) nv[k][pl][e] = max(a[e], b[e]);
          }
        }
      }
      lds_barrier();
#pragma unroll
      for (int k = 0; k < PXT; ++k) {
        const int px = tid + k * NT;
        if (px < npx_t) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) tab[pl * npl_px + px] = nv[k][pl];
        }
      }
      lds_barrier();
    }
    const int span = 1 << L;
    SP_T(t_lv2); SP_ADD(2, t_lv1, t_lv2);
    const int n_lvl = t_hi - t_lo, total = n_lvl * PW;
    // Items (task, bin column) of the level, in units of 64 consecutive ones per wave and iteration.
    // The <= 10 records (7 bin columns) a wave's 64 items touch are consecutive in the list; 20 lanes fetch them (16 bytes each) THREE
    // iterations ahead and the wave parks them in its own 2 x 320 bytes of LDS one iteration ahead: vector loads return in issue order with the
    // output stores, so a record fetched one iteration ahead would make every iteration wait for the previous one's 16 stores to be
    // acknowledged (measured: 63x63 map 152 -> 344 us); three iterations ahead the wave keeps ~48 stores in flight as the
    // chunked kernel's fire-and-forget stores did.  No workgroup barrier inside a level.
    // The 64-item units are taken from the END of the level's list (the classes are filed by ascending window height: the long units go
    // first, the level ends on short ones) and CLAIMED, not dealt: a wave takes the next unit of its workgroup's share (units z, z + n_zsplit,
    // ...) from an LDS counter whenever it is free — four ahead, the records follow three ahead.  Dealt round-robin, the spread of the
    // units' lengths (a class spans a factor of two in window rows and more beyond 8 rows) added up per wave: the phase clocks showed a
    // wave parked at the next level's barrier for 19 % of a workgroup's cycles on the 99x165 map, 4-5 units' worth per level (claimed: 7 %;
    // the call itself 1-3 % shorter: the parked wave's SIMD was serving its other three waves meanwhile).
    const int n_units_all = (total + 63) >> 6;
    const int n_units = n_units_all > zfirst ? (n_units_all - zfirst + n_zsplit - 1) / n_zsplit : 0;      // this workgroup's share
    if (n_units > 0) {
    const u32x4* lvl = my + 2 * (long)t_lo;
    const int lane = tid & 63;
    const int n_rec = (62 + PW) / PW + 1, slot = n_rec * 32;        // records a wave's 64 consecutive items can touch (7 columns: 10)
    char* wbuf = smem + ((((size_t)NPL * npl_px * sizeof(TE)) + 15) & ~(size_t)15) + (size_t)(tid >> 6) * (2 * slot);
    auto claim = [&]() {                                            // j-th claim of the workgroup -> first item of that unit, or -1
      int j = 0;
      if (lane == 0) j = atomicAdd(&s_next[L], 1);
      j = __builtin_amdgcn_readfirstlane(j);
      return j < n_units ? (zfirst + (n_units - 1 - j) * n_zsplit) * 64 : -1;
    };
    auto fetch = [&](int i_first) {                                 // lanes 0-19: their piece of the records of the unit at i_first
      const int te = min(max(i_first, 0) / PW + (lane >> 1), n_lvl - 1);
      return lvl[2 * te + (lane & 1)];
    };
    int un0 = claim(), un1 = claim(), un2 = claim(), un3 = claim();      // first items of the claimed units
    u32x4 r1 = u32x4{0u, 0u, 0u, 0u}, r2 = r1;
    if (lane < 2 * n_rec) { const u32x4 r0 = fetch(un0); r1 = fetch(un1); r2 = fetch(un2); *(u32x4*)(wbuf + lane * 16) = r0; }
    SP_T(t_lv3); SP_ADD(3, t_lv2, t_lv3);
    for (int k = 0; un0 >= 0; ++k) {
      u32x4 r3 = r2;
      if (lane < 2 * n_rec) r3 = fetch(un3);
      const int un4 = claim();
      const int i_first = un0;
      const int i = min(i_first + lane, total - 1);                 // lanes past the end redo the last item (same stores, same values)
      const int te = i / PW, pw = i - te * PW;
      const char* rec = wbuf + (k & 1) * slot + (te - i_first / PW) * 32;
      const u32x4 q0 = *(const u32x4*)rec;
      const unsigned int wb = *(const unsigned short*)(rec + 16 + 2 * pw);
      const int r = (int)(q0[0] & 0xFFFFu), ph = (int)(q0[0] >> 16);
      const int hs = (int)(q0[1] & 0xFFu), he = (int)(q0[1] >> 8);
      const int ws = (int)(wb & 0xFFu), we = (int)(wb >> 8);
      const int b = ph * PW + pw;
      const bool empty = (he <= hs) || (we <= ws);
      unsigned int best[CB];
#pragma unroll
      for (int q = 0; q < CB; ++q) best[q] = KEY_INIT;
#ifdef SW_TK_NOSCAN                     // development ablation (tools/build_variant.sh): no window scan
      if (!empty && hs == 250) {
#else
      if (!empty) {
#endif
        const int bw = we - ws;
        const int he_lds = min(he, y1);
        int cold_from = max(hs, y1);                                // first window row read from global memory
        if (bw >= span) {
          const int offB = bw - span;
          const int last = (he_lds - 1 - y0) * W + ws;
          for (int hh = hs; hh < he_lds; hh += 2) {
            const int i0 = (hh - y0) * W + ws, i1 = min(i0 + W, last);
            TE a0[NPL], b0[NPL], a1[NPL], b1[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
              a0[pl] = tab[pl * npl_px + i0]; b0[pl] = tab[pl * npl_px + i0 + offB];
              a1[pl] = tab[pl * npl_px + i1]; b1[pl] = tab[pl * npl_px + i1 + offB];
            }
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
              for (int e = 0; e < EW; ++e)
                best[EW * pl + e] = max(max(max(max(best[EW * pl + e], a0[pl][e]), b0[pl][e]), a1[pl][e]), b1[pl][e]);
            if (2 * span < bw) {                                    // only when the level was capped (SP_NLEV): spans in between
              for (int rr = 0; rr < 2; ++rr) {
                const int ib = rr ? i1 : i0;
                for (int x = span; x < offB; x += span) {
#pragma unroll
                  for (int pl = 0; pl < NPL; ++pl) {
                    const TE a = tab[pl * npl_px + ib + x];
#pragma unroll
                    for (int e = 0; e < EW; ++e) best[EW * pl + e] = max(best[EW * pl + e], a[e]);
                  }
                }
              }
            }
          }
        } else if (we == W) {                                       // clipped by the map's right edge: the span at ws ends at W
          for (int hh = hs; hh < he_lds; ++hh) {
            const int i0 = (hh - y0) * W + ws;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
              const TE a = tab[pl * npl_px + i0];
#pragma unroll
              for (int e = 0; e < EW; ++e) best[EW * pl + e] = max(best[EW * pl + e], a[e]);
            }
          }
        } else {
          cold_from = hs;                                           // narrower than the ROI's span for another reason: pixel loop
        }
        for (int hh = cold_from; hh < he; ++hh)
          for (int x = ws; x < we; ++x) {
            const int gi = hh * W + x;
            const unsigned int inv0 = 0xFFFEu - (unsigned)gi;
            unsigned int u[CB / 2];
            if constexpr (CB == 8) { const u32x4 w = *(const u32x4*)(fimg + (long)gi * C); u[0] = w[0]; u[1] = w[1]; u[2] = w[2]; u[3] = w[3]; }
            else if constexpr (CB == 4) { const u32x2 w = *(const u32x2*)(fimg + (long)gi * C); u[0] = w[0]; u[1] = w[1]; }
            else u[0] = *(const unsigned int*)(fimg + (long)gi * C);
#pragma unroll
            for (int k = 0; k < CB / 2; ++k) {
              best[2 * k] = max(best[2 * k], (key16_of(u[k] & 0xFFFFu) << 16) | inv0);
              best[2 * k + 1] = max(best[2 * k + 1], (key16_of(u[k] >> 16) << 16) | inv0);
            }
          }
      }
      float mv[CB]; int mi[CB];
      unsigned int lowest = best[0];
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const unsigned int m = (unsigned int)((int)best[q] >> 31);
        mv[q] = __uint_as_float((best[q] & 0xFFFF0000u) ^ (0x80000000u | (~m & 0x7FFF0000u)));
        mi[q] = (int)(0xFFFEu - (best[q] & 0xFFFFu));
        lowest = min(lowest, best[q]);
      }
      if (lowest == KEY_INIT) {
#pragma unroll
        for (int q = 0; q < CB; ++q)
          if (best[q] == KEY_INIT) mv[q] = empty ? 0.f : -FLT_MAX;
      }
      const float mul = __uint_as_float(q0[2]);
      const long o = (long)r * ld + (long)c0 * nb + b;
#ifdef SW_TK_NOSTORE                    // development ablation: the 2 x CB stores only for a value that does not occur
      if (lowest == 0x12345u)
#endif
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        Elem<unsigned short>::store(out + o + (long)q * nb, __fmul_rn(mv[q], mul));
        argmax[o + (long)q * nb] = ArgIdx<IT>::enc(mi[q]);
      }
      if (lane < 2 * n_rec) *(u32x4*)(wbuf + ((k + 1) & 1) * slot + lane * 16) = r1;     // iteration k + 1's records (fetched at k - 2)
      r1 = r2; r2 = r3;
      un0 = un1; un1 = un2; un2 = un3; un3 = un4;
    }
    SP_T(t_lv4); SP_ADD(4, t_lv3, t_lv4);
    }
  }     // levels
  SP_T(t_end); SP_ADD(6, t_begin, t_end);
}

// max |x| over n elements -> out[0] (f32; caller zero-fills).  |x| as IEEE bits is monotone => integer atomicMax; a NaN in x
// yields NaN, an Inf yields Inf.
template <typename T>
__global__ void absmax_kernel(long n, const T* __restrict__ x, float* __restrict__ out) {
  unsigned int m = 0u;                       // IEEE bits of |x|: NaN > Inf > finite, so a poisoned tensor reports it
  const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
  long done = 0;
  if ((((uintptr_t)x) & 15) == 0) {          // 16-byte loads over the aligned bulk, the tail element by element
    constexpr int V = 16 / (int)sizeof(T);
    const long nv = n / V;
    for (long i = tid; i < nv; i += nthr) {
      const u32x4 w = *(const u32x4*)(x + i * V);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (sizeof(T) == 2) m = max(m, max((w[t] << 16) & 0x7FFFFFFFu, w[t] & 0x7FFF0000u));      // bf16 pair as two f32 bit patterns
        else m = max(m, w[t] & 0x7FFFFFFFu);
      }
    }
    done = nv * V;
  }
  for (long i = done + tid; i < n; i += nthr) m = max(m, absbits(Elem<T>::load(x + i)));
  m = wave_reduce_max_u32(m);
  // one atomic per WORKGROUP (round 6: per wave, the 8192 atomics of a 2048-block launch on one address serialised to ~90 us for a 25 MB
  // gradient — the Stage-3 ROIAlign backward calls this once per iteration)
  __shared__ unsigned int s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax((unsigned int*)out, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));
}

}  // namespace

namespace {
// workgroups per (slab, image, band) = how many ways the ROI chunks are dealt out: enough workgroups to fill the chip a few times
// over (the chunks differ in work), no more (each one fetches its slab of the map)
inline int fwd_zsplit(int wg_per_z, int n_chunks, size_t lds) {
  // measured (tools/roi_bench_voc.py, SW_ROI_FWD_WGS sweep): one-workgroup-per-CU slabs 512-1024 workgroups (150x200 map: 1.36 ms
  // at 1024, 1.52 at 4096), two-per-CU slabs flat from 512 to 4096
  static const int forced = getenv("SW_ROI_FWD_WGS") ? atoi(getenv("SW_ROI_FWD_WGS")) : 0;      // development switch
  const int target = forced ? forced : (lds > 80 * 1024 ? 768 : 2048);
  int nz = (target + wg_per_z - 1) / wg_per_z;
  nz = nz < 1 ? 1 : nz;
  return nz > n_chunks ? n_chunks : nz;
}

template <typename T, int CB, typename IT>
int launch_fwd_plane(int nimg, int H, int W, int C, long ld, int PH, int PW, float scale, const void* feat, const float* rois, int R,
                     const float* row_scale, float row_scale_add, void* out, void* argmax, hipStream_t stream) {
  constexpr int NT = 1024, CHUNK = 256;
  const size_t lds = (((size_t)H * W * CB * sizeof(T) + 15) & ~(size_t)15) + (size_t)CHUNK * (4 + 2 * (PH + PW)) + 4 + (size_t)CHUNK * 4;
  const bool key = sizeof(T) == 2 && (long)H * W < 65535;
  auto kern = key ? roi_pool_fwd_plane_kernel<T, CB, NT, IT, (sizeof(T) == 2)> : roi_pool_fwd_plane_kernel<T, CB, NT, IT, false>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int nz = fwd_zsplit((C / CB) * nimg, (R + CHUNK - 1) / CHUNK, lds);
  dim3 grid(C / CB, nimg, nz), block(NT);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, H, W, C, ld, PH, PW, scale, (const T*)feat, rois, R, CHUNK, row_scale,
                     row_scale_add, (T*)out, (IT*)argmax, H, H, 1, nz);
  SW_CHECK_LAUNCH();
  return 0;
}

// band form (bf16, 8 channels per lane): rows per band from the LDS left beside the ROI tables and the owned-pair list
constexpr int BAND_CHUNK = 256;
inline size_t band_tables_bytes(int PH, int PW) {
  return (size_t)BAND_CHUNK * (4 + 2 * (PH + PW)) + 4 + (size_t)BAND_CHUNK * 4 + (size_t)BAND_CHUNK * PH * 4;
}
inline bool band_geometry(int H, int W, int PH, int PW, int* S, int* rows) {
  const size_t budget = 160 * 1024 - 1024 - band_tables_bytes(PH, PW);     // 1 KiB: the kernel's static LDS + alignment
  const int fit = (int)(budget / ((size_t)W * 16));
  const int halo = (H + PH - 1) / PH + 1;          // a bin window of a ROI inside the image: <= ceil(H / PH) + 1 rows
  if (fit - halo < 8 || PH > 255) return false;
  *S = fit - halo; *rows = fit;
  return true;
}
template <typename IT>
int launch_fwd_band(int nimg, int H, int W, int C, long ld, int PH, int PW, float scale, const void* feat, const float* rois, int R,
                    const float* row_scale, float row_scale_add, void* out, void* argmax, int S, int rows, hipStream_t stream) {
  constexpr int NT = 1024, CB = 8;
  typedef unsigned short T;
  const size_t lds = (((size_t)rows * W * 16 + 15) & ~(size_t)15) + band_tables_bytes(PH, PW);
  auto kern = roi_pool_fwd_plane_kernel<T, CB, NT, IT, true, true>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int n_bands = (H + S - 1) / S, n_chunks = (R + BAND_CHUNK - 1) / BAND_CHUNK;
  const int nz = fwd_zsplit((C / CB) * nimg * n_bands, n_chunks, lds);
  if ((long)n_bands * nz > 65535) return -6;
  dim3 grid(C / CB, nimg, n_bands * nz), block(NT);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, H, W, C, ld, PH, PW, scale, (const T*)feat, rois, R, BAND_CHUNK, row_scale,
                     row_scale_add, (T*)out, (IT*)argmax, S, rows, n_bands, nz);
  SW_CHECK_LAUNCH();
  return 0;
}

// sparse-table form: LDS = table planes + the image's sorted ROI list + one chunk's tables
inline size_t sparse_tables_bytes(int R, int PH, int PW, bool band, int chunk) {
  return (size_t)((R + 1) & ~1) * 2 + (size_t)chunk * (4 + 4 + 2 * (PH + PW)) + 4 + (band ? (size_t)chunk * PH * 4 : 0) + 16;
}
template <typename IT, int CB, bool BAND>
int launch_sparse_kernel(dim3 grid, size_t lds, int H, int W, int C, long ld, int PH, int PW, float scale, const void* feat, const float* rois,
                         int R, const float* row_scale, float row_scale_add, void* out, void* argmax, int S, int rows, int n_bands, int nz,
                         int chunk, hipStream_t stream) {
  auto kern = (PH == 7 && PW == 7) ? roi_pool_fwd_sparse_kernel<IT, CB, BAND, 7> : roi_pool_fwd_sparse_kernel<IT, CB, BAND, 0>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, dim3(SP_NT), lds, stream, H, W, C, ld, PH, PW, scale, (const unsigned short*)feat, rois, R, row_scale,
                     row_scale_add, (unsigned short*)out, (IT*)argmax, S, rows, n_bands, nz, chunk);
  SW_CHECK_LAUNCH();
  return 0;
}

template <typename IT>
int launch_fwd_sparse(int nimg, int H, int W, int C, long ld, int PH, int PW, float scale, const void* feat, const float* rois, int R,
                      const float* row_scale, float row_scale_add, void* out, void* argmax, hipStream_t stream) {
  constexpr size_t LDS_MAX = 160 * 1024 - 1536;                     // the kernel's static LDS (class tables) + alignment
  const int n_blocks = (R + 127) / 128;                             // row blocks of 128 ROIs are dealt to the workgroups of a (slab, image)
  // workgroups: ONE per CU (every form here holds most of a CU's LDS, and a workgroup's fixed work — its share of the ROI sort, the
  // slab fetch, the level advances — is paid per workgroup: 63x63 / 4000 ROIs 156 us at 256 workgroups, 188 at 768, 226 at 1536);
  // SW_ROI_FWD_WGS overrides the target
  static const int forced = getenv("SW_ROI_FWD_WGS") ? atoi(getenv("SW_ROI_FWD_WGS")) : 0;      // development switch
  auto zsplit = [&](int wg_per_z) {
    const int target = forced ? forced : 256;
    int nz = (target + wg_per_z / 2) / wg_per_z;
    nz = nz < 1 ? 1 : nz;
    return nz > n_blocks ? n_blocks : nz;
  };
  static const bool force_cb4 = getenv("SW_ROI_SPARSE_CB4") != nullptr;                // development switch: 4 channels per lane everywhere
  // ROIs per chunk: a run-time kernel argument since round 6 (SW_ROI_SPARSE_CHUNK, development switch).  The phase clocks of the band form
  // (99x165 map: 29 % of a workgroup's cycles in chunk set-up, 21 % at the barrier behind a scan; profiles/r06_roi_phases.txt) suggested
  // larger chunks; measured, they are slower on every map (99x165 / 8000 ROIs: 672 us at 128, 684 at 256, 792 at 512, 1041 at 1024 —
  // their tables take LDS from the band's rows; profiles/r06_roi_chunk_sweep.txt): 128 stays
  static const int chunk_env = getenv("SW_ROI_SPARSE_CHUNK") ? atoi(getenv("SW_ROI_SPARSE_CHUNK")) : 0;
  const int chunk_plane = chunk_env > 0 ? chunk_env : SP_CH, chunk_band = chunk_env > 0 ? chunk_env : SP_CH;
  const size_t plane8 = (size_t)H * W * 32 + sparse_tables_bytes(R, PH, PW, false, chunk_plane);
  if (!force_cb4 && plane8 <= LDS_MAX && (long)H * W <= 5 * SP_NT) {
    const int nz = zsplit((C / 8) * nimg);
    return launch_sparse_kernel<IT, 8, false>(dim3(C / 8, nimg, nz), plane8, H, W, C, ld, PH, PW, scale, feat, rois, R, row_scale,
                                              row_scale_add, out, argmax, H, H, 1, nz, chunk_plane, stream);
  }
  // row bands: 4 channels per lane (16 B per pixel).  8 channels (32 B per pixel: a third of the rows per band) measured slower on
  // every banded map (99x165 / 8000 ROIs: 956 vs 669 us with 9 vs 3 bands; 125x167: 525 vs 383; 76x114: 245 vs 219) — the bands'
  // redundant ROI tables and slab fetches outweigh the halved task count; SW_ROI_SPARSE_CB8BAND=1 tries it first (A/B timing)
  static const bool try_cb8_band = getenv("SW_ROI_SPARSE_CB8BAND") != nullptr;          // development switch
  const size_t tb = sparse_tables_bytes(R, PH, PW, true, chunk_band);
  if (PH > 255) return -100;
  const int halo = (H + PH - 1) / PH + 1;
  for (int cb = (try_cb8_band && !force_cb4) ? 8 : 4; cb >= 4; cb >>= 1) {
    const int pxb = cb * 4, pxt = cb == 8 ? 5 : 10;
    if (tb + (size_t)W * pxb * 8 > LDS_MAX) continue;
    int fit = (int)((LDS_MAX - tb) / ((size_t)W * pxb));
    if ((long)fit * W > (long)pxt * SP_NT) fit = pxt * SP_NT / W;
    int S, rows;
    if (fit >= H) { S = H; rows = H; }                              // the whole map fits: one band
    else { if (fit - halo < 8) continue; S = fit - halo; rows = fit; }
    const int n_bands = (H + S - 1) / S;
    const size_t lds = (size_t)rows * W * pxb + tb;
    const int nz = zsplit((C / cb) * nimg * n_bands);
    if ((long)n_bands * nz > 65535) continue;
    dim3 grid(C / cb, nimg, n_bands * nz);
    if (cb == 8)
      return launch_sparse_kernel<IT, 8, true>(grid, lds, H, W, C, ld, PH, PW, scale, feat, rois, R, row_scale, row_scale_add, out, argmax,
                                               S, rows, n_bands, nz, chunk_band, stream);
    return launch_sparse_kernel<IT, 4, true>(grid, lds, H, W, C, ld, PH, PW, scale, feat, rois, R, row_scale, row_scale_add, out, argmax,
                                             S, rows, n_bands, nz, chunk_band, stream);
  }
  return -100;
}

// prepared-task form (sw_roi_pool_fwd_ws): workspace = [nimg][TK_MAXB][TK_SEG] ints, then [nimg][R * PH] task records of 32 bytes
inline size_t tasks_seg_bytes(int nimg) { return (((size_t)nimg * TK_MAXB * TK_SEG * 4) + 255) & ~(size_t)255; }
inline size_t tasks_geo_bytes(int R) { return (((size_t)R * sizeof(RoiGeo)) + 255) & ~(size_t)255; }
inline size_t tasks_hist_bytes(int nimg, int R) {
  return ((((size_t)(R + TK_PREP_NT - 1) / TK_PREP_NT) * (size_t)nimg * TK_MAXB * SP_NCLS * 4) + 255) & ~(size_t)255;
}
inline bool tasks_shape_ok(int dtype, int nimg, int H, int W, int C, int PH, int PW, int R, const void* feat) {
  return dtype == SW_BF16 && nimg > 0 && (C % 8) == 0 && (long)H * W < 65535 && H <= 255 && W <= 255 && R <= 65535 && PH <= 8 && PW <= 8 &&
         (((uintptr_t)feat) & 15) == 0;
}

template <typename IT, int CB, int NT>
int launch_tasks_kernel(dim3 grid, size_t lds, int H, int W, int C, long ld, int PH, int PW, const void* feat, const u32x4* tasks,
                        const int* seg, long cap, void* out, void* argmax, int S, int rows, int n_bands, int nz, hipStream_t stream) {
  auto kern = (PH == 7 && PW == 7) ? roi_pool_fwd_tasks_kernel<IT, CB, NT, 7> : roi_pool_fwd_tasks_kernel<IT, CB, NT, 0>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, H, W, C, ld, PH, PW, (const unsigned short*)feat, tasks, seg, cap,
                     (unsigned short*)out, (IT*)argmax, S, rows, n_bands, nz);
  SW_CHECK_LAUNCH();
  return 0;
}

template <typename IT>
int launch_fwd_tasks(int nimg, int H, int W, int C, long ld, int PH, int PW, float scale, const void* feat, const float* rois, int R,
                     const float* row_scale, float row_scale_add, void* out, void* argmax, void* workspace, hipStream_t stream) {
  // development switches: SW_ROI_TASKS_CB (4 / 8 channels per lane on banded maps), SW_ROI_TASKS_NT (1024 / 512 threads: one / two
  // workgroups per CU), SW_ROI_TASKS_HALO (rows of overlap between bands; anything below ceil(H / PH) + 1 sends the tallest windows
  // to the global-memory loop), SW_ROI_FWD_WGS (workgroup target)
  static const int env_cb = getenv("SW_ROI_TASKS_CB") ? atoi(getenv("SW_ROI_TASKS_CB")) : 0;
  static const int env_nt = getenv("SW_ROI_TASKS_NT") ? atoi(getenv("SW_ROI_TASKS_NT")) : 0;
  static const int env_halo = getenv("SW_ROI_TASKS_HALO") ? atoi(getenv("SW_ROI_TASKS_HALO")) : 0;
  static const int env_wgs = getenv("SW_ROI_FWD_WGS") ? atoi(getenv("SW_ROI_FWD_WGS")) : 0;
  constexpr size_t LDS_ALL = 160 * 1024 - 1024;                     // the kernel's static LDS + alignment
  int* seg = (int*)workspace;
  RoiGeo* geo = (RoiGeo*)((char*)workspace + tasks_seg_bytes(nimg));
  int* hist = (int*)((char*)geo + tasks_geo_bytes(R));
  u32x4* tasks = (u32x4*)((char*)hist + tasks_hist_bytes(nimg, R));
  const long cap = (long)R * PH;
  // whole map at 32 B per pixel (8 channels per lane) where it fits, else row bands
  int cb, nt, S, rows, n_bands;
  if (PW < 3) return -100;                                          // a wave's 64 items then span more records than it has lanes to fetch
  auto ring_bytes = [&](int nthreads) { return (size_t)(nthreads / 64) * 2 * (size_t)(((62 + PW) / PW + 1) * 32); };   // per-wave record ring
  const bool plane8 = (size_t)H * W * 32 + ring_bytes(1024) <= LDS_ALL && (long)H * W <= 5 * 1024;
  // maps that fit LDS whole at 8 channels per lane: the sparse kernel's per-chunk work is small there (63x63 / 4000 ROIs: 151 us against
  // 155 + 18 us of list building), so they stay with it; SW_ROI_TASKS_PLANE=1 sends them here (A/B timing)
  static const bool env_plane = getenv("SW_ROI_TASKS_PLANE") != nullptr;
  if (plane8 && (env_cb == 0 || env_cb == 8)) {
    if (!env_plane) return -100;
    cb = 8; nt = 1024; S = H; rows = H; n_bands = 1;
  }
  else {
    cb = env_cb == 8 ? 8 : env_cb == 2 ? 2 : 4;
    nt = env_nt == 512 ? 512 : 1024;
    const size_t budget = (nt == 512 ? LDS_ALL / 2 : LDS_ALL) - ring_bytes(nt);
    const int pxb = cb * 4, pxt = (cb == 8 ? 5 : cb == 4 ? 10 : 20) * (1024 / nt);
    int fit = (int)(budget / ((size_t)W * pxb));
    if ((long)fit * W > (long)pxt * nt) fit = pxt * nt / W;
    if (fit >= H) { S = H; rows = H; n_bands = 1; }
    else {
      const int halo = env_halo > 0 ? env_halo : (H + PH - 1) / PH + 1;
      if (fit - halo < 4) return -100;
      n_bands = (H + (fit - halo) - 1) / (fit - halo);
      static const int env_bands = getenv("SW_ROI_TASKS_BANDS") ? atoi(getenv("SW_ROI_TASKS_BANDS")) : 0;   // development switch: more, overlapping bands
      if (env_bands > n_bands) n_bands = env_bands;
      S = (H + n_bands - 1) / n_bands;                              // equal bands
      n_bands = (H + S - 1) / S;
      static const bool tight = getenv("SW_ROI_TASKS_TIGHT") != nullptr;           // development switch: rows = S + halo as the sparse kernel
      rows = tight ? S + halo : fit;                                // all the rows LDS holds: more ROIs lie inside one band
      if (rows > H) rows = H;
      if (n_bands > TK_MAXB) return -100;
    }
  }
  if (nimg * n_bands > 64) return -100;
  const int prep_wgs = (R + TK_PREP_NT - 1) / TK_PREP_NT;
  hipLaunchKernelGGL(roi_pool_geo_kernel, dim3(prep_wgs), dim3(TK_PREP_NT), 0, stream, nimg, H, W, PH, PW, scale, rois, R, row_scale,
                     row_scale_add, S, rows, n_bands, geo, hist);
  hipLaunchKernelGGL(roi_pool_tasks_kernel, dim3(prep_wgs), dim3(TK_PREP_NT), 0, stream, nimg, PH, R, n_bands, (const RoiGeo*)geo,
                     (const int*)hist, tasks, seg, cap);
  SW_CHECK_LAUNCH();
  const size_t lds = ((((size_t)rows * W * cb * 4) + 15) & ~(size_t)15) + ring_bytes(nt);
  const int slabs = (C / cb) * nimg * n_bands;
  const int target = env_wgs ? env_wgs : (nt == 512 ? 512 : 256);
  int nz = (target + slabs / 2) / slabs;
  nz = nz < 1 ? 1 : nz;
  if ((long)n_bands * nz > 65535) return -6;
  dim3 grid(C / cb, nimg, n_bands * nz);
#define SW_TK(CBV, NTV) return launch_tasks_kernel<IT, CBV, NTV>(grid, lds, H, W, C, ld, PH, PW, feat, tasks, seg, cap, out, argmax, S, rows, \
                                                                 n_bands, nz, stream)
  if (cb == 8) { if (nt == 512) SW_TK(8, 512); SW_TK(8, 1024); }
  if (cb == 2) { if (nt == 512) SW_TK(2, 512); SW_TK(2, 1024); }
  if (nt == 512) SW_TK(4, 512);
  SW_TK(4, 1024);
#undef SW_TK
}

template <typename IT>
int roi_fwd_dispatch(int dtype, int nimg, int H, int W, int C, long ld, int PH, int PW, float spatial_scale, const void* feat,
                     const float* rois, int R, const float* row_scale, float row_scale_add, void* out, void* argmax,
                     hipStream_t stream) {
  // feature-stationary form with the widest channel slab (16 / 8 / 4 bytes per pixel) whose H*W plane fits LDS next to the 9 KiB
  // of ROI tables — ONE workgroup per CU if need be: the per-task work outside the window scan is paid per slab, so 8 channels
  // per lane at 16 waves per CU beat 4 channels at 32 (76x114 map, 4000 ROIs: 393 vs 439 us; 86x115: 560 vs 657 with 8- vs
  // 4-byte slabs), and on maps too large for any two-workgroup slab (125x167, 150x200) the 4-byte slab still beats the
  // ROI-stationary gather kernel below (1368 vs 1479, 1762 vs 2072 us) since the per-task overhead was cut (tools/roi_bench_voc.py)
  static const bool force_gather = getenv("SW_ROI_FWD_GATHER") != nullptr;    // development switch
  const size_t es = dtype == SW_BF16 ? 2 : 4;
  if (!force_gather && nimg > 0 && (((uintptr_t)feat) & 15) == 0 && H <= 255 && W <= 255) {      // 8-bit bin tables
    int pxb = 0;
    for (int cand = 16; cand >= 4 && !pxb; cand >>= 1)
      if ((C % (cand / (int)es)) == 0 && (size_t)H * W * cand <= 150 * 1024) pxb = cand;
    static const char* force_pxb = getenv("SW_ROI_FWD_PXB");                      // development switch: 16 / 8 / 4, 0 = gather form
    if (force_pxb) {
      const int want = atoi(force_pxb);
      pxb = (want && (C % (want / (int)es)) == 0 && (size_t)H * W * want <= 150 * 1024) ? want : 0;
    }
    // row sparse table form (bf16, packed candidates): the whole map at 32 B per pixel (8 channels per lane), else row bands at 16 B
    // per pixel (4 channels per lane); SW_ROI_FWD_SPARSE=0 keeps the scan forms below (A/B timing)
    static const char* sparse_sw = getenv("SW_ROI_FWD_SPARSE");                       // development switch
    if (!(sparse_sw && sparse_sw[0] == '0') && !force_pxb && dtype == SW_BF16 && (C % 8) == 0 && (long)H * W < 65535 && R <= 16384 &&
        PH <= 8 && PW <= 8) {
      const int rc = launch_fwd_sparse<IT>(nimg, H, W, C, ld, PH, PW, spatial_scale, feat, rois, R, row_scale, row_scale_add, out, argmax, stream);
      if (rc != -100) return rc;                                                   // -100: shape not covered, use the forms below
    }
    static const bool no_band = getenv("SW_ROI_FWD_NO_BAND") != nullptr;            // development switch
    int bS = 0, brows = 0;
    if (!no_band && !force_pxb && dtype == SW_BF16 && pxb < 16 && (C % 8) == 0 && (long)H * W < 65535 &&
        band_geometry(H, W, PH, PW, &bS, &brows))
      return launch_fwd_band<IT>(nimg, H, W, C, ld, PH, PW, spatial_scale, feat, rois, R, row_scale, row_scale_add, out, argmax,
                                 bS, brows, stream);
    if (pxb) {
#define SW_FWD_PLANE(T, CB) return launch_fwd_plane<T, CB, IT>(nimg, H, W, C, ld, PH, PW, spatial_scale, feat, rois, R, row_scale, \
                                                               row_scale_add, out, argmax, stream)
      if (dtype == SW_BF16) {
        if (pxb == 16) SW_FWD_PLANE(unsigned short, 8);
        if (pxb == 8) SW_FWD_PLANE(unsigned short, 4);
        SW_FWD_PLANE(unsigned short, 2);
      } else {
        if (pxb == 16) SW_FWD_PLANE(float, 4);
        if (pxb == 8) SW_FWD_PLANE(float, 2);
        SW_FWD_PLANE(float, 1);
      }
#undef SW_FWD_PLANE
    }
  }
  const int vec = dtype == SW_BF16 ? 2 : 1;
  if (C % vec) return -5;
  const int ch = 64 * vec;
  const size_t lds = (size_t)ch * PH * PW * (4 + es);
  if (lds > 64 * 1024) return -6;
  dim3 grid(R, (C + ch - 1) / ch), block(256);
  if (dtype == SW_BF16)
    hipLaunchKernelGGL((roi_pool_fwd_kernel<unsigned short, 2, IT>), grid, block, lds, stream, H, W, C, ld, PH, PW, spatial_scale,
                       (const unsigned short*)feat, rois, row_scale, row_scale_add, (unsigned short*)out, (IT*)argmax);
  else
    hipLaunchKernelGGL((roi_pool_fwd_kernel<float, 1, IT>), grid, block, lds, stream, H, W, C, ld, PH, PW, spatial_scale,
                       (const float*)feat, rois, row_scale, row_scale_add, (float*)out, (IT*)argmax);
  SW_CHECK_LAUNCH();
  return 0;
}

template <typename T, typename IT>
int roi_bwd_dispatch(int nimg, int H, int W, int C, long ld, int PH, int PW, const void* dout, const void* argmax, const float* rois,
                     int R, const float* row_scale, float row_scale_add, const void* relu_ref, const float* dout_absmax,
                     void* dfeat, float spatial_scale, hipStream_t stream) {
  // fixed-point path: CB in {8, 4} with H*W*CB*8 bytes of LDS; needs max|dout| (device scalar)
  static const bool float_atomics = getenv("SW_ROI_FLOAT_ATOMICS") != nullptr;    // development switch
  // bf16: a hi/lo pair of 32-bit accumulators (see the kernel) when PH*PW*R terms (every bin of every ROI on one pixel: the worst
  // case) leave >= 8 bits per word; fp32: one 64-bit word.  SW_ROI_BWD_ACC32: round 2's single 32-bit word (A/B timing only)
  static const bool acc32_single = getenv("SW_ROI_BWD_ACC32") != nullptr;          // development switch
  const bool small_terms = sizeof(T) == 2 && R > 0 && (long)PH * PW * R < (1L << 30) &&
                           (30 - (32 - __builtin_clz((unsigned)(PH * PW * R)))) >= 8;
  static const bool acc64 = getenv("SW_ROI_BWD_ACC64") != nullptr;                 // development switch: the fp32 form for bf16 too
  const int accmode = (!small_terms || acc64) ? 0 : (acc32_single ? 1 : 2);
  const size_t ab = accmode == 1 ? 4 : 8;
  int cbx = 8;
  constexpr size_t ACC_BUDGET = 150 * 1024;        // of the CU's 160 KiB: + 8 KiB ROI list + the static reduction scratch
  while (cbx > 4 && ((size_t)H * W * cbx * ab > ACC_BUDGET || (C % cbx) || (C / cbx) * nimg < 256)) cbx >>= 1;
  if (C % cbx) cbx = 0;
  // (2-channel slabs on maps that need >= 3 pixel ranges at 4 channels — half the ranges, fewer ROIs straddling a range boundary and
  // streamed twice — measured slower, round 6: 99x165 / 8000 ROIs 459 -> 486 us, 150x200 398 -> 458: twice the workgroups' fixed work)
  const int nsplit = cbx ? (int)(((size_t)H * W * cbx * ab + ACC_BUDGET - 1) / ACC_BUDGET) : 1;   // pixel ranges per plane
  if (cbx >= 4 && nsplit <= 16 && dout_absmax != nullptr && (((uintptr_t)dout & 7) == 0) && (((uintptr_t)argmax & 15) == 0) && (ld % 4) == 0 &&
      !float_atomics) {
    const int px_per = (H * W + nsplit - 1) / nsplit;
    const size_t ldsx = (size_t)px_per * cbx * ab + FX_CHUNK * 8;
    dim3 gridx(C / cbx, nimg, nsplit), blockx(1024);
    hipError_t ex;
#define SW_BWD_FX(MODE)                                                                                                       \
    {                                                                                                                           \
      auto k = roi_pool_bwd_fx_kernel<T, IT, MODE>;                                                                             \
      ex = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx);                          \
      if (ex != hipSuccess) return (int)ex;                                                                                     \
      hipLaunchKernelGGL(k, gridx, blockx, ldsx, stream, H, W, C, ld, PH * PW, cbx, (const T*)dout, (const IT*)argmax, rois, R, \
                         row_scale, row_scale_add, dout_absmax, (const T*)relu_ref, (T*)dfeat, spatial_scale);                  \
    }
    if (accmode == 2) SW_BWD_FX(2) else if (accmode == 1) SW_BWD_FX(1) else SW_BWD_FX(0)
#undef SW_BWD_FX
    SW_CHECK_LAUNCH();
    return 0;
  }
  // float-atomic path: channel slab per workgroup a power of two, H*W*CB*4 <= 64 KiB (two 1024-thread workgroups per
  // CU) and enough slabs to give every CU work (C/CB * nimg >= 512 where the map allows)
  int CB = 64;
  while (CB > 1 && ((size_t)H * W * CB * 4 > 64 * 1024 || (C % CB) || (C / CB) * nimg < 512)) CB >>= 1;
  const size_t lds = (size_t)H * W * CB * 4;
  if (lds > 160 * 1024) return -6;
  dim3 grid(C / CB, nimg), block(1024);
  auto k = roi_pool_bwd_kernel<T, IT>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k, grid, block, lds, stream, H, W, C, ld, PH * PW, CB, (const T*)dout, (const IT*)argmax, rois, R, row_scale,
                     row_scale_add, (const T*)relu_ref, (T*)dfeat);
  SW_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" long sw_roi_pool_fwd_workspace_bytes(int nimg, int R, int PH, int PW) {
  if (nimg <= 0 || R <= 0 || PH <= 0 || PW <= 0) return 0;
  return (long)(tasks_seg_bytes(nimg) + tasks_geo_bytes(R) + tasks_hist_bytes(nimg, R)) + (long)nimg * R * PH * 32;
}

extern "C" int sw_roi_pool_fwd(int dtype, int nimg, int H, int W, int C, int PH, int PW, float spatial_scale,
                               const void* feat, const float* rois, int R, const float* row_scale,
                               float row_scale_add, void* out, void* argmax, int argmax_bits, long ld_out,
                               hipStream_t stream) {
  return sw_roi_pool_fwd_ws(dtype, nimg, H, W, C, PH, PW, spatial_scale, feat, rois, R, row_scale, row_scale_add, out, argmax,
                            argmax_bits, ld_out, nullptr, 0, stream);
}

extern "C" int sw_roi_pool_fwd_ws(int dtype, int nimg, int H, int W, int C, int PH, int PW, float spatial_scale,
                                  const void* feat, const float* rois, int R, const float* row_scale,
                                  float row_scale_add, void* out, void* argmax, int argmax_bits, long ld_out,
                                  void* workspace, long workspace_bytes, hipStream_t stream) {
  SW_ENTER();
  if (R <= 0) return 0;
  const long ld = ld_out > 0 ? ld_out : (long)C * PH * PW;
  if (ld < (long)C * PH * PW) return -5;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  static const bool no_tasks = getenv("SW_ROI_FWD_TASKS") && getenv("SW_ROI_FWD_TASKS")[0] == '0';      // development switch (A/B timing)
  if (workspace && !no_tasks && (argmax_bits == 32 || argmax_bits == 16) && tasks_shape_ok(dtype, nimg, H, W, C, PH, PW, R, feat)) {
    if ((((uintptr_t)workspace) & 15) || workspace_bytes < sw_roi_pool_fwd_workspace_bytes(nimg, R, PH, PW)) return -5;
    const int rc = argmax_bits == 32
        ? launch_fwd_tasks<int>(nimg, H, W, C, ld, PH, PW, spatial_scale, feat, rois, R, row_scale, row_scale_add, out, argmax, workspace, stream)
        : launch_fwd_tasks<unsigned short>(nimg, H, W, C, ld, PH, PW, spatial_scale, feat, rois, R, row_scale, row_scale_add, out, argmax,
                                           workspace, stream);
    if (rc != -100) return rc;                                      // -100: shape not covered, the forms below take it
  }
  if (argmax_bits == 32)
    return roi_fwd_dispatch<int>(dtype, nimg, H, W, C, ld, PH, PW, spatial_scale, feat, rois, R, row_scale, row_scale_add, out,
                                 argmax, stream);
  if (argmax_bits != 16) return -1;
  if ((long)H * W >= 65535) return -6;
  return roi_fwd_dispatch<unsigned short>(dtype, nimg, H, W, C, ld, PH, PW, spatial_scale, feat, rois, R, row_scale,
                                          row_scale_add, out, argmax, stream);
}

extern "C" int sw_roi_pool_bwd(int dtype, int nimg, int H, int W, int C, int PH, int PW, const void* dout,
                               const void* argmax, int argmax_bits, long ld_in, const float* rois, int R,
                               const float* row_scale, float row_scale_add, const void* relu_ref,
                               const float* dout_absmax, void* dfeat, float spatial_scale, hipStream_t stream) {
  SW_ENTER();
  const long ld = ld_in > 0 ? ld_in : (long)C * PH * PW;
  if (ld < (long)C * PH * PW) return -5;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if (argmax_bits != 32 && argmax_bits != 16) return -1;
#define SW_BWD(T, IT) return roi_bwd_dispatch<T, IT>(nimg, H, W, C, ld, PH, PW, dout, argmax, rois, R, row_scale, row_scale_add, \
                                                     relu_ref, dout_absmax, dfeat, spatial_scale, stream)
  if (dtype == SW_BF16) { if (argmax_bits == 32) SW_BWD(unsigned short, int); SW_BWD(unsigned short, unsigned short); }
  if (argmax_bits == 32) SW_BWD(float, int);
  SW_BWD(float, unsigned short);
#undef SW_BWD
}

extern "C" int sw_absmax(int dtype, long n, const void* x, float* out, hipStream_t stream) {
  SW_ENTER();
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  if (n <= 0) return 0;
  long blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 1024) blocks = 1024;
  if (dtype == SW_BF16)
    hipLaunchKernelGGL(absmax_kernel<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, stream, n, (const unsigned short*)x, out);
  else if (dtype == SW_F32)
    hipLaunchKernelGGL(absmax_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, n, (const float*)x, out);
  else
    return -1;
  SW_CHECK_LAUNCH();
  return 0;
}

#ifdef SW_ROI_PHASES
extern "C" int sw_debug_roi_phases(unsigned long long* out8, int reset) {
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && out8) e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_roi_phase), 64);
  if (e == hipSuccess && reset) { unsigned long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_roi_phase), z, 64); }
  return e == hipSuccess ? 0 : -(int)e;
}
#endif
