// MFMA tile GEMM for gfx950: one kernel template serves the FC layers (fwd / dgrad / wgrad)
// and, through gathering loaders, the 3x3 convolutions as implicit GEMMs (fwd / dgrad / wgrad).
//
//   C[m][n] (+)= sum_k A(m,k) * B(k,n)
//
// Operand storage modes
//   OP_KCONTIG   : the operand's K index is the contiguous one  (A[m*lda+k]   / B[n*ldb+k])
//   OP_KSTRIDED  : K is the slow index                          (A[k*lda+m]   / B[k*ldb+n])
//   OP_CONV_A    : A(m,k) gathered from an NHWC tensor: m=(img,y,x), k=(tap,c)   (conv fwd / dgrad)
//   OP_CONV_B    : B(k,n) gathered from an NHWC tensor: k=(img,y,x), n=(tap,c)   (conv wgrad)
//
// Tile 128x128 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles),
// K-step of 128 bytes per row (64 bf16 / 32 f32), LDS double buffered, register-staged global
// loads issued before the MFMA block and written to the other LDS buffer after it (one barrier
// per K-step).  bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32 (exact f32).
// K-contiguous operands are read from LDS with ds_read_b128 through an XOR swizzle; K-strided
// operands keep their natural [k][m] image and are transposed on the fly by ds_read_b64_tr_b16
// (bf16) or read element-wise (f32).
#include <stdlib.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {

enum { OP_KCONTIG = 0, OP_KSTRIDED = 1, OP_CONV_A = 2, OP_CONV_B = 3 };

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  int tiles_m, tiles_n, patches_m, k_per_split;
  int cH, cW, cC, cDil;            // geometry of the gathered NHWC tensor (conv modes)
  const float* bias;               // per column n (or per row m when bias_on_m)
  const uint8_t* drop; long ldd; float drop_scale;
  const void* ref; long ldr; float ref_scale; int ref_bf16;
  int relu, out_bf16, atomic, oihw_cin;
};

template <typename T> struct GT;
template <> struct GT<unsigned short> { static constexpr int EPC = 8, BK = 64; };   // bf16
template <> struct GT<float> { static constexpr int EPC = 4, BK = 32; };

__device__ const u32x4 g_zero_chunk = {0u, 0u, 0u, 0u};   // source of zero fill for out-of-range 16-byte chunks

constexpr int TILE = 128;
constexpr int LDS_TILE_BYTES = 16384;

template <typename T, int MODE>
struct OperandGeom {   // how a 16 KiB LDS tile of this operand is cut into 16-byte chunks
  static constexpr bool KS = (MODE == OP_KSTRIDED || MODE == OP_CONV_B);
  static constexpr int CHUNKS_PER_ROW = KS ? (TILE * (int)sizeof(T) / 16) : 8;
  static constexpr int SHIFT = KS ? (sizeof(T) == 2 ? 4 : 5) : 3;
  __device__ static __forceinline__ int lds_off(int row, int chunk) {
    if (!KS) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    if (sizeof(T) == 2) return row * 256 + ((chunk ^ ((row & 3) << 2)) << 4);
    return row * 512 + (chunk << 4);
  }
};

// Global address (or nullptr => zero fill) of one 16-byte chunk of the A tile.
template <typename T, int MODE>
__device__ __forceinline__ const u32x4* a_chunk_ptr(const GemmArgs& g, int bm, int row, int chunk, int kbase, int kend,
                                                    int pb, int py, int px) {
  constexpr int EPC = GT<T>::EPC;
  const T* A = (const T*)g.A;
  if (MODE == OP_KCONTIG) {
    const int m = bm * TILE + row, k0 = kbase + chunk * EPC;
    if (m >= g.M || k0 >= kend) return nullptr;
    return (const u32x4*)(A + (long)m * g.lda + k0);
  } else if (MODE == OP_KSTRIDED) {
    const int k = kbase + row, m0 = bm * TILE + chunk * EPC;
    if (k >= kend || m0 >= g.M) return nullptr;
    return (const u32x4*)(A + (long)k * g.lda + m0);
  } else {  // OP_CONV_A : m -> (pb,py,px) precomputed by the caller
    const int m = bm * TILE + row, k0 = kbase + chunk * EPC;
    if (m >= g.M || k0 >= kend) return nullptr;
    const int tap = k0 / g.cC, c0 = k0 - tap * g.cC;
    const int ty = tap / 3, tx = tap - ty * 3;
    const int yy = py + (ty - 1) * g.cDil, xx = px + (tx - 1) * g.cDil;
    if (yy < 0 || yy >= g.cH || xx < 0 || xx >= g.cW) return nullptr;
    return (const u32x4*)(A + (((long)pb * g.cH + yy) * g.cW + xx) * g.cC + c0);
  }
}

template <typename T, int MODE>
__device__ __forceinline__ const u32x4* b_chunk_ptr(const GemmArgs& g, int bn, int row, int chunk, int kbase, int kend) {
  constexpr int EPC = GT<T>::EPC;
  const T* B = (const T*)g.B;
  if (MODE == OP_KCONTIG) {
    const int n = bn * TILE + row, k0 = kbase + chunk * EPC;
    if (n >= g.N || k0 >= kend) return nullptr;
    return (const u32x4*)(B + (long)n * g.ldb + k0);
  } else if (MODE == OP_KSTRIDED) {
    const int k = kbase + row, n0 = bn * TILE + chunk * EPC;
    if (k >= kend || n0 >= g.N) return nullptr;
    return (const u32x4*)(B + (long)k * g.ldb + n0);
  } else {  // OP_CONV_B : k -> pixel, n -> (tap, ci)
    const int k = kbase + row, n0 = bn * TILE + chunk * EPC;
    if (k >= kend || n0 >= g.N) return nullptr;
    const int hw = g.cH * g.cW;
    const int pb = k / hw, rem = k - pb * hw;
    const int py = rem / g.cW, px = rem - py * g.cW;
    const int tap = n0 / g.cC, c0 = n0 - tap * g.cC;
    const int ty = tap / 3, tx = tap - ty * 3;
    const int yy = py + (ty - 1) * g.cDil, xx = px + (tx - 1) * g.cDil;
    if (yy < 0 || yy >= g.cH || xx < 0 || xx >= g.cW) return nullptr;
    return (const u32x4*)(B + (((long)pb * g.cH + yy) * g.cW + xx) * g.cC + c0);
  }
}

// Fragment of a 32-row (A: rows = m, B: rows = n) sub-tile for K sub-step s (0..3) of the LDS tile.
// bf16: 8 elements k = 16 s + 8 h + j.   f32: 4 elements k = 8 s + 4 h + t.
template <typename T, bool KS>
__device__ __forceinline__ u32x4 load_frag(const char* tile, int sub_base, int s, int lane) {
  const int r = lane & 31, h = lane >> 5;
  if (!KS) {
    const int row = sub_base + r, chunk = 2 * s + h;
    return *(const u32x4*)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  } else if (sizeof(T) == 2) {
    // ds_read_b64_tr_b16: 16-lane group G reads a 4(k) x 16(m) block; lane 4q+p supplies row q, cols 4p..4p+3
    const int G = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col = sub_base + 16 * (G & 1) + 4 * p;
    const int chunk = col >> 3, half = (col >> 2) & 1;
    const int row0 = 16 * s + 8 * (G >> 1) + q;
    const int row1 = row0 + 4;
    const int off0 = row0 * 256 + ((chunk ^ ((row0 & 3) << 2)) << 4) + half * 8;
    const int off1 = row1 * 256 + ((chunk ^ ((row1 & 3) << 2)) << 4) + half * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off1));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 o; o[0] = l2[0]; o[1] = l2[1]; o[2] = h2[0]; o[3] = h2[1];
    return o;
  } else {
    const int col = sub_base + r;
    u32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = *(const unsigned int*)(tile + (8 * s + 4 * h + t) * 512 + col * 4);
    return o;
  }
}

template <typename T>
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
  if (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[t]), __uint_as_float(b[t]), acc, 0, 0, 0);
  }
}

template <typename T, int AMODE, int BMODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  using GA = OperandGeom<T, AMODE>;
  using GB = OperandGeom<T, BMODE>;
  constexpr int BK = GT<T>::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16K | B 16K]

  // ---- workgroup -> tile: XCD-contiguous chunks, 8x8 tile patches inside a chunk
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int swz = (bid & 7) * (nwg >> 3) + (bid >> 3);
  const int patch = swz >> 6, within = swz & 63;
  const int bm = (patch % g.patches_m) * 8 + (within & 7);
  const int bn = (patch / g.patches_m) * 8 + (within >> 3);
  if (bm >= g.tiles_m || bn >= g.tiles_n) return;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  if (kbeg >= kend) return;
  const int nt = (kend - kbeg + BK - 1) / BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- per-thread staging slots (4 chunks of A, 4 of B)
  int a_row[4], a_chk[4], b_row[4], b_chk[4];
  int a_pb[4], a_py[4], a_px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    a_row[i] = idx >> GA::SHIFT; a_chk[i] = idx & (GA::CHUNKS_PER_ROW - 1);
    b_row[i] = idx >> GB::SHIFT; b_chk[i] = idx & (GB::CHUNKS_PER_ROW - 1);
    a_pb[i] = a_py[i] = a_px[i] = 0;
    if (AMODE == OP_CONV_A) {
      const int m = bm * TILE + a_row[i];
      const int hw = g.cH * g.cW;
      a_pb[i] = m / hw; const int rem = m - a_pb[i] * hw;
      a_py[i] = rem / g.cW; a_px[i] = rem - a_py[i] * g.cW;
    }
  }

  // ---- gather index math kept out of the K loop (integer divisions would otherwise out-weigh the MFMAs):
  //  OP_CONV_A: the pixel of each staged row is fixed; when Cin % BK == 0 the tap is uniform per K-tile.
  //  OP_CONV_B: the (tap, ci) of each staged column is fixed; the pixel of each staged row advances by BK per K-tile.
  constexpr int EPC = GT<T>::EPC;
  const bool a_fast = (AMODE == OP_CONV_A) && (g.cC % BK == 0);
  long a_pixoff[4];
  int bq_pb[4], bq_py[4], bq_px[4], b_dy[4], b_dx[4], b_c0[4];
  bool b_colok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_pixoff[i] = (((long)a_pb[i] * g.cH + a_py[i]) * g.cW + a_px[i]) * g.cC;
    bq_pb[i] = bq_py[i] = bq_px[i] = b_dy[i] = b_dx[i] = b_c0[i] = 0; b_colok[i] = false;
    if (BMODE == OP_CONV_B) {
      const int n0 = bn * TILE + b_chk[i] * EPC;
      b_colok[i] = n0 < g.N;
      const int tap = n0 / g.cC; b_c0[i] = n0 - tap * g.cC;
      const int ty = tap / 3, tx = tap - ty * 3;
      b_dy[i] = (ty - 1) * g.cDil; b_dx[i] = (tx - 1) * g.cDil;
      const int k = kbeg + b_row[i];
      const int hw = g.cH * g.cW;
      bq_pb[i] = k / hw; const int rem = k - bq_pb[i] * hw;
      bq_py[i] = rem / g.cW; bq_px[i] = rem - bq_py[i] * g.cW;
    }
  }

  // two register sets: the global loads of K-tile t+2 are issued before the MFMAs of tile t and are only waited for
  // at the end of tile t+1 (counted vmcnt), i.e. a load has a full K-tile + one MFMA block to land.
  u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
  auto stage_load = [&](int kt, u32x4 (&ra)[4], u32x4 (&rb)[4]) {   // called with kt = 0, 1, 2, ... in order
    const int kb = kbeg + kt * BK;
    int f_dy = 0, f_dx = 0, f_c = 0;
    if (a_fast) {
      const int tap = kb / g.cC; f_c = kb - tap * g.cC;
      const int ty = tap / 3, tx = tap - ty * 3;
      f_dy = (ty - 1) * g.cDil; f_dx = (tx - 1) * g.cDil;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4* pa;
      if (a_fast) {
        const int m = bm * TILE + a_row[i], k0 = kb + a_chk[i] * EPC;
        const int yy = a_py[i] + f_dy, xx = a_px[i] + f_dx;
        const bool ok = m < g.M && k0 < kend && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
        pa = ok ? (const u32x4*)((const T*)g.A + a_pixoff[i] + ((long)f_dy * g.cW + f_dx) * g.cC + f_c + a_chk[i] * EPC) : nullptr;
      } else {
        pa = a_chunk_ptr<T, AMODE>(g, bm, a_row[i], a_chk[i], kb, kend, a_pb[i], a_py[i], a_px[i]);
      }
      const u32x4* pb;
      if (BMODE == OP_CONV_B) {
        const int yy = bq_py[i] + b_dy[i], xx = bq_px[i] + b_dx[i];
        const bool ok = b_colok[i] && (kb + b_row[i] < kend) && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
        pb = ok ? (const u32x4*)((const T*)g.B + (((long)bq_pb[i] * g.cH + yy) * g.cW + xx) * g.cC + b_c0[i]) : nullptr;
        bq_px[i] += BK;                                    // advance this row's pixel to the next K-tile
        while (bq_px[i] >= g.cW) { bq_px[i] -= g.cW; ++bq_py[i]; }
        while (bq_py[i] >= g.cH) { bq_py[i] -= g.cH; ++bq_pb[i]; }
      } else {
        pb = b_chunk_ptr<T, BMODE>(g, bn, b_row[i], b_chk[i], kb, kend);
      }
      // Out-of-range chunks read a 16-byte zero constant instead: the select is on the ADDRESS, never on the data.
      // (A conditional load makes hipcc branch around every load and wait for it; a select on the loaded value
      // makes it wait for the whole K-tile right after issue — either way the HBM/L2 latency is exposed per K-tile.)
      typedef const __attribute__((address_space(1))) u32x4* gptr;          // keep these global_load (not flat_load:
      ra[i] = *(gptr)(pa ? pa : &g_zero_chunk);                             //  flat ops also count on lgkmcnt and would
      rb[i] = *(gptr)(pb ? pb : &g_zero_chunk);                             //  be waited for before every ds_read use)
    }
  };
  auto stage_write = [&](int buf, const u32x4 (&ra)[4], const u32x4 (&rb)[4]) {
    char* sa = smem + buf * (2 * LDS_TILE_BYTES);
    char* sb = sa + LDS_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4*)(sa + GA::lds_off(a_row[i], a_chk[i])) = ra[i];
      *(u32x4*)(sb + GB::lds_off(b_row[i], b_chk[i])) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int buf) {
    const char* sa = smem + buf * (2 * LDS_TILE_BYTES);
    const char* sb = sa + LDS_TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = load_frag<T, GA::KS>(sa, wm * 64 + i * 32, s, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag<T, GB::KS>(sb, wn * 64 + j * 32, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma<T>(acc[i][j], fa[i], fb[j]);
    }
  };

  stage_load(0, ra0, rb0);
  stage_write(0, ra0, rb0);
  if (nt > 1) stage_load(1, ra1, rb1);
  __syncthreads();

  for (int kt = 0; kt < nt; kt += 2) {
    // even tile kt lives in LDS buffer 0; set 1 holds tile kt+1 (in flight or landed)
    if (kt + 2 < nt) stage_load(kt + 2, ra0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nt) stage_write(1, ra1, rb1);
    __syncthreads();
    if (kt + 1 >= nt) break;
    // odd tile kt+1 lives in LDS buffer 1; set 0 holds tile kt+2
    if (kt + 3 < nt) stage_load(kt + 3, ra1, rb1);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nt) stage_write(0, ra0, rb0);
    __syncthreads();
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = bn * TILE + wn * 64 + j * 32 + r;
      if (n >= g.N) continue;
      const float bcol = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm * TILE + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= g.M) continue;
        float v = acc[i][j][e] + bcol;
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.drop) v = g.drop[(long)m * g.ldd + n] ? v * g.drop_scale : 0.f;
        if (g.ref) {
          const float rv = g.ref_bf16 ? bf16_bits_to_f32(((const unsigned short*)g.ref)[(long)m * g.ldr + n])
                                      : ((const float*)g.ref)[(long)m * g.ldr + n];
          v = rv > 0.f ? v * g.ref_scale : 0.f;
        }
        long o;
        if (g.oihw_cin > 0) {        // conv wgrad: n = tap*Cin + ci  ->  OIHW flat index
          const int tap = n / g.oihw_cin, ci = n - tap * g.oihw_cin;
          o = (long)m * g.ldc + (long)ci * 9 + tap;
        } else {
          o = (long)m * g.ldc + n;
        }
        if (g.atomic) atomicAdd((float*)g.C + o, v);
        else if (g.out_bf16) ((unsigned short*)g.C)[o] = f32_to_bf16_bits(v);
        else ((float*)g.C)[o] = v;
      }
    }
}

// =====================================================================================================================
// gemm2: LDS-DMA (global_load_lds) staged, STAGES-deep ring, BM x BN tile with one 64x64 sub-tile per wave.
//
// Why: at 128x128x64 the v1 tile moves 32 KiB into the CU per 2 MFLOP (64 FLOP/B) and a CU pulls only ~20-50 GB/s with
// one K-tile in flight (measured: 1 WG/CU conv5_3 = 19 GB/s/CU, 2 WG/CU fc6 = 48 GB/s/CU) — the kernel was bound by
// bytes in flight per CU, not by the MFMA pipe.  gemm2 keeps STAGES-1 K-tiles in flight per workgroup without spending
// VGPRs (the DMA writes LDS directly), waits with a counted vmcnt, and uses ONE raw s_barrier per K-tile.
// The LDS image is byte-identical to v1's (same XOR swizzles); since an LDS-DMA wave-instruction writes 1 KiB linearly,
// the swizzle is applied on the per-lane SOURCE address (lane l of instruction q owns physical chunk q*64+l).
template <typename T, int MODE, int ROWS_MN>
struct Geom2 {   // tile of ROWS_MN rows (m or n) x BK
  static constexpr bool KS = (MODE == OP_KSTRIDED || MODE == OP_CONV_B);
  static constexpr int BK = GT<T>::BK;
  static constexpr int CPR = KS ? (ROWS_MN * (int)sizeof(T) / 16) : 8;      // 16-byte chunks per LDS row
  static constexpr int NROW = KS ? BK : ROWS_MN;                            // LDS rows
  static constexpr int BYTES = NROW * CPR * 16;
  static constexpr int ROW_BYTES = CPR * 16;
  __device__ static __forceinline__ int swz(int row) {
    if (!KS) return (row >> 1) & 7;
    if (sizeof(T) == 2) return (row & 3) << 2;
    return 0;
  }
};

template <typename T, bool KS, int ROW_BYTES>
__device__ __forceinline__ u32x4 load_frag2(const char* tile, int sub_base, int s, int lane) {
  const int r = lane & 31, h = lane >> 5;
  if (!KS) {
    const int row = sub_base + r, chunk = 2 * s + h;
    return *(const u32x4*)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  } else if (sizeof(T) == 2) {
    const int G = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col = sub_base + 16 * (G & 1) + 4 * p;
    const int chunk = col >> 3, half = (col >> 2) & 1;
    const int row0 = 16 * s + 8 * (G >> 1) + q;
    const int row1 = row0 + 4;
    const int off0 = row0 * ROW_BYTES + ((chunk ^ ((row0 & 3) << 2)) << 4) + half * 8;
    const int off1 = row1 * ROW_BYTES + ((chunk ^ ((row1 & 3) << 2)) << 4) + half * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off1));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 o; o[0] = l2[0]; o[1] = l2[1]; o[2] = h2[0]; o[3] = h2[1];
    return o;
  } else {
    const int col = sub_base + r;
    u32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = *(const unsigned int*)(tile + (8 * s + 4 * h + t) * ROW_BYTES + col * 4);
    return o;
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int AMODE, int BMODE, int BM, int BN, int STAGES>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, 1) void gemm2_kernel(GemmArgs g) {
  using GA = Geom2<T, AMODE, BM>;
  using GB = Geom2<T, BMODE, BN>;
  constexpr int BK = GT<T>::BK, EPC = GT<T>::EPC;
  constexpr int NWN = BN / 64, NW = (BM / 64) * NWN, NT = NW * 64;
  constexpr int A_SLOTS = GA::BYTES / 16 / NT, B_SLOTS = GB::BYTES / 16 / NT;     // 16-byte chunks per thread per K-tile
  constexpr int STAGE_BYTES = GA::BYTES + GB::BYTES;
  constexpr int GROUP = A_SLOTS + B_SLOTS;                                        // LDS-DMA instructions per wave per K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];                     // [STAGES][A | B]

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int swz = (bid & 7) * (nwg >> 3) + (bid >> 3);
  const int patch = swz >> 6, within = swz & 63;
  const int bm = (patch % g.patches_m) * 8 + (within & 7);
  const int bn = (patch / g.patches_m) * 8 + (within >> 3);
  if (bm >= g.tiles_m || bn >= g.tiles_n) return;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  if (kbeg >= kend) return;
  const int nt = (kend - kbeg + BK - 1) / BK;
  const int m0 = bm * BM, n0t = bn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- per-thread DMA slots: slot i owns physical chunk L = i*NT + tid of the operand's LDS image.
  // All address arithmetic is incremental: per K-tile a slot costs one pointer add, one bound compare and the DMA
  // itself (the first version recomputed pointers, taps and bounds per K-tile: ~160 VALU + ~200 SALU instructions per
  // wave per K-tile against 16 MFMAs — the kernel was instruction-issue bound, SQ counters in profiles/).
  typedef const __attribute__((address_space(1))) void* gvoid;
  typedef __attribute__((address_space(3))) void* lvoid;
  const bool a_fast = (AMODE == OP_CONV_A) && (g.cC % BK == 0);
  const int tiles_per_tap = a_fast ? g.cC / BK : 1;

  const char* a_ptr[A_SLOTS];      // source of this slot for the NEXT K-tile to be issued
  bool a_ok[A_SLOTS];              // row/column (and, conv: tap) validity
  int a_k[A_SLOTS];                // k index this slot starts at inside a K-tile (bound check against kend)
  int a_py[A_SLOTS], a_px[A_SLOTS], a_pb[A_SLOTS];
  const char* a_pix[A_SLOTS];
  long a_step = 0;
#pragma unroll
  for (int i = 0; i < A_SLOTS; ++i) {
    const int L = i * NT + tid;
    const int row = L / GA::CPR, chk = (L % GA::CPR) ^ GA::swz(row);      // logical chunk fetched into physical slot L
    a_py[i] = a_px[i] = a_pb[i] = 0; a_pix[i] = nullptr;
    if (AMODE == OP_KCONTIG) {
      const int m = m0 + row;
      a_ok[i] = m < g.M; a_k[i] = chk * EPC;
      a_ptr[i] = (const char*)((const T*)g.A + (long)(a_ok[i] ? m : 0) * g.lda + kbeg + chk * EPC);
      a_step = (long)BK * sizeof(T);
    } else if (AMODE == OP_KSTRIDED) {
      const int mm = m0 + chk * EPC;
      a_ok[i] = mm < g.M; a_k[i] = row;
      a_ptr[i] = (const char*)((const T*)g.A + (long)(kbeg + row) * g.lda + (a_ok[i] ? mm : 0));
      a_step = (long)BK * g.lda * sizeof(T);
    } else {
      const int m = m0 + row;
      const int hw = g.cH * g.cW;
      const int mc = m < g.M ? m : 0;
      a_pb[i] = mc / hw; const int rem = mc - a_pb[i] * hw;
      a_py[i] = rem / g.cW; a_px[i] = rem - a_py[i] * g.cW;
      a_ok[i] = m < g.M; a_k[i] = chk * EPC;
      a_pix[i] = (const char*)((const T*)g.A + (((long)a_pb[i] * g.cH + a_py[i]) * g.cW + a_px[i]) * g.cC + chk * EPC);
      a_ptr[i] = a_pix[i];
      a_step = (long)BK * sizeof(T);
    }
  }
  bool a_tapok[A_SLOTS];
  int a_tap = 0, a_tt = 0;          // conv fast path: current tap and K-tiles consumed inside it
  auto conv_a_set_tap = [&](int tap, int cofs) {
    const int ty = tap / 3, tx = tap - ty * 3;
    const int dy = (ty - 1) * g.cDil, dx = (tx - 1) * g.cDil;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
      const int yy = a_py[i] + dy, xx = a_px[i] + dx;
      a_tapok[i] = a_ok[i] && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
      a_ptr[i] = a_pix[i] + (((long)dy * g.cW + dx) * g.cC + cofs) * (long)sizeof(T);
    }
  };
  if (a_fast) {
    a_tap = kbeg / g.cC;
    const int cofs = kbeg - a_tap * g.cC;
    a_tt = cofs / BK;
    conv_a_set_tap(a_tap, cofs);
  }

  const char* b_ptr[B_SLOTS];
  bool b_ok[B_SLOTS];
  int b_k[B_SLOTS], bq_py[B_SLOTS], bq_px[B_SLOTS], b_dy[B_SLOTS], b_dx[B_SLOTS];
  long b_step = 0;
#pragma unroll
  for (int i = 0; i < B_SLOTS; ++i) {
    const int L = i * NT + tid;
    const int row = L / GB::CPR, chk = (L % GB::CPR) ^ GB::swz(row);
    bq_py[i] = bq_px[i] = b_dy[i] = b_dx[i] = 0;
    if (BMODE == OP_KCONTIG) {
      const int n = n0t + row;
      b_ok[i] = n < g.N; b_k[i] = chk * EPC;
      b_ptr[i] = (const char*)((const T*)g.B + (long)(b_ok[i] ? n : 0) * g.ldb + kbeg + chk * EPC);
      b_step = (long)BK * sizeof(T);
    } else if (BMODE == OP_KSTRIDED) {
      const int nn = n0t + chk * EPC;
      b_ok[i] = nn < g.N; b_k[i] = row;
      b_ptr[i] = (const char*)((const T*)g.B + (long)(kbeg + row) * g.ldb + (b_ok[i] ? nn : 0));
      b_step = (long)BK * g.ldb * sizeof(T);
    } else {   // OP_CONV_B: column = (tap, ci) fixed; row = pixel kbeg+row, advancing BK pixels per K-tile
      const int nn = n0t + chk * EPC;
      b_ok[i] = nn < g.N; b_k[i] = row;
      const int nc = b_ok[i] ? nn : 0;
      const int tap = nc / g.cC, c0 = nc - tap * g.cC;
      const int ty = tap / 3, tx = tap - ty * 3;
      b_dy[i] = (ty - 1) * g.cDil; b_dx[i] = (tx - 1) * g.cDil;
      const int k = kbeg + row;
      const int hw = g.cH * g.cW;
      const int pb = k / hw, rem = k - pb * hw;
      bq_py[i] = rem / g.cW; bq_px[i] = rem - bq_py[i] * g.cW;
      // shifted pixel = linear pixel + dy*W + dx whenever it is inside the image (checked per K-tile)
      b_ptr[i] = (const char*)((const T*)g.B + ((long)k + (long)b_dy[i] * g.cW + b_dx[i]) * g.cC + c0);
      b_step = (long)BK * g.cC * sizeof(T);
    }
  }

  auto stage_issue = [&](int kt) {          // kt = 0, 1, 2, ... in order (the slot state is incremental)
    const int kb = kbeg + kt * BK;
    char* sbase = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
      bool ok;
      const char* p = a_ptr[i];
      if (AMODE == OP_CONV_A && !a_fast) {        // generic path (first layer, Cin padded to 8/4): per chunk tap math
        const int k0 = kb + a_k[i];
        ok = a_ok[i] && k0 < kend;
        const int kc = ok ? k0 : 0;
        const int tap = kc / g.cC, c0 = kc - tap * g.cC;
        const int ty = tap / 3, tx = tap - ty * 3;
        const int dy = (ty - 1) * g.cDil, dx = (tx - 1) * g.cDil;
        const int yy = a_py[i] + dy, xx = a_px[i] + dx;
        ok = ok && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
        p = a_pix[i] + (((long)dy * g.cW + dx) * g.cC + c0 - a_k[i]) * (long)sizeof(T);
      } else if (AMODE == OP_CONV_A) {
        ok = a_tapok[i] && (kb + a_k[i] < kend);
      } else {
        ok = a_ok[i] && (kb + a_k[i] < kend);
      }
      const void* src = ok ? (const void*)p : (const void*)&g_zero_chunk;
      __builtin_amdgcn_global_load_lds((gvoid)src, (lvoid)(sbase + (i * NW + wave) * 1024), 16, 0, 0);
      a_ptr[i] += a_step;
    }
    if (a_fast) {
      if (++a_tt == tiles_per_tap) { a_tt = 0; ++a_tap; if (a_tap < 9) conv_a_set_tap(a_tap, 0); }
    }
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
      bool ok = b_ok[i] && (kb + b_k[i] < kend);
      if (BMODE == OP_CONV_B) {
        const int yy = bq_py[i] + b_dy[i], xx = bq_px[i] + b_dx[i];
        ok = ok && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
        bq_px[i] += BK;
        while (bq_px[i] >= g.cW) { bq_px[i] -= g.cW; ++bq_py[i]; }
        while (bq_py[i] >= g.cH) bq_py[i] -= g.cH;
      }
      const void* src = ok ? (const void*)b_ptr[i] : (const void*)&g_zero_chunk;
      __builtin_amdgcn_global_load_lds((gvoid)src, (lvoid)(sbase + GA::BYTES + (i * NW + wave) * 1024), 16, 0, 0);
      b_ptr[i] += b_step;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int kt) {
    const char* sa = smem + (kt % STAGES) * STAGE_BYTES;
    const char* sb = sa + GA::BYTES;
    u32x4 fa[2], fb[2], na[2], nb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = load_frag2<T, GA::KS, GA::ROW_BYTES>(sa, wm * 64 + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = load_frag2<T, GB::KS, GB::ROW_BYTES>(sb, wn * 64 + j * 32, 0, lane);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {       // fragments of the next K sub-step are in flight while this sub-step's MFMAs issue
#pragma unroll
        for (int i = 0; i < 2; ++i) na[i] = load_frag2<T, GA::KS, GA::ROW_BYTES>(sa, wm * 64 + i * 32, s + 1, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) nb[j] = load_frag2<T, GB::KS, GB::ROW_BYTES>(sb, wn * 64 + j * 32, s + 1, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma<T>(acc[i][j], fa[i], fb[j]);
      if (s < 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
      }
    }
  };

  // prologue: STAGES-1 K-tiles in flight
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t)
    if (t < nt) stage_issue(t);

  for (int kt = 0; kt < nt; ++kt) {
    // tile kt must have landed: groups kt .. min(kt+STAGES-2, nt-1) are outstanding
    if (kt + STAGES - 2 < nt) wait_vmcnt<GROUP * (STAGES - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();           // everybody's DMA for tile kt landed; everybody finished computing tile kt-1
    if (kt + STAGES - 1 < nt) stage_issue(kt + STAGES - 1);      // refill the stage tile kt-1 lived in
    compute(kt);
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0t + wn * 64 + j * 32 + r;
      if (n >= g.N) continue;
      const float bcol = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= g.M) continue;
        float v = acc[i][j][e] + bcol;
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.drop) v = g.drop[(long)m * g.ldd + n] ? v * g.drop_scale : 0.f;
        if (g.ref) {
          const float rv = g.ref_bf16 ? bf16_bits_to_f32(((const unsigned short*)g.ref)[(long)m * g.ldr + n])
                                      : ((const float*)g.ref)[(long)m * g.ldr + n];
          v = rv > 0.f ? v * g.ref_scale : 0.f;
        }
        long o;
        if (g.oihw_cin > 0) {        // conv wgrad: n = tap*Cin + ci  ->  OIHW flat index
          const int tap = n / g.oihw_cin, ci = n - tap * g.oihw_cin;
          o = (long)m * g.ldc + (long)ci * 9 + tap;
        } else {
          o = (long)m * g.ldc + n;
        }
        if (g.atomic) atomicAdd((float*)g.C + o, v);
        else if (g.out_bf16) ((unsigned short*)g.C)[o] = f32_to_bf16_bits(v);
        else ((float*)g.C)[o] = v;
      }
    }
}

template <typename T, int AMODE, int BMODE>
int launch(GemmArgs& g, int splitk, hipStream_t stream) {
  constexpr int BK = GT<T>::BK;
  g.tiles_m = (g.M + TILE - 1) / TILE;
  g.tiles_n = (g.N + TILE - 1) / TILE;
  g.patches_m = (g.tiles_m + 7) / 8;
  const int patches_n = (g.tiles_n + 7) / 8;
  if (splitk < 1) splitk = 1;
  int kps = (g.K + splitk - 1) / splitk;
  kps = ((kps + BK - 1) / BK) * BK;
  g.k_per_split = kps;
  splitk = (g.K + kps - 1) / kps;
  if (splitk > 1 && !g.atomic) return -2;
  dim3 grid(g.patches_m * patches_n * 64, 1, splitk), block(256);
  static bool attr_done = false;   // raising the dynamic-LDS cap is idempotent
  auto kern = gemm_kernel<T, AMODE, BMODE>;
  (void)attr_done;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * LDS_TILE_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, block, 4 * LDS_TILE_BYTES, stream, g);
  SW_CHECK_LAUNCH();
  return 0;
}

template <typename T, int AMODE, int BMODE, int BM, int BN, int STAGES>
int launch2(GemmArgs& g, int splitk, hipStream_t stream) {
  constexpr int BK = GT<T>::BK;
  using GA = Geom2<T, AMODE, BM>;
  using GB = Geom2<T, BMODE, BN>;
  constexpr int LDS = STAGES * (GA::BYTES + GB::BYTES);
  constexpr int NT = (BM / 64) * (BN / 64) * 64;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  g.patches_m = (g.tiles_m + 7) / 8;
  const int patches_n = (g.tiles_n + 7) / 8;
  if (splitk < 1) splitk = 1;
  int kps = (g.K + splitk - 1) / splitk;
  kps = ((kps + BK - 1) / BK) * BK;
  g.k_per_split = kps;
  splitk = (g.K + kps - 1) / kps;
  if (splitk > 1 && !g.atomic) return -2;
  dim3 grid(g.patches_m * patches_n * 64, 1, splitk), block(NT);
  auto kern = gemm2_kernel<T, AMODE, BMODE, BM, BN, STAGES>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, block, LDS, stream, g);
  SW_CHECK_LAUNCH();
  return 0;
}

// Tile choice.  Staging cost per MFMA falls with the tile (LDS-DMA instructions per 32x32x16 MFMA: 0.5 at 128x128,
// 0.375 at 256x128, 0.25 at 256x256) and so does the traffic into the CU, but a launch must still fill 256 CUs:
//   N <= 64 (conv1_x)                      -> 256x64,  3 stages
//   >= 200 tiles of 256x256 (FC layers)    -> 256x256, 2 stages (16 waves)
//   >= 400 tiles of 256x128                -> 256x128, 3 stages
//   >= 512 tiles of 128x128 (split-K wgrad)-> 128x128, 2 stages (64 KiB LDS: two workgroups per CU)
//   else (conv4/conv5 at batch 2)          -> 128x128, 4 stages
template <typename T, int AMODE, int BMODE>
int launch_auto(GemmArgs& g, int splitk, hipStream_t s) {
  static const char* v = getenv("SW_GEMM_V");               // development switch
  if (v && v[0] == '1') return launch<T, AMODE, BMODE>(g, splitk, s);            // register-staged v1 kernel
  const long sk = splitk < 1 ? 1 : splitk;
  auto tiles = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * sk; };
  if (v && v[0] == '2') return launch2<T, AMODE, BMODE, 128, 128, 4>(g, splitk, s);
  if (v && v[0] == '3') return launch2<T, AMODE, BMODE, 256, 128, 3>(g, splitk, s);
  if (v && v[0] == '4') return launch2<T, AMODE, BMODE, 256, 256, 2>(g, splitk, s);
  if (v && v[0] == '5') return launch2<T, AMODE, BMODE, 128, 128, 2>(g, splitk, s);
  if (g.N <= 64 && tiles(256, 64) >= 256) return launch2<T, AMODE, BMODE, 256, 64, 3>(g, splitk, s);
  if (g.N > 128 && tiles(256, 256) >= 200) return launch2<T, AMODE, BMODE, 256, 256, 2>(g, splitk, s);
  if (tiles(256, 128) >= 400) return launch2<T, AMODE, BMODE, 256, 128, 3>(g, splitk, s);
  if (tiles(128, 128) >= 512) return launch2<T, AMODE, BMODE, 128, 128, 2>(g, splitk, s);
  return launch2<T, AMODE, BMODE, 128, 128, 4>(g, splitk, s);
}

template <typename T>
int dispatch_modes(GemmArgs& g, int amode, int bmode, int splitk, hipStream_t s) {
  if (amode == OP_KCONTIG && bmode == OP_KCONTIG) return launch_auto<T, OP_KCONTIG, OP_KCONTIG>(g, splitk, s);
  if (amode == OP_KCONTIG && bmode == OP_KSTRIDED) return launch_auto<T, OP_KCONTIG, OP_KSTRIDED>(g, splitk, s);
  if (amode == OP_KSTRIDED && bmode == OP_KSTRIDED) return launch_auto<T, OP_KSTRIDED, OP_KSTRIDED>(g, splitk, s);
  if (amode == OP_CONV_A && bmode == OP_KCONTIG) return launch_auto<T, OP_CONV_A, OP_KCONTIG>(g, splitk, s);
  if (amode == OP_KSTRIDED && bmode == OP_CONV_B) return launch_auto<T, OP_KSTRIDED, OP_CONV_B>(g, splitk, s);
  return -3;
}

int check_align(const void* p) { return (((uintptr_t)p) & 15) ? -4 : 0; }

}  // namespace

extern "C" int sw_gemm(int dtype, int a_kstrided, int b_kstrided, int M, int N, int K, const void* A, long lda,
                       const void* B, long ldb, void* C, long ldc, const sw_epilogue* ep, int splitk,
                       hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int epc = dtype == SW_BF16 ? 8 : 4;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if (check_align(A) || check_align(B)) return -4;
  if ((lda % epc) || (ldb % epc)) return -5;
  if (!a_kstrided && (K % epc)) return -5;
  if (!b_kstrided && (K % epc)) return -5;
  if (a_kstrided && (M % epc)) return -5;
  if (b_kstrided && (N % epc)) return -5;
  if (a_kstrided && !b_kstrided) return -3;
  GemmArgs g = {};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  if (ep) {
    g.bias = ep->bias; g.drop = ep->drop_mask; g.ldd = ep->ld_drop; g.drop_scale = ep->drop_scale;
    g.ref = ep->relu_ref; g.ldr = ep->ld_ref; g.ref_scale = ep->ref_scale; g.ref_bf16 = ep->ref_dtype == SW_BF16;
    g.relu = ep->relu; g.out_bf16 = ep->out_dtype == SW_BF16; g.atomic = ep->accumulate_atomic;
  }
  const int am = a_kstrided ? OP_KSTRIDED : OP_KCONTIG, bmo = b_kstrided ? OP_KSTRIDED : OP_KCONTIG;
  return dtype == SW_BF16 ? dispatch_modes<unsigned short>(g, am, bmo, splitk, stream)
                          : dispatch_modes<float>(g, am, bmo, splitk, stream);
}

// conv3x3 (stride 1, pad = dilation) forward / data-gradient as an implicit GEMM over an NHWC tensor:
//   out[(img,y,x)][co] = sum_{tap,ci} in[img, y+(ty-1)d, x+(tx-1)d, ci] * Wk[co][tap][ci]
extern "C" int sw_conv3x3_igemm(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* in,
                                const void* wk, void* out, const sw_epilogue* ep, hipStream_t stream) {
  const int epc = dtype == SW_BF16 ? 8 : 4;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if (Cin % epc) return -5;
  if (check_align(in) || check_align(wk)) return -4;
  GemmArgs g = {};
  g.A = in; g.B = wk; g.C = out; g.M = nimg * H * W; g.N = Cout; g.K = 9 * Cin; g.lda = 0; g.ldb = 9L * Cin; g.ldc = Cout;
  g.cH = H; g.cW = W; g.cC = Cin; g.cDil = dilation;
  if (ep) {
    g.bias = ep->bias; g.drop = ep->drop_mask; g.ldd = ep->ld_drop; g.drop_scale = ep->drop_scale;
    g.ref = ep->relu_ref; g.ldr = ep->ld_ref; g.ref_scale = ep->ref_scale; g.ref_bf16 = ep->ref_dtype == SW_BF16;
    g.relu = ep->relu; g.out_bf16 = ep->out_dtype == SW_BF16; g.atomic = ep->accumulate_atomic;
  }
  return dtype == SW_BF16 ? dispatch_modes<unsigned short>(g, OP_CONV_A, OP_KCONTIG, 1, stream)
                          : dispatch_modes<float>(g, OP_CONV_A, OP_KCONTIG, 1, stream);
}

namespace {
// workspace [co][tap][ci] -> OIHW [co][ci][tap]  (reads coalesced along ci; 14.7 M elements for the whole VGG16)
__global__ void wk_to_oihw_kernel(int Cout, int Cin, const float* __restrict__ wk, float* __restrict__ out) {
  const long total = (long)Cout * 9 * Cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin); const long t = i / Cin;
    const int tap = (int)(t % 9); const int co = (int)(t / 9);
    out[((long)co * Cin + ci) * 9 + tap] = wk[i];
  }
}
}  // namespace

// conv3x3 weight gradient:  dW[co][ci][ty][tx] (OIHW, f32, overwritten)
//   = sum_{img,y,x} dY[(img,y,x)][co] * X[img, y+(ty-1)d, x+(tx-1)d, ci]
// Split-K partial tiles are accumulated with f32 atomics into workspace[co][tap][ci] — contiguous along ci, i.e.
// 128-byte segments per half-wave, the shape the memory-side atomic units serve at full rate; a 36-byte-strided
// OIHW scatter runs an order of magnitude slower — and permuted to OIHW by a small second kernel.
extern "C" int sw_conv3x3_wgrad(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x,
                                const void* dy, float* dw_oihw, float* workspace, int splitk, hipStream_t stream) {
  const int epc = dtype == SW_BF16 ? 8 : 4;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if ((Cin % epc) || (Cout % epc)) return -5;
  if (check_align(x) || check_align(dy)) return -4;
  const size_t nelem = (size_t)Cout * 9 * Cin;
  hipError_t e = hipMemsetAsync(workspace, 0, nelem * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  GemmArgs g = {};
  g.A = dy; g.B = x; g.C = workspace; g.M = Cout; g.N = 9 * Cin; g.K = nimg * H * W; g.lda = Cout; g.ldb = 0;
  g.ldc = 9L * Cin; g.cH = H; g.cW = W; g.cC = Cin; g.cDil = dilation; g.atomic = 1; g.oihw_cin = 0;
  const int rc = dtype == SW_BF16 ? dispatch_modes<unsigned short>(g, OP_KSTRIDED, OP_CONV_B, splitk, stream)
                                  : dispatch_modes<float>(g, OP_KSTRIDED, OP_CONV_B, splitk, stream);
  if (rc) return rc;
  long blocks = ((long)nelem + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(wk_to_oihw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, Cout, Cin, workspace, dw_oihw);
  SW_CHECK_LAUNCH();
  return 0;
}
