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

template <typename T>
int dispatch_modes(GemmArgs& g, int amode, int bmode, int splitk, hipStream_t s) {
  if (amode == OP_KCONTIG && bmode == OP_KCONTIG) return launch<T, OP_KCONTIG, OP_KCONTIG>(g, splitk, s);
  if (amode == OP_KCONTIG && bmode == OP_KSTRIDED) return launch<T, OP_KCONTIG, OP_KSTRIDED>(g, splitk, s);
  if (amode == OP_KSTRIDED && bmode == OP_KSTRIDED) return launch<T, OP_KSTRIDED, OP_KSTRIDED>(g, splitk, s);
  if (amode == OP_CONV_A && bmode == OP_KCONTIG) return launch<T, OP_CONV_A, OP_KCONTIG>(g, splitk, s);
  if (amode == OP_KSTRIDED && bmode == OP_CONV_B) return launch<T, OP_KSTRIDED, OP_CONV_B>(g, splitk, s);
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
