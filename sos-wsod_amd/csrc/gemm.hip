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
// Kernel structure, tile choice and the staging path are described at gemm2_tile below: operands are staged by LDS-DMA buffer
// loads (no register staging), K-step of 128 bytes per row (64 bf16 / 32 f32), a 2-stage LDS ring with counted vmcnt waits and raw
// barriers.  bf16: v_mfma_f32_16x16x32_bf16; f32: v_mfma_f32_32x32x2_f32 (exact f32).  K-contiguous operands are read from LDS
// with ds_read_b128 through an XOR swizzle applied on the DMA source side; K-strided operands keep their natural [k][m] image
// and are transposed on the fly by ds_read_b64_tr_b16 (bf16) or read element-wise (f32).  Two main loops: 16 waves of 64x64
// (one barrier per K-tile) and, for 256x256 tiles with a K-contiguous A operand, 8 waves of 128x64 in two staggered groups
// (ping-pong: one group issues MFMAs while the other loads).  gemm2_grouped_kernel walks a list of conv weight-gradient
// problems with the same tile code.
#include <stdlib.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {

enum { OP_KCONTIG = 0, OP_KSTRIDED = 1, OP_CONV_A = 2, OP_CONV_B = 3, OP_CONV_A_GEN = 4 };   // _GEN: Cin % BK != 0 (first layer)

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  int tiles_m, tiles_n, patches_m, k_per_split;
  int patch_m_log2;                // an XCD patch is (1 << patch_m_log2) x (64 >> patch_m_log2) tiles (launch2 picks the shape)
  int cH, cW, cC, cDil;            // geometry of the gathered NHWC tensor (conv modes)
  const float* bias;               // per column n (or per row m when bias_on_m)
  const uint8_t* drop; long ldd; float drop_scale;
  unsigned long long drop_seed_mixed, drop_offset; float drop_p;   // hash dropout: splitmix64(seed) pre-mixed on the host side
  const unsigned long long* drop_offset_dev;                       // optional device-resident part of the stream position
  const void* ref; long ldr; float ref_scale; int ref_bf16;
  const void* res; long ldres; int res_bf16;            // residual added before the ReLU
  int relu, out_bf16, atomic, oihw_cin, staged_out;
  unsigned a_bytes, b_bytes;       // extents of the A / B operands (buffer descriptors' num_records)
  long slab_stride;                // > 0: split z stores its partial tile to C + z*slab_stride (plain stores, no atomics)
  int vgrid;                       // persistent form: virtual workgroup count walked by gridDim.x resident workgroups (0 = off)
  int krot;                        // conv: rotate the channel-chunk order by the M-tile index (L2 channel spread)
  float* absmax;                   // optional: atomicMax of |stored value| (IEEE bits of a non-negative float are monotone)
  // fused SGD update (round 6, ping-pong form, plain f32 "output"): C[m][n] is the gradient of the [M][N] parameter sgd_param; the
  // epilogue applies torch.optim.SGD's momentum update to it and rewrites the parameter's compute-dtype copies — the gradient is never
  // written (sw_epilogue.sgd_fused)
  float* sgd_param; float* sgd_mom; long sgd_ld;
  unsigned short* sgd_st0; long sgd_ld0; unsigned short* sgd_st1; long sgd_ld1;
  const float* sgd_hyper; float sgd_lr, sgd_wd, sgd_momentum, sgd_gscale; int sgd_first;
};

template <typename T> struct GT;
template <> struct GT<unsigned short> { static constexpr int EPC = 8, BK = 64; };   // bf16
template <> struct GT<float> { static constexpr int EPC = 4, BK = 32; };

__device__ const u32x4 g_zero_chunk = {0u, 0u, 0u, 0u};   // source of zero fill for out-of-range 16-byte chunks


// MFMA shape per dtype.  bf16: v_mfma_f32_16x16x32_bf16 (16 cycles, K = 32) — on MI355X the chip holds a higher clock
// on this shape than on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md "DVFS give-back" item 7); set
// -DSW_MFMA32 to build the 32x32x16 variant for A/B runs.  f32: v_mfma_f32_32x32x2_f32 (exact f32).
template <typename T> struct Mma;
#ifndef SW_MFMA32
template <> struct Mma<unsigned short> {
  static constexpr int TS = 16, NSTEP = 2, NACC = 4;         // sub-tile size, K sub-steps per K-tile, acc regs per lane
  typedef f32x4 Acc;
  __device__ static __forceinline__ void mma(Acc& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
  // C/D map: col = lane&15, row = (lane>>4)*4 + e
  __device__ static __forceinline__ int row(int lane, int e) { return (lane >> 4) * 4 + e; }
  __device__ static __forceinline__ int col(int lane) { return lane & 15; }
};
#else
template <> struct Mma<unsigned short> {
  static constexpr int TS = 32, NSTEP = 4, NACC = 16;
  typedef f32x16 Acc;
  __device__ static __forceinline__ void mma(Acc& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ int row(int lane, int e) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
  __device__ static __forceinline__ int col(int lane) { return lane & 31; }
};
#endif
template <> struct Mma<float> {
  static constexpr int TS = 32, NSTEP = 4, NACC = 16;
  typedef f32x16 Acc;
  __device__ static __forceinline__ void mma(Acc& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[t]), __uint_as_float(b[t]), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ int row(int lane, int e) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
  __device__ static __forceinline__ int col(int lane) { return lane & 31; }
};

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
    if (sizeof(T) == 2) return ((row & 3) << 2) | (Mma<T>::TS == 16 ? ((row >> 3) & 1) << 1 : 0);
    return 0;
  }
};

// Fragment of a TS-row sub-tile (A: rows = m, B: rows = n) for K sub-step s of the LDS tile.
//   TS = 32: bf16 8 elements k = 16 s + 8 h + j (h = lane>>5); f32 4 elements k = 8 s + 4 h + t
//   TS = 16: bf16 8 elements k = 32 s + 8 g + j (g = lane>>4)
template <typename T, bool KS, int ROW_BYTES>
__device__ __forceinline__ u32x4 load_frag2(const char* tile, int sub_base, int s, int lane) {
  constexpr int TS = Mma<T>::TS;
  if (!KS) {
    const int row = sub_base + (lane & (TS - 1));
    const int chunk = TS == 32 ? 2 * s + (lane >> 5) : 4 * s + (lane >> 4);
    return *(const u32x4*)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  } else if (sizeof(T) == 2) {
    // ds_read_b64_tr_b16: 16-lane group G reads a 4(k) x 16(m) block; lane 4q+p supplies row q, cols 4p..4p+3
    const int G = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col = TS == 32 ? sub_base + 16 * (G & 1) + 4 * p : sub_base + 4 * p;
    const int chunk = col >> 3, half = (col >> 2) & 1;
    const int row0 = TS == 32 ? 16 * s + 8 * (G >> 1) + q : 32 * s + 8 * G + q;
    const int row1 = row0 + 4;
    const int off0 = row0 * ROW_BYTES + ((chunk ^ Geom2<T, OP_KSTRIDED, 128>::swz(row0)) << 4) + half * 8;
    const int off1 = row1 * ROW_BYTES + ((chunk ^ Geom2<T, OP_KSTRIDED, 128>::swz(row1)) << 4) + half * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off1));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 o; o[0] = l2[0]; o[1] = l2[1]; o[2] = h2[0]; o[3] = h2[1];
    return o;
  } else {
    const int r = lane & 31, h = lane >> 5;
    const int col = sub_base + r;
    u32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = *(const unsigned int*)(tile + (8 * s + 4 * h + t) * ROW_BYTES + col * 4);
    return o;
  }
}

template <int V> struct IntC { static constexpr int value = V; };

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One output tile (bm, bn) of K-split `zsplit`: the whole main loop and epilogue.  Called by gemm2_kernel (one problem, the
// workgroup -> tile map of the plain / split-K / persistent launches) and by gemm2_grouped_kernel (a list of problems).
template <typename T, int AMODE, int BMODE, int BM, int BN, int STAGES, int WTM, int WTN>
__device__ __forceinline__ void gemm2_tile(const GemmArgs& g, const int bm, const int bn, const int zsplit, char* const smem) {
  using GA = Geom2<T, AMODE, BM>;
  using GB = Geom2<T, BMODE, BN>;
  constexpr int BK = GT<T>::BK, EPC = GT<T>::EPC;
  constexpr int NWN = BN / WTN, NW = (BM / WTM) * NWN, NT = NW * 64;      // one WTM x WTN sub-tile per wave
  using MM = Mma<T>;
  constexpr int TS = MM::TS, NSTEP = MM::NSTEP, NACC = MM::NACC;
  constexpr int MI = WTM / TS, NI = WTN / TS;
  constexpr bool PREFETCH_FRAGS = (NW <= 8);       // 16-wave tiles run 4 waves per SIMD under a 128-VGPR cap:
  constexpr bool PREFETCH_A = (NW > 8) && sizeof(T) == 2 && BMODE != OP_CONV_B;   // there only the next sub-step's A fragments
                                                                                   // fit (111-126 VGPRs, no spill): +1-2 %
  constexpr int A_SLOTS = GA::BYTES / 16 / NT, B_SLOTS = GB::BYTES / 16 / NT;     // 16-byte chunks per thread per K-tile
  constexpr int STAGE_BYTES = GA::BYTES + GB::BYTES;
  constexpr int GROUP = A_SLOTS + B_SLOTS;                                        // LDS-DMA instructions per wave per K-tile

  const int kbeg = zsplit * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  if (kbeg >= kend) return;
  const int nt = (kend - kbeg + BK - 1) / BK;
  const int m0 = bm * BM, n0t = bn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- per-thread DMA slots: slot i owns physical chunk L = i*NT + tid of the operand's LDS image.
  // The DMA is a raw BUFFER load to LDS (buffer_load_dwordx4 ... offen lds): per slot a 32-bit byte offset in a VGPR
  // (fixed for the plain GEMM modes), the K-tile advance in the wave-uniform SGPR soffset, and zero fill for rows /
  // taps / K-tails outside the operand by handing the hardware an offset beyond num_records (out-of-range buffer loads
  // return 0) — no pointer selects, no 64-bit adds.  (Recomputing pointers, taps and bounds per K-tile cost ~160 VALU +
  // ~200 SALU instructions per wave per K-tile against 8-16 MFMAs: the kernel was instruction-issue bound; profiles/.)
  typedef __attribute__((address_space(3))) void* lvoid;
  constexpr unsigned INVALID = 0xFFFFFF00u;                // >= num_records of every operand (checked by the host)
  constexpr unsigned ES = (unsigned)sizeof(T);
  constexpr bool a_fast = (AMODE == OP_CONV_A);            // Cin % BK == 0: the tap is uniform per K-tile
  const int tiles_per_tap = a_fast ? g.cC / BK : 1;
  // (sw_dma16, common.h: the DMA goes out through inline assembly — with the builtin the compiler put s_waitcnt vmcnt(0) in front of the
  // first transposed fragment read after every DMA issue, i.e. the K-strided operand forms waited for the K-tile they had just requested)
  const sw_i32x4 rsA = sw_make_rsrc(g.A, g.a_bytes), rsB = sw_make_rsrc(g.B, g.b_bytes);

  unsigned a_v[A_SLOTS];           // byte offset of this slot (valid rows) or INVALID
  bool a_ok[A_SLOTS];
  int a_k[A_SLOTS];                // k index this slot starts at inside a K-tile (tail check against kend)
  int a_py[A_SLOTS], a_px[A_SLOTS];
  unsigned a_pix[A_SLOTS];         // conv: byte offset of (pixel, logical chunk) before the tap shift
  unsigned a_sstep = 0;            // soffset advance per K-tile
#pragma unroll
  for (int i = 0; i < A_SLOTS; ++i) {
    const int L = i * NT + tid;
    const int row = L / GA::CPR, chk = (L % GA::CPR) ^ GA::swz(row);      // logical chunk fetched into physical slot L
    a_py[i] = a_px[i] = 0; a_pix[i] = 0;
    if (AMODE == OP_KCONTIG) {
      const int m = m0 + row;
      a_ok[i] = m < g.M; a_k[i] = chk * EPC;
      a_v[i] = a_ok[i] ? (unsigned)(((long)m * g.lda + kbeg + chk * EPC) * ES) : INVALID;
      a_sstep = BK * ES;
    } else if (AMODE == OP_KSTRIDED) {
      const int mm = m0 + chk * EPC;
      a_ok[i] = mm < g.M; a_k[i] = row;
      a_v[i] = a_ok[i] ? (unsigned)(((long)(kbeg + row) * g.lda + mm) * ES) : INVALID;
      a_sstep = (unsigned)((long)BK * g.lda * ES);
    } else {   // conv gather: row = output pixel
      const int m = m0 + row;
      const int hw = g.cH * g.cW;
      const int mc = m < g.M ? m : 0;
      const int pb = mc / hw, rem = mc - pb * hw;
      a_py[i] = rem / g.cW; a_px[i] = rem - a_py[i] * g.cW;
      a_ok[i] = m < g.M; a_k[i] = chk * EPC;
      a_pix[i] = (unsigned)(((((long)pb * g.cH + a_py[i]) * g.cW + a_px[i]) * g.cC + chk * EPC) * ES);
      a_v[i] = INVALID;
      a_sstep = BK * ES;
    }
  }
  int a_tap = 0, a_tt = 0;          // conv fast path: current tap and K-tiles consumed inside it
  // K rotation: inside a tap the Cin/BK channel chunks are visited starting at chunk (bm mod Cin/BK).  An NHWC row is
  // Cin*2 bytes (1 KiB for 512 channels) and a K-tile touches 128 bytes of each row: with every workgroup on the same chunk
  // all rows of all tiles sit on the same 4 of the 16 L2 channels of the XCD.  Rotating by the M-tile index spreads the
  // concurrent workgroups over all 128-byte columns (A and B use the same order, so the sum is unchanged up to rounding).
  const int k_rot = (a_fast && g.krot) ? bm % tiles_per_tap : 0;
  auto a_chunk = [&]() { const int c = a_tt + k_rot; return c >= tiles_per_tap ? c - tiles_per_tap : c; };
  auto conv_a_set_tap = [&](int tap) {
    const int ty = tap / 3, tx = tap - ty * 3;
    const int dy = (ty - 1) * g.cDil, dx = (tx - 1) * g.cDil;
    const int shift = (dy * g.cW + dx) * g.cC * (int)ES;                 // wave-uniform byte shift of this tap
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
      const int yy = a_py[i] + dy, xx = a_px[i] + dx;
      const bool ok = a_ok[i] && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
      a_v[i] = ok ? a_pix[i] + (unsigned)shift : INVALID;
    }
  };
  if (a_fast) {
    a_tap = kbeg / g.cC;
    a_tt = (kbeg - a_tap * g.cC) / BK;
    conv_a_set_tap(a_tap);
  }

  unsigned b_v[B_SLOTS];
  bool b_ok[B_SLOTS];
  int b_k[B_SLOTS], bq_py[B_SLOTS], bq_px[B_SLOTS], b_dy[B_SLOTS], b_dx[B_SLOTS];
  unsigned b_sstep = 0, b_vstep = 0;
#pragma unroll
  for (int i = 0; i < B_SLOTS; ++i) {
    const int L = i * NT + tid;
    const int row = L / GB::CPR, chk = (L % GB::CPR) ^ GB::swz(row);
    bq_py[i] = bq_px[i] = b_dy[i] = b_dx[i] = 0;
    if (BMODE == OP_KCONTIG) {
      const int n = n0t + row;
      b_ok[i] = n < g.N; b_k[i] = chk * EPC;
      b_v[i] = b_ok[i] ? (unsigned)(((long)n * g.ldb + kbeg + chk * EPC) * ES) : INVALID;
      b_sstep = BK * ES;
    } else if (BMODE == OP_KSTRIDED) {
      const int nn = n0t + chk * EPC;
      b_ok[i] = nn < g.N; b_k[i] = row;
      b_v[i] = b_ok[i] ? (unsigned)(((long)(kbeg + row) * g.ldb + nn) * ES) : INVALID;
      b_sstep = (unsigned)((long)BK * g.ldb * ES);
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
      // shifted pixel = linear pixel + dy*W + dx whenever it is inside the image (checked per K-tile); 32-bit
      // wrap-around arithmetic: an out-of-image (possibly "negative") offset is never used
      b_v[i] = (unsigned)((((long)k + (long)b_dy[i] * g.cW + b_dx[i]) * g.cC + c0) * ES);
      b_vstep = (unsigned)((long)BK * g.cC * ES);
    }
  }

  const int adv_q = (BMODE == OP_CONV_B) ? BK / g.cW : 0, adv_r = (BMODE == OP_CONV_B) ? BK - adv_q * g.cW : 0;
  // DMA issue of one K-tile, split into per-slot pieces so that a deep ring can spread them between the MFMA sub-steps
  int is_kb = 0, is_kt = 0; char* is_base = nullptr; bool is_tail = false;
  auto issue_begin = [&](int kt) {          // kt = 0, 1, 2, ... in order (the slot state is incremental)
    is_kt = kt;
    is_kb = kbeg + kt * BK;
    is_base = smem + (kt % STAGES) * STAGE_BYTES;
    is_tail = is_kb + BK > kend;            // only the last K-tile of a split can be partial (wave-uniform)
    if constexpr (BM == 256 && BN == 256 && WTM == 128 && WTN == 64) is_tail = false;   // ping-pong form: K % BK == 0 (launch_auto)
  };
  auto issue_a = [&](int i) {
    unsigned voff = a_v[i], soff;
    if (AMODE == OP_CONV_A_GEN) {               // generic path (first layer, Cin padded to 8/4): per chunk tap math
      const int k0 = is_kb + a_k[i];
      bool ok = a_ok[i] && k0 < kend;
      const int kc = ok ? k0 : 0;
      const int tap = kc / g.cC, c0 = kc - tap * g.cC;
      const int ty = tap / 3, tx = tap - ty * 3;
      const int dy = (ty - 1) * g.cDil, dx = (tx - 1) * g.cDil;
      const int yy = a_py[i] + dy, xx = a_px[i] + dx;
      ok = ok && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
      voff = ok ? a_pix[i] + (unsigned)(((dy * g.cW + dx) * g.cC + c0 - a_k[i]) * (int)ES) : INVALID;
      soff = 0;
    } else if (AMODE == OP_CONV_A) {
      soff = (unsigned)a_chunk() * BK * ES;     // K = 9*Cin is a multiple of BK here: no partial K-tile
    } else {
      soff = (unsigned)is_kt * a_sstep;
      if (is_tail && !(is_kb + a_k[i] < kend)) voff = INVALID;
    }
    sw_dma16(rsA, is_base + (i * NW + wave) * 1024, voff, soff);
  };
  auto issue_b = [&](int i) {
    unsigned voff = b_v[i], soff = 0;
    if (BMODE == OP_CONV_B) {
      const int yy = bq_py[i] + b_dy[i], xx = bq_px[i] + b_dx[i];
      bool ok = b_ok[i] && yy >= 0 && yy < g.cH && xx >= 0 && xx < g.cW;
      if (is_tail) ok = ok && (is_kb + b_k[i] < kend);
      if (!ok) voff = INVALID;
      // advance this row's pixel by BK, branch-free: BK = adv_q * W + adv_r (wave-uniform), one carry each
      int px = bq_px[i] + adv_r, py = bq_py[i] + adv_q;
      const bool cx = px >= g.cW;
      px = cx ? px - g.cW : px; py = cx ? py + 1 : py;
      py = py >= g.cH ? py - g.cH : py;
      py = py >= g.cH ? py - g.cH : py;           // adv_q + 1 <= 2*H is checked by the host wrapper
      bq_px[i] = px; bq_py[i] = py;
      b_v[i] += b_vstep;
    } else {
      soff = (a_fast && BMODE == OP_KCONTIG) ? (unsigned)(a_tap * tiles_per_tap + a_chunk()) * b_sstep : (unsigned)is_kt * b_sstep;
      if (is_tail && !(is_kb + b_k[i] < kend)) voff = INVALID;
    }
    sw_dma16(rsB, is_base + GA::BYTES + (i * NW + wave) * 1024, voff, soff);
  };
  auto issue_end = [&]() {
    if (a_fast) {
      if (++a_tt == tiles_per_tap) { a_tt = 0; ++a_tap; if (a_tap < 9) conv_a_set_tap(a_tap); }
    }
  };
  // slots [lo, hi) of the combined A|B slot list
  auto issue_range = [&](int lo, int hi) {
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i)
      if (i >= lo && i < hi) issue_a(i);
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i)
      if (A_SLOTS + i >= lo && A_SLOTS + i < hi) issue_b(i);
  };
  auto stage_issue = [&](int kt) { issue_begin(kt); issue_range(0, GROUP); issue_end(); };

  typename MM::Acc acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < NACC; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int kt, bool refill) {
    const char* sa = smem + (kt % STAGES) * STAGE_BYTES;
    const char* sb = sa + GA::BYTES;
    u32x4 fa[MI], fb[NI], na[MI], nb[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = load_frag2<T, GA::KS, GA::ROW_BYTES>(sa, wm * WTM + i * TS, 0, lane);
#pragma unroll
    for (int j = 0; j < NI; ++j) fb[j] = load_frag2<T, GB::KS, GB::ROW_BYTES>(sb, wn * WTN + j * TS, 0, lane);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (PREFETCH_A && s < NSTEP - 1) {
#pragma unroll
        for (int i = 0; i < MI; ++i) na[i] = load_frag2<T, GA::KS, GA::ROW_BYTES>(sa, wm * WTM + i * TS, s + 1, lane);
      }
      if (PREFETCH_FRAGS && s < NSTEP - 1) {   // fragments of the next K sub-step are in flight while these MFMAs issue
#pragma unroll
        for (int i = 0; i < MI; ++i) na[i] = load_frag2<T, GA::KS, GA::ROW_BYTES>(sa, wm * WTM + i * TS, s + 1, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) nb[j] = load_frag2<T, GB::KS, GB::ROW_BYTES>(sb, wn * WTN + j * TS, s + 1, lane);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) MM::mma(acc[i][j], fa[i], fb[j]);
      if (refill) issue_range(s * GROUP / NSTEP, (s + 1) * GROUP / NSTEP);     // this sub-step's share of the next DMA group
      if (s < NSTEP - 1) {
        if (PREFETCH_FRAGS) {
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = na[i];
#pragma unroll
          for (int j = 0; j < NI; ++j) fb[j] = nb[j];
        } else if (PREFETCH_A) {
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = na[i];
#pragma unroll
          for (int j = 0; j < NI; ++j) fb[j] = load_frag2<T, GB::KS, GB::ROW_BYTES>(sb, wn * WTN + j * TS, s + 1, lane);
        } else {
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[i] = load_frag2<T, GA::KS, GA::ROW_BYTES>(sa, wm * WTM + i * TS, s + 1, lane);
#pragma unroll
          for (int j = 0; j < NI; ++j) fb[j] = load_frag2<T, GB::KS, GB::ROW_BYTES>(sb, wn * WTN + j * TS, s + 1, lane);
        }
      }
    }
  };

  // ---------------------------------------------------------------------------------------------------------------------
  // Ping-pong main loop (256x256 tile, EIGHT waves of 128x64, two per SIMD, up to 256 VGPRs each).
  // The 16-wave loop below runs four waves per SIMD under a 128-VGPR cap: no room to prefetch fragments, every wave alternates
  // a ds_read burst and an MFMA burst and all sixteen do so in step after each K-tile barrier (MFMA pipe 55-69 % busy).  Here
  // the waves form two groups (wave < 4 / wave >= 4 = one wave of each group on every SIMD) that run the SAME program shifted
  // by one barrier: a K-tile is 4 phases (one 64x32 quadrant of the wave tile x K = 64 = 16 MFMAs), every phase is a LOAD
  // segment (the quadrant's new fragments: 12 / 4 / 8 / 0 ds_read_b128, plus 2 of the wave's 8 LDS-DMA pieces of a later
  // K-tile) and an MFMA segment, separated by barriers — while one group issues its 16 MFMAs, the other loads.  Fragment reads
  // per K-tile drop from 256 to 192 KiB as well (128x64 wave tiles; the B fragments of both column halves stay in registers).
  // LDS-DMA schedule (2 buffers): the pieces of tile t+2 go out in phase 3 of tile t (its buffer was last read in phase 2: the
  // second group drains those reads before the barrier) and phases 0-2 of tile t+1; each wave waits for them with a counted
  // vmcnt at the end of tile t+1, one barrier before the first read.
  constexpr bool PP = (BM == 256 && BN == 256 && WTM == 128 && WTN == 64 && STAGES == 2 && sizeof(T) == 2 && (TS == 16 || TS == 32) &&
                       AMODE <= OP_KSTRIDED && BMODE <= OP_KSTRIDED);
  constexpr int QI = 64 / TS, QJ = 32 / TS;             // sub-tiles of a 64x32 quadrant (16x16x32: 4 x 2 x 2 K sub-steps; 32x32x16: 2 x 1 x 4)
  if constexpr (PP) {
    const int grp = wm;                                   // 0: waves 0-3, 1: waves 4-7
    auto dma_pair = [&](int tile, int j) {
      if (tile < nt) { issue_begin(tile); issue_range(2 * j, 2 * j + 2); }
    };
    auto tile_wait = [&](int kt) {                        // everything of tile kt+1 landed; the first pair of kt+2 may fly on
      if (kt + 2 < nt) wait_vmcnt<2>(); else wait_vmcnt<0>();
    };
    if (nt > 0) { issue_begin(0); issue_range(0, GROUP); }
    dma_pair(1, 0);
    tile_wait(-1);
    __builtin_amdgcn_s_barrier();
#ifndef SW_PP_NOSTAGGER
    if (grp == 1) __builtin_amdgcn_s_barrier();           // the stagger: group 1 runs one segment behind
#endif
    u32x4 fa[QI][NSTEP], fb[2][QJ][NSTEP];
    for (int kt = 0; kt < nt; ++kt) {
      const char* sa = smem + (kt & 1) * STAGE_BYTES;
      const char* sb = sa + GA::BYTES;
      auto phase = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        constexpr int mq = p >> 1, nq = (p == 1 || p == 2) ? 1 : 0;       // quadrant order (0,0) (0,1) (1,1) (1,0)
        // ---- LOAD segment
        __builtin_amdgcn_sched_barrier(0);
        if (p == 0 || p == 2) {
#pragma unroll
          for (int ii = 0; ii < QI; ++ii)
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) fa[ii][st] = load_frag2<T, GA::KS, GA::ROW_BYTES>(sa, wm * WTM + (mq * QI + ii) * TS, st, lane);
        }
        if (p == 0 || p == 1) {
#pragma unroll
          for (int jj = 0; jj < QJ; ++jj)
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) fb[nq][jj][st] = load_frag2<T, GB::KS, GB::ROW_BYTES>(sb, wn * WTN + (nq * QJ + jj) * TS, st, lane);
        }
        if (p == 3) dma_pair(kt + 2, 0); else dma_pair(kt + 1, p + 1);
        if (grp == 1 && p == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // tile kt's last reads done before its buffer is refilled
        if (grp == 1 && p == 3) tile_wait(kt);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA segment
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef SW_PP_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int st = 0; st < NSTEP; ++st)
#pragma unroll
          for (int ii = 0; ii < QI; ++ii)
#pragma unroll
            for (int jj = 0; jj < QJ; ++jj) MM::mma(acc[mq * QI + ii][nq * QJ + jj], fa[ii][st], fb[nq][jj][st]);
#ifndef SW_PP_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (grp == 0 && p == 3) tile_wait(kt);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      phase(IntC<0>{}); phase(IntC<1>{}); phase(IntC<2>{}); phase(IntC<3>{});
    }
#ifndef SW_PP_NOSTAGGER
    if (grp == 0) __builtin_amdgcn_s_barrier();           // both groups executed the same number of barriers
#endif
  } else {
  // prologue: STAGES-1 K-tiles in flight
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t)
    if (t < nt) stage_issue(t);

  for (int kt = 0; kt < nt; ++kt) {
    // tile kt must have landed: groups kt .. min(kt+STAGES-2, nt-1) are outstanding
    if (kt + STAGES - 2 < nt) wait_vmcnt<GROUP * (STAGES - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();           // everybody's DMA for tile kt landed; everybody finished computing tile kt-1
    const bool refill = kt + STAGES - 1 < nt;                    // refill the stage tile kt-1 lived in
    if (STAGES >= 3) {                                           // deep ring: spread the DMA issue over the MFMA sub-steps
      if (refill) issue_begin(kt + STAGES - 1);
      compute(kt, refill);
      if (refill) issue_end();
    } else {                                                     // 2 stages: the DMA needs the whole tile time to land
      if (refill) stage_issue(kt + STAGES - 1);                  // (measured: interleaving costs 10-25 % here)
      compute(kt, false);
    }
  }
  }

  // ---- epilogue (C/D element map: Mma<T>::row / col)
  const int r = MM::col(lane);
  unsigned int vmax = 0u;                                    // max |stored value| as IEEE bits (NaN / Inf survive)
  const unsigned long long drop_base = g.drop_seed_mixed + g.drop_offset + ((g.drop_p > 0.f && g.drop_offset_dev) ? *g.drop_offset_dev : 0ull);
  if constexpr (PP) {
    // Row-wise epilogue through LDS (the ring is free now).  128 accumulator registers plus a fully unrolled 128-element
    // epilogue do not fit 256 VGPRs (the compiler parked every accumulator in scratch: ~55 us per tile); here a wave moves its
    // tile 32 rows at a time into a private 32 x 68 f32 LDS image and walks it in row pieces: a lane owns 4 (f32 output) or 8
    // (bf16 output) consecutive columns of one row, applies bias / ReLU / dropout / mask to them and stores 16 bytes.
    constexpr int LDW = WTN + 4;                                   // padded pitch: the 4 row groups of a store hit different banks
    float* wt = (float*)smem + wave * (32 * LDW);
    // fused SGD: the updated weights of a quarter (32 rows x WTN columns) as bf16, column-major, for the TRANSPOSED compute copy
    constexpr int TP = 40;                                         // halfwords per column (32 rows + pad; 80 B: 16-byte column starts)
    unsigned short* tT = (unsigned short*)((float*)smem + NW * (32 * LDW)) + wave * (WTN * TP);
    static_assert((long)NW * (32 * LDW) * 4 + (long)NW * WTN * TP * 2 <= (long)STAGES * STAGE_BYTES, "epilogue staging fits the ring");
    float sgd_lr = g.sgd_lr, sgd_wd = g.sgd_wd;
    if (g.sgd_param && g.sgd_hyper) { sgd_lr = g.sgd_hyper[0]; sgd_wd = g.sgd_hyper[1]; }
    __syncthreads();                                               // every wave is done reading the last K-tile
    const bool vec_ok = g.staged_out;                              // host: 16-byte row pieces are aligned and inside the row pitch
    auto quarter = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      // fused SGD: the quarter's parameter / momentum pieces are requested BEFORE the accumulators go through LDS — sixteen 16-byte
      // loads in flight per lane instead of two per piece (alone: the epilogue is a one-workgroup-per-CU streaming phase, latency bound)
      constexpr int NPC = (32 * (WTN / 4)) / 64;                    // row pieces of 4 columns per lane and quarter
      f32x4 pw[NPC], pm[NPC];
      if (g.sgd_param) {
        const int nbq = n0t + wn * WTN + (lane % (WTN / 4)) * 4;
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
          const int mq = m0 + wm * WTM + q * 32 + (it * 64 + lane) / (WTN / 4);
          const bool okq = mq < g.M && nbq < g.N;
          const long po = (long)mq * g.sgd_ld + nbq;
          pw[it] = okq ? *(const f32x4*)(g.sgd_param + po) : f32x4{0.f, 0.f, 0.f, 0.f};
          pm[it] = (okq && !g.sgd_first) ? *(const f32x4*)(g.sgd_mom + po) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int ii = 0; ii < 32 / TS; ++ii)                          // the sub-tiles of this 32-row quarter (two of 16 rows / one of 32)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < NACC; ++e) wt[(ii * TS + MM::row(lane, e)) * LDW + j * TS + r] = acc[(32 / TS) * q + ii][j][e];
      __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): own LDS writes before own reads (same wave)
      // a lane owns CP consecutive columns of one row: 8 for bf16 rows (16-byte stores), 4 for f32 rows.  CP and "no
      // per-element work" are compile-time cases: left as run-time tests inside the element loop, hipcc keeps that loop rolled
      // and selects v[t] through compare chains (~60 instructions per element: 50 us per tile)
      auto pieces = [&](auto cpc, auto plainc) {
        constexpr int CP = decltype(cpc)::value;
        constexpr bool PLAIN = decltype(plainc)::value != 0;
        constexpr int PPR = WTN / CP;
        // 64 % PPR == 0: a lane keeps the SAME CP columns for every row piece it handles — its bias values are loaded once per
        // quarter, not once per element, and the ReLU-mask reference of a full 16-byte-aligned piece is ONE load (fc6's data
        // gradient 1.258 -> 1.23 ms inside the step; tools/probes/fc7_probe.py: a bias / ReLU / mask epilogue costs 10 us per round of
        // 256x256 tiles over the plain one)
        const int c0 = (lane % PPR) * CP;
        const int nb = n0t + wn * WTN + c0;
        float bcol[CP];
#pragma unroll
        for (int t = 0; t < CP; ++t) bcol[t] = (!PLAIN && g.bias && nb + t < g.N) ? g.bias[nb + t] : 0.f;
        const bool ref_vec = !PLAIN && g.ref && g.ref_bf16 && CP == 8 && (g.ldr % 8) == 0 && ((((uintptr_t)g.ref) & 15) == 0);
#pragma unroll
        for (int it = 0; it < (32 * PPR) / 64; ++it) {
          const int pc = it * 64 + lane;
          const int row = pc / PPR;
          const int m = m0 + wm * WTM + q * 32 + row;
          if (m >= g.M || nb >= g.N) continue;
          float v[CP];
          *(f32x4*)&v[0] = *(const f32x4*)&wt[row * LDW + c0];
          if (CP == 8) *(f32x4*)&v[CP - 4] = *(const f32x4*)&wt[row * LDW + c0 + 4];
          const bool full = nb + CP <= g.N;
          u32x4 refw = {0u, 0u, 0u, 0u};
          const bool ref_piece = ref_vec && full;
          if (ref_piece) refw = *(const u32x4*)((const unsigned short*)g.ref + (long)m * g.ldr + nb);
#pragma unroll
          for (int t = 0; t < CP; ++t) {
            const int n = nb + t;
            const bool ok = n < g.N;
            float x = v[t];
            if (!PLAIN) {
              x += bcol[t];
              if (g.res && ok) x += g.res_bf16 ? bf16_bits_to_f32(((const unsigned short*)g.res)[(long)m * g.ldres + n])
                                               : ((const float*)g.res)[(long)m * g.ldres + n];
              if (g.relu) x = fmaxf(x, 0.f);
              if (g.drop && ok) x = g.drop[(long)m * g.ldd + n] ? x * g.drop_scale : 0.f;
              if (g.drop_p > 0.f && ok) {                          // identical Bernoulli stream to dropout_mask_kernel (elementwise.hip)
                unsigned long long z = drop_base + (unsigned long long)((long)m * g.N + n);
                z += 0x9E3779B97F4A7C15ull;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
                const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
                x = u >= g.drop_p ? x * g.drop_scale : 0.f;
              }
              if (g.ref && ok) {
                float rv;
                if (ref_piece) rv = __uint_as_float((t & 1) ? (refw[(t >> 1) & 3] & 0xFFFF0000u) : (refw[(t >> 1) & 3] << 16));
                else rv = g.ref_bf16 ? bf16_bits_to_f32(((const unsigned short*)g.ref)[(long)m * g.ldr + n])
                                     : ((const float*)g.ref)[(long)m * g.ldr + n];
                x = rv > 0.f ? x * g.ref_scale : 0.f;
              }
            }
            if (ok) vmax = max(vmax, absbits(x));
            v[t] = x;
          }
          const long o = (long)m * g.ldc + nb;
          if (CP == 4 && PLAIN && g.sgd_param) {
            // ---- fused SGD (sgd_tile_t_kernel's arithmetic, elementwise.hip: identical bits): host guarantees full 16-byte pieces
            const long po = (long)m * g.sgd_ld + nb;
            const f32x4 w4 = pw[it * CP / 4 < NPC ? it * CP / 4 : 0];     // (CP == 4 here: piece `it` of the quarter)
            f32x4 m4 = pm[it * CP / 4 < NPC ? it * CP / 4 : 0];
            f32x4 wn;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float dd = v[t] * g.sgd_gscale + sgd_wd * w4[t];
              m4[t] = g.sgd_first ? dd : g.sgd_momentum * m4[t] + dd;
              wn[t] = w4[t] - sgd_lr * m4[t];
            }
            *(f32x4*)(g.sgd_mom + po) = m4;
            *(f32x4*)(g.sgd_param + po) = wn;
            const unsigned b01 = (unsigned)f32_to_bf16_bits(wn[0]) | ((unsigned)f32_to_bf16_bits(wn[1]) << 16);
            const unsigned b23 = (unsigned)f32_to_bf16_bits(wn[2]) | ((unsigned)f32_to_bf16_bits(wn[3]) << 16);
            if (g.sgd_st0) *(u32x2*)(g.sgd_st0 + (long)m * g.sgd_ld0 + nb) = u32x2{b01, b23};
            if (g.sgd_st1) {
              tT[(c0 + 0) * TP + row] = (unsigned short)(b01 & 0xFFFFu); tT[(c0 + 1) * TP + row] = (unsigned short)(b01 >> 16);
              tT[(c0 + 2) * TP + row] = (unsigned short)(b23 & 0xFFFFu); tT[(c0 + 3) * TP + row] = (unsigned short)(b23 >> 16);
            }
            continue;
          }
          if (CP == 4) {
            if (g.atomic && g.slab_stride <= 0) {
#pragma unroll
              for (int t = 0; t < 4; ++t) if (nb + t < g.N) atomicAdd((float*)g.C + o + t, v[t]);
            } else {                                               // f32 rows (plain or K-split slab)
              float* dst = (float*)g.C + o + (g.slab_stride > 0 ? (long)zsplit * g.slab_stride : 0);
              // nontemporal: the 33-68 MB burst a round of tiles writes retires faster streamed past the write-back L2s, and the next
              // tiles' loads queue behind it on the in-order vmcnt (fc6 data gradient 1.36 -> 1.25 ms; -D SW_EP_PLAIN_STORES for A/B).
              // Unconditional on purpose: `if (flag) nontemporal else plain` is merged into one plain store by the optimiser, and
              // arms kept apart by an asm barrier stop the pieces' LDS reads and stores from overlapping (no gain left)
#ifndef SW_EP_PLAIN_STORES
              if (full && vec_ok) __builtin_nontemporal_store(*(const f32x4*)&v[0], (f32x4*)dst);
#else
              if (full && vec_ok) *(f32x4*)dst = *(const f32x4*)&v[0];
#endif
              else {
#pragma unroll
                for (int t = 0; t < 4; ++t) if (nb + t < g.N) dst[t] = v[t];
              }
            }
          } else {                                                 // bf16 rows
            unsigned short* dst = (unsigned short*)g.C + o;
            if (full && vec_ok) {
              u32x4 w;
#pragma unroll
              for (int t = 0; t < 4; ++t) w[t] = (unsigned)f32_to_bf16_bits(v[2 * t]) | ((unsigned)f32_to_bf16_bits(v[2 * t + 1]) << 16);
#ifndef SW_EP_PLAIN_STORES
              __builtin_nontemporal_store(w, (u32x4*)dst);
#else
              *(u32x4*)dst = w;
#endif
            } else {
#pragma unroll
              for (int t = 0; t < CP; ++t) if (nb + t < g.N) dst[t] = f32_to_bf16_bits(v[t]);
            }
          }
        }
      };
      const bool plain_ep = !g.bias && !g.relu && !g.drop && !(g.drop_p > 0.f) && !g.ref && !g.res;
      const bool bf16_rows = g.out_bf16 && g.slab_stride <= 0 && !g.atomic;
      if (bf16_rows) { if (plain_ep) pieces(IntC<8>{}, IntC<1>{}); else pieces(IntC<8>{}, IntC<0>{}); }
      else { if (plain_ep) pieces(IntC<4>{}, IntC<1>{}); else pieces(IntC<4>{}, IntC<0>{}); }
      if (g.sgd_param && g.sgd_st1) {
        // the quarter's updated weights, transposed: column c of the tile = row (n0 + c) of the transposed copy, its 32 rows one 64-byte
        // run; lane -> (column lane >> 2 of 16 per pass, 8 rows = 16 bytes)
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // own LDS writes before own reads
        const int part = lane & 3;
        const int mrow = m0 + wm * WTM + q * 32 + part * 8;
#pragma unroll
        for (int ps = 0; ps < WTN / 16; ++ps) {
          const int c = ps * 16 + (lane >> 2);
          const int n = n0t + wn * WTN + c;
          if (n < g.N && mrow < g.M)
            *(u32x4*)(g.sgd_st1 + (long)n * g.sgd_ld1 + mrow) = *(const u32x4*)(tT + c * TP + part * 8);
        }
      }
    };
    quarter(IntC<0>{}); quarter(IntC<1>{}); quarter(IntC<2>{}); quarter(IntC<3>{});
  } else {
  // bf16 outputs go through LDS (the ring is free now): the accumulator layout gives a lane one 2-byte element per
  // store (64-byte segments); staged, every lane stores 16 contiguous bytes of one output row (8x fewer, full-line stores)
  const bool staged = g.staged_out;                          // host: bf16 out, plain store, N % 8 == 0, ldc % 8 == 0
  unsigned short* stile = (unsigned short*)smem + wave * (WTM * WTN);
  if (staged) __syncthreads();                               // every wave is done reading the last K-tile
  // residual of a staged bf16 tile: fetched with the store phase's 16-byte row pieces INTO the staging tile, read per element from
  // LDS and overwritten in place by the result (per-element 2-byte global loads cost the memory-bound 1x1-convolution GEMMs
  // 0.7 ms per Stage-3 iteration); the wave's own LDS accesses execute in order: no barrier
  const bool res_staged = staged && g.res && g.res_bf16 && (g.ldres % 8) == 0 && ((((uintptr_t)g.res) & 15) == 0);
  // the same for the ReLU-mask reference when there is no residual (a data gradient masked by its consumer's input)
  const bool ref_ok = staged && g.ref && g.ref_bf16 && (g.ldr % 8) == 0 && ((((uintptr_t)g.ref) & 15) == 0);
  const bool ref_staged = ref_ok && !g.res;
  // residual AND mask (a bottleneck's block-input gradient: shortcut gradient added, then masked by the block input's ReLU): the
  // mask reference goes to a second staging tile when the ring has room for two (the 8-wave tiles: 2 x 32 KiB)
  constexpr bool TWO_TILES = 2L * NW * WTM * WTN * 2 <= (long)STAGES * STAGE_BYTES;
  const bool ref_staged2 = TWO_TILES && ref_ok && res_staged;
  unsigned short* stile2 = stile + NW * (WTM * WTN);
  auto fetch_tile = [&](const unsigned short* src, const long lds_, unsigned short* dst) {
    constexpr int CPRW = WTN / 8;
#pragma unroll
    for (int q = 0; q < (WTM * CPRW) / 64; ++q) {
      const int idx = q * 64 + lane, lrow = idx / CPRW, ch = idx % CPRW;
      const int m = m0 + wm * WTM + lrow, n = n0t + wn * WTN + ch * 8;
      u32x4 rv = {0u, 0u, 0u, 0u};
      if (m < g.M && n < g.N) rv = *(const u32x4*)(src + (long)m * lds_ + n);
      *(u32x4*)(dst + lrow * WTN + ch * 8) = rv;
    }
  };
  if (res_staged || ref_staged) {
    if (res_staged) fetch_tile((const unsigned short*)g.res, g.ldres, stile);
    else fetch_tile((const unsigned short*)g.ref, g.ldr, stile);
    if (ref_staged2) fetch_tile((const unsigned short*)g.ref, g.ldr, stile2);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (also keeps the compiler from moving the 2-byte reads above these stores)
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0t + wn * WTN + j * TS + r;
      const bool nok = n < g.N;
      const float bcol = (g.bias && nok) ? g.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < NACC; ++e) {
        const int lrow = i * TS + MM::row(lane, e);
        const int m = m0 + wm * WTM + lrow;
        const bool ok = nok && m < g.M;
        float v = acc[i][j][e] + bcol;
        if (res_staged) v += bf16_bits_to_f32(stile[lrow * WTN + j * TS + r]);
        else if (g.res && ok) v += g.res_bf16 ? bf16_bits_to_f32(((const unsigned short*)g.res)[(long)m * g.ldres + n])
                                              : ((const float*)g.res)[(long)m * g.ldres + n];
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.drop && ok) v = g.drop[(long)m * g.ldd + n] ? v * g.drop_scale : 0.f;
        if (g.drop_p > 0.f && ok) {                        // identical Bernoulli stream to dropout_mask_kernel (elementwise.hip)
          unsigned long long z = drop_base + (unsigned long long)((long)m * g.N + n);
          z += 0x9E3779B97F4A7C15ull;
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
          z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
          z ^= z >> 31;
          const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
          v = u >= g.drop_p ? v * g.drop_scale : 0.f;
        }
        if (ref_staged) {
          v = bf16_bits_to_f32(stile[lrow * WTN + j * TS + r]) > 0.f ? v * g.ref_scale : 0.f;
        } else if (ref_staged2) {
          v = bf16_bits_to_f32(stile2[lrow * WTN + j * TS + r]) > 0.f ? v * g.ref_scale : 0.f;
        } else if (g.ref && ok) {
          const float rv = g.ref_bf16 ? bf16_bits_to_f32(((const unsigned short*)g.ref)[(long)m * g.ldr + n])
                                      : ((const float*)g.ref)[(long)m * g.ldr + n];
          v = rv > 0.f ? v * g.ref_scale : 0.f;
        }
        if (ok) vmax = max(vmax, absbits(v));
        if (staged) { stile[lrow * WTN + j * TS + r] = f32_to_bf16_bits(v); continue; }
        if (!ok) continue;
        const long o = (long)m * g.ldc + n;
#ifndef SW_EP_PLAIN_STORES
        if (g.slab_stride > 0) __builtin_nontemporal_store(v, (float*)g.C + o + (long)zsplit * g.slab_stride);   // slabs: written once, folded once
        else if (g.atomic) atomicAdd((float*)g.C + o, v);
        else if (g.out_bf16) __builtin_nontemporal_store(f32_to_bf16_bits(v), (unsigned short*)g.C + o);
        else __builtin_nontemporal_store(v, (float*)g.C + o);
#else
        if (g.slab_stride > 0) ((float*)g.C)[o + (long)zsplit * g.slab_stride] = v;
        else if (g.atomic) atomicAdd((float*)g.C + o, v);
        else if (g.out_bf16) ((unsigned short*)g.C)[o] = f32_to_bf16_bits(v);
        else ((float*)g.C)[o] = v;
#endif
      }
    }
  if (staged) {
    // the wave re-reads its own WTM x WTN tile: lane -> (row, 16-byte chunk); same-wave LDS write->read needs no barrier
    constexpr int CPRW = WTN / 8;                            // chunks per tile row
#pragma unroll
    for (int q = 0; q < (WTM * CPRW) / 64; ++q) {
      const int idx = q * 64 + lane, lrow = idx / CPRW, ch = idx % CPRW;
      const int m = m0 + wm * WTM + lrow, n = n0t + wn * WTN + ch * 8;
      if (m < g.M && n < g.N) {
#ifndef SW_EP_PLAIN_STORES
        __builtin_nontemporal_store(*(const u32x4*)(stile + lrow * WTN + ch * 8), (u32x4*)((unsigned short*)g.C + (long)m * g.ldc + n));
#else
        *(u32x4*)((unsigned short*)g.C + (long)m * g.ldc + n) = *(const u32x4*)(stile + lrow * WTN + ch * 8);
#endif
      }
    }
  }
  }
  if (g.absmax) {
    vmax = wave_reduce_max_u32(vmax);
    if (lane == 0) atomicMax((unsigned int*)g.absmax, vmax);
  }
}

template <typename T, int AMODE, int BMODE, int BM, int BN, int STAGES, int WTM, int WTN>
__global__ __launch_bounds__((BM / WTM) * (BN / WTN) * 64, 1) void gemm2_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];                     // [STAGES][A | B]
  // Workgroup -> (tile, K-split).  The dispatcher deals workgroups round-robin to the 8 XCDs in flat launch order and
  // every XCD has a private 4 MiB L2:
  //  * no split-K: XCD x takes a contiguous run of 8x8-tile patches, so its 32 CUs share A rows / B columns in L2;
  //  * split-K (few tiles, long K: the weight gradients): the grid holds valid tiles only and XCD x takes a contiguous
  //    run of the split-major work list, i.e. ONE or two K-ranges for all tiles — its L2 then holds just that K-range
  //    of both operands.  (With every XCD walking all of K the conv3 wgrad fetched 6x its operands from HBM; profiles/.)
  const int vgrid = (g.vgrid > 0 && gridDim.z == 1) ? g.vgrid : (int)gridDim.x;
  for (int vbid = blockIdx.x; vbid < vgrid; vbid += gridDim.x) {
  int bm, bn, zsplit = 0;
  if (gridDim.z > 1) {
    const int ntv = gridDim.x, tot = ntv * (int)gridDim.z;
    const int f = blockIdx.z * ntv + blockIdx.x;
    const int x = f & 7, j = f >> 3;
    const int w = x * (tot >> 3) + min(x, tot & 7) + j;
    zsplit = w / ntv;
    const int t = w - zsplit * ntv;
    bm = t % g.tiles_m; bn = t / g.tiles_m;
  } else {
    const int nwg = vgrid;
    const int bid = vbid;
    const int swz = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int patch = swz >> 6, within = swz & 63;
    const int pml = g.patch_m_log2;
    bm = ((patch % g.patches_m) << pml) + (within & ((1 << pml) - 1));
    bn = ((patch / g.patches_m) << (6 - pml)) + (within >> pml);
    if (bm >= g.tiles_m || bn >= g.tiles_n) continue;
  }
  gemm2_tile<T, AMODE, BMODE, BM, BN, STAGES, WTM, WTN>(g, bm, bn, zsplit, smem);
  if (vbid + (int)gridDim.x < vgrid) __syncthreads();       // persistent form: the staged epilogue is done with the LDS ring
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped conv weight gradients: ALL trainable conv layers x ALL view batches of a backward pass as ONE launch of resident
// workgroups (one per CU) walking a list of (problem, K-split, 256x256 tile) items.
// Why: launched one by one, a conv4 / conv5 weight gradient has 36 tiles of 256x256 (or 144 of 128x128): to fill 256 CUs it
// was cut into 3-28 K-splits of 128x128 tiles, whose 64 FLOP per staged byte keep the XCD L2s, not the MFMA pipe, busy
// (0.17 of the MFMA peak, 18 launches + 9 slab folds per step).  All layers together are ~600-1200 items of 256x256 x ~64
// K-tiles: 128 FLOP per staged byte, every CU busy for the whole launch, few slabs.
struct GroupedProblem {
  const void* A; const void* B; float* C;        // dy [pixels][Cout], x NHWC, slabs [nsplit][Cout][9*Cin]
  int M, N, K, cH, cW, cC, cDil, k_per_split, nsplit, tiles_m, tiles_n;
  unsigned a_bytes, b_bytes;
  int lda, ldb;                                  // plain problems (both operands K-strided: sw_gemm_kk_grouped); conv: lda = M, ldb unused
};
constexpr int GROUPED_MAX = 40;
struct GroupedArgs {
  int n_problems, n_items;
  int first_item[GROUPED_MAX + 1];               // prefix sums of the problems' item counts (items = nsplit * tiles)
  GroupedProblem p[GROUPED_MAX];
};

template <typename T, int BMODE = OP_CONV_B>
__global__ __launch_bounds__(1024, 1) void gemm2_grouped_kernel(GroupedArgs ga) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // resident workgroup b serves the virtual ids b, b + G, b + 2G, ... (G = gridDim.x, a multiple of 8: all on one XCD).  Inside a
  // round of G ids, XCD x (= id & 7) takes the contiguous run [x * G/8, (x+1) * G/8) of the locality-ordered item list
  // (problem-major, K-split-major, tile rows fastest): its CUs share one problem's K-range — A rows / B columns — in L2.
  const int G = gridDim.x;
  for (int vbid = blockIdx.x; vbid < ga.n_items; vbid += G) {
    const int round = vbid / G, within = vbid - round * G;
    const int in_round = min(G, ga.n_items - round * G);          // the last round may be short: keep the map a bijection
    const int x = within & 7, j = within >> 3;
    const int per = in_round >> 3, extra = in_round & 7;
    if (j >= per + (x < extra ? 1 : 0)) continue;
    const int t = round * G + x * per + min(x, extra) + j;
    int pi = 0;
    while (pi + 1 < ga.n_problems && t >= ga.first_item[pi + 1]) ++pi;
    const GroupedProblem& P = ga.p[pi];
    const int local = t - ga.first_item[pi];
    const int ntile = P.tiles_m * P.tiles_n;
    const int zsplit = local / ntile, tl = local - zsplit * ntile;
    GemmArgs g = {};
    g.A = P.A; g.B = P.B; g.C = P.C; g.M = P.M; g.N = P.N; g.K = P.K; g.lda = P.lda; g.ldb = P.ldb; g.ldc = P.N;
    g.cH = P.cH; g.cW = P.cW; g.cC = P.cC; g.cDil = P.cDil; g.k_per_split = P.k_per_split;
    g.tiles_m = P.tiles_m; g.tiles_n = P.tiles_n; g.a_bytes = P.a_bytes; g.b_bytes = P.b_bytes;
    g.slab_stride = (long)P.M * P.N; g.drop_scale = 1.f; g.ref_scale = 1.f;
    gemm2_tile<T, OP_KSTRIDED, BMODE, 256, 256, 2, 64, 64>(g, tl % P.tiles_m, tl / P.tiles_m, zsplit, smem);
    __syncthreads();                                              // the next item restarts the LDS ring
  }
}


template <typename T, int AMODE, int BMODE, int BM, int BN, int STAGES, int WTM = 64, int WTN = 64>
int launch2(GemmArgs& g, int splitk, hipStream_t stream) {
  constexpr int BK = GT<T>::BK;
  using GA = Geom2<T, AMODE, BM>;
  using GB = Geom2<T, BMODE, BN>;
  constexpr int LDS = STAGES * (GA::BYTES + GB::BYTES);
  constexpr int NT = (BM / WTM) * (BN / WTN) * 64;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  // XCD patch shape: 64 tiles as pm x pn.  8 x 8 (least operand traffic per tile) unless one side of the tile grid is shorter than
  // 8: then the patch spans that side and grows along the other — the two tile columns of a ResNet 1x1 convolution (N = 256) under
  // 8 x 8 patches launched 4 workgroups for every one that had a tile.
  int pml = 3;
  if (g.tiles_n < 8) { int l = 0; while ((1 << l) < g.tiles_n) ++l; pml = 6 - l; }
  else if (g.tiles_m < 8) { int l = 0; while ((1 << l) < g.tiles_m) ++l; pml = l; }
  g.patch_m_log2 = pml;
  g.patches_m = (g.tiles_m + (1 << pml) - 1) >> pml;
  const int patches_n = (g.tiles_n + (64 >> pml) - 1) / (64 >> pml);
  if (splitk < 1) splitk = 1;
  int kps = (g.K + splitk - 1) / splitk;
  kps = ((kps + BK - 1) / BK) * BK;
  g.k_per_split = kps;
  splitk = (g.K + kps - 1) / kps;
  if (splitk > 1 && !g.atomic && g.slab_stride <= 0) return -2;
  g.staged_out = (g.out_bf16 && !g.atomic && g.slab_stride <= 0 && (g.N % 8) == 0 && (g.ldc % 8) == 0 &&
                  (((uintptr_t)g.C) & 15) == 0 && (long)NT / 64 * WTM * WTN * 2 <= (long)LDS) ? 1 : 0;
  if (BM == 256 && BN == 256 && WTM == 128 && WTN == 64 && sizeof(T) == 2)      // ping-pong form: "16-byte row pieces are legal"
    g.staged_out = ((((uintptr_t)g.C) & 15) == 0 && (g.ldc % (g.out_bf16 && g.slab_stride <= 0 ? 8 : 4)) == 0 &&
                    (g.slab_stride % 4) == 0) ? 1 : 0;
  dim3 grid(splitk > 1 ? g.tiles_m * g.tiles_n : g.patches_m * patches_n * 64, 1, splitk), block(NT);
  // Persistent form for the one-workgroup-per-CU tile (256x256, 128 KiB LDS) when there is more than one round of tiles: one
  // workgroup per CU walks the tile list (same XCD: virtual id = resident id + k * CUs keeps id & 7) instead of being
  // re-dispatched per tile — fc6's data gradient (12.25 rounds of 64 K-tiles) -2 %, the step -0.08 ms.
  {
    static const char* ps = getenv("SW_GEMM_PERSIST");          // development switch: resident workgroups, "0" = off
    static int ncu = 0;
    if (!ncu) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
      else ncu = 256;
    }
    const int np = ps ? atoi(ps) : ((ncu % 8) == 0 ? ncu : 0);
    if (np > 0 && splitk == 1 && BM == 256 && BN == 256 && (int)grid.x > np) { g.vgrid = (int)grid.x; grid.x = np; }
  }
  auto kern = gemm2_kernel<T, AMODE, BMODE, BM, BN, STAGES, WTM, WTN>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, grid, block, LDS, stream, g);
  SW_CHECK_LAUNCH();
  return 0;
}

// Tile choice (measured, tools/gemm_bench.py):
//   * >= 200 tiles of 256x256 and N > 128 (the FC layers): 256x256, 16 waves of 64x64, 2 stages (128 KiB LDS).
//     LDS-DMA instructions per MFMA and bytes staged per FLOP are lowest here (0.25 / 128 FLOP per byte).
//   * everything else (all convolutions at batch 2, split-K weight gradients, predictor GEMMs): 128x128, 8 waves of
//     32x64, 2 stages = 64 KiB LDS so that TWO workgroups (also of two different kernels on two streams) share a CU.
//   * N <= 64 (conv1_1 / conv1_2 at full resolution): 256x64, 8 waves of 32x64, 2 stages = 80 KiB; a 128-wide N tile
//     would spend half of its MFMAs on zero columns.
//     With ~250 tiles per conv4/conv5 launch, wave-level parallelism beat deeper prefetch: 2 stages == 4 stages in time.
template <typename T, int AMODE, int BMODE>
int launch_auto(GemmArgs& g, int splitk, hipStream_t s) {
  static const char* v = getenv("SW_GEMM_V");               // development switch (tile override)
  const long sk = splitk < 1 ? 1 : splitk;
  auto tiles = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * sk; };
  // the 256x256 tile pays off once the K loop is long enough to hide its prologue / epilogue: the 1x1 convolutions of a ResNet
  // (121 600 pixels x 256 channels, K = 64 .. 256: memory bound) ran 4x slower on it than on the 128x128 tile, two workgroups per
  // CU (244 vs 57 us; tools/probes/gemm_1x1_probe.py).  Every fc / conv shape of the OICR+ step has K >= 1152.
  const bool big = (v && v[0] == '4') ? true : (v && v[0] == '8') ? false : (g.N > 128 && tiles(256, 256) >= 200 && g.K >= 1024);
  if (big) {
#ifdef SW_GEMM_TRY_4WAVE
    // experiment (tools/build_variant.sh w4 -DSW_GEMM_TRY_4WAVE, SW_GEMM_W4=1): four waves of 128x128, 256 accumulator registers per lane
    { static const bool w4 = getenv("SW_GEMM_W4") != nullptr;
      if constexpr (sizeof(T) == 2 && AMODE == OP_KCONTIG && BMODE == OP_KCONTIG)
        if (w4 && sk == 1) return launch2<T, AMODE, BMODE, 256, 256, 2, 128, 128>(g, splitk, s); }
#endif
    static const char* pp = getenv("SW_GEMM_PP");             // development switch: "0" = the 16-wave loop
    // ping-pong form (8 waves of 128x64, two staggered groups): forward / data-gradient shapes (A K-contiguous) run 4-6 % faster
    // on it; the weight gradients (both operands transposed on the fly: twice the LDS instructions per fragment in the load
    // segments) stay on the 16-wave loop, +1 % there.  SW_GEMM_PP=0 / =2: never / also for the weight gradients.
    if constexpr (sizeof(T) == 2 && AMODE <= OP_KSTRIDED && BMODE <= OP_KSTRIDED) {
      const bool want = pp ? (pp[0] == '2' || (pp[0] != '0' && AMODE == OP_KCONTIG)) : AMODE == OP_KCONTIG;
      if (want && (g.K % 64) == 0 && sk == 1) return launch2<T, AMODE, BMODE, 256, 256, 2, 128, 64>(g, splitk, s);
    }
    return launch2<T, AMODE, BMODE, 256, 256, 2>(g, splitk, s);
  }
  const bool narrow = (v && v[0] == '6') || (!v && g.N <= 64 && sk == 1);     // 64-wide outputs (conv1_x): no half-empty N tile
  if (narrow) return launch2<T, AMODE, BMODE, 256, 64, 2, 32, 64>(g, splitk, s);
  // Few tiles and a long K (a ResNet's res4 / res5 1x1 convolutions at batch 1-2: 30-60 tiles x 16-32 K-tiles): one workgroup per
  // CU at most, and its K loop runs at one memory round trip per K-tile (~1 us against 0.15 us of MFMA work) on the 2-buffer ring.
  // A 4-buffer ring keeps three K-tiles in flight (128 KiB of LDS: fine when no second workgroup would share the CU anyway).
  if constexpr (sizeof(T) == 2 && AMODE <= OP_KSTRIDED && BMODE <= OP_KSTRIDED) {
    static const char* dp = getenv("SW_GEMM_DEEP");           // development switch: "0" = off, else the largest tile count
    const long lim = dp ? atol(dp) : 256;
    if (tiles(128, 128) <= lim && g.K >= 512) return launch2<T, AMODE, BMODE, 128, 128, 4, 32, 64>(g, splitk, s);
  }
  return launch2<T, AMODE, BMODE, 128, 128, 2, 32, 64>(g, splitk, s);
}

template <typename T>
int dispatch_modes(GemmArgs& g, int amode, int bmode, int splitk, hipStream_t s) {
  if (amode == OP_KCONTIG && bmode == OP_KCONTIG) return launch_auto<T, OP_KCONTIG, OP_KCONTIG>(g, splitk, s);
  if (amode == OP_KCONTIG && bmode == OP_KSTRIDED) return launch_auto<T, OP_KCONTIG, OP_KSTRIDED>(g, splitk, s);
  if (amode == OP_KSTRIDED && bmode == OP_KSTRIDED) return launch_auto<T, OP_KSTRIDED, OP_KSTRIDED>(g, splitk, s);
  if (amode == OP_CONV_A && bmode == OP_KCONTIG) return launch_auto<T, OP_CONV_A, OP_KCONTIG>(g, splitk, s);
  if (amode == OP_CONV_A_GEN && bmode == OP_KCONTIG) return launch_auto<T, OP_CONV_A_GEN, OP_KCONTIG>(g, splitk, s);
  if (amode == OP_KSTRIDED && bmode == OP_CONV_B) return launch_auto<T, OP_KSTRIDED, OP_CONV_B>(g, splitk, s);
  return -3;
}

__global__ void scale_rows_kernel(int M, int N, float* __restrict__ C, long ldc, const float* __restrict__ row_scale) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / N;
    float* p = C + m * ldc + (i - m * N);
    *p = __fmul_rn(*p, row_scale[m]);
  }
}

// deterministic split-K: C[m][n] = sum_z ws[z][m][n], slabs added in a fixed order (pairs of four partial sums)
// row_scale (may be NULL): C[m][n] = row_scale[m] * sum — the FrozenBN fold of a weight gradient (dW = scale * dW_eff) without a pass of its own
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int M, int N, int nslab, const float* __restrict__ ws,
                                                            float* __restrict__ C, long ldc, int vec, const float* __restrict__ row_scale) {
  const long slab = (long)M * N;
  if (vec) {
    const long nv = slab >> 2;
    const int nq = N >> 2;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nv; q += (long)gridDim.x * blockDim.x) {
      const f32x4* src = (const f32x4*)ws + q;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      int z = 0;
      for (; z + 4 <= nslab; z += 4) {
        const f32x4 v0 = src[(long)z * nv], v1 = src[(long)(z + 1) * nv], v2 = src[(long)(z + 2) * nv], v3 = src[(long)(z + 3) * nv];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
      }
      for (; z < nslab; ++z) a0 += src[(long)z * nv];
      const long m = q / nq; const int n4 = (int)(q - m * nq);
      f32x4 r = (a0 + a1) + (a2 + a3);
      if (row_scale) { const float sc = row_scale[m]; r[0] = __fmul_rn(r[0], sc); r[1] = __fmul_rn(r[1], sc); r[2] = __fmul_rn(r[2], sc); r[3] = __fmul_rn(r[3], sc); }
      *(f32x4*)(C + m * ldc + n4 * 4) = r;
    }
  } else {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < slab; i += (long)gridDim.x * blockDim.x) {
      float a = 0.f;
      for (int z = 0; z < nslab; ++z) a += ws[(long)z * slab + i];
      const long m = i / N;
      C[m * ldc + (i - m * N)] = row_scale ? __fmul_rn(a, row_scale[m]) : a;
    }
  }
}

// The same ordered fold for a split-K GEMM WITH an epilogue (sw_gemm: bias / residual / ReLU / ReLU-mask reference, f32 or bf16 C):
// x = row_scale[m] * sum + bias[n] + residual[m][n]; ReLU; mask — the order of the GEMM's own epilogue.  4 columns per thread
// (N % 4 == 0, 16-byte aligned rows; checked by the host).  `res` may be C itself (f32 accumulation into a gradient that exists).
struct FoldEp {
  int M, N, nslab; const float* ws; void* C; long ldc; int out_bf16;
  const float* bias; const float* row_scale;
  const void* res; long ldres; int res_bf16;
  int relu; const void* ref; long ldr; int ref_bf16; float ref_scale;
};
__global__ __launch_bounds__(256) void splitk_fold_ep_kernel(FoldEp f) {
  const long nv = ((long)f.M * f.N) >> 2;
  const int nq = f.N >> 2;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < nv; q += (long)gridDim.x * blockDim.x) {
    const f32x4* src = (const f32x4*)f.ws + q;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int z = 0;
    for (; z + 4 <= f.nslab; z += 4) {
      const f32x4 v0 = src[(long)z * nv], v1 = src[(long)(z + 1) * nv], v2 = src[(long)(z + 2) * nv], v3 = src[(long)(z + 3) * nv];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; z < f.nslab; ++z) a0 += src[(long)z * nv];
    const long m = q / nq; const int n = (int)(q - m * nq) * 4;
    const f32x4 r = (a0 + a1) + (a2 + a3);
    float x[4] = {r[0], r[1], r[2], r[3]};
    if (f.row_scale) { const float sc = f.row_scale[m];
#pragma unroll
      for (int t = 0; t < 4; ++t) x[t] = __fmul_rn(x[t], sc); }
    if (f.bias) {
#pragma unroll
      for (int t = 0; t < 4; ++t) x[t] += f.bias[n + t]; }
    if (f.res) {
      if (f.res_bf16) { const unsigned short* p = (const unsigned short*)f.res + m * f.ldres + n;
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] += bf16_bits_to_f32(p[t]); }
      else { const f32x4 v = *(const f32x4*)((const float*)f.res + m * f.ldres + n);
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] += v[t]; }
    }
    if (f.relu) {
#pragma unroll
      for (int t = 0; t < 4; ++t) x[t] = fmaxf(x[t], 0.f); }
    if (f.ref) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float rv = f.ref_bf16 ? bf16_bits_to_f32(((const unsigned short*)f.ref)[m * f.ldr + n + t]) : ((const float*)f.ref)[m * f.ldr + n + t];
        x[t] = rv > 0.f ? x[t] * f.ref_scale : 0.f;
      }
    }
    if (f.out_bf16) {
      unsigned int w0 = (unsigned)f32_to_bf16_bits(x[0]) | ((unsigned)f32_to_bf16_bits(x[1]) << 16);
      unsigned int w1 = (unsigned)f32_to_bf16_bits(x[2]) | ((unsigned)f32_to_bf16_bits(x[3]) << 16);
      unsigned int* d = (unsigned int*)((unsigned short*)f.C + m * f.ldc + n);
      d[0] = w0; d[1] = w1;
    } else {
      f32x4 o = {x[0], x[1], x[2], x[3]};
      *(f32x4*)((float*)f.C + m * f.ldc + n) = o;
    }
  }
}

// tail peel geometry of the plain f32-output GEMMs on the 256x256 tile (see sw_gemm): r peeled tile columns, sk K-splits
bool peel_geometry(int M, int N, int* r_out, long* sk_out) {
  const long tm = (M + 255) / 256, tn = (N + 255) / 256, tiles = tm * tn;
  if (!(tiles >= 512 && (tiles % 256) != 0 && (tiles % 256) <= 160)) return false;
  int r = 0;
  for (int c = 1; c <= 4 && !r; ++c)
    if (tn - c >= 2 && (tm * (tn - c)) % 256 == 0) r = c;
  if (!r) return false;
  const int N2 = N - (int)((tn - r) * 256);
  // the tail runs on 128x128 tiles, two workgroups per CU: K-splits to ~512 workgroups (fc6: 4 splits of 128 tiles =
  // 60 us; 8 splits put it on 256x256 tiles: 81 us; tools/probes/wgrad_tail.py)
  const long t128 = ((M + 127) / 128) * ((N2 + 127) / 128);
  long sk = 512 / t128;
  sk = sk < 1 ? 1 : (sk > 8 ? 8 : sk);
  while (sk > 1 && tm * r * sk >= 200) --sk;              // stay below launch_auto's switch to the 256x256 tile
  *r_out = r; *sk_out = sk;
  return true;
}

int effective_splits(int dtype, int K, int splitk) {
  const int bk = dtype == SW_BF16 ? 64 : 32;
  if (splitk < 1) splitk = 1;
  int kps = (K + splitk - 1) / splitk;
  kps = ((kps + bk - 1) / bk) * bk;
  return (K + kps - 1) / kps;
}

int check_align(const void* p) { return (((uintptr_t)p) & 15) ? -4 : 0; }

}  // namespace

int sw_sgd_tile_t_block(int rows, int cols, float* param, const float* grad, float* buf, long ld_src, float lr, float wd, int first,
                        float mom, float gscale, unsigned short* st0, long ld0, unsigned short* st1, long ld1, const float* hyper,
                        hipStream_t stream);                                 // elementwise.hip

// shapes the fused-SGD epilogue covers: launch_auto's ping-pong form (the only epilogue that implements it), whole 16-byte pieces
static bool sgd_fused_shape_ok(int dtype, int a_kstrided, int b_kstrided, int M, int N, int K) {
  static const char* pp = getenv("SW_GEMM_PP");
  static const char* v = getenv("SW_GEMM_V");
  if (dtype != SW_BF16 || a_kstrided || (pp && pp[0] == '0') || v) return false;
  (void)b_kstrided;
  if ((M % 32) || (N % 64) || (K % 64) || K < 1024 || N <= 128) return false;
  return ((long)((M + 255) / 256) * ((N + 255) / 256)) >= 200;
}
extern "C" int sw_gemm_sgd_fused_supported(int dtype, int a_kstrided, int b_kstrided, int M, int N, int K) {
  return sgd_fused_shape_ok(dtype, a_kstrided, b_kstrided, M, N, K) ? 1 : 0;
}

extern "C" int sw_gemm(int dtype, int a_kstrided, int b_kstrided, int M, int N, int K, const void* A, long lda,
                       const void* B, long ldb, void* C, long ldc, const sw_epilogue* ep, int splitk,
                       hipStream_t stream) {
  SW_ENTER();
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
  // Tail peel (plain f32-output GEMMs on the 256x256 tile, i.e. the fc weight gradients): with T tiles on 256 CUs the last
  // ceil(T/256)-th round runs (T mod 256)/256 full — fc6's weight gradient has 16 x 98 = 1568 tiles = 6.125 rounds and paid
  // for 7.  The last r tile columns (r <= 4, chosen so that the rest is a whole number of rounds) are computed by a second
  // launch on 128x128 tiles with split-K sized to fill the chip once: deterministic slabs + ordered fold when the caller
  // gave a split-K workspace, else f32 atomics into the zeroed column block.
  const bool plain = (!ep || (ep->out_dtype == SW_F32 && !ep->bias && !ep->relu && !ep->drop_mask && !(ep->drop_hash_p > 0.f) && !ep->relu_ref &&
                              !ep->accumulate_atomic && !ep->absmax_out && !ep->residual));
  const bool has_row_scale = ep && ep->fold_row_scale;
  float* const det_ws = (ep && plain) ? ep->splitk_workspace : nullptr;
  {
    static const bool no_peel = getenv("SW_GEMM_NO_PEEL") != nullptr;    // development switch
    int r = 0; long sk = 1;
    // (Peeling the bf16-output launches the same way — fc6's data gradient has 32 x 98 tiles = 12.25 rounds — gained 3 % alone
    // and lost 1 % inside the step: the persistent form already spreads its quarter round; not done.)
    if (!no_peel && plain && !has_row_scale && splitk <= 1 && peel_geometry(M, N, &r, &sk)) {
      const long es = dtype == SW_BF16 ? 2 : 4;
      const long tn = (N + 255) / 256;
      const int N1 = (int)((tn - r) * 256), N2 = N - N1;
      const char* B2 = (const char*)B + (b_kstrided ? (long)N1 * es : (long)N1 * ldb * es);
      sw_epilogue ep1 = {};
      ep1.out_dtype = SW_F32; ep1.drop_scale = 1.f; ep1.ref_scale = 1.f;
      const sw_sgd_tensor* fz = ep ? ep->sgd_fused : nullptr;
      if (fz) {                                     // the whole rounds update the parameter in their epilogue; the tail below goes through C
        if (!sgd_fused_shape_ok(dtype, a_kstrided, b_kstrided, M, N1, K)) return -5;
        ep1.sgd_fused = fz; ep1.sgd_momentum = ep->sgd_momentum; ep1.sgd_grad_scale = ep->sgd_grad_scale;
      }
      int rc = sw_gemm(dtype, a_kstrided, b_kstrided, M, N1, K, A, lda, B, ldb, C, ldc, &ep1, 1, stream);
      if (rc) return rc;
      float* C2 = (float*)C + N1;
      ep1.sgd_fused = nullptr;
      sw_epilogue ep2 = ep1;
      if (det_ws) ep2.splitk_workspace = det_ws;
      else if (sk > 1) {
        hipError_t e = hipMemset2DAsync(C2, (size_t)ldc * 4, 0, (size_t)N2 * 4, (size_t)M, stream);
        if (e != hipSuccess) return (int)e;
        ep2.accumulate_atomic = 1;
      }
      rc = sw_gemm(dtype, a_kstrided, b_kstrided, M, N2, K, A, lda, B2, ldb, C2, ldc, &ep2, (int)sk, stream);
      if (rc || !fz) return rc;
      // the tail columns' gradient sits in C2: the tiled update kernel on that column block (same arithmetic, same copies)
      return sw_sgd_tile_t_block(M, N2, fz->param + N1, C2, fz->momentum_buf + N1, ldc, fz->lr, fz->weight_decay, fz->first_step,
                                 ep->sgd_momentum, ep->sgd_grad_scale, fz->stage0 ? (unsigned short*)fz->stage0 + N1 : nullptr, fz->ld0,
                                 (unsigned short*)fz->stage1 + (long)N1 * fz->ld1, fz->ld1, fz->hyper_dev, stream);
    }
  }
  if (ep && ep->sgd_fused) {
    const sw_sgd_tensor* fz = ep->sgd_fused;
    if (!plain || splitk > 1 || !sgd_fused_shape_ok(dtype, a_kstrided, b_kstrided, M, N, K)) return -5;
    if (fz->stage_kind != 3 || fz->stage_dtype != SW_BF16 || !fz->stage1 || fz->d0 < N || !fz->param || !fz->momentum_buf) return -5;
    if ((ldc % 4) || (fz->ld0 % 4) || (fz->ld1 % 8) || ((((uintptr_t)fz->param) | ((uintptr_t)fz->momentum_buf)) & 15) ||
        (((uintptr_t)fz->stage0) & 7) || (((uintptr_t)fz->stage1) & 15)) return -4;
  }
  const int eff = effective_splits(dtype, K, splitk);
  // split-K with an epilogue (a few-tile, long-K GEMM whose result is not a plain f32 matrix: the 1x1 convolutions of res4 / res5, a
  // gradient accumulated into an existing one): plain slabs, the epilogue runs in the fold
  // (also with ONE slab when a row scale meets a residual: C = residual + row_scale * A B has no single-launch epilogue)
  const bool fold_ep = ep && !plain && (eff > 1 || (ep->fold_row_scale && ep->residual)) && ep->splitk_workspace &&
                       !ep->accumulate_atomic && !ep->drop_mask && !(ep->drop_hash_p > 0.f) && !ep->absmax_out;
  if (fold_ep) {
    if ((N % 4) || (ldc % 4) || (((uintptr_t)C) & 7) || (((uintptr_t)ep->splitk_workspace) & 15)) return -5;
    if (ep->residual && ((ep->ld_res % 4) || (((uintptr_t)ep->residual) & 7))) return -5;
    float* const slabs = ep->splitk_workspace;       // (an f32 residual aliasing C is legal: the slab launch does not touch C)
    GemmArgs gs = {};
    gs.A = A; gs.B = B; gs.M = M; gs.N = N; gs.K = K; gs.lda = lda; gs.ldb = ldb;
    gs.C = slabs; gs.ldc = N; gs.slab_stride = (long)M * N; gs.drop_scale = 1.f; gs.ref_scale = 1.f;
    {
      const long es = dtype == SW_BF16 ? 2 : 4;
      const long ab = (long)(a_kstrided ? K : M) * lda * es, bb = (long)(b_kstrided ? K : N) * ldb * es;
      if (ab >= 0xFFFFFF00L || bb >= 0xFFFFFF00L) return -6;
      gs.a_bytes = (unsigned)ab; gs.b_bytes = (unsigned)bb;
    }
    const int am = a_kstrided ? OP_KSTRIDED : OP_KCONTIG, bmo = b_kstrided ? OP_KSTRIDED : OP_KCONTIG;
    const int rc = dtype == SW_BF16 ? dispatch_modes<unsigned short>(gs, am, bmo, eff, stream) : dispatch_modes<float>(gs, am, bmo, eff, stream);
    if (rc) return rc;
    FoldEp f = {};
    f.M = M; f.N = N; f.nslab = eff; f.ws = slabs; f.C = C; f.ldc = ldc; f.out_bf16 = ep->out_dtype == SW_BF16;
    f.bias = ep->bias; f.row_scale = ep->fold_row_scale;
    f.res = ep->residual; f.ldres = ep->ld_res; f.res_bf16 = ep->res_dtype == SW_BF16;
    f.relu = ep->relu; f.ref = ep->relu_ref; f.ldr = ep->ld_ref; f.ref_bf16 = ep->ref_dtype == SW_BF16; f.ref_scale = ep->ref_scale;
    long blocks = (((long)M * N >> 2) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(splitk_fold_ep_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f);
    SW_CHECK_LAUNCH();
    return 0;
  }
  const bool det = det_ws != nullptr && eff > 1;
  if (det && (((uintptr_t)det_ws) & 15)) return -4;
  GemmArgs g = {};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  if (ep) {
    g.bias = ep->bias; g.drop = ep->drop_mask; g.ldd = ep->ld_drop; g.drop_scale = ep->drop_scale;
    g.ref = ep->relu_ref; g.ldr = ep->ld_ref; g.ref_scale = ep->ref_scale; g.ref_bf16 = ep->ref_dtype == SW_BF16;
    g.res = ep->residual; g.ldres = ep->ld_res; g.res_bf16 = ep->res_dtype == SW_BF16;
    g.relu = ep->relu; g.out_bf16 = ep->out_dtype == SW_BF16; g.atomic = ep->accumulate_atomic; g.absmax = ep->absmax_out;
    if (ep->sgd_fused) {
      const sw_sgd_tensor* fz = ep->sgd_fused;
      g.sgd_param = fz->param; g.sgd_mom = fz->momentum_buf; g.sgd_ld = ldc;
      g.sgd_st0 = (unsigned short*)fz->stage0; g.sgd_ld0 = fz->ld0; g.sgd_st1 = (unsigned short*)fz->stage1; g.sgd_ld1 = fz->ld1;
      g.sgd_hyper = fz->hyper_dev; g.sgd_lr = fz->lr; g.sgd_wd = fz->weight_decay; g.sgd_first = fz->first_step;
      g.sgd_momentum = ep->sgd_momentum; g.sgd_gscale = ep->sgd_grad_scale;
    }
    if (!ep->drop_mask && ep->drop_hash_p > 0.f) {
      unsigned long long x = ep->drop_seed;                  // splitmix64(seed), as the mask kernel mixes it
      x += 0x9E3779B97F4A7C15ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      x ^= x >> 31;
      g.drop_seed_mixed = x; g.drop_offset = ep->drop_offset; g.drop_p = ep->drop_hash_p;
      g.drop_offset_dev = (const unsigned long long*)ep->drop_offset_dev;
    }
  }
  {
    const long es = dtype == SW_BF16 ? 2 : 4;
    const long ab = (long)(a_kstrided ? K : M) * lda * es, bb = (long)(b_kstrided ? K : N) * ldb * es;
    if (ab >= 0xFFFFFF00L || bb >= 0xFFFFFF00L) return -6;           // 32-bit buffer offsets
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
  }
  const int am = a_kstrided ? OP_KSTRIDED : OP_KCONTIG, bmo = b_kstrided ? OP_KSTRIDED : OP_KCONTIG;
  if (det) {                                   // every K-split stores into its own slab [z][M][N]; ordered fold into C below
    g.C = det_ws; g.ldc = N; g.slab_stride = (long)M * N; g.atomic = 0;
  }
  const int rc = dtype == SW_BF16 ? dispatch_modes<unsigned short>(g, am, bmo, det ? eff : splitk, stream)
                                  : dispatch_modes<float>(g, am, bmo, det ? eff : splitk, stream);
  if (rc) return rc;
  if (!det) {
    if (ep && ep->fold_row_scale) {            // no slab fold in this launch: the row scale as a pass of its own (plain f32 C only)
      if (!plain || ep->out_dtype != SW_F32) return -5;
      long blocks = ((long)M * N + 255) / 256;
      blocks = blocks > 4096 ? 4096 : blocks;
      hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, M, N, (float*)C, ldc, ep->fold_row_scale);
      SW_CHECK_LAUNCH();
    }
    return 0;
  }
  const int vec = ((N % 4) == 0 && (ldc % 4) == 0 && (((uintptr_t)C) & 15) == 0) ? 1 : 0;
  long blocks = (((long)M * N >> (vec ? 2 : 0)) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, M, N, eff, det_ws, (float*)C, ldc, vec,
                     ep ? ep->fold_row_scale : nullptr);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" long sw_gemm_splitk_workspace_floats(int M, int N, int K, int splitk) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (splitk > 1) return (long)splitk * M * N;
  int r = 0; long sk = 1;
  if (peel_geometry(M, N, &r, &sk) && sk > 1) {
    const long tn = (N + 255) / 256;
    const int N2 = N - (int)((tn - r) * 256);
    return sk * (long)M * N2;
  }
  return 0;
}

// conv_direct.hip: halo-reusing direct kernel for the bf16 conv3..conv5 shapes (1 = launched, 0 = not covered, < 0 error)
int sw_conv3x3_direct_try(int nimg, int H, int W, int Cin, int Cout, int dilation, const void* in, const void* wk, void* out,
                          const sw_epilogue* ep, hipStream_t stream);

// conv3x3 (stride 1, pad = dilation) forward / data-gradient as an implicit GEMM over an NHWC tensor:
//   out[(img,y,x)][co] = sum_{tap,ci} in[img, y+(ty-1)d, x+(tx-1)d, ci] * Wk[co][tap][ci]
extern "C" int sw_conv3x3_igemm(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* in,
                                const void* wk, void* out, const sw_epilogue* ep, hipStream_t stream) {
  SW_ENTER();
  const int epc = dtype == SW_BF16 ? 8 : 4;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if (Cin % epc) return -5;
  if (check_align(in) || check_align(wk)) return -4;
  if (dtype == SW_BF16) {
    const int rc = sw_conv3x3_direct_try(nimg, H, W, Cin, Cout, dilation, in, wk, out, ep, stream);
    if (rc != 0) return rc < 0 ? -rc : 0;
  }
  GemmArgs g = {};
  g.A = in; g.B = wk; g.C = out; g.M = nimg * H * W; g.N = Cout; g.K = 9 * Cin; g.lda = 0; g.ldb = 9L * Cin; g.ldc = Cout;
  g.cH = H; g.cW = W; g.cC = Cin; g.cDil = dilation;
  // bf16 only: the fp32 parity mode keeps the natural K order, whose f32 rounding tracks the reference's conv closely enough
  // that ReLU masks / pool argmax of near-zero activations do not flip (test_backbone_backward_matches_autograd)
  { static const bool no_rot = getenv("SW_CONV_NO_KROT") != nullptr; g.krot = (no_rot || dtype != SW_BF16) ? 0 : 1; }
  if (ep) {
    g.bias = ep->bias; g.drop = ep->drop_mask; g.ldd = ep->ld_drop; g.drop_scale = ep->drop_scale;
    g.ref = ep->relu_ref; g.ldr = ep->ld_ref; g.ref_scale = ep->ref_scale; g.ref_bf16 = ep->ref_dtype == SW_BF16;
    g.relu = ep->relu; g.out_bf16 = ep->out_dtype == SW_BF16; g.atomic = ep->accumulate_atomic;
  }
  {
    const long es = dtype == SW_BF16 ? 2 : 4;
    const long ab = (long)nimg * H * W * Cin * es, bb = (long)Cout * 9 * Cin * es;
    if (ab >= 0xFFFFFF00L || bb >= 0xFFFFFF00L) return -6;
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
  }
  const int bk = dtype == SW_BF16 ? 64 : 32;
  const int amode = (Cin % bk == 0) ? OP_CONV_A : OP_CONV_A_GEN;
  return dtype == SW_BF16 ? dispatch_modes<unsigned short>(g, amode, OP_KCONTIG, 1, stream)
                          : dispatch_modes<float>(g, amode, OP_KCONTIG, 1, stream);
}

namespace {
// sum the split-K slabs [z][co][tap][ci] in a fixed order and permute to OIHW [co][ci][tap] (deterministic, no atomics).
// A workgroup owns one output channel and one of gridDim.y input-channel ranges: its 9 x CI partial sums are read as 16-byte
// pieces (eight slabs in flight per iteration; a scalar one-slab-at-a-time loop was latency bound once the launches grew to
// 14-28 slabs), folded, transposed through LDS and written as ONE contiguous run of 16-byte stores (the direct form
// scattered 4-byte stores 36 bytes apart: 2.4 M write transactions per conv4 layer).  Requires (Cin / gridDim.y) % 4 == 0.
__device__ __forceinline__ void wgrad_reduce_body(int Cout, int Cin, int nslab, const float* __restrict__ slabs,
                                                  float* __restrict__ out, int co, int parts, int part, float* s_t,
                                                  const float* __restrict__ cout_scale = nullptr, const int accumulate = 0) {
  const int CI = Cin / parts, ci0 = part * CI;
  const int cv = CI >> 2, nv = 9 * cv;                              // 16-byte pieces per tap / per workgroup
  const long slab_f = (long)Cout * 9 * Cin;
  const float* src = slabs + (long)co * 9 * Cin + ci0;
  for (int q = threadIdx.x; q < nv; q += 256) {
    const int tap = q / cv, c4 = q - tap * cv;
    const float* p = src + (long)tap * Cin + c4 * 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int z = 0;
    for (; z + 8 <= nslab; z += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(p + (long)(z + u) * slab_f);
      a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
      a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
    }
    for (; z < nslab; ++z) a0 += *(const f32x4*)(p + (long)z * slab_f);
    f32x4 r = (a0 + a1) + (a2 + a3);
    if (cout_scale) { const float sc = cout_scale[co]; r[0] = __fmul_rn(r[0], sc); r[1] = __fmul_rn(r[1], sc); r[2] = __fmul_rn(r[2], sc); r[3] = __fmul_rn(r[3], sc); }
    *(f32x4*)(s_t + tap * CI + c4 * 4) = r;
  }
  __syncthreads();
  f32x4* dst = (f32x4*)(out + ((long)co * Cin + ci0) * 9);
  for (int q = threadIdx.x; q < nv; q += 256) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = 4 * q + j, ci = e / 9, tap = e - 9 * ci;
      o[j] = s_t[tap * CI + ci];
    }
    if (accumulate) o += dst[q];                                    // a gradient that exists already (the second pass of one backward)
    dst[q] = o;
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(int Cout, int Cin, int nslab, const float* __restrict__ slabs,
                                                           float* __restrict__ out, const float* __restrict__ cout_scale, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float s_t[];       // [9][CI]
  wgrad_reduce_body(Cout, Cin, nslab, slabs, out, blockIdx.x, (int)gridDim.y, blockIdx.y, s_t, cout_scale, accumulate);
}

// every fold of a backward pass in ONE launch: workgroup -> (parameter, output channel, input-channel range)
constexpr int FOLD_MAX = 32;
struct FoldMulti {
  int n;
  int first_wg[FOLD_MAX + 1];
  int Cin[FOLD_MAX], Cout[FOLD_MAX], nslab[FOLD_MAX], parts[FOLD_MAX];
  const float* ws[FOLD_MAX];
  float* out[FOLD_MAX];
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(FoldMulti f) {
  extern __shared__ __attribute__((aligned(16))) float s_t[];
  int i = 0;
  while (i + 1 < f.n && (int)blockIdx.x >= f.first_wg[i + 1]) ++i;
  const int local = blockIdx.x - f.first_wg[i];
  wgrad_reduce_body(f.Cout[i], f.Cin[i], f.nslab[i], f.ws[i], f.out[i], local / f.parts[i], f.parts[i], local % f.parts[i], s_t);
}

}  // namespace

extern "C" long sw_conv3x3_wgrad_workspace_floats(int dtype, int nimg, int H, int W, int Cin, int Cout, int splitk) {
  const int bk = dtype == SW_BF16 ? 64 : 32;
  const long K = (long)nimg * H * W;
  if (splitk < 1) splitk = 1;
  long kps = (K + splitk - 1) / splitk;
  kps = ((kps + bk - 1) / bk) * bk;
  const long eff = (K + kps - 1) / kps;
  return eff * Cout * 9L * Cin;
}

// conv3x3 weight gradient:  dW[co][ci][ty][tx] (OIHW, f32, overwritten)
//   = sum_{img,y,x} dY[(img,y,x)][co] * X[img, y+(ty-1)d, x+(tx-1)d, ci]
// Every K-split stores its partial [co][tap][ci] tile into its own slab of the workspace with plain coalesced stores;
// a second kernel adds the slabs in fixed order and permutes to OIHW.  (f32 atomics were measured 3-5x slower here:
// the splits of one tile finish together and collide on the same addresses; this form is also deterministic.)
extern "C" int sw_conv3x3_wgrad_slabs(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x,
                                      const void* dy, float* workspace, int splitk, hipStream_t stream) {
  SW_ENTER();
  const int epc = dtype == SW_BF16 ? 8 : 4;
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if ((Cin % epc) || (Cout % epc)) return -5;
  if (check_align(x) || check_align(dy) || check_align(workspace)) return -4;
  if ((64 / W) + 1 > 2 * H) return -6;          // pixel-advance carry logic of the gather (tiny maps only)
  const long nelem = (long)Cout * 9 * Cin;
  const int nslab = (int)(sw_conv3x3_wgrad_workspace_floats(dtype, nimg, H, W, Cin, Cout, splitk) / nelem);
  GemmArgs g = {};
  g.A = dy; g.B = x; g.C = workspace; g.M = Cout; g.N = 9 * Cin; g.K = nimg * H * W; g.lda = Cout; g.ldb = 0;
  g.ldc = 9L * Cin; g.cH = H; g.cW = W; g.cC = Cin; g.cDil = dilation; g.atomic = 0; g.oihw_cin = 0;
  g.slab_stride = nelem;
  {
    const long es = dtype == SW_BF16 ? 2 : 4;
    const long ab = (long)nimg * H * W * Cout * es, bb = (long)nimg * H * W * Cin * es;
    if (ab >= 0xFFFFFF00L || bb >= 0xFFFFFF00L) return -6;
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
  }
  return dtype == SW_BF16 ? dispatch_modes<unsigned short>(g, OP_KSTRIDED, OP_CONV_B, nslab, stream)
                          : dispatch_modes<float>(g, OP_KSTRIDED, OP_CONV_B, nslab, stream);
}

static int wgrad_fold_impl(int Cin, int Cout, int nslab, const float* workspace, float* dw_oihw, const float* cout_scale,
                           hipStream_t stream, int accumulate = 0);
extern "C" int sw_conv3x3_wgrad_fold(int Cin, int Cout, int nslab, const float* workspace, float* dw_oihw,
                                     hipStream_t stream) {
  SW_ENTER();
  return wgrad_fold_impl(Cin, Cout, nslab, workspace, dw_oihw, nullptr, stream);
}
extern "C" int sw_conv3x3_wgrad_fold_acc(int Cin, int Cout, int nslab, const float* workspace, float* dw_oihw, const float* cout_scale,
                                        int accumulate, hipStream_t stream) {
  SW_ENTER();
  return wgrad_fold_impl(Cin, Cout, nslab, workspace, dw_oihw, cout_scale, stream, accumulate);
}
static int wgrad_fold_impl(int Cin, int Cout, int nslab, const float* workspace, float* dw_oihw, const float* cout_scale,
                           hipStream_t stream, int accumulate) {
  if (nslab < 1 || (Cin % 4)) return -5;
  if ((size_t)36 * Cin > 65536 || (((uintptr_t)dw_oihw) & 15) || (((uintptr_t)workspace) & 15)) return -5;
  int parts = 1;                                              // input-channel ranges per output channel: >= 1024 workgroups
  while (Cout * parts < 1024 && (Cin % (parts * 2 * 4)) == 0 && Cin / (parts * 2) >= 32) parts *= 2;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)Cout, (unsigned)parts), dim3(256), (size_t)36 * Cin / parts, stream, Cout, Cin,
                     nslab, workspace, dw_oihw, cout_scale, accumulate);
  SW_CHECK_LAUNCH();
  return 0;
}

int sw_conv3x3_wgrad_direct_try(int n_problems, const sw_wgrad_problem* problems, const int* eff, hipStream_t stream);   // conv_wgrad_direct.hip

extern "C" int sw_conv3x3_wgrad_grouped(int dtype, int n_problems, const sw_wgrad_problem* problems, hipStream_t stream) {
  SW_ENTER();
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if (n_problems <= 0) return 0;
  if (dtype == SW_BF16 && n_problems <= 256) {
    // round 6: the direct weight-gradient kernel (input rows staged once for all nine taps) when it covers every problem of the list;
    // it writes the same slabs (count and layout) as the implicit GEMM below
    int eff[256];
    for (int i = 0; i < n_problems; ++i) {
      const sw_wgrad_problem& q = problems[i];
      eff[i] = (int)(sw_conv3x3_wgrad_workspace_floats(dtype, q.nimg, q.H, q.W, q.Cin, q.Cout, q.nsplit) / ((long)q.Cout * 9 * q.Cin));
    }
    const int rc = sw_conv3x3_wgrad_direct_try(n_problems, problems, eff, stream);
    if (rc < 0) return -rc;
    if (rc == 1) return 0;
  }
  const int epc = dtype == SW_BF16 ? 8 : 4, bk = dtype == SW_BF16 ? 64 : 32;
  const long es = dtype == SW_BF16 ? 2 : 4;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
    else ncu = 256;
  }
  constexpr int LDS = 2 * (Geom2<unsigned short, OP_KSTRIDED, 256>::BYTES + Geom2<unsigned short, OP_CONV_B, 256>::BYTES);
  static_assert(LDS == 2 * (Geom2<float, OP_KSTRIDED, 256>::BYTES + Geom2<float, OP_CONV_B, 256>::BYTES), "one LDS size for both types");
  for (int base = 0; base < n_problems; base += GROUPED_MAX) {
    GroupedArgs ga = {};
    ga.n_problems = n_problems - base < GROUPED_MAX ? n_problems - base : GROUPED_MAX;
    int items = 0;
    for (int i = 0; i < ga.n_problems; ++i) {
      const sw_wgrad_problem& q = problems[base + i];
      if ((q.Cin % epc) || (q.Cout % epc)) return -5;
      if (check_align(q.x) || check_align(q.dy) || check_align(q.slabs)) return -4;
      if ((64 / q.W) + 1 > 2 * q.H) return -6;          // pixel-advance carry logic of the gather (tiny maps only)
      GroupedProblem& P = ga.p[i];
      P.A = q.dy; P.B = q.x; P.C = q.slabs; P.lda = q.Cout; P.ldb = 0;
      P.M = q.Cout; P.N = 9 * q.Cin; P.K = q.nimg * q.H * q.W; P.cH = q.H; P.cW = q.W; P.cC = q.Cin; P.cDil = q.dilation;
      int ns = q.nsplit < 1 ? 1 : q.nsplit;
      long kps = (P.K + ns - 1) / ns;
      kps = ((kps + bk - 1) / bk) * bk;
      P.k_per_split = (int)kps;
      P.nsplit = (int)((P.K + kps - 1) / kps);
      P.tiles_m = (P.M + 255) / 256; P.tiles_n = (P.N + 255) / 256;
      const long ab = (long)P.K * q.Cout * es, bb = (long)P.K * q.Cin * es;
      if (ab >= 0xFFFFFF00L || bb >= 0xFFFFFF00L) return -6;
      P.a_bytes = (unsigned)ab; P.b_bytes = (unsigned)bb;
      ga.first_item[i] = items;
      items += P.nsplit * P.tiles_m * P.tiles_n;
    }
    ga.first_item[ga.n_problems] = items;
    ga.n_items = items;
    int G = ncu - (ncu % 8);
    if (G < 8) G = 8;
    hipError_t e;
    if (dtype == SW_BF16) {
      e = hipFuncSetAttribute((const void*)gemm2_grouped_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(gemm2_grouped_kernel<unsigned short>, dim3(G), dim3(1024), LDS, stream, ga);
    } else {
      e = hipFuncSetAttribute((const void*)gemm2_grouped_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(gemm2_grouped_kernel<float>, dim3(G), dim3(1024), LDS, stream, ga);
    }
    SW_CHECK_LAUNCH();
  }
  return 0;
}

// The same resident-grid launch for PLAIN weight-gradient GEMMs (both operands K-strided: C = A^T B over K = pixels): the 1x1
// convolutions of a ResNet bottleneck — three weights per block, each used by every forward pass of the iteration.
extern "C" int sw_gemm_kk_grouped(int dtype, int n_problems, const sw_gemm_kk_problem* problems, hipStream_t stream) {
  SW_ENTER();
  if (dtype != SW_BF16 && dtype != SW_F32) return -1;
  if (n_problems <= 0) return 0;
  const int epc = dtype == SW_BF16 ? 8 : 4, bk = dtype == SW_BF16 ? 64 : 32;
  const long es = dtype == SW_BF16 ? 2 : 4;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
    else ncu = 256;
  }
  constexpr int LDS = 4 * Geom2<unsigned short, OP_KSTRIDED, 256>::BYTES;
  static_assert(LDS == 4 * Geom2<float, OP_KSTRIDED, 256>::BYTES, "one LDS size for both types");
  for (int base = 0; base < n_problems; base += GROUPED_MAX) {
    GroupedArgs ga = {};
    ga.n_problems = n_problems - base < GROUPED_MAX ? n_problems - base : GROUPED_MAX;
    int items = 0;
    for (int i = 0; i < ga.n_problems; ++i) {
      const sw_gemm_kk_problem& q = problems[base + i];
      if (q.M <= 0 || q.N <= 0 || q.K <= 0) return -5;
      if ((q.M % epc) || (q.N % epc) || (q.lda % epc) || (q.ldb % epc) || q.lda < q.M || q.ldb < q.N) return -5;
      if (check_align(q.A) || check_align(q.B) || check_align(q.slabs)) return -4;
      GroupedProblem& P = ga.p[i];
      P.A = q.A; P.B = q.B; P.C = q.slabs; P.lda = (int)q.lda; P.ldb = (int)q.ldb;
      P.M = q.M; P.N = q.N; P.K = q.K;
      int ns = q.nsplit < 1 ? 1 : q.nsplit;
      long kps = (P.K + ns - 1) / ns;
      kps = ((kps + bk - 1) / bk) * bk;
      P.k_per_split = (int)kps;
      P.nsplit = (int)((P.K + kps - 1) / kps);
      P.tiles_m = (P.M + 255) / 256; P.tiles_n = (P.N + 255) / 256;
      const long ab = (long)P.K * q.lda * es, bb = (long)P.K * q.ldb * es;
      if (ab >= 0xFFFFFF00L || bb >= 0xFFFFFF00L) return -6;
      P.a_bytes = (unsigned)ab; P.b_bytes = (unsigned)bb;
      ga.first_item[i] = items;
      items += P.nsplit * P.tiles_m * P.tiles_n;
    }
    ga.first_item[ga.n_problems] = items;
    ga.n_items = items;
    int G = ncu - (ncu % 8);
    if (G < 8) G = 8;
    hipError_t e;
    if (dtype == SW_BF16) {
      e = hipFuncSetAttribute((const void*)gemm2_grouped_kernel<unsigned short, OP_KSTRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((gemm2_grouped_kernel<unsigned short, OP_KSTRIDED>), dim3(G), dim3(1024), LDS, stream, ga);
    } else {
      e = hipFuncSetAttribute((const void*)gemm2_grouped_kernel<float, OP_KSTRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL((gemm2_grouped_kernel<float, OP_KSTRIDED>), dim3(G), dim3(1024), LDS, stream, ga);
    }
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" long sw_gemm_kk_grouped_slabs(int dtype, int K, int nsplit) {        // slabs problem (K, nsplit) really writes
  const int bk = dtype == SW_BF16 ? 64 : 32;
  int ns = nsplit < 1 ? 1 : nsplit;
  long kps = ((long)K + ns - 1) / ns;
  kps = ((kps + bk - 1) / bk) * bk;
  return K <= 0 ? 0 : ((long)K + kps - 1) / kps;
}

// n ordered folds (splitk_reduce_kernel: C = [C +] row_scale * sum of slabs) in ONE launch; 16-byte pieces only
namespace {
constexpr int SFOLD_MAX = 24;
struct SplitkFoldMulti {
  int n;
  int first_wg[SFOLD_MAX + 1];
  int M[SFOLD_MAX], N[SFOLD_MAX], nslab[SFOLD_MAX], acc[SFOLD_MAX];
  const float* ws[SFOLD_MAX]; float* C[SFOLD_MAX]; const float* row_scale[SFOLD_MAX];
  long ldc[SFOLD_MAX];
};
__global__ __launch_bounds__(256) void splitk_fold_multi_kernel(SplitkFoldMulti f) {
  int i = 0;
  while (i + 1 < f.n && (int)blockIdx.x >= f.first_wg[i + 1]) ++i;
  const int blk = blockIdx.x - f.first_wg[i], nblk = f.first_wg[i + 1] - f.first_wg[i];
  const int M = f.M[i], N = f.N[i], nslab = f.nslab[i];
  const long nv = ((long)M * N) >> 2;
  const int nq = N >> 2;
  const float* rs = f.row_scale[i];
  for (long q = blk * 256L + threadIdx.x; q < nv; q += (long)nblk * 256) {
    const f32x4* src = (const f32x4*)f.ws[i] + q;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int z = 0;
    for (; z + 4 <= nslab; z += 4) {
      const f32x4 v0 = src[(long)z * nv], v1 = src[(long)(z + 1) * nv], v2 = src[(long)(z + 2) * nv], v3 = src[(long)(z + 3) * nv];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; z < nslab; ++z) a0 += src[(long)z * nv];
    const long m = q / nq; const int n4 = (int)(q - m * nq);
    f32x4 r = (a0 + a1) + (a2 + a3);
    if (rs) { const float sc = rs[m]; r[0] = __fmul_rn(r[0], sc); r[1] = __fmul_rn(r[1], sc); r[2] = __fmul_rn(r[2], sc); r[3] = __fmul_rn(r[3], sc); }
    f32x4* dst = (f32x4*)(f.C[i] + m * f.ldc[i] + n4 * 4);
    if (f.acc[i]) r += *dst;
    *dst = r;
  }
}
}  // namespace

extern "C" int sw_splitk_fold_multi(int n, const sw_splitk_fold* folds, hipStream_t stream) {
  SW_ENTER();
  for (int base = 0; base < n; base += SFOLD_MAX) {
    SplitkFoldMulti f = {};
    f.n = n - base < SFOLD_MAX ? n - base : SFOLD_MAX;
    int wgs = 0;
    for (int i = 0; i < f.n; ++i) {
      const sw_splitk_fold& q = folds[base + i];
      if (q.M <= 0 || q.N <= 0 || q.nslab < 1 || (q.N % 4) || (q.ldc % 4) || (((uintptr_t)q.C) & 15) || (((uintptr_t)q.workspace) & 15)) return -5;
      f.M[i] = q.M; f.N[i] = q.N; f.nslab[i] = q.nslab; f.acc[i] = q.accumulate; f.ws[i] = q.workspace; f.C[i] = q.C;
      f.row_scale[i] = q.row_scale; f.ldc[i] = q.ldc;
      long b = (((long)q.M * q.N >> 2) + 255) / 256;
      b = b < 1 ? 1 : (b > 1024 ? 1024 : b);
      f.first_wg[i] = wgs;
      wgs += (int)b;
    }
    f.first_wg[f.n] = wgs;
    hipLaunchKernelGGL(splitk_fold_multi_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, f);
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int sw_conv3x3_wgrad_fold_multi(int n, const sw_wgrad_fold* folds, hipStream_t stream) {
  SW_ENTER();
  for (int base = 0; base < n; base += FOLD_MAX) {
    FoldMulti f = {};
    f.n = n - base < FOLD_MAX ? n - base : FOLD_MAX;
    int wgs = 0;
    size_t lds = 0;
    for (int i = 0; i < f.n; ++i) {
      const sw_wgrad_fold& q = folds[base + i];
      if (q.nslab < 1 || (q.Cin % 4) || (size_t)36 * q.Cin > 65536 || (((uintptr_t)q.dw_oihw) & 15) || (((uintptr_t)q.workspace) & 15)) return -5;
      int parts = 1;
      while (q.Cout * parts < 1024 && (q.Cin % (parts * 2 * 4)) == 0 && q.Cin / (parts * 2) >= 32) parts *= 2;
      f.Cin[i] = q.Cin; f.Cout[i] = q.Cout; f.nslab[i] = q.nslab; f.parts[i] = parts; f.ws[i] = q.workspace; f.out[i] = q.dw_oihw;
      f.first_wg[i] = wgs;
      wgs += q.Cout * parts;
      const size_t need = (size_t)36 * q.Cin / parts;
      lds = need > lds ? need : lds;
    }
    f.first_wg[f.n] = wgs;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)wgs), dim3(256), lds, stream, f);
    SW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int sw_conv3x3_wgrad(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x,
                                const void* dy, float* dw_oihw, float* workspace, int splitk, hipStream_t stream) {
  return sw_conv3x3_wgrad_scaled(dtype, nimg, H, W, Cin, Cout, dilation, x, dy, dw_oihw, workspace, splitk, nullptr, stream);
}

extern "C" int sw_conv3x3_wgrad_scaled(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x,
                                       const void* dy, float* dw_oihw, float* workspace, int splitk, const float* cout_scale,
                                       hipStream_t stream) {
  return sw_conv3x3_wgrad_acc(dtype, nimg, H, W, Cin, Cout, dilation, x, dy, dw_oihw, workspace, splitk, cout_scale, 0, stream);
}

extern "C" int sw_conv3x3_wgrad_acc(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x,
                                    const void* dy, float* dw_oihw, float* workspace, int splitk, const float* cout_scale,
                                    int accumulate, hipStream_t stream) {
  const int rc = sw_conv3x3_wgrad_slabs(dtype, nimg, H, W, Cin, Cout, dilation, x, dy, workspace, splitk, stream);
  if (rc) return rc;
  const long nelem = (long)Cout * 9 * Cin;
  const int nslab = (int)(sw_conv3x3_wgrad_workspace_floats(dtype, nimg, H, W, Cin, Cout, splitk) / nelem);
  if (nslab < 1 || (Cin % 4)) return -5;
  return wgrad_fold_impl(Cin, Cout, nslab, workspace, dw_oihw, cout_scale, stream, accumulate);
}
