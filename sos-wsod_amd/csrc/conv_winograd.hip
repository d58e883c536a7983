// Winograd F(2x2, 3x3) convolution (stride 1, pad = dilation, NHWC, bf16) for gfx950: forward and data gradient of the
// 256 / 512-channel layers (conv3_2 .. conv5_3), where the direct kernel (conv_direct.hip) is MFMA-issue bound at 0.39 of peak
// and nothing but fewer MFMAs moves it: 16 multiplications per 2x2 outputs instead of 36 (2.25x fewer MFMA cycles).
//
//   Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A        d: 4x4 input tile, g: 3x3 filter, Y: 2x2 outputs
//
// The filters arrive transformed (sw_winograd_weight_prep: U[xi][co][ci] = G g G^T from the f32 masters, rounded to bf16 once).
// A workgroup (16 waves) owns 8 x 8 tiles (16 x 16 output pixels of one image) x 64 output channels and walks the input channels
// in chunks of 32:
//   T  every thread transforms one (tile, channel pair): 16 patch pixels from the LDS-DMA-staged (18 x 18) patch -> f32 B^T d B ->
//      bf16 -> V[xi][tile][ci] in LDS (the one extra rounding of this form: V is the MFMA operand)
//   M  wave w owns transform position xi = w: acc[xi] (64 tiles x 64 channels, f32) += V[xi] (64 x 32) . U[xi] (32 x 64), 16 MFMAs
// Three barriers per chunk; every LDS-DMA transfer (the patch, the two 32-channel halves of U) is in flight for a whole phase
// before its wait.  After the last chunk the
// 16 accumulator sets meet through LDS (two halves of 32 channels), every thread forms A^T m A for one tile and channel pair, adds
// the bias, applies ReLU (forward) or the ReLU mask of the producer (data gradient) and stores bf16 pairs.
// Dilation 2 (conv5): the four parity classes of the pixel grid are four independent dilation-1 problems on sub-grids; a
// workgroup works on one class (pixel step 2 in memory).  Image borders / ragged edges: DMA offsets beyond num_records -> zeros.
// Numerics: inputs and outputs bf16 as in the direct kernel, f32 transforms and accumulation; the transformed operands are
// rounded to bf16 (relative L2 of a layer's output against float64: 3.6e-3 vs 1.8e-3 for the direct form; the output's own bf16
// rounding is 1.7e-3).  fp32 mode never comes here.
#include <stdlib.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {

constexpr int WT = 8, WPS = 2 * WT + 2, WNPX = WPS * WPS;      // tiles per side, patch side (18), patch pixels (324)
constexpr int WCK = 16, WTN = 64, WNT = 1024;                  // channels per phase, output channels per workgroup, threads
constexpr int W_RAW_INSTR = (WNPX + 31) / 32;                  // 1 KiB LDS-DMA instructions per patch sub-chunk: 32 rows of 32 B (11)
constexpr int W_RAW_BYTES = W_RAW_INSTR * 1024;                // per buffer
constexpr int W_V_BYTES = 16 * 64 * 32, W_U_BYTES = 16 * WTN * 32;      // per buffer: [xi][64 rows][32 B]
constexpr int W_LDS = 2 * (W_RAW_BYTES + W_V_BYTES + W_U_BYTES);
constexpr unsigned W_INVALID = 0xFFFFFF00u;

struct WinoArgs {
  const void* in; const void* U; void* out;
  const float* bias; const void* ref;
  int nimg, H, W, Cin, Cout, relu, dil;
  int bly, blx, n_px_blocks, n_co_blocks, total;
  unsigned in_bytes, u_bytes;
};

// 32-byte LDS rows (16 channels): row r keeps its two 16-byte halves swapped when bit 3 of r is set — the 16 lanes of every
// ds_read_b128 service group then hit 16 distinct 16-byte slots of the 256-byte bank window (rows r and r + 8 / r + 24 share a slot pair)
__device__ __forceinline__ int w_swz(int row) { return (row >> 3) & 1; }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {          // ONE v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
  return __builtin_bit_cast(unsigned int, v);
}

// Phase p (one 16-channel sub-chunk, Cin / 16 of them), ONE barrier per phase, everything double buffered:
//   at the barrier: V[p & 1] (transform of sub-chunk p), U[p & 1], patch[(p + 1) & 1] are complete
//   issue LDS-DMA: U(p + 1) -> U[(p + 1) & 1], patch(p + 2) -> patch[p & 1]            (both buffers were last read one phase ago)
//   M(p): wave xi multiplies V[p & 1][xi] (64 tiles x 16) by U[p & 1][xi] (16 x 64): 4 x v_mfma_f32_32x32x16_bf16
//   T(p + 1): waves 0-7, one (tile, channel pair) per thread: patch[(p + 1) & 1] -> f32 B^T d B -> bf16 -> V[(p + 1) & 1]
// The transform's vector work issues in the shadow of the wave's own MFMAs and of the other waves' — the two-phase form
// (transform, barrier, multiply, barrier on single buffers of 32 channels) ran them one after the other: 42 vs 38 us for the direct kernel.
__global__ __launch_bounds__(WNT) void conv3x3_winograd_kernel(WinoArgs g) {
  typedef __attribute__((address_space(3))) void* lvoid;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sRAW = smem;                                       // [2][11 KiB]
  char* const sV = smem + 2 * W_RAW_BYTES;                       // [2][32 KiB]
  char* const sU = sV + 2 * W_V_BYTES;                           // [2][32 KiB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (channel block, pixel block): XCD x takes a contiguous run of the work list in which the pixel block runs
  // fastest, so an XCD streams ONE channel block's transformed filters (16 x 64 x Cin: 1 MiB at Cin = 512) from its own L2
  const unsigned bid = blockIdx.x;
  const int per_xcd = (g.total + 7) >> 3;
  const int f = (bid & 7) * per_xcd + (bid >> 3);
  if ((int)(bid >> 3) >= per_xcd || f >= g.total) return;
  const int co_blk = f / g.n_px_blocks;
  int pb = f - co_blk * g.n_px_blocks;
  const int co0 = co_blk * WTN;
  const int d = g.dil, ncls = d * d;
  const int bx = pb % g.blx; pb /= g.blx;
  const int by = pb % g.bly; pb /= g.bly;
  const int cls = pb % ncls; const int img = pb / ncls;
  const int py = cls / d, px = cls - py * d;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.in, 0, (int)g.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)g.U, 0, (int)g.u_bytes, 0x00020000);

  // ---- fixed per-lane DMA offsets.  Patch image: row r' = u * 18 + perm(v), perm = even columns first: the patch pixels of
  // horizontally adjacent tiles (v and v + 2) are adjacent rows, so a wave's 8 tiles x 8 channel pairs read 256 contiguous bytes.
  // Instruction q (waves 0 .. 10 issue one each) carries rows 32 q .. 32 q + 31, two 16-byte halves per row.
  unsigned a_v;
  {
    const int r = wave * 32 + (lane >> 1), half = lane & 1;
    const int u = r / WPS, pv = r - u * WPS;
    const int v = pv < 9 ? 2 * pv : 2 * (pv - 9) + 1;
    const int y = py + d * (2 * WT * by - 1 + u), x = px + d * (2 * WT * bx - 1 + v);
    const bool ok = wave < W_RAW_INSTR && r < WNPX && y >= 0 && y < g.H && x >= 0 && x < g.W;
    a_v = ok ? (unsigned)(((((long)img * g.H + y) * g.W + x) * g.Cin + half * 8) * 2) : W_INVALID;
  }
  unsigned u_v[2];                                               // 32 instructions per sub-chunk, two per wave: rows xi * 64 + co
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = wave + 16 * s;
    const int row = q * 32 + (lane >> 1), slot = lane & 1;
    const int xi = row >> 6, co = row & 63;
    const int src = slot ^ w_swz(co);
    u_v[s] = (co0 + co < g.Cout) ? (unsigned)((((long)xi * g.Cout + co0 + co) * g.Cin + src * 8) * 2) : W_INVALID;
  }
  auto issue_raw = [&](int sub) {
    if (wave < W_RAW_INSTR)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lvoid)(sRAW + (sub & 1) * W_RAW_BYTES + wave * 1024), 16, (int)a_v,
                                               (int)((unsigned)sub * (WCK * 2)), 0, 0);
  };
  auto issue_u = [&](int sub) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (lvoid)(sU + (sub & 1) * W_U_BYTES + (wave + 16 * s) * 1024), 16, (int)u_v[s],
                                               (int)((unsigned)sub * (WCK * 2)), 0, 0);
  };

  // ---- transform side (threads 0 .. 511): thread = (tile t, channel pair cp of the sub-chunk's 8)
  const int t_tile = (tid >> 3) & 63, cp = tid & 7;
  const int tty = t_tile >> 3, ttx = t_tile & 7;
  // patch pixel (2 tty + a, 2 ttx + b) sits in row (2 tty + a) * 18 + (b & 1) * 9 + ttx + (b >> 1): a constant distance from the
  // tile's first pixel for every (a, b)
  const int raw_base = ((2 * tty) * WPS + ttx) * 32 + cp * 4;
  const int v_off = t_tile * 32 + (((cp >> 2) ^ w_swz(t_tile)) << 4) + (cp & 3) * 4;       // + xi * 2048
  auto transform = [&](int sub) {
    const char* const R = sRAW + (sub & 1) * W_RAW_BYTES + raw_base;
    char* const Vw = sV + (sub & 1) * W_V_BYTES + v_off;
    float lo[4][4], hi[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned int w = *(const unsigned int*)(R + (a * WPS + (b & 1) * 9 + (b >> 1)) * 32);
        lo[a][b] = __uint_as_float(w << 16); hi[a][b] = __uint_as_float(w & 0xFFFF0000u);
      }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float l0 = lo[0][b] - lo[2][b], l1 = lo[1][b] + lo[2][b], l2 = lo[2][b] - lo[1][b], l3 = lo[1][b] - lo[3][b];
      lo[0][b] = l0; lo[1][b] = l1; lo[2][b] = l2; lo[3][b] = l3;
      const float h0 = hi[0][b] - hi[2][b], h1 = hi[1][b] + hi[2][b], h2 = hi[2][b] - hi[1][b], h3 = hi[1][b] - hi[3][b];
      hi[0][b] = h0; hi[1][b] = h1; hi[2][b] = h2; hi[3][b] = h3;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float l0 = lo[a][0] - lo[a][2], l1 = lo[a][1] + lo[a][2], l2 = lo[a][2] - lo[a][1], l3 = lo[a][1] - lo[a][3];
      const float h0 = hi[a][0] - hi[a][2], h1 = hi[a][1] + hi[a][2], h2 = hi[a][2] - hi[a][1], h3 = hi[a][1] - hi[a][3];
      char* dst = Vw + (a * 4) * 2048;
      *(unsigned int*)(dst) = pack_bf16x2(l0, h0);
      *(unsigned int*)(dst + 2048) = pack_bf16x2(l1, h1);
      *(unsigned int*)(dst + 4096) = pack_bf16x2(l2, h2);
      *(unsigned int*)(dst + 6144) = pack_bf16x2(l3, h3);
    }
  };

  // ---- MFMA side: wave = transform position xi; 32x32x16: lane (row = lane & 31, K half = lane >> 5)
  const int l31 = lane & 31, kh = lane >> 5;
  const int frag_off = wave * 2048 + l31 * 32 + ((kh ^ w_swz(l31)) << 4);                   // + block * 1024 (32 rows of 32 B)
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nsub = g.Cin / WCK;
  issue_raw(0);
  issue_u(0);
  if (nsub > 1) issue_raw(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < 512) transform(0);
  for (int p = 0; p < nsub; ++p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // own pieces of U(p) and patch(p + 1), issued a phase ago
    __syncthreads();                                            // V[p & 1] written (LDS drained), U(p), patch(p + 1) complete; phase p - 1's readers are done
    if (p + 1 < nsub) issue_u(p + 1);
    if (p + 2 < nsub) issue_raw(p + 2);
    {
      const char* const Vb = sV + (p & 1) * W_V_BYTES + frag_off;
      const char* const Ub = sU + (p & 1) * W_U_BYTES + frag_off;
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(Vb + i * 1024);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *(const u32x4*)(Ub + j * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                              acc[i][j], 0, 0, 0);
    }
    if (p + 1 < nsub && tid < 512) transform(p + 1);
  }
  // ---- output transform: the 16 positions of a (tile, channel) meet through LDS, 32 channels at a time
  float* const X = (float*)smem;                                 // [xi][tile 64][32 channels] f32 = 128 KiB
  unsigned short* const out = (unsigned short*)g.out;
  const unsigned short* const ref = (const unsigned short*)g.ref;
  const int o_tile = tid >> 4, ocp = tid & 15;                   // output side: thread = (tile, channel pair of the half's 16)
  const int oty = o_tile >> 3, otx = o_tile & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                             // h = 0: every wave is done with V / U; h = 1: X has been read
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e)                               // C/D map of 32x32: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
        X[(wave * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + l31] = acc[i][h][e];
    __syncthreads();
    float m[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      const float2 v2 = *(const float2*)(X + (xi * 64 + o_tile) * 32 + 2 * ocp);
      m[xi][0] = v2.x; m[xi][1] = v2.y;
    }
    const int co = co0 + 32 * h + 2 * ocp;
    float bv[2] = {0.f, 0.f};
    if (g.bias && co < g.Cout) { bv[0] = g.bias[co]; bv[1] = g.bias[co + 1]; }
    float y[2][2][2];                                            // [a][b][channel]
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      float s[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i][0] = (m[4 * i + 0][ch] + m[4 * i + 1][ch]) + m[4 * i + 2][ch];
        s[i][1] = (m[4 * i + 1][ch] - m[4 * i + 2][ch]) - m[4 * i + 3][ch];
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        y[0][b][ch] = (s[0][b] + s[1][b]) + s[2][b];
        y[1][b][ch] = (s[1][b] - s[2][b]) - s[3][b];
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int yy = py + d * (2 * WT * by + 2 * oty + a), xx = px + d * (2 * WT * bx + 2 * otx + b);
        if (yy < g.H && xx < g.W && co < g.Cout) {
          float v0 = y[a][b][0] + bv[0], v1 = y[a][b][1] + bv[1];
          if (g.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          const long o = (((long)img * g.H + yy) * g.W + xx) * g.Cout + co;
          unsigned int pk = pack_bf16x2(v0, v1);
          if (ref) {
            const unsigned int r = *(const unsigned int*)(ref + o);
            const bool lo_on = __uint_as_float(r << 16) > 0.f, hi_on = __uint_as_float(r & 0xFFFF0000u) > 0.f;
            pk = (lo_on ? (pk & 0xFFFFu) : 0u) | (hi_on ? (pk & 0xFFFF0000u) : 0u);
          }
          *(unsigned int*)(out + o) = pk;
        }
      }
  }
}

// U[xi = 4 i + j][out][in] = (G g G^T)[i][j] in bf16 from the f32 OIHW master; mode 1: the data gradient's filters
// g'[out = ci][in = co][ky][kx] = g[co][ci][2 - ky][2 - kx].  One thread per (out, in) pair, `in` fastest (coalesced writes).
struct WinoPrep { const float* w; unsigned short* U; int Cout, Cin, mode; unsigned first_block; };
constexpr int WINO_PREP_MAX = 24;
struct WinoPrepMulti { int n; unsigned total_blocks; WinoPrep p[WINO_PREP_MAX]; };

__global__ __launch_bounds__(256) void winograd_weight_prep_kernel(WinoPrepMulti m) {
  int k = 0;
  while (k + 1 < m.n && blockIdx.x >= m.p[k + 1].first_block) ++k;
  const WinoPrep q = m.p[k];
  const int n_out = q.mode == 0 ? q.Cout : q.Cin, n_in = q.mode == 0 ? q.Cin : q.Cout;
  const long idx = (long)(blockIdx.x - q.first_block) * 256 + threadIdx.x;
  if (idx >= (long)n_out * n_in) return;
  const int o = (int)(idx / n_in), i = (int)(idx - (long)o * n_in);
  float gk[3][3];
  const float* src = q.mode == 0 ? q.w + ((long)o * q.Cin + i) * 9 : q.w + ((long)i * q.Cin + o) * 9;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) gk[a][b] = q.mode == 0 ? src[a * 3 + b] : src[(2 - a) * 3 + (2 - b)];
  float t[4][3];                                              // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = gk[0][b];
    t[1][b] = __fmul_rn(0.5f, __fadd_rn(__fadd_rn(gk[0][b], gk[1][b]), gk[2][b]));
    t[2][b] = __fmul_rn(0.5f, __fadd_rn(__fsub_rn(gk[0][b], gk[1][b]), gk[2][b]));
    t[3][b] = gk[2][b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u0 = t[a][0];
    const float u1 = __fmul_rn(0.5f, __fadd_rn(__fadd_rn(t[a][0], t[a][1]), t[a][2]));
    const float u2 = __fmul_rn(0.5f, __fadd_rn(__fsub_rn(t[a][0], t[a][1]), t[a][2]));
    const float u3 = t[a][2];
    const long base = ((long)(a * 4) * n_out + o) * n_in + i;
    const long st = (long)n_out * n_in;
    q.U[base] = f32_to_bf16_bits(u0); q.U[base + st] = f32_to_bf16_bits(u1);
    q.U[base + 2 * st] = f32_to_bf16_bits(u2); q.U[base + 3 * st] = f32_to_bf16_bits(u3);
  }
}

}  // namespace

// Returns 1 if the Winograd kernel took the launch, 0 if the shape / epilogue is not covered (caller falls back), < 0 on error.
extern "C" int sw_conv3x3_winograd(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* in, const void* U,
                                   void* out, const sw_epilogue* ep, hipStream_t stream) {
  SW_ENTER();
  if (dtype != SW_BF16 || !ep || (Cin % 32) || (Cout % 2) || (dilation != 1 && dilation != 2) || nimg < 1 || H < 1 || W < 1) return 0;
  if (ep->out_dtype != SW_BF16 || ep->drop_mask || ep->accumulate_atomic || ep->absmax_out || ep->drop_hash_p > 0.f || ep->residual) return 0;
  if (ep->relu_ref && (ep->ref_dtype != SW_BF16 || ep->ld_ref != Cout || ep->ref_scale != 1.0f)) return 0;
  if ((((uintptr_t)in | (uintptr_t)U | (uintptr_t)out | (uintptr_t)ep->relu_ref) & 15)) return 0;
  WinoArgs g = {};
  g.in = in; g.U = U; g.out = out; g.bias = ep->bias; g.ref = ep->relu_ref; g.relu = ep->relu;
  g.nimg = nimg; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.dil = dilation;
  const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;      // the largest parity class' sub-grid
  g.bly = (hs + 2 * WT - 1) / (2 * WT); g.blx = (ws + 2 * WT - 1) / (2 * WT);
  g.n_px_blocks = nimg * dilation * dilation * g.bly * g.blx;
  g.n_co_blocks = (Cout + WTN - 1) / WTN;
  g.total = g.n_px_blocks * g.n_co_blocks;
  const long ib = (long)nimg * H * W * Cin * 2, ub = (long)16 * Cout * Cin * 2;
  if (ib >= 0xFFFFFF00L || ub >= 0xFFFFFF00L) return 0;
  g.in_bytes = (unsigned)ib; g.u_bytes = (unsigned)ub;
  const size_t lds = (size_t)W_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_winograd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    attr_set = true;
  }
  const int per_xcd = (g.total + 7) / 8;
  hipLaunchKernelGGL(conv3x3_winograd_kernel, dim3(per_xcd * 8), dim3(WNT), lds, stream, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}

// Transformed filters of n layers in one launch.  w: f32 OIHW master (Cout, Cin, 3, 3); U: 16 * Cout * Cin bf16.
extern "C" int sw_winograd_weight_prep(int n, const sw_winograd_prep* descs, hipStream_t stream) {
  SW_ENTER();
  for (int i0 = 0; i0 < n; i0 += WINO_PREP_MAX) {
    WinoPrepMulti m = {};
    m.n = n - i0 < WINO_PREP_MAX ? n - i0 : WINO_PREP_MAX;
    unsigned blocks = 0;
    for (int i = 0; i < m.n; ++i) {
      const sw_winograd_prep& d = descs[i0 + i];
      if (!d.w || !d.U || d.Cout < 1 || d.Cin < 1 || (d.mode != 0 && d.mode != 1)) return -5;
      m.p[i] = WinoPrep{d.w, (unsigned short*)d.U, d.Cout, d.Cin, d.mode, blocks};
      blocks += (unsigned)(((long)d.Cout * d.Cin + 255) / 256);
    }
    m.total_blocks = blocks;
    if (blocks) hipLaunchKernelGGL(winograd_weight_prep_kernel, dim3(blocks), dim3(256), 0, stream, m);
    SW_CHECK_LAUNCH();
  }
  return 0;
}
