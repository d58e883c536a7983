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
constexpr int WCK = 32, WTN = 64, WNT = 1024;                  // channels per chunk, output channels per workgroup, threads
constexpr int W_RAW_INSTR = (WNPX + 15) / 16;                  // 1 KiB LDS-DMA instructions per patch chunk (21)
constexpr int W_RAW_BYTES = 32 * 1024;                         // 2 instructions per wave x 16 waves (21 carry pixels, the rest zero fill)
constexpr int W_V_BYTES = 16 * 64 * 64, W_U_BYTES = 16 * WTN * 64;
constexpr unsigned W_INVALID = 0xFFFFFF00u;

struct WinoArgs {
  const void* in; const void* U; void* out;
  const float* bias; const void* ref;
  int nimg, H, W, Cin, Cout, relu, dil;
  int bly, blx, n_px_blocks, n_co_blocks, total;
  unsigned in_bytes, u_bytes;
};

__device__ __forceinline__ int w_swz(int row) { return ((row >> 2) & 1) << 1; }     // 16-byte slot XOR (conv_direct.hip)

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {          // ONE v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
  return __builtin_bit_cast(unsigned int, v);
}

__global__ __launch_bounds__(WNT) void conv3x3_winograd_kernel(WinoArgs g) {
  typedef __attribute__((address_space(3))) void* lvoid;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sRAW = smem;
  char* const sV = smem + W_RAW_BYTES;
  char* const sU = sV + W_V_BYTES;
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
  // horizontally adjacent tiles (v and v + 2) are adjacent rows, so a wave's 4 tiles x 16 channel pairs read 256 contiguous bytes
  unsigned a_v[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = wave + 16 * s;
    const int r = q * 16 + (lane >> 2), slot = lane & 3;
    const int u = r / WPS, pv = r - u * WPS;
    const int v = pv < 9 ? 2 * pv : 2 * (pv - 9) + 1;
    const int y = py + d * (2 * WT * by - 1 + u), x = px + d * (2 * WT * bx - 1 + v);
    const bool ok = r < WNPX && y >= 0 && y < g.H && x >= 0 && x < g.W;      // (instructions 21 .. 31 carry no pixel: zero fill)
    a_v[s] = ok ? (unsigned)(((((long)img * g.H + y) * g.W + x) * g.Cin + slot * 8) * 2) : W_INVALID;
  }
  // U is staged in two halves of 32 output channels (32 instructions of 1 KiB each, two per wave): while the MFMAs of one half
  // run, the other half's buffer is being refilled — a whole-chunk buffer could only be refilled between two chunks, with the
  // MFMAs waiting for it (44 vs 38 us for the direct kernel at conv5_3)
  unsigned u_v[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = wave + 16 * s;                            // 32 instructions per half: xi = q >> 1, 16 channel rows each
      const int xi = q >> 1, co = hf * 32 + (q & 1) * 16 + (lane >> 2), slot = lane & 3;
      const int src = slot ^ w_swz(co);
      u_v[hf][s] = (co0 + co < g.Cout) ? (unsigned)((((long)xi * g.Cout + co0 + co) * g.Cin + src * 8) * 2) : W_INVALID;
    }
  auto issue_raw = [&](int chunk) {                           // every wave issues 2 (uniform vmcnt accounting); surplus ones: zero fill
    const unsigned soff = (unsigned)chunk * (WCK * 2);
#pragma unroll
    for (int s = 0; s < 2; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lvoid)(sRAW + (wave + 16 * s) * 1024), 16, (int)a_v[s], (int)soff, 0, 0);
  };
  auto issue_u = [&](int chunk, int hf) {                     // LDS image of a half: [xi][32 channels][64 B]
    const unsigned soff = (unsigned)chunk * (WCK * 2);
#pragma unroll
    for (int s = 0; s < 2; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (lvoid)(sU + hf * (W_U_BYTES / 2) + (wave + 16 * s) * 1024), 16, (int)u_v[hf][s],
                                               (int)soff, 0, 0);
  };

  // ---- transform side: thread = (tile t, channel pair cp)
  const int t_tile = tid >> 4, cp = tid & 15;
  const int tty = t_tile >> 3, ttx = t_tile & 7;
  // patch pixel (2 tty + a, 2 ttx + b) sits in row (2 tty + a) * 18 + (b & 1) * 9 + ttx + (b >> 1): a constant distance from the
  // tile's first pixel for every (a, b)
  const char* const raw_base = sRAW + ((2 * tty) * WPS + ttx) * 64 + cp * 4;
  const int v_off = t_tile * 64 + (((cp >> 2) ^ w_swz(t_tile)) << 4) + (cp & 3) * 4;      // + xi * 4096

  // ---- MFMA side: wave = transform position xi
  const int l15 = lane & 15, kq = lane >> 4;
  const int frag_off = l15 * 64 + ((kq ^ w_swz(l15)) << 4);                                // + block * 1024 (16 rows of 64 B)
  const char* const Vx = sV + wave * 4096;
  const char* const Ux = sU + wave * 2048;                       // + half * 32 KiB, + block * 1024
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunk = g.Cin / WCK;
  auto mma_half = [&](int hf) {
    u32x4 fa[4], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *(const u32x4*)(Vx + i * 1024 + frag_off);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *(const u32x4*)(Ux + hf * (W_U_BYTES / 2) + j * 1024 + frag_off);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][2 * hf + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                                     acc[i][2 * hf + j], 0, 0, 0);
  };
  // DMA in flight, oldest first, at each wait (2 instructions per wave each):
  //   B1  patch(c) | U_lo(c)                 -> vmcnt(2): patch(c) landed;       then issue U_hi(c)
  //   B2  U_lo(c) | U_hi(c)                  -> vmcnt(2): U_lo(c) landed;        then issue patch(c + 1)
  //   B3  U_hi(c) | patch(c + 1)             -> vmcnt(2): U_hi(c) landed;        then issue U_lo(c + 1)
  issue_raw(0);
  issue_u(0, 0);
  for (int c = 0; c < nchunk; ++c) {
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // B1: patch(c) complete; every wave is done with M_hi(c - 1): V and U_hi are free
    issue_u(c, 1);
    {
      // T: B^T d B in f32 for two channels; rows first (index a), then columns (index b)
      float lo[4][4], hi[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned int w = *(const unsigned int*)(raw_base + (a * WPS + (b & 1) * 9 + (b >> 1)) * 64);
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
        char* dst = sV + (a * 4) * 4096 + v_off;
        *(unsigned int*)(dst) = pack_bf16x2(l0, h0);
        *(unsigned int*)(dst + 4096) = pack_bf16x2(l1, h1);
        *(unsigned int*)(dst + 8192) = pack_bf16x2(l2, h2);
        *(unsigned int*)(dst + 12288) = pack_bf16x2(l3, h3);
      }
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __syncthreads();                                            // B2: V complete (LDS writes drained), U_lo(c) complete, the patch buffer is free
    if (c + 1 < nchunk) issue_raw(c + 1);
    mma_half(0);
    if (c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // B3: every wave is done with U_lo(c); U_hi(c) complete
    if (c + 1 < nchunk) issue_u(c + 1, 0);
    mma_half(1);
  }
  // ---- output transform: the 16 positions of a (tile, channel) meet through LDS, 32 channels at a time
  float* const X = (float*)smem;                                 // [xi][tile 64][32 channels] f32 = 128 KiB
  unsigned short* const out = (unsigned short*)g.out;
  const unsigned short* const ref = (const unsigned short*)g.ref;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                                             // h = 0: every wave is done with V / U; h = 1: X has been read
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          X[(wave * 64 + i * 16 + kq * 4 + e) * 32 + jj * 16 + l15] = acc[i][2 * h + jj][e];
    __syncthreads();
    float m[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      const float2 v2 = *(const float2*)(X + (xi * 64 + t_tile) * 32 + 2 * cp);
      m[xi][0] = v2.x; m[xi][1] = v2.y;
    }
    const int co = co0 + 32 * h + 2 * cp;
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
        const int yy = py + d * (2 * WT * by + 2 * tty + a), xx = px + d * (2 * WT * bx + 2 * ttx + b);
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
  if (dtype != SW_BF16 || !ep || (Cin % WCK) || (Cout % 2) || (dilation != 1 && dilation != 2) || nimg < 1 || H < 1 || W < 1) return 0;
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
  const size_t lds = (size_t)W_RAW_BYTES + W_V_BYTES + W_U_BYTES;
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
