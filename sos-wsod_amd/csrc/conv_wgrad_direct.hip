// Direct 3x3 convolution WEIGHT GRADIENT (stride 1, pad = dilation, NHWC, bf16) for gfx950 — round 6.
//
//   dW[co][tap][ci] = sum over pixels (img, y, x) of  dY[img][y][x][co] * X[img][y + (ty-1) d][x + (tx-1) d][ci]       (zero outside the map)
//   (reference: autograd's conv2d backward of vgg.py:104-122's convolutions; the slab / fold contract of sw_conv3x3_wgrad_grouped)
//
// Why a second weight-gradient kernel.  As an implicit GEMM (gemm.hip, OP_CONV_B) the reduction runs over flat pixels and the N axis is
// (tap, ci): every input pixel is gathered NINE times per K-tile row (once per tap column block), each gather slot re-derives its pixel's
// (y, x) bounds per K-tile, and both operands are read back transposed by 16 waves that cannot hold a fragment ahead — measured 0.30-0.40 of
// the MFMA peak, counter traffic 1.86x the operands.  Here, as in the forward kernel (conv_direct.hip), a row of input pixels is staged ONCE
// and serves all nine taps as shifted windows of the same LDS image:
//
//   * a workgroup (4 waves, two workgroups per CU) owns a block of 64 output channels x 64 input channels x all 9 taps: wave w = input-channel
//     quarter w, 9 taps x 4 output-channel sub-tiles of 16 x 16 = 144 accumulator registers per lane;
//   * it walks a vertical strip of the map, 32 pixels wide, one image row per step (K = 32 pixels = one v_mfma_f32_16x16x32_bf16 K extent):
//     per step ONE new input row segment of (32 + 2d) pixels x 64 channels (4.6 KB) joins a ring of 8 rows in LDS — the rows y-d .. y+d of
//     the step are the three tap rows — and one dY row segment of 32 pixels x 64 channels (4 KB) a ring of 4;
//   * both operands have K (the pixel) as the slow index of their LDS image (128-byte pixel rows): fragments are read with
//     ds_read_b64_tr_b16 (two per fragment); 32-byte channel pairs are XOR-swizzled by pixel bits 1 and 3, so that the 8 pixel rows a 32-lane
//     half touches hit distinct banks for every tap shift (exhaustive check offline; SQ_LDS_BANK_CONFLICT = 0 measured);
//   * per step and wave: 4 dY fragments + 9 input fragments (26 transposed reads) feed 36 MFMAs; an input fragment is requested two taps
//     before its MFMAs, the next step's dY fragments and first two input fragments during taps 4-8; LDS-DMA loads run three steps ahead
//     (counted vmcnt), one barrier per step, placed after tap 3 — no wave opens a step by waiting for LDS;
//   * work items = (problem, pixel split, block): the step list (image, strip, row) of a problem is cut into `nsplit` ranges, each range
//     x block writes its partial [64][9][64] tile into the range's slab with plain stores — the slab layout and the ordered fold of the
//     implicit-GEMM path (sw_conv3x3_wgrad_fold*) are unchanged, results are deterministic.
// History of the round (profiles/r06_wgrad_*): one 8-wave workgroup per CU (128 x 64 blocks) ran 0.21-0.29 of peak — the two waves of a SIMD
// sat in the same phase of the same program; two independent 4-wave workgroups 0.30-0.39; the compiler's s_waitcnt vmcnt(0) in front of the
// first transposed read after every DMA issue (dma16 below) was the rest: 0.45-0.50 of peak by the layers' true FLOP, MFMA pipe 63 % busy.
#include <stdlib.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {

constexpr int CO_B = 64, CI_B = 64;
constexpr int XSLOT = 40 * 128;                      // one input row segment: up to 36 pixels x 128 B, padded to 5 DMA instructions
constexpr int NSLOT = 8;                             // input rows in the ring
constexpr int DYB = 32 * 128, NDY = 4;               // one dY row segment: 32 pixels x 128 B
constexpr int WD_LDS = NSLOT * XSLOT + NDY * DYB;    // 57 344 B: two workgroups per CU
constexpr unsigned INVALID = 0xFFFFFF00u;
constexpr int WD_MAX = 32;

struct WdProblem {
  const void* dy; const void* x; float* slabs;
  int nimg, H, W, Cin, Cout, dil;
  int strips, steps, per_split, nsplit, co_blocks, ci_blocks;
  unsigned dy_bytes, x_bytes;
};
struct WdArgs {
  int n_problems, n_items;
  int first_item[WD_MAX + 1];
  WdProblem p[WD_MAX];
};

typedef __attribute__((address_space(3))) void* lvoid;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ u32x2 tr_read(const char* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
__device__ __forceinline__ int fsw(int pix) { return ((pix >> 1) & 1) | (((pix >> 3) & 1) << 1); }      // 128-byte rows: XOR of the 32-byte pair

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DIL>
__device__ __forceinline__ void wd_item(const WdProblem& P, const int co_blk, const int ci_blk, const int z, char* const smem) {
  constexpr int PW = 32 + 2 * DIL, NPRIME = 2 * DIL + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = input-channel quarter of the block
  const int G = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
  char* const sX = smem;
  char* const sD = smem + NSLOT * XSLOT;
  const int H = P.H, W = P.W, Cin = P.Cin, Cout = P.Cout;
  const int co0 = co_blk * CO_B, ci0 = ci_blk * CI_B;

  // ---- this item's step range (step = one image row of one 32-pixel strip of one image; rows fastest); boundaries are moved off the
  // first / last two rows of a strip: no run (consecutive rows of one strip) of 1-2 steps
  auto bound = [&](int b) {
    if (b >= P.steps) return P.steps;
    const int m = b % H;
    if (m != 0 && m < 3) b -= m; else if (m > H - 3) b += H - m;
    return b;
  };
  const int k0 = bound(z * P.per_split), k1 = bound((z + 1) * P.per_split);

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses (fixed per lane): rows of both LDS images are pixels of 128 B (64 channels), pair p' = p ^ fsw(pixel)
  int a_addr[4], b_addr[3][2];
  {
    const int row0 = 8 * G + q4;
#pragma unroll
    for (int j = 0; j < 4; ++j) a_addr[j] = row0 * 128 + ((j ^ fsw(row0)) << 5) + p4 * 8;       // second read: + 4 pixels = + 512 (same XOR)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pix = tx * DIL + row0 + 4 * h;
        b_addr[tx][h] = pix * 128 + ((wave ^ fsw(pix)) << 5) + p4 * 8;
      }
  }
  auto read_a = [&](const char* D, int j) {
    const u32x2 lo = tr_read(D + a_addr[j]), hi = tr_read(D + a_addr[j] + 512);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  auto read_b = [&](const char* X, int tx) {
    const u32x2 lo = tr_read(X + b_addr[tx][0]), hi = tr_read(X + b_addr[tx][1]);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  // ---- DMA lane geometry: instruction q of a row image, LDS chunk q*64 + lane = (pixel pr, physical chunk pc) <- logical chunk lc
  auto lane_pix = [&](int q) { return (q * 64 + lane) >> 3; };
  auto lane_lc = [&](int q) { const int pr = (q * 64 + lane) >> 3, pc = lane & 7; return (((pc >> 1) ^ fsw(pr)) << 1) | (pc & 1); };
  const sw_i32x4 rsX = sw_make_rsrc(P.x, P.x_bytes), rsD = sw_make_rsrc(P.dy, P.dy_bytes);      // (sw_dma16: common.h)
  const unsigned x_row_bytes = (unsigned)(W * Cin * 2), d_row_bytes = (unsigned)(W * Cout * 2);

  int k = k0;
  int img, strip, r0;
  { const int q = k0 / H; r0 = k0 - q * H; img = q / P.strips; strip = q - img * P.strips; }
  while (k < k1) {
    // ================= one run: rows r0 .. r1-1 of (img, strip)
    const int r1 = min(H, r0 + (k1 - k));
    const int c0 = strip * 32;
    const unsigned x_img = (unsigned)(img * H) * x_row_bytes, d_img = (unsigned)(img * H) * d_row_bytes;
    auto x_off = [&](int q) -> unsigned {
      const int pr = lane_pix(q), col = c0 - DIL + pr;
      return (pr < PW && col >= 0 && col < W) ? (unsigned)((col * Cin + ci0 + lane_lc(q) * 8) * 2) : INVALID;
    };
    // steady state: wave w issues input instruction w and dY instruction w of the step's rows, wave 0 also input instruction 4
    const unsigned xo_own = x_off(wave), xo_4 = x_off(4);
    unsigned d_off;
    { const int pr = lane_pix(wave); d_off = (c0 + pr < W) ? (unsigned)(((c0 + pr) * Cout + co0 + lane_lc(wave) * 8) * 2) : INVALID; }
    auto load_x_row = [&](int y, int q, unsigned vo) {           // input row y, instruction q -> its ring slot
      const bool yok = y >= 0 && y < H;
      sw_dma16(rsX, sX + ((y + DIL) & (NSLOT - 1)) * XSLOT + q * 1024, yok ? vo : INVALID, yok ? x_img + (unsigned)y * x_row_bytes : 0u);
    };
    auto load_step = [&](int r) {                                // the one new input row (r + d) and the dY row of step r
      load_x_row(r + DIL, wave, xo_own);
      if (wave == 0) load_x_row(r + DIL, 4, xo_4);
      sw_dma16(rsD, sD + (r & (NDY - 1)) * DYB + wave * 1024, d_off, d_img + (unsigned)r * d_row_bytes);
    };
    __builtin_amdgcn_s_barrier();                                // the previous run's (item's) readers are done with the rings
    // prime: rows r0-d .. r0+d-1 (the step's own load brings r0+d), then steps r0, r0+1, r0+2 in flight
#pragma unroll 1
    for (int j = 0; j < NPRIME - 1; ++j) {
      load_x_row(r0 - DIL + j, wave, xo_own);
      if (wave == 0) load_x_row(r0 - DIL + j, 4, xo_4);
    }
    load_step(r0);
    if (r0 + 1 < r1) load_step(r0 + 1);
    if (r0 + 2 < r1) load_step(r0 + 2);
    {
      const int later = (r0 + 1 < r1) + (r0 + 2 < r1);          // steps whose loads may stay in flight
      if (wave == 0) { if (later == 2) wait_vm<6>(); else if (later == 1) wait_vm<3>(); else wait_vm<0>(); }
      else           { if (later == 2) wait_vm<4>(); else if (later == 1) wait_vm<2>(); else wait_vm<0>(); }
    }
    __builtin_amdgcn_s_barrier();
    u32x4 fa[4], fbq[3];                                        // fbq: the input fragments of taps t, t+1, t+2 (ring; 9 taps = 3 turns)
    {
      const char* const D = sD + (r0 & (NDY - 1)) * DYB;
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = read_a(D, j);
      const char* const X0 = sX + (r0 & (NSLOT - 1)) * XSLOT;
      fbq[0] = read_b(X0, 0);
      fbq[1] = read_b(X0, 1);
    }
    // ---- one step = 9 taps x 4 MFMAs.  A tap's input fragment is requested two taps before its MFMAs.  After tap 3: "the loads of step
    // r+1 landed" + barrier (every wave is past step r-1 and the first tap row of step r) + the DMA of step r+3; behind it the fragments
    // of step r+1 (dY: taps 4..7, first input taps: 7, 8) are read — no wave opens a step by waiting for LDS.
    // (two steps per loop turn with the roles of fa / fb swapped: no register copies of the prefetched dY fragments)
    auto step = [&](const int r, u32x4 (&fcur)[4], u32x4 (&fnxt)[4]) {
      const char* X[3];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) X[ty] = sX + ((r + ty * DIL) & (NSLOT - 1)) * XSLOT;
      const char* const Xn = sX + ((r + 1) & (NSLOT - 1)) * XSLOT;
      const char* const Dn = sD + ((r + 1) & (NDY - 1)) * DYB;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const u32x4 fbc = fbq[t % 3];
        if (t < 7) fbq[(t + 2) % 3] = read_b(X[(t + 2) / 3], (t + 2) % 3);
        else fbq[(t + 2) % 3] = read_b(Xn, t - 7);              // (past the run's last row: stale bytes, never multiplied)
        if (t >= 4 && t < 8) fnxt[t - 4] = read_a(Dn, t - 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fcur[j]), __builtin_bit_cast(bf16x8, fbc), acc[t][j], 0, 0, 0);
        if (t == 3) {
          // outstanding: the loads of steps r+1 and r+2; r+1's must have landed
          if (r + 2 < r1) { if (wave == 0) wait_vm<3>(); else wait_vm<2>(); }
          else wait_vm<0>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (r + 3 < r1) load_step(r + 3);
        }
      }
    };
    u32x4 fb2[4];
    int r = r0;
    for (; r + 1 < r1; r += 2) { step(r, fa, fb2); step(r + 1, fb2, fa); }
    if (r < r1) step(r, fa, fb2);
    k += r1 - r0;
    r0 = 0;
    if (++strip == P.strips) { strip = 0; ++img; }
  }

  // ---- partial [64][9][64] tile -> this split's slab ([Cout][9][Cin] f32), plain stores
  float* const S = P.slabs + (long)z * Cout * 9 * Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + j * 16 + (lane >> 4) * 4 + e;
        const int ci = ci0 + wave * 16 + (lane & 15);
        S[((long)co * 9 + t) * Cin + ci] = acc[t][j][e];
      }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_direct_kernel(WdArgs wa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // resident workgroup b serves the virtual ids b, b + G, ...; inside a round of G ids XCD x (= id & 7) takes a contiguous run of the item
  // list (problem-major, split-major, blocks fastest): its CUs share one pixel range of dY and X in that XCD's L2
  const int G = gridDim.x;
  for (int vbid = blockIdx.x; vbid < wa.n_items; vbid += G) {
    const int round = vbid / G, within = vbid - round * G;
    const int in_round = min(G, wa.n_items - round * G);
    const int x = within & 7, j = within >> 3;
    const int per = in_round >> 3, extra = in_round & 7;
    if (j >= per + (x < extra ? 1 : 0)) continue;
    const int t = round * G + x * per + min(x, extra) + j;
    int pi = 0;
    while (pi + 1 < wa.n_problems && t >= wa.first_item[pi + 1]) ++pi;
    const WdProblem& P = wa.p[pi];
    const int local = t - wa.first_item[pi];
    const int nblk = P.co_blocks * P.ci_blocks;
    const int z = local / nblk, blk = local - z * nblk;
    const int co_blk = blk % P.co_blocks, ci_blk = blk / P.co_blocks;
    if (P.dil == 2) wd_item<2>(P, co_blk, ci_blk, z, smem);
    else wd_item<1>(P, co_blk, ci_blk, z, smem);
  }
}

}  // namespace

// Returns 1 if the direct kernel took the whole problem list, 0 if some problem is not covered (the caller runs the implicit GEMM), < 0 on
// error.  `eff[i]` = slabs problem i must write (sw_conv3x3_wgrad_workspace_floats / (Cout * 9 * Cin)).
int sw_conv3x3_wgrad_direct_try(int n_problems, const sw_wgrad_problem* problems, const int* eff, hipStream_t stream) {
  static const char* sw = getenv("SW_WGRAD_DIRECT");          // development switch: "0" = never
  if (sw && sw[0] == '0') return 0;
  if (n_problems <= 0) return 1;
  for (int i = 0; i < n_problems; ++i) {
    const sw_wgrad_problem& q = problems[i];
    if ((q.Cout % CO_B) || (q.Cin % CI_B) || (q.dilation != 1 && q.dilation != 2) || q.H < 8 || q.W < 1 || eff[i] < 1) return 0;
    const long steps = (long)q.nimg * ((q.W + 31) / 32) * q.H;
    if ((steps + eff[i] - 1) / eff[i] < 8) return 0;
    if ((((uintptr_t)q.x | (uintptr_t)q.dy | (uintptr_t)q.slabs) & 15)) return 0;
    const long xb = (long)q.nimg * q.H * q.W * q.Cin * 2, db = (long)q.nimg * q.H * q.W * q.Cout * 2;
    if (xb >= 0xFFFFFF00L || db >= 0xFFFFFF00L) return 0;
  }
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
    else ncu = 256;
  }
  hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_direct_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS);
  if (e != hipSuccess) return -(int)e;
  for (int base = 0; base < n_problems; base += WD_MAX) {
    WdArgs wa = {};
    wa.n_problems = n_problems - base < WD_MAX ? n_problems - base : WD_MAX;
    int items = 0;
    for (int i = 0; i < wa.n_problems; ++i) {
      const sw_wgrad_problem& q = problems[base + i];
      WdProblem& P = wa.p[i];
      P.dy = q.dy; P.x = q.x; P.slabs = q.slabs;
      P.nimg = q.nimg; P.H = q.H; P.W = q.W; P.Cin = q.Cin; P.Cout = q.Cout; P.dil = q.dilation;
      P.strips = (q.W + 31) / 32;
      P.steps = q.nimg * P.strips * q.H;
      P.nsplit = eff[base + i];
      P.per_split = (P.steps + P.nsplit - 1) / P.nsplit;
      P.co_blocks = q.Cout / CO_B; P.ci_blocks = q.Cin / CI_B;
      P.x_bytes = (unsigned)((long)q.nimg * q.H * q.W * q.Cin * 2);
      P.dy_bytes = (unsigned)((long)q.nimg * q.H * q.W * q.Cout * 2);
      wa.first_item[i] = items;
      items += P.nsplit * P.co_blocks * P.ci_blocks;
    }
    wa.first_item[wa.n_problems] = items;
    wa.n_items = items;
    int G = ncu - (ncu % 8);
    if (G < 8) G = 8;
    hipLaunchKernelGGL(conv_wgrad_direct_kernel, dim3(2 * G), dim3(256), WD_LDS, stream, wa);
    e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
  }
  return 1;
}
