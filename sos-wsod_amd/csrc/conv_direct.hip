// Direct 3x3 convolution (stride 1, pad = dilation, NHWC, bf16) for gfx950: forward and data gradient of conv1_2..conv5_3.
//
// Why a second conv kernel.  The implicit GEMM (gemm.hip) re-fetches every input pixel once per tap: a 128x128 output tile
// streams 128 x 9*Cin input elements + 128 x 9*Cin weights through the CU.  With 63x63 / 64x64 maps a launch has only
// ~250 tiles, so tiles cannot grow, and the measured limit is the XCD's L2: conv5_3 issues 4.7 M 128-byte L2 reads per
// launch (600 MB in 54 us = 11 TB/s, TCC busy 73 %, profiles/) while the MFMA pipe idles at 28 %.
// Here a workgroup owns an 8 x 32 pixel patch of ONE image x 64 output channels and walks the input channels in chunks of
// 32: the (8+2d) x (32+2d) input patch of a chunk is staged ONCE and serves all 9 taps as shifted row windows of the same
// LDS image; the weights of a chunk are staged one tap row (3 taps) at a time.  Per 256 x 64 outputs the CU now pulls
// 432 x Cin + 64 x 9 Cin elements instead of 2 x 128 x 9 Cin + ... : 2.3x fewer L2 bytes, and the grid is still 256
// workgroups for a 64 x 64 x 512 layer.
//
// Structure: 4 waves, wave w = tile rows 2w, 2w+1 (64 pixels) x 64 channels = 4 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16,
// one MFMA consumes the whole 32-channel chunk of a tap).  LDS: input patch double buffered per chunk, weights double
// buffered per (chunk, tap row); both staged with LDS-DMA buffer loads whose per-lane offsets are fixed for the whole
// kernel (image border / ragged edge = offset beyond num_records -> zero fill), the chunk / tap row advance is the
// wave-uniform SGPR offset.  One barrier per 3 taps (48 MFMAs per wave).  79.9 KiB LDS -> two workgroups per CU.
// 64-byte LDS rows (one pixel or one output channel x 32 input channels): 16-byte position = K chunk ^ (((row >> 2) & 1) << 1).
// ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md): with this
// XOR the 16 lanes of every group hit 64 distinct banks for ANY start row, i.e. for every tap shift (found by exhaustive
// search; the obvious (row >> 2) & 3 measured 45 % of the LDS cycles as bank conflicts).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "soswsod_hip.h"

namespace {

constexpr int TH = 8, TW = 32, CK = 32;                   // pixel tile, input channels per chunk (output channels: template TN)
constexpr unsigned INVALID = 0xFFFFFF00u;

struct DirectArgs {
  const void* in; const void* wk; void* out;
  const float* bias; const void* ref;
  int nimg, H, W, Cin, Cout, relu;
  int pool;                            // 1: `out` is the 2x2 / stride-2 max-pooled map [nimg][(H-2)/2+1][(W-2)/2+1][Cout] (KG = 1 form only)
  int out_f32;                         // 1: `out` (and `ref`, if any) hold f32: the bf16x3 convolutions of the fp32 mode (six-product form,
                                       //    Cin = 6 x the layer's channels): the accumulators are stored as they are
  int tiles_x, tiles_y, n_px_tiles, n_co_blocks, total;
  unsigned in_bytes, wk_bytes;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (Deeper weight rings — 3 / 4 tap-row buffers with counted vmcnt — were tried for the one-workgroup-per-CU layers and
// dropped: equal or slower than two waves per SIMD, and the K-group form below needs the LDS.)
//
// KG = 2 (the 63x63 / 64x64 layers, conv4_x / conv5_x at a 512x512 input): EIGHT waves, two groups of four; group k walks the
// input-channel chunks 2c + k with its own staging buffers (exactly the four-wave kernel, on alternate chunks, sharing the
// barriers), and the two partial accumulators are exchanged through LDS at the end (group k finishes tile row 2w + k of its
// wave pair).  Those layers have only ~256 tiles of 256 pixels x 64 channels: one four-wave workgroup per CU leaves one wave
// per SIMD (nothing to overlap LDS latency with), 32-channel tiles give two waves per SIMD but read 0.75 KiB of LDS per MFMA
// (6 fragments per 8 MFMAs).  Splitting K inside the workgroup gives two waves per SIMD at 0.5 KiB per MFMA and 30 % fewer bytes
// staged per output.  160 KiB LDS at dilation 2.  Counters of the conv5_3 launch (tools/probes/pmc_kernel.sh conv5_3): MFMA pipe 51 % busy,
// LDS 27 % busy with 0 bank-conflict cycles, 2.4 M non-MFMA VALU instructions against 2.36 M MFMAs — with ONE workgroup per CU the
// ~7 us of prologue (first patch + weights from HBM) and epilogue (partial-sum exchange, store) of the 38 us launch overlap with
// nothing; inside the training step the second backbone stream's launches fill them (two concurrent launches: 62 us for both).
template <int DIL, int TN, int KG = 1>
__device__ __forceinline__ void conv3x3_direct_body(const DirectArgs& g, const unsigned bid) {
  static_assert(KG == 1 || (KG == 2 && TN == 64), "K groups: two groups of four waves, 64-channel tiles");
  constexpr int DEPTH = 2;                                  // weight tap-row buffers
  constexpr int NI = TN / 16;                              // MFMA tiles along the output channels
  constexpr int B_INSTR = 3 * TN * 4 / 64, B_PER_WAVE = (B_INSTR + 3) / 4;
  constexpr int PH = TH + 2 * DIL, PW = TW + 2 * DIL, P = PH * PW;
  constexpr int A_INSTR = (P * 4 + 63) / 64;               // 1 KiB LDS-DMA instructions per input patch
  constexpr int A_PER_WAVE = (A_INSTR + 3) / 4;             // every wave issues this many (uniform vmcnt accounting);
  constexpr int A_BYTES = A_PER_WAVE * 4 * 1024;            //   the surplus ones zero-fill the padding of the image
  constexpr int B_BYTES = 3 * TN * 64;                      // one tap row: 3 taps x 64 channels x 64 B
  typedef __attribute__((address_space(3))) void* lvoid;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // A[2] | B[2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KG == 1 ? 0 : (wave_all >> 2), wave = wave_all & 3;     // K group, wave inside the group (= pixel rows 2w, 2w+1)
  char* const sA = smem + grp * (2 * A_BYTES);             // per group: A[2]
  char* const sB = smem + KG * 2 * A_BYTES + grp * (DEPTH * B_BYTES);     // per group: [DEPTH] tap-row buffers

  // workgroup -> (pixel tile, channel block): XCD x (= blockIdx & 7) takes a contiguous run of the work list in which the
  // channel block runs fastest, so the 8 channel blocks of a pixel tile share its input patch in that XCD's L2
  const int per_xcd = (g.total + 7) >> 3;
  const int f = (bid & 7) * per_xcd + (bid >> 3);
  if ((int)(bid >> 3) >= per_xcd || f >= g.total) return;
  const int px_tile = f / g.n_co_blocks, co_blk = f - px_tile * g.n_co_blocks;
  const int img = px_tile / (g.tiles_x * g.tiles_y);
  const int trem = px_tile - img * (g.tiles_x * g.tiles_y);
  const int ty0 = (trem / g.tiles_x) * TH, tx0 = (trem % g.tiles_x) * TW;
  const int co0 = co_blk * TN;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.in, 0, (int)g.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.wk, 0, (int)g.wk_bytes, 0x00020000);

  // ---- fixed per-lane DMA offsets
  unsigned a_v[A_PER_WAVE];
#pragma unroll
  for (int s = 0; s < A_PER_WAVE; ++s) {
    const int q = wave + 4 * s;                             // instruction index inside the patch image
    const int idx = q * 64 + lane;
    const int prow = idx >> 2, p = idx & 3;
    const int src = p ^ (((prow >> 2) & 1) << 1);                  // LDS position p of row prow holds source chunk src
    const int py = prow / PW, px = prow - py * PW;
    const int y = ty0 - DIL + py, x = tx0 - DIL + px;
    const bool ok = prow < P && y >= 0 && y < g.H && x >= 0 && x < g.W;
    a_v[s] = ok ? (unsigned)(((((long)img * g.H + y) * g.W + x) * g.Cin + src * 8) * 2) : INVALID;
  }
  unsigned b_v[B_PER_WAVE];
#pragma unroll
  for (int s = 0; s < B_PER_WAVE; ++s) {
    const int idx = (wave + 4 * s) * 64 + lane;             // row r = tap_x * TN + co, position p
    const int r = idx >> 2, p = idx & 3;
    const int tap_x = r / TN, co = r - tap_x * TN;
    const int src = p ^ (((co >> 2) & 1) << 1);
    const bool ok = tap_x < 3 && co0 + co < g.Cout;
    b_v[s] = ok ? (unsigned)((((long)(co0 + co) * 9 + tap_x) * g.Cin + src * 8) * 2) : INVALID;
  }
  auto issue_a = [&](int chunk) {                           // chunk = the group's local chunk index
    char* dst = sA + (chunk & 1) * A_BYTES;
    const unsigned soff = (unsigned)(chunk * KG + grp) * (CK * 2);
#pragma unroll
    for (int s = 0; s < A_PER_WAVE; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lvoid)(dst + (wave + 4 * s) * 1024), 16, (int)a_v[s], (int)soff, 0, 0);
  };
  auto issue_b = [&](int step) {                            // step = chunk * 3 + tap row
    const int chunk = step / 3, ty = step - chunk * 3;
    char* dst = sB + (step % DEPTH) * B_BYTES;
    const unsigned soff = (unsigned)((ty * 3 * g.Cin + (chunk * KG + grp) * CK) * 2);
#pragma unroll
    for (int s = 0; s < B_PER_WAVE; ++s)
      if (wave + 4 * s < B_INSTR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lvoid)(dst + (wave + 4 * s) * 1024), 16, (int)b_v[s], (int)soff, 0, 0);
  };

  f32x4 acc[4][NI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunk = g.Cin / (CK * KG), nstep = nchunk * 3;     // per group
  const int l15 = lane & 15, kq = lane >> 4;                // fragment row, 16-byte K position
  // B fragment byte offsets inside a tap (fixed): channel j*16 + l15
  int b_off[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = j * 16 + l15;
    b_off[j] = co * 64 + ((kq ^ (((co >> 2) & 1) << 1)) << 4);
  }
  // Per step (= 3 taps of one chunk): all loads of the step landed (issued one step earlier) -> barrier -> issue the next
  // step's weights (and, opening a chunk, the next chunk's patch) -> 3 x (4 + NI fragment reads, 4 x NI MFMAs); the
  // compiler interleaves the reads of a tap with the MFMAs of the previous one.  (An explicit one-tap-ahead register
  // pipeline with the barrier moved into the last tap was measured 4-12 % slower: 256 VGPRs, longer dependency stalls; round 6: the
  // fragments of tap tx + 1 requested before the MFMAs of tap tx inside a step, sched_barrier-fenced, 170 VGPRs: every layer of the
  // 512x512 and 800x1333 views within +-2 %.  On the 100x166 maps the loop runs at 0.54 of peak per ISSUED MFMA; what the true-FLOP
  // figure loses there is the 8x32 tiling of the map (1.20x), not the loop — and not the round count either: whole rounds of 64-channel tiles + a launch of
  // 32-channel tiles for the remaining pixel tiles ran slower, 133 -> 143 us at 106x141, 140 -> 150 at 100x166.)
  issue_a(0);
  issue_b(0);
  // Ragged right / bottom edge (round 6): a tile whose columns tx0 + 16 .. tx0 + 31 all lie outside the image runs the LEFT form of the
  // loop — the 16-pixel sub-tiles i = 1, 3 are neither read nor multiplied (half the MFMAs of the tile: a 166-pixel-wide map issues 5.5
  // tile columns instead of 6) — and a wave whose two rows lie below the image runs the EMPTY form (staging and barriers only; its
  // SIMD's matrix pipe goes to the other workgroup of the CU).  Uniform per wave; the full tile's code is unchanged.
  constexpr int FORM_FULL = 0, FORM_LEFT = 1, FORM_EMPTY = 2;
  // Fragment addresses without vector arithmetic in the loop (round 6; the counters of the conv5 launch showed 0.8 non-MFMA vector
  // instructions per MFMA — two address adds per input fragment — on an issue port the MFMAs of two waves already fill to 13 of 16
  // cycles).  A patch row of the fragment (tap, sub-tile) is u + K with u = 2 * wave * PW + l15 per lane and K a compile-time constant;
  // the XOR bit of its 16-byte position is bit 2 of (u + K), which depends on K only through K mod 8: EIGHT per-lane base pointers
  // (four at dilation 2, where K is even), and every read is `base[K mod 8] + immediate`, the buffer parity of the chunk / step included
  // (the chunk loop runs two chunks per turn).
  const char* pa[8];
  {
    const int u = 2 * wave * PW + l15;
#pragma unroll
    for (int k = 0; k < 8; ++k) pa[k] = sA + u * 64 + ((kq ^ ((((u + k) >> 2) & 1) << 1)) << 4);
  }
  const char* pb[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) pb[j] = sB + b_off[j];
  auto main_loop = [&](auto form_c) {
    constexpr int FORM = decltype(form_c)::value;
    auto compute_step = [&](auto par_c, auto ty_c) {           // par = chunk & 1 (compile time), ty = tap row
      constexpr int PAR = decltype(par_c)::value, ty = decltype(ty_c)::value;
      constexpr int BPAR = (PAR * 3 + ty) & 1;                  // step & 1 for step = chunk * 3 + ty
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        u32x4 fa[4], fb[NI];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (FORM == FORM_LEFT && (i & 1)) continue;
          const int K = ((i >> 1) + ty * DIL) * PW + (i & 1) * 16 + tx * DIL;
          fa[i] = *(const u32x4*)(pa[K & 7] + (PAR * A_BYTES + K * 64));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = *(const u32x4*)(pb[j] + (BPAR * B_BYTES + tx * (TN * 64)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (FORM == FORM_LEFT && (i & 1)) continue;
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                                acc[i][j], 0, 0, 0);
        }
      }
    };
    auto chunk_steps = [&](auto par_c, int chunk) {
      auto one = [&](auto ty_c) {
        constexpr int ty = decltype(ty_c)::value;
        const int step = chunk * 3 + ty;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (step + 1 < nstep) issue_b(step + 1);
        if (ty == 0 && chunk + 1 < nchunk) issue_a(chunk + 1);
        if (FORM != FORM_EMPTY) compute_step(par_c, ty_c);
      };
      one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
    };
    for (int chunk = 0; chunk < nchunk; chunk += 2) {
      chunk_steps(std::integral_constant<int, 0>{}, chunk);
      if (chunk + 1 < nchunk) chunk_steps(std::integral_constant<int, 1>{}, chunk + 1);
    }
  };
  {
    static_assert(TW == 32 && TH == 8, "edge forms: 16-pixel sub-tiles, two rows per wave");
    const bool left_only = tx0 + 16 >= g.W;                  // workgroup-uniform
    const bool no_rows = ty0 + 2 * wave >= g.H;              // wave-uniform
    if (no_rows) main_loop(std::integral_constant<int, FORM_EMPTY>{});
    else if (left_only) main_loop(std::integral_constant<int, FORM_LEFT>{});
    else main_loop(std::integral_constant<int, FORM_FULL>{});
  }
  __syncthreads();                                          // every wave is done with the staging buffers

  unsigned short* out = (unsigned short*)g.out;
  const unsigned short* ref = (const unsigned short*)g.ref;
  if (KG == 2) {
    // ---- two K groups: exchange halves through LDS.  Wave (grp, w) keeps the 16-pixel sub-tiles i = 2*grp, 2*grp+1 (tile row
    // 2w + grp) and hands the other two to its partner wave (1 - grp, w).
    f32x4* X = (f32x4*)smem;                                  // [8 waves][2][NI][64 lanes], 8 KiB per wave
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const f32x4 give = grp == 0 ? acc[2 + i2][j] : acc[i2][j];
        X[((wave_all * 2 + i2) * NI + j) * 64 + lane] = give;
      }
    __syncthreads();
    f32x4 fin[2][NI];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const f32x4 got = X[(((wave_all ^ 4) * 2 + i2) * NI + j) * 64 + lane];
        const f32x4 mine = grp == 0 ? acc[i2][j] : acc[2 + i2][j];
        // group 0's partial first: the sum does not depend on which wave finishes it
        fin[i2][j] = grp == 0 ? f32x4{mine[0] + got[0], mine[1] + got[1], mine[2] + got[2], mine[3] + got[3]}
                              : f32x4{got[0] + mine[0], got[1] + mine[1], got[2] + mine[2], got[3] + mine[3]};
      }
    if (g.out_f32) {
      // f32 output: straight from the accumulator layout (row = kq * 4 + e of the 16-pixel sub-tile, column = l15): 16 lanes write
      // 64 contiguous bytes of a pixel; the launch runs 6x the K of a bf16 layer, the 4-byte stores do not show
      float* outf = (float*)g.out;
      const float* reff = (const float*)g.ref;
      const int y = ty0 + 2 * wave + grp;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int co = co0 + j * 16 + l15;
        const float bv = (g.bias && co < g.Cout) ? g.bias[co] : 0.f;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int x = tx0 + i2 * 16 + kq * 4 + e;
            if (y < g.H && x < g.W && co < g.Cout) {
              float v = fin[i2][j][e] + bv;
              if (g.relu) v = fmaxf(v, 0.f);
              const long o = (((long)img * g.H + y) * g.W + x) * g.Cout + co;
              if (reff && !(reff[o] > 0.f)) v = 0.f;
              outf[o] = v;
            }
          }
      }
      return;
    }
    unsigned short* S = (unsigned short*)(smem + 8 * 2 * NI * 64 * 16 + wave_all * (32 * TN * 2));      // [32 pixels][TN]
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = j * 16 + l15;
      const float bv = (g.bias && co0 + col < g.Cout) ? g.bias[co0 + col] : 0.f;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = fin[i2][j][e] + bv;
          if (g.relu) v = fmaxf(v, 0.f);
          const int pl = i2 * 16 + kq * 4 + e;
          S[pl * TN + (col ^ ((pl & (TN / 8 - 1)) << 3))] = f32_to_bf16_bits(v);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int it = 0; it < TN / 16; ++it) {
      const int idx = it * 64 + lane;
      const int pl = idx / (TN / 8), ch = idx % (TN / 8);
      const int y = ty0 + 2 * wave + grp, x = tx0 + pl;
      const int co = co0 + ch * 8;
      if (y < g.H && x < g.W && co < g.Cout) {
        u32x4 v = *(const u32x4*)(S + pl * TN + ((ch ^ (pl & (TN / 8 - 1))) << 3));
        const long o = (((long)img * g.H + y) * g.W + x) * g.Cout + co;
        if (ref) {
          const u32x4 r = *(const u32x4*)(ref + o);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool lo = __uint_as_float(r[t] << 16) > 0.f, hi = __uint_as_float(r[t] & 0xFFFF0000u) > 0.f;
            v[t] = (lo ? (v[t] & 0xFFFFu) : 0u) | (hi ? (v[t] & 0xFFFF0000u) : 0u);
          }
        }
        *(u32x4*)(out + o) = v;
      }
    }
    return;
  }
  if (g.out_f32) {                                            // (see the K-group form above)
    float* outf = (float*)g.out;
    const float* reff = (const float*)g.ref;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int co = co0 + j * 16 + l15;
      const float bv = (g.bias && co < g.Cout) ? g.bias[co] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pl = i * 16 + kq * 4 + e;
          const int y = ty0 + 2 * wave + (pl >> 5), x = tx0 + (pl & 31);
          if (y < g.H && x < g.W && co < g.Cout) {
            float v = acc[i][j][e] + bv;
            if (g.relu) v = fmaxf(v, 0.f);
            const long o = (((long)img * g.H + y) * g.W + x) * g.Cout + co;
            if (reff && !(reff[o] > 0.f)) v = 0.f;
            outf[o] = v;
          }
        }
    }
    return;
  }
  // ---- epilogue: bias / ReLU in registers, bf16 tile through LDS, 16-byte stores with the ReLU-backward mask
  unsigned short* S = (unsigned short*)(smem + wave * (64 * TN * 2));        // [64 pixels][TN channels]
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = j * 16 + l15;
    const float bv = (g.bias && co0 + col < g.Cout) ? g.bias[co0 + col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[i][j][e] + bv;
        if (g.relu) v = fmaxf(v, 0.f);
        const int pl = i * 16 + kq * 4 + e;                 // C/D map: row = (lane>>4)*4 + e, col = lane&15
        S[pl * TN + (col ^ ((pl & (TN / 8 - 1)) << 3))] = f32_to_bf16_bits(v);      // 16-byte groups XOR-swizzled by the row
      }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): own LDS writes visible to own reads (same wave)
  if (g.pool) {
    // 2x2 / stride-2 max pool inside the epilogue (frozen layers: nobody reads the unpooled map).  The wave's 64 pixels are the image
    // rows 2w, 2w+1 of the tile x 32 columns = 16 whole windows (tile origins are multiples of 8 x 32): a lane takes the windows'
    // 16-byte channel groups, maximum in the order of sw_maxpool2x2_fwd ((y,x), (y,x+1), (y+1,x), (y+1,x+1)) on the bf16 values the
    // unfused pair would have stored and re-read: identical bits.  A window that does not fit the image (odd H / W) has no output.
    const int OH = (g.H - 2) / 2 + 1, OW = (g.W - 2) / 2 + 1;
#pragma unroll
    for (int it = 0; it < (16 * (TN / 8) + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      const int pp = idx / (TN / 8), ch = idx % (TN / 8);
      const int oy = (ty0 + 2 * wave) >> 1, ox = (tx0 >> 1) + pp;
      const int co = co0 + ch * 8;
      if (pp < 16 && oy < OH && ox < OW && co < g.Cout) {
        float a[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int pl = (t >> 1) * 32 + 2 * pp + (t & 1);
          const u32x4 v = *(const u32x4*)(S + pl * TN + ((ch ^ (pl & (TN / 8 - 1))) << 3));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = __uint_as_float(v[q] << 16), hi = __uint_as_float(v[q] & 0xFFFF0000u);
            a[2 * q] = t == 0 ? lo : fmaxf(a[2 * q], lo);
            a[2 * q + 1] = t == 0 ? hi : fmaxf(a[2 * q + 1], hi);
          }
        }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (unsigned)f32_to_bf16_bits(a[2 * q]) | ((unsigned)f32_to_bf16_bits(a[2 * q + 1]) << 16);
        *(u32x4*)(out + (((long)img * OH + oy) * OW + ox) * g.Cout + co) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < TN / 8; ++it) {
    const int idx = it * 64 + lane;
    const int pl = idx / (TN / 8), ch = idx % (TN / 8);     // pixel of this wave's 64, 16-byte channel group
    const int y = ty0 + 2 * wave + (pl >> 5), x = tx0 + (pl & 31);
    const int co = co0 + ch * 8;
    if (y < g.H && x < g.W && co < g.Cout) {
      u32x4 v = *(const u32x4*)(S + pl * TN + ((ch ^ (pl & (TN / 8 - 1))) << 3));
      const long o = (((long)img * g.H + y) * g.W + x) * g.Cout + co;
      if (ref) {
        const u32x4 r = *(const u32x4*)(ref + o);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool lo = __uint_as_float(r[t] << 16) > 0.f, hi = __uint_as_float(r[t] & 0xFFFF0000u) > 0.f;
          v[t] = (lo ? (v[t] & 0xFFFFu) : 0u) | (hi ? (v[t] & 0xFFFF0000u) : 0u);
        }
      }
      *(u32x4*)(out + o) = v;
    }
  }
}

template <int DIL, int TN, int KG = 1>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void conv3x3_direct_kernel(DirectArgs g) {
  conv3x3_direct_body<DIL, TN, KG>(g, blockIdx.x);
}

// Several convolutions (different maps, possibly different weights) as ONE launch: the FPN levels of a detector — the RPN head's
// shared 3x3 convolution on p2..p6, the four FPN output convolutions — are independent launches of which only the finest map fills
// the chip; side by side the small ones run in its shadow.  Workgroup ranges per problem are multiples of 8 (XCD map of the body).
constexpr int DIRECT_MULTI_MAX = 8;
struct DirectMulti {
  int n;
  unsigned first[DIRECT_MULTI_MAX + 1];
  DirectArgs p[DIRECT_MULTI_MAX];
};
__global__ __launch_bounds__(256, 2) void conv3x3_direct_multi_kernel(DirectMulti m) {
  int i = 0;
  while (i + 1 < m.n && blockIdx.x >= m.first[i + 1]) ++i;
  conv3x3_direct_body<1, 64, 1>(m.p[i], blockIdx.x - m.first[i]);
}

// First layer (conv1_1: 3 real input channels padded to 8, 64 output channels, forward only).  As an implicit GEMM its K is
// 72 and nearly all time went into per-chunk tap arithmetic of the generic loader (100 us for 2 x 512 x 512 pixels whose
// output alone is 67 MB = 13 us of HBM).  Here K = 9 taps x 8 channels is laid out as 3 MFMA K-steps of 4 taps: the A
// fragment of (16 pixels, K-step s) is ONE 16-byte load per lane — the 8 channels of the pixel shifted by tap 4s + (lane>>4)
// (taps 9..11 and out-of-image pixels read as zero) — straight from global memory (the 4 MB input sits in L2), no LDS for
// the operands; the 12 weight fragments live in registers for the whole kernel.  A wave computes 64 pixels x 64 channels
// per iteration and writes them as 128-byte pixel rows through a private 8 KiB LDS tile.
__global__ __launch_bounds__(256) void conv3x3_first_kernel(int nimg, int H, int W, const unsigned short* __restrict__ in,
                                                            const unsigned short* __restrict__ wk,      // [64][9][8]
                                                            const float* __restrict__ bias, int relu,
                                                            unsigned short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short S_all[4][64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  unsigned short* S = S_all[wave];
  u32x4 fb[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int tap = s3 * 4 + kq;
      fb[j][s3] = tap < 9 ? *(const u32x4*)(wk + ((long)(j * 16 + l15) * 9 + tap) * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = bias ? bias[j * 16 + l15] : 0.f;
  int dy[3], dx[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) { const int tap = s3 * 4 + kq; dy[s3] = tap / 3 - 1; dx[s3] = tap % 3 - 1; }
  const int segs = (W + 63) / 64;                            // 64-pixel row segments
  const long nseg = (long)nimg * H * segs;
  for (long sg = (long)blockIdx.x * 4 + wave; sg < nseg; sg += (long)gridDim.x * 4) {
    const int xs = (int)(sg % segs) * 64; const long t = sg / segs;
    const int y = (int)(t % H), img = (int)(t / H);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = xs + i * 16 + l15;
      u32x4 fa[3];
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int yy = y + dy[s3], xx = x + dx[s3];
        const bool ok = s3 * 4 + kq < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W;
        fa[s3] = ok ? *(const u32x4*)(in + (((long)img * H + yy) * W + xx) * 8) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[s3]), __builtin_bit_cast(bf16x8, fb[j][s3]),
                                                              acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e] + bv[j];
          if (relu) v = fmaxf(v, 0.f);
          const int pl = i * 16 + kq * 4 + e, col = j * 16 + l15;
          S[pl * 64 + (col ^ ((pl & 7) << 3))] = f32_to_bf16_bits(v);
        }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane;
      const int pl = idx >> 3, ch = idx & 7;
      const int x = xs + pl;
      if (x < W)
        *(u32x4*)(out + (((long)img * H + y) * W + x) * 64 + ch * 8) = *(const u32x4*)(S + pl * 64 + ((ch ^ (pl & 7)) << 3));
    }
  }
}

}  // namespace

// Returns 1 if the direct kernel took the launch, 0 if the shape is not covered (caller falls back), < 0 on error.
int sw_conv3x3_direct_try(int nimg, int H, int W, int Cin, int Cout, int dilation, const void* in, const void* wk, void* out,
                          const sw_epilogue* ep, hipStream_t stream) {
  static const char* sw = getenv("SW_CONV_DIRECT");          // development switch: "0" off, "1" every covered shape
  if (sw && sw[0] == '0') return 0;
  if (Cin == 8 && Cout == 64 && dilation == 1 && ep && ep->out_dtype == SW_BF16 && !ep->drop_mask && !ep->relu_ref &&
      !ep->accumulate_atomic && !ep->absmax_out && !(ep->drop_hash_p > 0.f) && !(sw && sw[0] == '2') &&
      (((uintptr_t)in | (uintptr_t)wk | (uintptr_t)out) & 15) == 0) {
    const long nseg = (long)nimg * H * ((W + 63) / 64);
    long blocks = (nseg + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_first_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, nimg, H, W, (const unsigned short*)in,
                       (const unsigned short*)wk, ep->bias, ep->relu, (unsigned short*)out);
    hipError_t e1 = hipGetLastError();
    return e1 == hipSuccess ? 1 : -(int)e1;
  }
  if ((Cin % CK) || (Cout % 8) || (dilation != 1 && dilation != 2)) return 0;
  if (ep && ((ep->out_dtype != SW_BF16 && ep->out_dtype != SW_F32) || ep->drop_mask || ep->accumulate_atomic || ep->absmax_out)) return 0;
  if (!ep) return 0;
  // the ReLU-backward reference in the output's type (bf16 maps: bf16; the fp32 mode's bf16x3 convolutions: f32)
  if (ep->relu_ref && (ep->ref_dtype != ep->out_dtype || ep->ld_ref != Cout || ep->ref_scale != 1.0f)) return 0;
  if (ep->residual || ep->drop_hash_p > 0.f) return 0;
  if ((((uintptr_t)in | (uintptr_t)wk | (uintptr_t)out | (uintptr_t)ep->relu_ref) & 15)) return 0;
  if (!(sw && sw[0] == '1') && Cin < 64) return 0;           // (Cin % 32 == 0 leaves only conv1_1 to the implicit GEMM)
  DirectArgs g = {};
  g.in = in; g.wk = wk; g.out = out; g.bias = ep->bias; g.ref = ep->relu_ref; g.relu = ep->relu;
  g.out_f32 = ep->out_dtype == SW_F32;
  g.nimg = nimg; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
  g.tiles_x = (W + TW - 1) / TW; g.tiles_y = (H + TH - 1) / TH;
  g.n_px_tiles = g.tiles_x * g.tiles_y * nimg;
  // 64 output channels per workgroup, or 32 when that leaves less than 1.5 workgroups per CU (63x63 / 64x64 maps): twice the
  // workgroups = two per CU = two waves per SIMD to overlap LDS latency with MFMAs
  static const char* tsw = getenv("SW_CONV_DIRECT_TN");       // development switch
  static const char* ksw = getenv("SW_CONV_DIRECT_KG");       // development switch: "1" = never split K inside the workgroup
  const bool few = g.n_px_tiles * ((Cout + 63) / 64) <= 384;
  // few tiles: two K groups of four waves on 64-channel tiles (header); needs an even number of 32-channel chunks
  const int kg = ((few || (ksw && ksw[0] == '2')) && !tsw && !(ksw && ksw[0] == '1') && (Cin % (2 * CK)) == 0) ? 2 : 1;
  const int tn = tsw ? atoi(tsw) : ((few && kg == 1) ? 32 : 64);
  g.n_co_blocks = (Cout + tn - 1) / tn;
  g.total = g.n_px_tiles * g.n_co_blocks;
  const long ib = (long)nimg * H * W * Cin * 2, wb = (long)Cout * 9 * Cin * 2;
  if (ib >= 0xFFFFFF00L || wb >= 0xFFFFFF00L) return 0;
  g.in_bytes = (unsigned)ib; g.wk_bytes = (unsigned)wb;
  const int per_xcd = (g.total + 7) / 8;
  const int d = dilation;
  const int P = (TH + 2 * d) * (TW + 2 * d);
  const int apw = ((P * 4 + 63) / 64 + 3) / 4;
  const size_t lds = (size_t)kg * ((size_t)2 * apw * 4096 + (size_t)2 * (3 * tn * 64));
  hipError_t e = hipSuccess;
#define SW_LAUNCH_DIRECT(D, TNV, KGV)                                                                                     \
  do {                                                                                                                     \
    auto kern = conv3x3_direct_kernel<D, TNV, KGV>;                                                                        \
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    if (e != hipSuccess) return -(int)e;                                                                                   \
    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(256 * kg), lds, stream, g);                                           \
  } while (0)
  if (kg == 2) { if (d == 1) SW_LAUNCH_DIRECT(1, 64, 2); else SW_LAUNCH_DIRECT(2, 64, 2); }
  else if (tn == 32) { if (d == 1) SW_LAUNCH_DIRECT(1, 32, 1); else SW_LAUNCH_DIRECT(2, 32, 1); }
  else if (tn == 64) { if (d == 1) SW_LAUNCH_DIRECT(1, 64, 1); else SW_LAUNCH_DIRECT(2, 64, 1); }
  else return 0;
#undef SW_LAUNCH_DIRECT
  e = hipGetLastError();
  if (e != hipSuccess) return -(int)e;
  return 1;
}


// 3x3 convolution (stride 1, dilation 1, bf16) + bias + ReLU + 2x2 / stride-2 max pool in ONE launch: `out` is the pooled map only.  For
// layers nobody differentiates through (the frozen conv1_2 / conv2_2 of the VGG16 backbone, vgg.py:104-122 with FREEZE_AT 2): the
// unpooled map (67 MB per 512x512 view pair at conv1_2) is never written or read back.  Returns 1 = launched, 0 = shape not covered
// (the caller runs sw_conv3x3_igemm + sw_maxpool2x2_fwd), < 0 on error.  Results identical to that pair, bit for bit.
extern "C" int sw_conv3x3_relu_pool2(int dtype, int nimg, int H, int W, int Cin, int Cout, const void* in, const void* wk, const float* bias,
                                     void* out_pooled, hipStream_t stream) {
  SW_ENTER();
  static const char* sw = getenv("SW_CONV_DIRECT");
  static const char* fsw = getenv("SW_CONV_POOL_FUSED");      // development switch: "0" = never
  if ((sw && sw[0] == '0') || (fsw && fsw[0] == '0')) return 0;
  if (dtype != SW_BF16 || (Cin % CK) || Cin < 64 || (Cout % 64) || H < 2 || W < 2) return 0;
  if ((((uintptr_t)in | (uintptr_t)wk | (uintptr_t)out_pooled) & 15)) return 0;
  DirectArgs g = {};
  g.in = in; g.wk = wk; g.out = out_pooled; g.bias = bias; g.ref = nullptr; g.relu = 1; g.pool = 1;
  g.nimg = nimg; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
  g.tiles_x = (W + TW - 1) / TW; g.tiles_y = (H + TH - 1) / TH;
  g.n_px_tiles = g.tiles_x * g.tiles_y * nimg;
  g.n_co_blocks = Cout / 64;
  g.total = g.n_px_tiles * g.n_co_blocks;
  // few tiles: sw_conv3x3_igemm would split K inside the workgroup or take 32-channel tiles (another order of additions): not covered,
  // so that the fused launch always equals the unfused pair bit for bit
  if (g.total <= 384) return 0;
  const long ib = (long)nimg * H * W * Cin * 2, wb = (long)Cout * 9 * Cin * 2;
  if (ib >= 0xFFFFFF00L || wb >= 0xFFFFFF00L) return 0;
  g.in_bytes = (unsigned)ib; g.wk_bytes = (unsigned)wb;
  const int per_xcd = (g.total + 7) / 8;
  constexpr int P = (TH + 2) * (TW + 2);
  constexpr int apw = ((P * 4 + 63) / 64 + 3) / 4;
  const size_t lds = (size_t)2 * apw * 4096 + (size_t)2 * (3 * 64 * 64);
  auto kern = conv3x3_direct_kernel<1, 64, 1>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return -(int)e;
  hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(256), lds, stream, g);
  e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}


// n stride-1, dilation-1 bf16 convolutions in one launch (see conv3x3_direct_multi_kernel).  Returns 1 if launched, 0 if some problem is
// not covered by the direct kernel (the caller then launches them one by one), < 0 on error.
extern "C" int sw_conv3x3_multi(int dtype, int n, const sw_conv_problem* probs, hipStream_t stream) {
  SW_ENTER();
  if (n <= 0) return 1;
  if (dtype != SW_BF16 || n > DIRECT_MULTI_MAX) return 0;
  static const char* sw = getenv("SW_CONV_DIRECT");
  if (sw && sw[0] == '0') return 0;
  DirectMulti m = {};
  m.n = n;
  unsigned wgs = 0;
  for (int i = 0; i < n; ++i) {
    const sw_conv_problem& q = probs[i];
    const sw_epilogue* ep = q.ep;
    if (!ep || q.nimg <= 0 || q.H <= 0 || q.W <= 0) return 0;
    if ((q.Cin % CK) || q.Cin < 64 || (q.Cout % 8)) return 0;
    if (ep->out_dtype != SW_BF16 || ep->drop_mask || ep->accumulate_atomic || ep->absmax_out || (ep->drop_hash_p > 0.f) || ep->residual) return 0;
    if (ep->relu_ref && (ep->ref_dtype != SW_BF16 || ep->ld_ref != q.Cout || ep->ref_scale != 1.0f)) return 0;
    if ((((uintptr_t)q.in | (uintptr_t)q.wk | (uintptr_t)q.out | (uintptr_t)ep->relu_ref) & 15)) return 0;
    DirectArgs& g = m.p[i];
    g.in = q.in; g.wk = q.wk; g.out = q.out; g.bias = ep->bias; g.ref = ep->relu_ref; g.relu = ep->relu;
    g.nimg = q.nimg; g.H = q.H; g.W = q.W; g.Cin = q.Cin; g.Cout = q.Cout;
    g.tiles_x = (q.W + TW - 1) / TW; g.tiles_y = (q.H + TH - 1) / TH;
    g.n_px_tiles = g.tiles_x * g.tiles_y * q.nimg;
    g.n_co_blocks = (q.Cout + 63) / 64;
    g.total = g.n_px_tiles * g.n_co_blocks;
    const long ib = (long)q.nimg * q.H * q.W * q.Cin * 2, wb = (long)q.Cout * 9 * q.Cin * 2;
    if (ib >= 0xFFFFFF00L || wb >= 0xFFFFFF00L) return 0;
    g.in_bytes = (unsigned)ib; g.wk_bytes = (unsigned)wb;
    m.first[i] = wgs;
    wgs += (unsigned)(((g.total + 7) / 8) * 8);
  }
  m.first[n] = wgs;
  constexpr int P = (TH + 2) * (TW + 2);
  constexpr int apw = ((P * 4 + 63) / 64 + 3) / 4;
  constexpr size_t lds = (size_t)2 * apw * 4096 + (size_t)2 * (3 * 64 * 64);
  hipError_t e = hipFuncSetAttribute((const void*)conv3x3_direct_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return -(int)e;
  hipLaunchKernelGGL(conv3x3_direct_multi_kernel, dim3(wgs), dim3(256), lds, stream, m);
  e = hipGetLastError();
  if (e != hipSuccess) return -(int)e;
  return 1;
}
