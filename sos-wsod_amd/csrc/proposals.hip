// Stage-3 detector, the index side on the device: what the reference does with torch sort / nonzero / randperm /
// fancy indexing and a host round trip each —
//   * RPN proposal selection      detectron2/detectron2/modeling/proposal_generator/proposal_utils.py:22-130 (find_top_rpn_proposals:
//                                 per-level top-k by objectness, decode, clip, drop empty) -> sw_rpn_select_pack
//   * RPN anchor labels           .../proposal_generator/rpn.py:305-360 (label_and_sample_anchors) over modeling/matcher.py:60-126 and
//                                 modeling/sampling.py:8-54 (subsample_labels) -> sw_rpn_label_anchors
//   * ROI-head proposal sampling  unbias/ubteacher/modeling/roi_heads/roi_heads.py:324-375 (label_and_sample_proposals) over the same
//                                 matcher / sampler -> sw_roi_label_sample
// Shared machinery: an exact segmented "k smallest (key, index)" selection — three radix passes over 32-bit keys (11 + 11 + 10 bits,
// LDS histograms), ties at the threshold key resolved by ASCENDING INDEX through an ordered two-level compaction — which is both
// torch.sort(descending, stable)[:k] on the objectness logits (key = order-inverted float bits) and
// argsort(random keys, stable)[:num] of the label sampler (key = 24-bit hash of (seed, position in the candidate list)).
// Random keys: u = splitmix64(seed + position) >> 40, the closed form of the oracle's generator (oracle/detgen.py), so that the
// fixtures' sampled sets are reproduced bit for bit; the product draws `seed` per candidate list from its own counter.
#include <float.h>
#include <stdlib.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {

constexpr int CHUNK = 1024;                  // elements per workgroup of the streaming passes
constexpr int NBIN = 2048;                   // 11-bit radix digits
constexpr unsigned KEY_NONE = 0xFFFFFFFFu;   // "not a candidate"

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  unsigned long long z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned hash_key24(unsigned long long seed, unsigned long long pos) {
  return (unsigned)(splitmix64(seed + pos) >> 40);
}
// ascending unsigned order == DESCENDING float order (larger logits first)
__device__ __forceinline__ unsigned desc_key(float v) {
  unsigned b = __float_as_uint(v);
  if (b == 0x80000000u) b = 0u;                      // -0.0 ties with +0.0, as under torch.sort
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0u;    // NaN sorts first in a descending torch.sort (and is then refused / filtered)
  const unsigned mono = (b & 0x80000000u) ? ~b : (b | 0x80000000u);    // ascending float order
  const unsigned k = ~mono;
  return k == KEY_NONE ? KEY_NONE - 1 : k;
}

// ---- segmented selection of the k smallest (key, index) pairs
struct Seg {              // one selection problem: candidates = elements i of [0, n) with keys[off + i] != KEY_NONE
  long off;               // first key of the segment
  long mark_off;          // first mark byte of the segment (sel_write: mark[mark_off + i] = mark_val for selected i)
  int n;
  int k;                  // k >= 0: how many to take; k < 0: take k_dev[segment] (a count computed on the device)
  int mark_val, pad;
};
constexpr int MAX_SEG = 40;
struct SegTab { Seg s[MAX_SEG]; int n_seg; int max_chunks; };      // by value in the kernel arguments: no table upload, no sync
struct SelState {
  unsigned prefix;        // threshold key bits decided so far (after pass 2: the threshold key T)
  unsigned k_rem;         // how many of the candidates whose key == T are taken (lowest indices first)
  unsigned mode;          // 0 = threshold form, 1 = every candidate is selected (segment holds <= k), 2 = none (k == 0)
  unsigned total;         // candidates in the segment
  unsigned n_sel;         // selected count, written by the scan
  unsigned pad[3];
};

__device__ __forceinline__ void digit_of(int pass, int& shift, unsigned& pmask, unsigned& dmask) {
  shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
  pmask = pass == 0 ? 0u : pass == 1 ? 0xFFE00000u : 0xFFFFFC00u;
  dmask = pass == 2 ? 0x3FFu : 0x7FFu;
}

__global__ __launch_bounds__(256) void sel_hist_kernel(SegTab t, const unsigned* __restrict__ keys, const SelState* __restrict__ st,
                                                       unsigned* __restrict__ hist, int pass) {
  __shared__ unsigned h[NBIN];
  const int s = blockIdx.y;
  const Seg sg = t.s[s];
  const long c0 = (long)blockIdx.x * CHUNK;
  if (c0 >= sg.n) return;
  if (pass > 0 && st[s].mode != 0) return;
  for (int i = threadIdx.x; i < NBIN; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned prefix = pass > 0 ? st[s].prefix : 0u;
  int shift; unsigned pmask, dmask;
  digit_of(pass, shift, pmask, dmask);
  for (int i = threadIdx.x; i < CHUNK && c0 + i < sg.n; i += 256) {
    const unsigned k = keys[sg.off + c0 + i];
    if (k != KEY_NONE && (k & pmask) == (prefix & pmask)) atomicAdd(&h[(k >> shift) & dmask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NBIN; i += 256)
    if (h[i]) atomicAdd(&hist[(long)s * NBIN + i], h[i]);
}

// one workgroup per segment: the digit at which the running count reaches the wanted rank; clears the histogram for the next pass
__global__ __launch_bounds__(256) void sel_pick_kernel(SegTab t, const int* __restrict__ k_dev, SelState* __restrict__ st,
                                                       unsigned* __restrict__ hist, int pass) {
  __shared__ unsigned part[256];
  __shared__ unsigned s_digit, s_before, s_total;
  const int s = blockIdx.x;
  unsigned* h = hist + (long)s * NBIN;
  SelState me = st[s];
  if (pass > 0 && me.mode != 0) return;                 // (the histogram was not touched in this pass)
  unsigned loc[8]; unsigned sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { loc[j] = h[threadIdx.x * 8 + j]; sum += loc[j]; }
  part[threadIdx.x] = sum;
  if (threadIdx.x == 0) s_digit = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int i = 0; i < 256; ++i) { const unsigned v = part[i]; part[i] = run; run += v; }
    s_total = run;
  }
  __syncthreads();
  unsigned want = me.k_rem;
  if (pass == 0) {
    const int kk = t.s[s].k >= 0 ? t.s[s].k : k_dev[s];
    want = kk > 0 ? (unsigned)kk : 0u;
    me.total = s_total;
    me.mode = want == 0 ? 2u : (s_total <= want ? 1u : 0u);
    me.prefix = 0; me.k_rem = want;
  }
  if (me.mode == 0) {
    unsigned run = part[threadIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (want > run && want <= run + loc[j]) { s_digit = threadIdx.x * 8 + j; s_before = run; }
      run += loc[j];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (me.mode == 0) {
      int shift; unsigned pmask, dmask;
      digit_of(pass, shift, pmask, dmask);
      me.prefix |= s_digit << shift;
      me.k_rem = want - s_before;
    }
    st[s] = me;
  }
  for (int i = threadIdx.x; i < NBIN; i += 256) h[i] = 0;
}

// per chunk: (# candidates with key < T, # with key == T)  [mode 1: (# candidates, 0); mode 2: (0, 0)]
__global__ __launch_bounds__(256) void sel_count_kernel(SegTab t, const unsigned* __restrict__ keys, const SelState* __restrict__ st,
                                                        unsigned* __restrict__ chunk_cnt) {
  __shared__ unsigned s_lt, s_eq;
  const int s = blockIdx.y;
  const Seg sg = t.s[s];
  const long c0 = (long)blockIdx.x * CHUNK;
  if (c0 >= sg.n) return;
  if (threadIdx.x == 0) { s_lt = 0; s_eq = 0; }
  __syncthreads();
  const SelState me = st[s];
  unsigned lt = 0, eq = 0;
  for (int i = threadIdx.x; i < CHUNK && c0 + i < sg.n; i += 256) {
    const unsigned k = keys[sg.off + c0 + i];
    if (k == KEY_NONE || me.mode == 2) continue;
    if (me.mode == 1 || k < me.prefix) ++lt;
    else if (k == me.prefix) ++eq;
  }
  if (lt) atomicAdd(&s_lt, lt);
  if (eq) atomicAdd(&s_eq, eq);
  __syncthreads();
  if (threadIdx.x == 0) {
    chunk_cnt[((long)s * t.max_chunks + blockIdx.x) * 2] = s_lt;
    chunk_cnt[((long)s * t.max_chunks + blockIdx.x) * 2 + 1] = s_eq;
  }
}

// one thread per segment walks its chunks: (selected before, equal-to-T before) per chunk; ties are taken in index order until k_rem
__global__ void sel_scan_kernel(SegTab t, SelState* __restrict__ st, unsigned* __restrict__ chunk_cnt) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= t.n_seg) return;
  const int nch = (t.s[s].n + CHUNK - 1) / CHUNK;
  SelState me = st[s];
  const unsigned take_eq = me.mode == 0 ? me.k_rem : 0u;
  unsigned eq_run = 0, sel_run = 0;
  for (int c = 0; c < nch; ++c) {
    unsigned* p = chunk_cnt + ((long)s * t.max_chunks + c) * 2;
    const unsigned lt = p[0], eq = p[1];
    const unsigned eq_take = eq_run >= take_eq ? 0u : min(eq, take_eq - eq_run);
    p[0] = sel_run; p[1] = eq_run;
    sel_run += lt + eq_take; eq_run += eq;
  }
  me.n_sel = sel_run;
  st[s] = me;
}

// block-wide exclusive scan of 4 flags per thread (CHUNK = 1024 = 256 threads x 4 consecutive elements)
__device__ __forceinline__ void block_scan4(const unsigned f[4], unsigned out[4], unsigned* sh /*[256]*/) {
  const unsigned mine = f[0] + f[1] + f[2] + f[3];
  sh[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned v = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = sh[threadIdx.x] - mine;
#pragma unroll
  for (int j = 0; j < 4; ++j) { out[j] = run; run += f[j]; }
  __syncthreads();
}

// the selected elements in INDEX order: sel_idx[seg * sel_stride + pos] = index inside the segment; mark[mark_off + i] = mark_val
__global__ __launch_bounds__(256) void sel_write_kernel(SegTab t, const unsigned* __restrict__ keys, const SelState* __restrict__ st,
                                                        const unsigned* __restrict__ chunk_cnt, int* __restrict__ sel_idx, long sel_stride,
                                                        int8_t* __restrict__ mark) {
  __shared__ unsigned sh[256];
  const int s = blockIdx.y;
  const Seg sg = t.s[s];
  const long c0 = (long)blockIdx.x * CHUNK;
  if (c0 >= sg.n) return;
  const SelState me = st[s];
  if (me.mode == 2) return;
  const unsigned sel_before = chunk_cnt[((long)s * t.max_chunks + blockIdx.x) * 2], eq_before = chunk_cnt[((long)s * t.max_chunks + blockIdx.x) * 2 + 1];
  unsigned k[4], feq[4], req[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long i = c0 + threadIdx.x * 4 + j;
    k[j] = i < sg.n ? keys[sg.off + i] : KEY_NONE;
    feq[j] = (me.mode == 0 && k[j] != KEY_NONE && k[j] == me.prefix) ? 1u : 0u;
  }
  block_scan4(feq, req, sh);
  unsigned fsel[4], rsel[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bool sel = false;
    if (k[j] != KEY_NONE) sel = me.mode == 1 ? true : (k[j] < me.prefix || (feq[j] && eq_before + req[j] < me.k_rem));
    fsel[j] = sel ? 1u : 0u;
  }
  block_scan4(fsel, rsel, sh);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (fsel[j]) {
      const long i = c0 + threadIdx.x * 4 + j;
      if (sel_idx) sel_idx[(long)s * sel_stride + sel_before + rsel[j]] = (int)i;
      if (mark) mark[sg.mark_off + i] = (int8_t)sg.mark_val;
    }
}

// workspace of one selection run: [SelState n_seg][hist n_seg x NBIN][chunk counts n_seg x max_chunks x 2]
inline size_t sel_workspace_bytes(int n_seg, int max_chunks) {
  return (size_t)n_seg * (sizeof(SelState) + NBIN * 4 + (size_t)max_chunks * 8) + 256;
}
int sel_run(const SegTab& t, char* ws, const unsigned* keys, const int* k_dev, int* sel_idx, long sel_stride, int8_t* mark,
            SelState** st_out, hipStream_t stream) {
  SelState* st = (SelState*)ws;
  unsigned* hist = (unsigned*)(ws + (size_t)t.n_seg * sizeof(SelState));
  unsigned* chunk_cnt = hist + (size_t)t.n_seg * NBIN;
  hipError_t e = hipMemsetAsync(st, 0, (size_t)t.n_seg * (sizeof(SelState) + NBIN * 4), stream);
  if (e != hipSuccess) return (int)e;
  dim3 grid(t.max_chunks, t.n_seg);
  for (int pass = 0; pass < 3; ++pass) {
    hipLaunchKernelGGL(sel_hist_kernel, grid, dim3(256), 0, stream, t, keys, st, hist, pass);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(t.n_seg), dim3(256), 0, stream, t, k_dev, st, hist, pass);
  }
  hipLaunchKernelGGL(sel_count_kernel, grid, dim3(256), 0, stream, t, keys, st, chunk_cnt);
  hipLaunchKernelGGL(sel_scan_kernel, dim3((t.n_seg + 63) / 64), dim3(64), 0, stream, t, st, chunk_cnt);
  hipLaunchKernelGGL(sel_write_kernel, grid, dim3(256), 0, stream, t, keys, st, chunk_cnt, sel_idx, sel_stride, mark);
  SW_CHECK_LAUNCH();
  if (st_out) *st_out = st;
  return 0;
}

// ---------------------------------------------------------------------------------------------------- boxes
// structures/boxes.py:337-370 pairwise_iou, one pair, every operation rounded to f32 as torch evaluates it
__device__ __forceinline__ float iou_pair(const float4 a, const float4 b) {
  const float a1 = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
  const float a2 = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
  const float w = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
  const float h = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
  const float inter = __fmul_rn(w, h);
  return inter > 0.f ? __fdiv_rn(inter, __fsub_rn(__fadd_rn(a1, a2), inter)) : 0.f;
}

// ---------------------------------------------------------------------------------------------------- RPN proposal selection
struct RpnLevels {
  const float* logits[8];      // per level [N][n_l]
  const float* deltas[8];      // per level [N][n_l][4]
  const float* anchors[8];     // per level [n_l][4]
  int n[8];
  int L, N;
  long img_stride;             // anchors between two images' rows (0: n[l], i.e. each level dense on its own)
};
// keys of every (image, level) segment, segment-major: seg = img * L + level
__global__ void rpn_keys_kernel(RpnLevels lv, SegTab t, unsigned* __restrict__ keys) {
  const int s = blockIdx.y, img = s / lv.L, l = s - img * lv.L;
  const int n = lv.n[l];
  const float* lg = lv.logits[l] + (long)img * (lv.img_stride ? lv.img_stride : n);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[t.s[s].off + i] = desc_key(lg[i]);
}

// One workgroup per (image, level): the selected anchors (index order) sorted by (key, index) — torch.sort(descending, stable) —
// then decoded (box_regression.py:88-116 apply_deltas), and written as the candidate rows detect_postprocess takes: row r of image
// `img` = scores [L + 1] (-inf except column `level`; -inf there too when the clipped box is empty, proposal_utils.py:96-106) and
// boxes [4 L] (the box repeated).  finite[img] &= every selected box / score is finite (:86-94).
__global__ __launch_bounds__(1024) void rpn_sort_pack_kernel(int CAP, RpnLevels lv, SegTab t, const SelState* __restrict__ st,
                                                             const unsigned* __restrict__ keys, const int* __restrict__ sel_idx, long sel_stride,
                                                             int pre_topk, float w0, float w1, float w2, float w3, float scale_clamp,
                                                             const int* __restrict__ img_hw, float* __restrict__ scores, float* __restrict__ boxes,
                                                             int* __restrict__ finite) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];       // [CAP]
  const int s = blockIdx.x, img = s / lv.L, l = s - img * lv.L;
  const int L = lv.L;
  const int n_sel = (int)st[s].n_sel;
  const int n = lv.n[l];
  for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
    unsigned long long v = ~0ull;
    if (i < n_sel) { const int a = sel_idx[(long)s * sel_stride + i]; v = ((unsigned long long)keys[t.s[s].off + a] << 32) | (unsigned)a; }
    sk[i] = v;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= CAP; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = sk[i], b = sk[p];
          const bool up = (i & k2) == 0;
          if ((a > b) == up) { sk[i] = b; sk[p] = a; }
        }
      }
      __syncthreads();
    }
  // rows of this level inside the image's candidate block: levels in order, pre_topk slots each
  const float H = (float)img_hw[2 * img], W = (float)img_hw[2 * img + 1];
  const float ninf = -__uint_as_float(0x7F800000u);
  bool fin = true;
  for (int r = threadIdx.x; r < pre_topk; r += blockDim.x) {
    const long row = ((long)img * L + l) * pre_topk + r;
    float* sc = scores + row * (L + 1);
    float* bx = boxes + row * 4 * L;
    for (int c = 0; c <= L; ++c) sc[c] = ninf;
    if (r >= n_sel) { for (int c = 0; c < 4 * L; ++c) bx[c] = 0.f; continue; }
    const int a = (int)(sk[r] & 0xFFFFFFFFu);
    const long irow = (long)img * (lv.img_stride ? lv.img_stride : n) + a;
    const float* d = lv.deltas[l] + irow * 4;
    const float* an = lv.anchors[l] + (long)a * 4;
    const float logit = lv.logits[l][irow];
    // apply_deltas: widths / heights / centres of the anchor, deltas / weights, clamp dw, dh
    const float aw = __fsub_rn(an[2], an[0]), ah = __fsub_rn(an[3], an[1]);
    const float cx = __fadd_rn(an[0], __fmul_rn(0.5f, aw)), cy = __fadd_rn(an[1], __fmul_rn(0.5f, ah));
    const float dx = __fdiv_rn(d[0], w0), dy = __fdiv_rn(d[1], w1);
    float dw = __fdiv_rn(d[2], w2), dh = __fdiv_rn(d[3], w3);
    dw = dw > scale_clamp ? scale_clamp : dw; dh = dh > scale_clamp ? scale_clamp : dh;      // torch.clamp(max=): NaN stays NaN (fminf drops it)
    const float pcx = __fadd_rn(__fmul_rn(dx, aw), cx), pcy = __fadd_rn(__fmul_rn(dy, ah), cy);
    const float pw = __fmul_rn(expf(dw), aw), ph = __fmul_rn(expf(dh), ah);
    const float x1 = __fsub_rn(pcx, __fmul_rn(0.5f, pw)), y1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
    const float x2 = __fadd_rn(pcx, __fmul_rn(0.5f, pw)), y2 = __fadd_rn(pcy, __fmul_rn(0.5f, ph));
    const bool f4 = (x1 - x1 == 0.f) && (y1 - y1 == 0.f) && (x2 - x2 == 0.f) && (y2 - y2 == 0.f) && (logit - logit == 0.f);
    fin = fin && f4;
    const float cw = __fsub_rn(fminf(fmaxf(x2, 0.f), W), fminf(fmaxf(x1, 0.f), W));
    const float ch = __fsub_rn(fminf(fmaxf(y2, 0.f), H), fminf(fmaxf(y1, 0.f), H));
    sc[l] = (f4 && cw > 0.f && ch > 0.f) ? logit : ninf;
    for (int c = 0; c < L; ++c) { bx[4 * c] = x1; bx[4 * c + 1] = y1; bx[4 * c + 2] = x2; bx[4 * c + 3] = y2; }
  }
  if (!fin) atomicAnd(&finite[img], 0);
}

// ---------------------------------------------------------------------------------------------------- RPN anchor labels
constexpr int LABEL_MAX_IMG = 8;
struct LabelArgs {
  long A; int N;
  const float* anchors;          // [A][4]
  const float* gt;               // [sum G][4]
  int g_off[9]; int g_cnt[8];    // per image rows of gt
  float thr_lo, thr_hi;
  int batch, max_pos;
  unsigned long long seed_pos[8], seed_neg[8];
};
// pass 1: per-gt best IoU over all anchors (matcher.py:112-126 set_low_quality_matches_)
__global__ __launch_bounds__(256) void anchor_best_kernel(LabelArgs g, unsigned* __restrict__ best /*[sum G] float bits*/) {
  const int img = blockIdx.y;
  const int G = g.g_cnt[img];
  if (G == 0) return;
  const float4* an = (const float4*)g.anchors;
  const float4* gt = (const float4*)g.gt + g.g_off[img];
  for (int gi = 0; gi < G; ++gi) {
    const float4 b = gt[gi];
    float m = 0.f;
    for (long a = (long)blockIdx.x * blockDim.x + threadIdx.x; a < g.A; a += (long)gridDim.x * blockDim.x) m = fmaxf(m, iou_pair(b, an[a]));
    m = wave_reduce_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(&best[g.g_off[img] + gi], __float_as_uint(m));
  }
}
// pass 2: label before sampling (1 / 0 / -1), matched gt box, candidate flags; chunk counts of positives / negatives
__global__ __launch_bounds__(256) void anchor_label_kernel(LabelArgs g, const unsigned* __restrict__ best, int8_t* __restrict__ pre /*[N][A]*/,
                                                           float* __restrict__ matched /*[N][A][4]*/, unsigned* __restrict__ list_cnt /*[2N][chunks]*/,
                                                           int max_chunks) {
  __shared__ unsigned s_pos, s_neg;
  const int img = blockIdx.y;
  const int G = g.g_cnt[img];
  if (threadIdx.x == 0) { s_pos = 0; s_neg = 0; }
  __syncthreads();
  const float4* an = (const float4*)g.anchors;
  const float4* gt = (const float4*)g.gt + g.g_off[img];
  unsigned np = 0, nn = 0;
  const long c0 = (long)blockIdx.x * CHUNK;
  for (int i = threadIdx.x; i < CHUNK; i += 256) {
    const long a = c0 + i;
    if (a >= g.A) break;
    const float4 box = an[a];
    float mv = -1.f; int mi = 0; bool low = false;
    for (int gi = 0; gi < G; ++gi) {
      const float v = iou_pair(gt[gi], box);
      if (v > mv) { mv = v; mi = gi; }
      if (__float_as_uint(v) == best[g.g_off[img] + gi]) low = true;       // == the gt's best IoU (0 included: matcher.py:121-126)
    }
    int lab;
    if (G == 0) lab = 0;                                                       // matcher.py:76-85: no gt -> every anchor background
    else {
      lab = mv < g.thr_lo ? 0 : (mv < g.thr_hi ? -1 : 1);
      if (low) lab = 1;
    }
    pre[(long)img * g.A + a] = (int8_t)lab;
    float4 mb = G ? gt[mi] : make_float4(0.f, 0.f, 0.f, 0.f);
    ((float4*)matched)[(long)img * g.A + a] = mb;
    np += lab == 1; nn += lab == 0;
  }
  if (np) atomicAdd(&s_pos, np);
  if (nn) atomicAdd(&s_neg, nn);
  __syncthreads();
  if (threadIdx.x == 0) {
    list_cnt[((long)(2 * img) * max_chunks + blockIdx.x)] = s_pos;
    list_cnt[((long)(2 * img + 1) * max_chunks + blockIdx.x)] = s_neg;
  }
}
// per list: exclusive scan of the chunk counts; how many to draw (sampling.py:33-41)
__global__ void label_scan_kernel(LabelArgs g, unsigned* __restrict__ list_cnt, int max_chunks, int n_chunks, int* __restrict__ k_dev) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= g.N) return;
  unsigned tot[2];
  for (int w = 0; w < 2; ++w) {
    unsigned* p = list_cnt + (long)(2 * img + w) * max_chunks;
    unsigned run = 0;
    for (int c = 0; c < n_chunks; ++c) { const unsigned v = p[c]; p[c] = run; run += v; }
    tot[w] = run;
  }
  const int num_pos = min((int)tot[0], g.max_pos);
  const int num_neg = min((int)tot[1], g.batch - num_pos);
  k_dev[2 * img] = num_pos; k_dev[2 * img + 1] = num_neg;
}
// random key of every candidate: position in its list (ascending anchor index) -> 24-bit hash; labels reset to -1
__global__ __launch_bounds__(256) void label_keys_kernel(LabelArgs g, const int8_t* __restrict__ pre, const unsigned* __restrict__ list_cnt,
                                                         int max_chunks, unsigned* __restrict__ keys /*[2N][A]*/, int8_t* __restrict__ labels) {
  __shared__ unsigned sh[256];
  const int img = blockIdx.y;
  const long c0 = (long)blockIdx.x * CHUNK;
  int lab[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long a = c0 + threadIdx.x * 4 + j;
    lab[j] = a < g.A ? (int)pre[(long)img * g.A + a] : -2;
  }
  for (int w = 0; w < 2; ++w) {
    unsigned f[4], r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = lab[j] == (w == 0 ? 1 : 0) ? 1u : 0u;
    block_scan4(f, r, sh);
    const unsigned base = list_cnt[(long)(2 * img + w) * max_chunks + blockIdx.x];
    const unsigned long long seed = w == 0 ? g.seed_pos[img] : g.seed_neg[img];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long a = c0 + threadIdx.x * 4 + j;
      if (a < g.A) keys[((long)(2 * img + w)) * g.A + a] = f[j] ? hash_key24(seed, (unsigned long long)(base + r[j])) : KEY_NONE;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long a = c0 + threadIdx.x * 4 + j;
    if (a < g.A) labels[(long)img * g.A + a] = -1;
  }
}

// ---------------------------------------------------------------------------------------------------- ROI-head label + sample
// One workgroup per image over its P proposals (+ G ground-truth boxes appended, roi_heads.py:343-345): IoU >= thr -> foreground with
// the matched gt's class, else background K (matcher thresholds [0.5], labels [0, 1]); up to max_pos foreground and batch - that many
// background rows with the smallest random keys, each list in key order (sampling.py:49-54 + roi_heads.py:300-322).
constexpr int ROI_MAX_IMG = 64;          // per-image tables by value: 1.5 KiB of kernel arguments
struct RoiSampleArgs {
  int n_img, p_stride;
  int g_off[ROI_MAX_IMG], g_cnt[ROI_MAX_IMG];
  unsigned long long seeds[2 * ROI_MAX_IMG];
};
template <int CAP>
__global__ __launch_bounds__(1024) void roi_sample_kernel(RoiSampleArgs ra, const int* __restrict__ p_cnt, const float* __restrict__ props,
                                                          const float* __restrict__ gt, const int* __restrict__ gt_cls, int append_gt,
                                                          float thr, int K, int batch, int max_pos, int out_stride,
                                                          int* __restrict__ out_cnt, int* __restrict__ out_idx, int* __restrict__ out_cls,
                                                          float* __restrict__ out_box, float* __restrict__ out_gt) {
  __shared__ unsigned long long sk[CAP];
  __shared__ short s_cls[CAP];
  __shared__ short s_m[CAP];
  __shared__ unsigned sh[1024];
  const int img = blockIdx.x;
  const int G = ra.g_cnt[img];
  const int P = min(p_cnt[img], CAP - (append_gt ? G : 0));
  const int n = P + (append_gt ? G : 0);
  const float4* pb = (const float4*)props + (long)img * ra.p_stride;
  const float4* gb = (const float4*)gt + ra.g_off[img];
  const int* gc = gt_cls + ra.g_off[img];
  auto box_of = [&](int i) { return i < P ? pb[i] : gb[i - P]; };
  for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
    int cls = -2, mi = 0;
    if (i < n) {
      const float4 b = box_of(i);
      float mv = -1.f;
      for (int gi = 0; gi < G; ++gi) { const float v = iou_pair(gb[gi], b); if (v > mv) { mv = v; mi = gi; } }
      cls = (G > 0 && mv >= thr) ? gc[mi] : K;
    }
    s_cls[i] = (short)cls; s_m[i] = (short)mi;
  }
  __syncthreads();
  // positions inside the two candidate lists (ascending row order): serial prefix per 1024-slab through a block scan
  int base_p = 0, base_n = 0;
  for (int c0 = 0; c0 < CAP; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int cls = i < CAP ? (int)s_cls[i] : -2;
    for (int w = 0; w < 2; ++w) {
      const unsigned f = (w == 0) ? (cls >= 0 && cls != K) : (cls == K);
      sh[threadIdx.x] = f;
      __syncthreads();
      for (int o = 1; o < 1024; o <<= 1) {
        unsigned v = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
      }
      const unsigned incl = sh[threadIdx.x], total = sh[1023];
      __syncthreads();
      if (f && i < CAP) {
        const unsigned pos = (w == 0 ? base_p : base_n) + incl - 1;
        const unsigned key = hash_key24(ra.seeds[2 * img + w], pos);
        // sort key: list (0 = foreground first), random key, position; payload row
        sk[i] = ((unsigned long long)w << 63) | ((unsigned long long)key << 36) | ((unsigned long long)(pos & 0xFFFFFu) << 16) | (unsigned)i;
      }
      if (w == 0) base_p += (int)total; else base_n += (int)total;
    }
    if (i < CAP && !((cls >= 0 && cls != K) || cls == K)) sk[i] = ~0ull;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= CAP; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = sk[i], b = sk[p];
          const bool up = (i & k2) == 0;
          if ((a > b) == up) { sk[i] = b; sk[p] = a; }
        }
      }
      __syncthreads();
    }
  const int n_pos = base_p, n_neg = base_n;
  const int num_pos = min(n_pos, max_pos), num_neg = min(n_neg, batch - num_pos);
  if (threadIdx.x == 0) out_cnt[img] = num_pos + num_neg;
  for (int r = threadIdx.x; r < num_pos + num_neg; r += blockDim.x) {
    const unsigned long long v = r < num_pos ? sk[r] : sk[n_pos + (r - num_pos)];
    const int i = (int)(v & 0xFFFFu);
    const long o = (long)img * out_stride + r;
    out_idx[o] = i;
    out_cls[o] = (int)s_cls[i];
    ((float4*)out_box)[o] = box_of(i);
    ((float4*)out_gt)[o] = G > 0 ? gb[s_m[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // rows the sampler did not fill (more than max_pos foreground candidates AND fewer than batch - max_pos background ones: sampling.py
  // returns num_pos + num_neg < batch rows) are defined, never stale memory: class -1 (= ignore), empty boxes.  The caller reads out_cnt.
  for (int r = num_pos + num_neg + threadIdx.x; r < batch; r += blockDim.x) {
    const long o = (long)img * out_stride + r;
    out_idx[o] = -1; out_cls[o] = -1;
    ((float4*)out_box)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    ((float4*)out_gt)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}


// ---------------------------------------------------------------------------------------------------- RPN head output <-> anchor order
// The RPN head's two 1x1 convolutions run as ONE GEMM over the pixels of all levels: y [rows][ld] f32, row = row_off[l] + img * hw[l] +
// pixel, columns [objectness a | delta a * 4 + b] (A anchors per location).  The losses and the proposal selection want the anchor order
// of the reference (level-major, location-major, anchor-minor per image): logits [N][At], deltas [N][At][4].
struct RpnMap { int row_off[8], hw[8], a_off[9]; int L, N, A; long ld; };
__global__ void rpn_unpack_kernel(RpnMap m, const float* __restrict__ y, float* __restrict__ logits, float* __restrict__ deltas) {
  const long At = m.a_off[m.L];
  const long total = (long)m.N * At;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int img = (int)(i / At); const int j = (int)(i - (long)img * At);
    int l = 0;
    while (l + 1 < m.L && j >= m.a_off[l + 1]) ++l;
    const int q = j - m.a_off[l], pix = q / m.A, a = q - pix * m.A;
    const float* row = y + ((long)m.row_off[l] + (long)img * m.hw[l] + pix) * m.ld;
    logits[i] = row[a];
    const float* d = row + m.A + 4 * a;
    ((float4*)deltas)[i] = make_float4(d[0], d[1], d[2], d[3]);
  }
}
// dy [rows][ld] from (dlogits, ddeltas) (either may be NULL = zero); padding columns written as 0
__global__ void rpn_unpack_bwd_kernel(RpnMap m, long rows, const float* __restrict__ dlogits, const float* __restrict__ ddeltas,
                                      const float* __restrict__ g_logits, const float* __restrict__ g_deltas, float* __restrict__ dy) {
  const long At = m.a_off[m.L];
  const long total = rows * m.ld;
  const float sl = g_logits ? g_logits[0] : 1.f, sd = g_deltas ? g_deltas[0] : 1.f;     // cotangents of the two losses (device scalars)
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / m.ld; const int c = (int)(i - r * m.ld);
    float v = 0.f;
    if (c < 5 * m.A) {
      int l = 0;
      while (l + 1 < m.L && r >= m.row_off[l + 1]) ++l;
      const long rr = r - m.row_off[l];
      const int img = (int)(rr / m.hw[l]), pix = (int)(rr - (long)img * m.hw[l]);
      if (c < m.A) { if (dlogits) v = __fmul_rn(dlogits[(long)img * At + m.a_off[l] + (long)pix * m.A + c], sl); }
      else {
        const int a = (c - m.A) >> 2, b = (c - m.A) & 3;
        if (ddeltas) v = __fmul_rn(ddeltas[((long)img * At + m.a_off[l] + (long)pix * m.A + a) * 4 + b], sd);
      }
    }
    dy[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------- FPN level of every ROI
// poolers.py:17-50 assign_boxes_to_levels + convert_boxes_to_pooler_format: dense ROI rows [R][5] = (image, box) gathered from the
// images' box blocks, level = clamp(floor(4 + log2(sqrt(area) / 224 + 1e-8)), 2, 5) - 2, and per level the rows of that level in
// ascending order (as nonzero lists them) with their count — one workgroup (R <= 2^20, <= 64 images).
struct LevelArgs { int n_img; int row_off[ROI_MAX_IMG], row_cnt[ROI_MAX_IMG]; long box_off[ROI_MAX_IMG]; };
__global__ __launch_bounds__(1024) void roi_levels_kernel(LevelArgs g, int R, const float* __restrict__ boxes, float* __restrict__ rois,
                                                          int* __restrict__ level_of, int* __restrict__ sel /*[4][R]*/, int* __restrict__ sel_cnt) {
  __shared__ unsigned sh[1024];
  __shared__ int s_base[4];
  if (threadIdx.x < 4) s_base[threadIdx.x] = 0;
  __syncthreads();
  for (int r0 = 0; r0 < R; r0 += 1024) {
    const int r = r0 + threadIdx.x;
    int lv = -1;
    if (r < R) {
      int img = 0;
      while (img + 1 < g.n_img && r >= g.row_off[img + 1]) ++img;
      const float4 b = ((const float4*)(boxes + g.box_off[img]))[r - g.row_off[img]];
      float* o = rois + (long)r * 5;
      o[0] = (float)img; o[1] = b.x; o[2] = b.y; o[3] = b.z; o[4] = b.w;
      const float size = sqrtf(__fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y)));
      const float t = floorf(__fadd_rn(4.0f, log2f(__fadd_rn(__fdiv_rn(size, 224.0f), 1e-8f))));
      lv = (int)fminf(fmaxf(t, 2.0f), 5.0f) - 2;                 // (a NaN size — a degenerate box — lands on level 0 like torch.clamp of NaN -> ... see test)
      level_of[r] = lv;
    }
    for (int l = 0; l < 4; ++l) {
      const unsigned f = lv == l ? 1u : 0u;
      sh[threadIdx.x] = f;
      __syncthreads();
      for (int o = 1; o < 1024; o <<= 1) {
        const unsigned v = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
      }
      const unsigned incl = sh[threadIdx.x], tot = sh[1023];
      __syncthreads();
      if (f) sel[(long)l * R + s_base[l] + incl - 1] = r;
      __syncthreads();
      if (threadIdx.x == 0) s_base[l] += (int)tot;
      __syncthreads();
    }
  }
  if (threadIdx.x < 4) sel_cnt[threadIdx.x] = s_base[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------- WSDDN scores, general backward
// scores = softmax_k(C) * softmax_r(D) per image (fast_rcnn_wsddn.py:564-567), logits [R][ld] = [C (K) | D (K)].  With A = softmax_k(C),
// B = softmax_r(D) and g the cotangent of the scores:  dC = A (gB - sum_k A g B),  dD = B (gA - sum_r B g A).
// (The training path never needs this: sw_wsddn_mil emits the BCE loss with its own gradient.  It serves the stand-alone predictor API,
// where a caller builds its own loss on the scores.)
// Arithmetic in double: with peaky heads the gradient is a difference of nearly equal terms (g B - sum A g B with A ~ 1 on one class);
// float sums left ~1e-7 of absolute noise on weight gradients that are themselves ~1e-7 (an off-path API: the cost is irrelevant).
constexpr int WS_ROWS = 128;           // rows per workgroup of the row passes (K + 1 doubles of LDS per row: K <= 159)
__global__ __launch_bounds__(256) void wsddn_colstat_kernel(int R, int K, const float* __restrict__ lg, long ld, double* __restrict__ colstat) {
  __shared__ float red[32];
  __shared__ double dred[4];
  __shared__ float s_m;
  const int k = blockIdx.x;
  float m = -FLT_MAX;
  for (int r = threadIdx.x; r < R; r += 256) m = fmaxf(m, lg[(long)r * ld + K + k]);
  m = block_reduce_max(m, red);
  if (threadIdx.x == 0) s_m = m;
  __syncthreads();
  m = s_m;
  double s = 0.0;
  for (int r = threadIdx.x; r < R; r += 256) s += exp((double)lg[(long)r * ld + K + k] - (double)m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { colstat[2 * k] = (double)m; colstat[2 * k + 1] = (dred[0] + dred[1]) + (dred[2] + dred[3]); }
}
// phase 0: dC, and per-workgroup partial column sums t_part[wg][k] = sum_r B g A; phase 1: dD with the folded t
__global__ __launch_bounds__(WS_ROWS) void wsddn_bwd_rows_kernel(int R, int K, const float* __restrict__ lg, long ld, const float* __restrict__ g,
                                                                 long ldg, const double* __restrict__ colstat, float* __restrict__ dl, long ldd,
                                                                 double* __restrict__ t_part, const double* __restrict__ t_col, int phase) {
  extern __shared__ double shd[];          // [WS_ROWS][K + 1] products B g A of this workgroup's rows (phase 0)
  const int r = blockIdx.x * WS_ROWS + threadIdx.x;
  const bool ok = r < R;
  double rm = -1e300, rs = 0.0;
  if (ok) {
    for (int k = 0; k < K; ++k) rm = fmax(rm, (double)lg[(long)r * ld + k]);
    for (int k = 0; k < K; ++k) rs += exp((double)lg[(long)r * ld + k] - rm);
  }
  if (phase == 0) {
    double u = 0.0;
    if (ok)
      for (int k = 0; k < K; ++k) {
        const double A = exp((double)lg[(long)r * ld + k] - rm) / rs, B = exp((double)lg[(long)r * ld + K + k] - colstat[2 * k]) / colstat[2 * k + 1];
        u += A * (double)g[(long)r * ldg + k] * B;
      }
    for (int k = 0; k < K; ++k) {
      double p = 0.0;
      if (ok) {
        const double A = exp((double)lg[(long)r * ld + k] - rm) / rs, B = exp((double)lg[(long)r * ld + K + k] - colstat[2 * k]) / colstat[2 * k + 1];
        const double gv = (double)g[(long)r * ldg + k];
        dl[(long)r * ldd + k] = (float)(A * (gv * B - u));
        p = B * gv * A;
      }
      shd[threadIdx.x * (K + 1) + k] = p;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += WS_ROWS) {          // ordered sum over the workgroup's rows
      double t = 0.0;
      for (int i = 0; i < WS_ROWS; ++i) t += shd[i * (K + 1) + k];
      t_part[(long)blockIdx.x * K + k] = t;
    }
  } else if (ok) {
    for (int k = 0; k < K; ++k) {
      const double A = exp((double)lg[(long)r * ld + k] - rm) / rs, B = exp((double)lg[(long)r * ld + K + k] - colstat[2 * k]) / colstat[2 * k + 1];
      dl[(long)r * ldd + K + k] = (float)(B * ((double)g[(long)r * ldg + k] * A - t_col[k]));
    }
  }
}
__global__ void wsddn_tfold_kernel(int n_wg, int K, const double* __restrict__ t_part, double* __restrict__ t_col) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double t = 0.0;
  for (int i = 0; i < n_wg; ++i) t += t_part[(long)i * K + k];
  t_col[k] = t;
}

// ---------------------------------------------------------------------------------------------------- 3x3 weight gradient of tiny maps
// dw[co][ci][ty][tx] = sum over (img, y, x) of dy[img][y][x][co] * x[img][y + ty - 1][x + tx - 1][ci] for maps of a few pixels (p5 / p6
// of small images: 4x4, 2x2 — below the tile geometry of the gathering MFMA loader): one thread per (co, ci), nine running sums over the
// <= a few hundred pixels; the operands sit in L2.  cout_scale (may be NULL): the FrozenBN fold, dw[co] *= cout_scale[co].
template <typename T>
__global__ void conv3x3_wgrad_small_kernel(int N, int H, int W, int Cin, int Cout, const T* __restrict__ x, const T* __restrict__ dy,
                                           const float* __restrict__ cout_scale, float* __restrict__ dw, const int accumulate) {
  const long total = (long)Cout * Cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int img = 0; img < N; ++img)
      for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx) {
          const float g = Elem<T>::load(dy + (((long)img * H + y) * W + xx) * Cout + co);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xs = xx + t % 3 - 1;
            if (yy >= 0 && yy < H && xs >= 0 && xs < W) acc[t] += g * Elem<T>::load(x + (((long)img * H + yy) * W + xs) * Cin + ci);
          }
        }
    const float sc = cout_scale ? cout_scale[co] : 1.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float* d = dw + ((long)co * Cin + ci) * 9 + t;
      const float v = cout_scale ? __fmul_rn(acc[t], sc) : acc[t];
      *d = accumulate ? v + *d : v;
    }
  }
}

// out[m][n] = in[m][n] * (n < split ? g0[0] : g1[0]): the two cotangents of the ROI heads' (classification, box) losses applied to the
// unit gradient of the packed logits in one pass
__global__ void scale_col_blocks_kernel(long M, int N, int split, const float* __restrict__ in, long ld, const float* __restrict__ g0,
                                        const float* __restrict__ g1, float* __restrict__ out) {
  const float a = g0[0], b = g1[0];
  const long total = M * ld;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % ld);
    out[i] = n < N ? __fmul_rn(in[i], n < split ? a : b) : 0.f;
  }
}

}  // namespace

// =================================================================================================== entry points
extern "C" long sw_rpn_select_workspace_bytes(int N, int L, const int* n_per_level) {
  long tot = 0; int mx = 0;
  for (int l = 0; l < L; ++l) { tot += n_per_level[l]; mx = n_per_level[l] > mx ? n_per_level[l] : mx; }
  const int max_chunks = (mx + CHUNK - 1) / CHUNK;
  if (L >= 1 && L <= 8 && N * L > MAX_SEG) N = MAX_SEG / L;           // sw_rpn_select_pack walks image ranges of this size
  return (long)((sel_workspace_bytes(N * L, max_chunks) + 15) & ~(size_t)15) + (long)N * tot * 4 + 64;
}

extern "C" int sw_rpn_select_pack(int N, int L, const float* const* logits, const float* const* deltas, const float* const* anchors,
                                  const int* n_per_level, long img_stride, int pre_topk, const float* weights4, float scale_clamp,
                                  const int* img_hw_dev, float* cand_scores, float* cand_boxes, int* finite_dev, int* sel_idx,
                                  void* workspace, long workspace_bytes, hipStream_t stream) {
  SW_ENTER();
  if (N < 1 || L < 1 || L > 8 || pre_topk < 1 || pre_topk > 16384 || img_stride < 0) return -5;
  if (N * L > MAX_SEG) {
    // the segment table travels by value (MAX_SEG entries): more images run as consecutive calls over image ranges; every output
    // is per image and the workspace is reused in stream order
    const int per = MAX_SEG / L;
    const long rows = (long)L * pre_topk;
    for (int i0 = 0; i0 < N; i0 += per) {
      const int nc = N - i0 < per ? N - i0 : per;
      const float* lg[8]; const float* dl[8];
      for (int l = 0; l < L; ++l) {
        const long adv = (long)i0 * (img_stride ? img_stride : n_per_level[l]);
        lg[l] = logits[l] + adv; dl[l] = deltas[l] + adv * 4;
      }
      const int rc = sw_rpn_select_pack(nc, L, lg, dl, anchors, n_per_level, img_stride, pre_topk, weights4, scale_clamp, img_hw_dev + 2 * i0,
                                        cand_scores + (long)i0 * rows * (L + 1), cand_boxes + (long)i0 * rows * 4 * L, finite_dev + i0,
                                        sel_idx + (long)i0 * L * pre_topk, workspace, workspace_bytes, stream);
      if (rc) return rc;
    }
    return 0;
  }
  if (workspace_bytes < sw_rpn_select_workspace_bytes(N, L, n_per_level)) return -6;
  RpnLevels lv = {};
  lv.L = L; lv.N = N; lv.img_stride = img_stride;
  long tot = 0; int mx = 0;
  for (int l = 0; l < L; ++l) {
    lv.logits[l] = logits[l]; lv.deltas[l] = deltas[l]; lv.anchors[l] = anchors[l]; lv.n[l] = n_per_level[l];
    tot += n_per_level[l]; mx = n_per_level[l] > mx ? n_per_level[l] : mx;
  }
  SegTab t = {};
  t.n_seg = N * L; t.max_chunks = (mx + CHUNK - 1) / CHUNK;
  for (int img = 0; img < N; ++img) {
    long off = (long)img * tot;
    for (int l = 0; l < L; ++l) {
      Seg& sg = t.s[img * L + l];
      sg.off = off; sg.mark_off = 0; sg.n = n_per_level[l]; sg.k = n_per_level[l] < pre_topk ? n_per_level[l] : pre_topk; sg.mark_val = 0;
      off += n_per_level[l];
    }
  }
  char* ws = (char*)workspace;
  unsigned* keys = (unsigned*)(ws + ((sel_workspace_bytes(t.n_seg, t.max_chunks) + 15) & ~(size_t)15));
  hipError_t e = hipMemsetAsync(finite_dev, 0xFF, (size_t)N * sizeof(int), stream);
  if (e != hipSuccess) return (int)e;
  const int kb = (mx + 255) / 256 > 1024 ? 1024 : (mx + 255) / 256;
  hipLaunchKernelGGL(rpn_keys_kernel, dim3(kb, t.n_seg), dim3(256), 0, stream, lv, t, keys);
  SelState* st = nullptr;
  const int rc = sel_run(t, ws, keys, nullptr, sel_idx, pre_topk, nullptr, &st, stream);
  if (rc) return rc;
  int cap = 1024;
  while (cap < pre_topk) cap <<= 1;
  const size_t lds = (size_t)cap * 8;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute((const void*)rpn_sort_pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(rpn_sort_pack_kernel, dim3(t.n_seg), dim3(1024), lds, stream, cap, lv, t, st, keys, sel_idx, (long)pre_topk, pre_topk,
                     weights4[0], weights4[1], weights4[2], weights4[3], scale_clamp, img_hw_dev, cand_scores, cand_boxes, finite_dev);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" long sw_rpn_label_workspace_bytes(int N, long A, int total_gt) {
  const int max_chunks = (int)((A + CHUNK - 1) / CHUNK);
  if (N > LABEL_MAX_IMG) N = LABEL_MAX_IMG;                             // sw_rpn_label_anchors walks image ranges of this size (total_gt: an upper bound)
  return (long)((sel_workspace_bytes(2 * N, max_chunks) + 15) & ~(size_t)15) + (long)2 * N * A * 4 + (long)2 * N * max_chunks * 4 +
         (long)(total_gt + 8) * 4 + (long)2 * N * 8 + (long)N * A + 512;
}

extern "C" int sw_rpn_label_anchors(int N, long A, const float* anchors, const float* gt_boxes, const int* gt_count_per_image,
                                    float thr_lo, float thr_hi, int batch_size, int max_pos, const uint64_t* seeds,
                                    int8_t* labels, float* matched, void* workspace, long workspace_bytes, hipStream_t stream) {
  SW_ENTER();
  if (N < 1 || A < 1 || A > 0x3FFFFFFFL) return -5;
  if (N > LABEL_MAX_IMG) {
    // per-image tables travel by value (LABEL_MAX_IMG entries): more images run as consecutive calls over image ranges (outputs are
    // per image, seeds per image, the workspace is reused in stream order)
    int g0 = 0;
    for (int i0 = 0; i0 < N; i0 += LABEL_MAX_IMG) {
      const int nc = N - i0 < LABEL_MAX_IMG ? N - i0 : LABEL_MAX_IMG;
      const int rc = sw_rpn_label_anchors(nc, A, anchors, gt_boxes ? gt_boxes + (long)g0 * 4 : nullptr, gt_count_per_image + i0, thr_lo, thr_hi,
                                          batch_size, max_pos, seeds + 2 * i0, labels + (long)i0 * A, matched + (long)i0 * A * 4, workspace,
                                          workspace_bytes, stream);
      if (rc) return rc;
      for (int i = 0; i < nc; ++i) g0 += gt_count_per_image[i0 + i];
    }
    return 0;
  }
  int total_gt = 0;
  for (int i = 0; i < N; ++i) total_gt += gt_count_per_image[i];
  if (workspace_bytes < sw_rpn_label_workspace_bytes(N, A, total_gt)) return -6;
  LabelArgs g = {};
  g.A = A; g.N = N; g.anchors = anchors; g.gt = gt_boxes; g.thr_lo = thr_lo; g.thr_hi = thr_hi; g.batch = batch_size; g.max_pos = max_pos;
  int off = 0;
  for (int i = 0; i < N; ++i) {
    g.g_off[i] = off; g.g_cnt[i] = gt_count_per_image[i]; off += gt_count_per_image[i];
    g.seed_pos[i] = seeds[2 * i]; g.seed_neg[i] = seeds[2 * i + 1];
  }
  g.g_off[N] = off;
  SegTab t = {};
  t.n_seg = 2 * N; t.max_chunks = (int)((A + CHUNK - 1) / CHUNK);
  for (int i = 0; i < N; ++i)
    for (int wl = 0; wl < 2; ++wl) {
      Seg& sg = t.s[2 * i + wl];
      sg.off = (long)(2 * i + wl) * A; sg.mark_off = (long)i * A; sg.n = (int)A; sg.k = -1; sg.mark_val = wl == 0 ? 1 : 0;
    }
  char* w = (char*)workspace;
  char* sel_ws = w;
  w += (sel_workspace_bytes(t.n_seg, t.max_chunks) + 15) & ~(size_t)15;
  unsigned* keys = (unsigned*)w; w += (size_t)t.n_seg * A * 4;
  unsigned* list_cnt = (unsigned*)w; w += (size_t)t.n_seg * t.max_chunks * 4;
  unsigned* best = (unsigned*)w; w += (size_t)(total_gt + 8) * 4;
  int* k_dev = (int*)w; w += (size_t)t.n_seg * 8;
  int8_t* pre = (int8_t*)w;
  hipError_t e = hipMemsetAsync(best, 0, (size_t)(total_gt + 8) * 4, stream);
  if (e != hipSuccess) return (int)e;
  const int gb = (int)((A + 255) / 256) > 2048 ? 2048 : (int)((A + 255) / 256);
  if (total_gt > 0) hipLaunchKernelGGL(anchor_best_kernel, dim3(gb, N), dim3(256), 0, stream, g, best);
  hipLaunchKernelGGL(anchor_label_kernel, dim3(t.max_chunks, N), dim3(256), 0, stream, g, best, pre, matched, list_cnt, t.max_chunks);
  hipLaunchKernelGGL(label_scan_kernel, dim3(1), dim3(64), 0, stream, g, list_cnt, t.max_chunks, t.max_chunks, k_dev);
  hipLaunchKernelGGL(label_keys_kernel, dim3(t.max_chunks, N), dim3(256), 0, stream, g, pre, list_cnt, t.max_chunks, keys, labels);
  SW_CHECK_LAUNCH();
  return sel_run(t, sel_ws, keys, k_dev, nullptr, 0, labels, nullptr, stream);
}

extern "C" int sw_roi_label_sample(int n_img, const int* p_cnt_dev, int p_stride, const float* proposals, const int* g_off,
                                   const int* g_cnt, const float* gt_boxes, const int32_t* gt_classes, int append_gt,
                                   float iou_thresh, int num_classes, int batch_size, int max_pos, const uint64_t* seeds, int out_stride,
                                   int32_t* out_count, int32_t* out_index, int32_t* out_classes, float* out_boxes, float* out_gt_boxes,
                                   hipStream_t stream) {
  SW_ENTER();
  if (n_img < 1 || out_stride < batch_size || batch_size < 1 || max_pos < 0 || max_pos > batch_size) return -5;
  if (n_img > ROI_MAX_IMG) {                                            // image ranges (one workgroup per image: nothing is shared)
    for (int i0 = 0; i0 < n_img; i0 += ROI_MAX_IMG) {
      const int nc = n_img - i0 < ROI_MAX_IMG ? n_img - i0 : ROI_MAX_IMG;
      const int rc = sw_roi_label_sample(nc, p_cnt_dev + i0, p_stride, proposals + (long)i0 * p_stride * 4, g_off + i0, g_cnt + i0, gt_boxes,
                                         gt_classes, append_gt, iou_thresh, num_classes, batch_size, max_pos, seeds + 2 * i0, out_stride,
                                         out_count + i0, out_index + (long)i0 * out_stride, out_classes + (long)i0 * out_stride,
                                         out_boxes + (long)i0 * out_stride * 4, out_gt_boxes + (long)i0 * out_stride * 4, stream);
      if (rc) return rc;
    }
    return 0;
  }
  RoiSampleArgs ra = {};
  ra.n_img = n_img; ra.p_stride = p_stride;
  for (int i = 0; i < n_img; ++i) {
    ra.g_off[i] = g_off[i]; ra.g_cnt[i] = g_cnt[i]; ra.seeds[2 * i] = seeds[2 * i]; ra.seeds[2 * i + 1] = seeds[2 * i + 1];
    if (p_stride + (append_gt ? g_cnt[i] : 0) > 4096) return -6;       // proposals + appended ground truth of one image: LDS-resident sort
  }
  hipLaunchKernelGGL(roi_sample_kernel<4096>, dim3(n_img), dim3(1024), 0, stream, ra, p_cnt_dev, proposals, gt_boxes, gt_classes, append_gt,
                     iou_thresh, num_classes, batch_size, max_pos, out_stride, out_count, out_index, out_classes, out_boxes, out_gt_boxes);
  SW_CHECK_LAUNCH();
  return 0;
}

static RpnMap make_rpn_map(int N, int L, int A, const int* hw_per_level, long ld) {
  RpnMap m = {};
  m.L = L; m.N = N; m.A = A; m.ld = ld;
  int ro = 0, ao = 0;
  for (int l = 0; l < L; ++l) { m.row_off[l] = ro; m.hw[l] = hw_per_level[l]; m.a_off[l] = ao; ro += N * hw_per_level[l]; ao += hw_per_level[l] * A; }
  m.a_off[L] = ao;
  return m;
}

extern "C" int sw_rpn_unpack(int N, int L, int A, const int* hw_per_level, const float* y, long ld, float* logits, float* deltas,
                             hipStream_t stream) {
  SW_ENTER();
  if (N < 1 || L < 1 || L > 8 || A < 1 || ld < 5 * A) return -5;
  const RpnMap m = make_rpn_map(N, L, A, hw_per_level, ld);
  const long total = (long)N * m.a_off[L];
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(rpn_unpack_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, m, y, logits, deltas);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_rpn_unpack_bwd(int N, int L, int A, const int* hw_per_level, const float* dlogits, const float* ddeltas,
                                 const float* g_logits_dev, const float* g_deltas_dev, float* dy, long ld, hipStream_t stream) {
  SW_ENTER();
  if (N < 1 || L < 1 || L > 8 || A < 1 || ld < 5 * A) return -5;
  const RpnMap m = make_rpn_map(N, L, A, hw_per_level, ld);
  long rows = 0;
  for (int l = 0; l < L; ++l) rows += (long)N * hw_per_level[l];
  const long blocks = (rows * ld + 255) / 256;
  hipLaunchKernelGGL(rpn_unpack_bwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, m, rows, dlogits, ddeltas, g_logits_dev, g_deltas_dev, dy);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_roi_assign_levels(int n_img, const int* row_cnt, const long* box_off_floats, const float* boxes, float* rois,
                                    int32_t* level_of, int32_t* sel, int32_t* sel_cnt, hipStream_t stream) {
  SW_ENTER();
  if (n_img < 1 || n_img > ROI_MAX_IMG) return -5;                      // (the per-level lists run over all images: one call, one table)
  LevelArgs g = {};
  g.n_img = n_img;
  int R = 0;
  for (int i = 0; i < n_img; ++i) { g.row_off[i] = R; g.row_cnt[i] = row_cnt[i]; g.box_off[i] = box_off_floats[i]; R += row_cnt[i]; }
  if (R > (1 << 20)) return -6;
  if (R == 0) return (int)hipMemsetAsync(sel_cnt, 0, 16, stream);
  hipLaunchKernelGGL(roi_levels_kernel, dim3(1), dim3(1024), 0, stream, g, R, boxes, rois, level_of, sel, sel_cnt);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_scale_col_blocks(long M, int N, int split, const float* in, long ld, const float* g0_dev, const float* g1_dev, float* out,
                                   hipStream_t stream) {
  SW_ENTER();
  if (M <= 0) return 0;
  if (N < 1 || ld < N || split < 0 || split > N) return -5;
  const long blocks = (M * ld + 255) / 256;
  hipLaunchKernelGGL(scale_col_blocks_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, M, N, split, in, ld,
                     g0_dev, g1_dev, out);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" long sw_wsddn_scores_bwd_workspace_floats(int R, int K) {
  const long n_wg = (R + WS_ROWS - 1) / WS_ROWS;
  return 2 * (2L * K + n_wg * K + K) + 64;             // doubles, counted in floats
}

extern "C" int sw_wsddn_scores_bwd(int R, int K, const float* logits, long ld, const float* g_scores, long ld_g, float* dlogits, long ld_d,
                                   float* workspace, hipStream_t stream) {
  SW_ENTER();
  if (R <= 0) return 0;
  if (K < 1 || K > 1024 || ld < 2 * K || ld_d < 2 * K || ld_g < K || (((uintptr_t)workspace) & 7)) return -5;
  const int n_wg = (R + WS_ROWS - 1) / WS_ROWS;
  double* colstat = (double*)workspace; double* t_part = colstat + 2 * K; double* t_col = t_part + (long)n_wg * K;
  const size_t lds = (size_t)WS_ROWS * (K + 1) * 8;
  if (lds > 160 * 1024) return -6;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)wsddn_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(wsddn_colstat_kernel, dim3(K), dim3(256), 0, stream, R, K, logits, ld, colstat);
  hipLaunchKernelGGL(wsddn_bwd_rows_kernel, dim3(n_wg), dim3(WS_ROWS), lds, stream, R, K, logits, ld, g_scores, ld_g, colstat, dlogits, ld_d, t_part,
                     (const double*)nullptr, 0);
  hipLaunchKernelGGL(wsddn_tfold_kernel, dim3((K + 63) / 64), dim3(64), 0, stream, n_wg, K, t_part, t_col);
  hipLaunchKernelGGL(wsddn_bwd_rows_kernel, dim3(n_wg), dim3(WS_ROWS), 0, stream, R, K, logits, ld, g_scores, ld_g, colstat, dlogits, ld_d, t_part, t_col, 1);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_conv3x3_wgrad_small_acc(int dtype, int nimg, int H, int W, int Cin, int Cout, const void* x, const void* dy,
                                          const float* cout_scale, float* dw_oihw, int accumulate, hipStream_t stream) {
  SW_ENTER();
  if (nimg < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (long)nimg * H * W > 4096) return -5;
  const long blocks = ((long)Cout * Cin + 255) / 256;
  if (dtype == SW_BF16)
    hipLaunchKernelGGL(conv3x3_wgrad_small_kernel<unsigned short>, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, nimg, H, W,
                       Cin, Cout, (const unsigned short*)x, (const unsigned short*)dy, cout_scale, dw_oihw, accumulate);
  else if (dtype == SW_F32)
    hipLaunchKernelGGL(conv3x3_wgrad_small_kernel<float>, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, nimg, H, W, Cin,
                       Cout, (const float*)x, (const float*)dy, cout_scale, dw_oihw, accumulate);
  else return -1;
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_conv3x3_wgrad_small(int dtype, int nimg, int H, int W, int Cin, int Cout, const void* x, const void* dy,
                                      const float* cout_scale, float* dw_oihw, hipStream_t stream) {
  return sw_conv3x3_wgrad_small_acc(dtype, nimg, H, W, Cin, Cout, x, dy, cout_scale, dw_oihw, 0, stream);
}
