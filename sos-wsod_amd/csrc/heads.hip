// Fused per-proposal kernels of the WSDDN / OICR heads on gfx950 (all f32 arithmetic):
//   sw_wsddn_mil        dual softmax (classes x proposals), image-level BCE and its gradient, view-mean scores
//   sw_oicr_mean_probs  view-mean softmax scores of every refinement head (the next round's mining input)
//   sw_oicr_mine_label  top-p% mining, score threshold, class-agnostic NMS, IoU matching -> labels (all rounds at once)
//   sw_oicr_refine_loss weighted CE + L1 box loss and gradients (all rounds at once)
// They are latency-bound glue in the reference (dozens of tiny kernels + host syncs); here the whole
// chain is 8 launches with wave64 shuffle reductions, no host round trip, and the K refinement rounds side by side
// (a round's mining scores depend on the logits only, not on the previous round's labels).
#include <float.h>
#include "common.h"
#include "soswsod_hip.h"

namespace {

constexpr int KMAX = 128;      // max classes (+1) held in LDS column accumulators
constexpr int NWAVE = 16;      // 1024-thread workgroups

// ------------------------------------------------------------------------------------------- WSDDN
// The softmax over PROPOSALS couples all R rows of a view, the one over classes all K columns of a row.  A single
// workgroup per view (the first version) spent 220 us on 4 CUs; here the rows are cut into chunks of WS_CH and the
// coupling goes through a few hundred floats of workspace:
//   1. wsddn_stats_kernel   (chunk, view): per detection column the chunk's max and sum exp(x - max)
//   2. wsddn_scores_kernel  (chunk, view): merges the chunk statistics (fixed order), thread = proposal row:
//        scores, per-chunk column sums of the scores; mean_scores_kernel then averages the views (mining input)
//   3. wsddn_grad_kernel    (chunk, view): image-level score -> clamped BCE, its gradient through both softmaxes
// Every reduction has a fixed order => deterministic.
constexpr int WS_CH = 256;     // rows per workgroup
constexpr int WS_VMAX = 8;     // views held in registers by the scores kernel

// Column statistics of one chunk of rows: the chunk's detection logits are copied into LDS by a flat, coalesced sweep (a thread
// walking its own row issued K dependent 4-byte loads 1.7 KB apart), then K x G threads reduce column k over row group g and the G
// partials are combined in group order (deterministic).
constexpr int WS_GMAX = 16;
__global__ __launch_bounds__(WS_CH) void wsddn_stats_kernel(int R, int K, const float* __restrict__ logits, long ld,
                                                            int det_col, float* __restrict__ pm, float* __restrict__ ps) {
  extern __shared__ float tile[];                        // [WS_CH][K]
  __shared__ float s_part[WS_GMAX][KMAX];
  __shared__ float s_m[KMAX];
  const int chunk = blockIdx.x, v = blockIdx.y, nchunk = gridDim.x;
  const int row0 = chunk * WS_CH, nrow = min(WS_CH, R - row0);
  const float* L = logits + ((long)v * R + row0) * ld + det_col;
  for (int i = threadIdx.x; i < nrow * K; i += WS_CH) {
    const int r = i / K, k = i - r * K;
    tile[i] = L[(long)r * ld + k];
  }
  __syncthreads();
  const int G = min(WS_GMAX, WS_CH / K), rpg = (nrow + G - 1) / G;
  const int t = threadIdx.x, k = t % K, g = t / K;
  const int ra = min(nrow, g * rpg), rb = min(nrow, ra + rpg);
  if (t < G * K) {
    float m = -FLT_MAX;
    for (int r = ra; r < rb; ++r) m = fmaxf(m, tile[r * K + k]);
    s_part[g][k] = m;
  }
  __syncthreads();
  if (t < K) {
    float m = s_part[0][t];
    for (int w = 1; w < G; ++w) m = fmaxf(m, s_part[w][t]);
    s_m[t] = m;
  }
  __syncthreads();
  if (t < G * K) {
    float z = 0.f;
    const float m = s_m[k];
    for (int r = ra; r < rb; ++r) z += expf(tile[r * K + k] - m);
    s_part[g][k] = z;
  }
  __syncthreads();
  if (t < K) {
    float z = s_part[0][t];
    for (int w = 1; w < G; ++w) z += s_part[w][t];
    pm[((long)v * nchunk + chunk) * K + t] = s_m[t];
    ps[((long)v * nchunk + chunk) * K + t] = z;
  }
}

// merged column statistics of view v: max over chunks, sum rescaled to that max (chunk order)
__device__ __forceinline__ void merge_stats(int nchunk, int K, int v, int k, const float* pm, const float* ps, float* m_out,
                                            float* z_out) {
  float m = -FLT_MAX;
  for (int c = 0; c < nchunk; ++c) m = fmaxf(m, pm[((long)v * nchunk + c) * K + k]);
  float z = 0.f;
  for (int c = 0; c < nchunk; ++c) z += ps[((long)v * nchunk + c) * K + k] * expf(pm[((long)v * nchunk + c) * K + k] - m);
  *m_out = m; *z_out = z;
}

// one workgroup = 256 proposal rows of ONE view.  The 2 x K logits of a row are fetched by a flat, coalesced copy into LDS
// (a thread walking its own row issued 2K dependent 4-byte loads: 100 us for 8 workgroups); the per-column sums of the
// scores are taken from the LDS tile by K x 8 threads, 32 rows each, then 8 partials in fixed order.
__global__ __launch_bounds__(WS_CH) void wsddn_scores_kernel(int V, int R, int K, const float* __restrict__ logits, long ld,
                                                             int cls_col, int det_col, const float* __restrict__ pm,
                                                             const float* __restrict__ ps, float* __restrict__ cm,
                                                             float* __restrict__ cz, float* __restrict__ pS,
                                                             float* __restrict__ scores) {
  extern __shared__ float tile[];                        // [WS_CH][K]: class logits -> scores
  __shared__ float s_max[KMAX], s_sum[KMAX], s_part[8][KMAX];
  const int chunk = blockIdx.x, nchunk = gridDim.x, v = blockIdx.y;
  const int row0 = chunk * WS_CH, nrow = min(WS_CH, R - row0);
  float* tc = tile;
  const float* L = logits + ((long)v * R + row0) * ld;
  for (int i = threadIdx.x; i < nrow * K; i += WS_CH) {
    const int r = i / K, k = i - r * K;
    tc[i] = L[(long)r * ld + cls_col + k];
  }
  for (int k = threadIdx.x; k < K; k += WS_CH) {
    float m, z;
    merge_stats(nchunk, K, v, k, pm, ps, &m, &z);
    s_max[k] = m; s_sum[k] = z;
    if (chunk == 0) { cm[v * K + k] = m; cz[v * K + k] = z; }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < nrow) {
    float* c = tc + t * K; const float* d = L + (long)t * ld + det_col;
    float m = -FLT_MAX;
    for (int k = 0; k < K; ++k) m = fmaxf(m, c[k]);
    float z = 0.f;
    for (int k = 0; k < K; ++k) z += expf(c[k] - m);
    for (int k = 0; k < K; ++k) {
      const float p = expf(c[k] - m) / z;
      const float q = expf(d[k] - s_max[k]) / s_sum[k];
      c[k] = p * q;
    }
  }
  __syncthreads();
  float* S = scores + ((long)v * R + row0) * K;          // the tile is one contiguous run of the (V, R, K) score tensor
  for (int i = threadIdx.x; i < nrow * K; i += WS_CH) S[i] = tc[i];
  for (int i = threadIdx.x; i < K * 8; i += WS_CH) {     // column k, rows [part*32, part*32+32)
    const int k = i % K, part = i / K;
    float a = 0.f;
    for (int r = part * 32; r < min(nrow, part * 32 + 32); ++r) a += tc[r * K + k];
    s_part[part][k] = a;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += WS_CH) {
    float a = s_part[0][k];
    for (int w = 1; w < 8; ++w) a += s_part[w][k];
    pS[((long)v * nchunk + chunk) * K + k] = a;
  }
}

// out[r][k] = (((s0+s1)+s2)+s3)/V over the views' score matrices, rows of pitch ld_out (mining input of round 0)
__global__ void mean_scores_kernel(int V, int R, int K, const float* __restrict__ scores, float* __restrict__ out,
                                   long ld_out) {
  const long n = (long)R * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a = scores[i];
    for (int v = 1; v < V; ++v) a += scores[(long)v * n + i];
    const long r = i / K; const int k = (int)(i - r * K);
    out[r * ld_out + k] = __fdiv_rn(a, (float)V);
    if (k == 0) for (long j = K; j < ld_out; ++j) out[r * ld_out + j] = 0.f;   // pad columns (the rounds share a K+1 pitch)
  }
}

__global__ __launch_bounds__(WS_CH) void wsddn_grad_kernel(int V, int R, int K, const float* __restrict__ logits, long ld,
                                                           int cls_col, int det_col, const float* __restrict__ onehot,
                                                           const float* __restrict__ cm, const float* __restrict__ cz,
                                                           const float* __restrict__ pS, int nchunk,
                                                           const float* __restrict__ scores, float* __restrict__ loss_view,
                                                           float* __restrict__ dlogits, long ld_d,
                                                           const float* __restrict__ grad_scale) {
  __shared__ float s_S[KMAX], s_g[KMAX], s_max[KMAX], s_sum[KMAX];
  __shared__ float red[32];
  const int chunk = blockIdx.x, v = blockIdx.y;
  // BCE on the clamped image-level score (fast_rcnn_wsddn.py:340-375); gradient is zero where clamped
  float lpart = 0.f;
  const float gs = (dlogits && grad_scale) ? grad_scale[0] / (float)V : 0.f;
  for (int k = threadIdx.x; k < K; k += WS_CH) {
    float raw = 0.f;
    for (int c = 0; c < nchunk; ++c) raw += pS[((long)v * nchunk + c) * K + k];
    const float y = fminf(fmaxf(raw, 1e-6f), 1.0f - 1e-6f);
    const float t = onehot[k];
    lpart += -(t * fmaxf(logf(y), -100.f) + (1.f - t) * fmaxf(logf(1.f - y), -100.f));
    const bool inside = (raw >= 1e-6f) && (raw <= 1.0f - 1e-6f);
    s_S[k] = raw;
    s_g[k] = inside ? gs * (-(t / y - (1.f - t) / (1.f - y)) / (float)K) : 0.f;
    s_max[k] = cm[v * K + k]; s_sum[k] = cz[v * K + k];
  }
  const float ltot = block_reduce_sum(lpart, red);
  if (chunk == 0 && threadIdx.x == 0) loss_view[v] = ltot / (float)K;
  __syncthreads();
  if (!dlogits) return;
  const int r = chunk * WS_CH + threadIdx.x;
  if (r >= R) return;
  // backward of the two softmaxes
  const float* c = logits + ((long)v * R + r) * ld + cls_col;
  const float* d = logits + ((long)v * R + r) * ld + det_col;
  const float* S = scores + ((long)v * R + r) * K;
  float* DL = dlogits + ((long)v * R + r) * ld_d;
  float m = -FLT_MAX;
  for (int k = 0; k < K; ++k) m = fmaxf(m, c[k]);
  float z = 0.f;
  for (int k = 0; k < K; ++k) z += expf(c[k] - m);
  float dot = 0.f;                              // sum_j g_j s_rj
  for (int k = 0; k < K; ++k) dot += s_g[k] * S[k];
  for (int k = 0; k < K; ++k) {
    const float p = expf(c[k] - m) / z;
    const float q = expf(d[k] - s_max[k]) / s_sum[k];
    DL[cls_col + k] = p * (s_g[k] * q - dot);
    DL[det_col + k] = s_g[k] * q * (p - s_S[k]);
  }
}

// mean over views of every refinement head's own softmax:  out[k][r][j] = (((p0+p1)+p2)+p3)/V,
// p_v = softmax_j(logits[v][r][cls_col0 + k*col_stride + j])   (predict_probs fast_rcnn_oicr.py:702-716, view
// average roi_heads_oicrplus.py:390-395).  These are the mining scores of round k+1; they depend on the logits only,
// so all rounds are produced by one launch and the rounds' mining runs concurrently.
constexpr int MP_ROWS = 64;      // rows per workgroup of mean_probs_kernel (x V views = threads)
__global__ __launch_bounds__(MP_ROWS * WS_VMAX) void mean_probs_kernel(int V, int R, int K, const float* __restrict__ logits,
                                                                       long ld, int cls_col0, int col_stride,
                                                                       float* __restrict__ out) {
  extern __shared__ float ptile[];                       // [V][MP_ROWS][K+1] softmax of every (view, row)
  const int K1 = K + 1, k = blockIdx.y;
  const int row0 = blockIdx.x * MP_ROWS, nrow = min(MP_ROWS, R - row0);
  const int v = threadIdx.x / MP_ROWS, t = threadIdx.x % MP_ROWS;
  if (v < V && t < nrow) {
    const float* x = logits + ((long)v * R + row0 + t) * ld + cls_col0 + k * col_stride;
    float* p = ptile + ((long)v * MP_ROWS + t) * K1;
    float m = -FLT_MAX;
    for (int j = 0; j < K1; ++j) { const float xv = x[j]; p[j] = xv; m = fmaxf(m, xv); }
    float z = 0.f;
    for (int j = 0; j < K1; ++j) z += expf(p[j] - m);
    for (int j = 0; j < K1; ++j) p[j] = expf(p[j] - m) / z;
  }
  __syncthreads();
  float* o = out + ((long)k * R + row0) * K1;            // contiguous run of the (rounds, R, K+1) output
  for (int i = threadIdx.x; i < nrow * K1; i += blockDim.x) {
    float a = ptile[i];
    for (int vv = 1; vv < V; ++vv) a += ptile[(long)vv * MP_ROWS * K1 + i];
    o[i] = __fdiv_rn(a, (float)V);
  }
}

// ------------------------------------------------------------------------------------------- OICR refine loss
// grid = (row groups of 4, prediction views, refinement rounds).  A WAVE owns one proposal row of prediction view pv (lanes = logit
// columns: the row's K+1 class logits and its 4K box entries are read and its gradient row written as contiguous runs; a thread
// per row made every access a 4-byte strided one, 29 us for 13 MB) and serves every target view v with pred_view[v] == pv in
// order, so the (reference-quirk) double use of view 2's logits needs no atomics.  Per-row loss terms go to a scratch array and
// are summed in fixed order by refine_reduce_kernel (deterministic, no float atomics).
__global__ __launch_bounds__(256) void refine_loss_kernel(int V, int R, int K, const float* __restrict__ logits, long ld,
                                                          int cls_col, int box_col, const float* __restrict__ boxes,
                                                          const int* __restrict__ lab_class,
                                                          const float* __restrict__ lab_weight,
                                                          const int* __restrict__ lab_index,
                                                          const int* __restrict__ pred_view,
                                                          float wx, float wy, float ww, float wh,
                                                          float* __restrict__ row_loss,
                                                          float* __restrict__ dlogits, long ld_d,
                                                          const float* __restrict__ grad_scale, int col_stride) {
  const int pv = blockIdx.y, round = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const int K1 = K + 1;
  cls_col += round * col_stride; box_col += round * col_stride;
  lab_class += (long)round * R; lab_weight += (long)round * R; lab_index += (long)round * R;
  row_loss += (long)round * 2 * V * R;
  if (grad_scale) grad_scale += 2 * round;
  const float* x = logits + ((long)pv * R + r) * ld + cls_col;
  float* DL = dlogits ? dlogits + ((long)pv * R + r) * ld_d : nullptr;
  const float gs_cls = (dlogits && grad_scale) ? grad_scale[0] / (float)V / (float)R : 0.f;
  const float gs_box = (dlogits && grad_scale) ? grad_scale[1] / (float)V / (float)R : 0.f;
  // own softmax (predict_probs, fast_rcnn_oicr.py:702-716)
  float m = -FLT_MAX;
  for (int j = lane; j < K1; j += 64) m = fmaxf(m, x[j]);
  m = wave_reduce_max(m);
  float z = 0.f;
  for (int j = lane; j < K1; j += 64) z += expf(x[j] - m);
  z = wave_reduce_sum(z);
  const float logz = logf(z);
  const int gt = lab_class[r];
  const float w = gt == -1 ? 0.f : lab_weight[r];                       // fast_rcnn_oicr.py:217-218
  int ntgt = 0;
  for (int v = 0; v < V; ++v) ntgt += (pred_view[v] == pv) ? 1 : 0;
  const float ce = gt >= 0 ? -((x[gt] - m) - logz) * w : 0.f;            // CE(ignore_index=-1) * weight
  // box terms: lanes 0-3 = the four deltas of the gt class, per served target view; sign sums accumulate in view order
  const bool fg = gt >= 0 && gt < K;
  float dsum = 0.f;
  for (int v = 0; v < V; ++v) {
    if (pred_view[v] != pv) continue;
    float lb = 0.f;
    if (fg) {                                                            // foreground: L1 on the gt-class deltas
      const float* B = boxes + (long)v * R * 4;
      const float* s = B + (long)r * 4;
      const float* t = B + (long)lab_index[r] * 4;                       // target = this view's proposal[gt_index]
      const float sw_ = s[2] - s[0], sh_ = s[3] - s[1];
      const float sx = s[0] + 0.5f * sw_, sy = s[1] + 0.5f * sh_;
      const float tw_ = t[2] - t[0], th_ = t[3] - t[1];
      const float tx = t[0] + 0.5f * tw_, ty = t[1] + 0.5f * th_;
      float tgt[4];
      tgt[0] = __fdiv_rn(wx * (tx - sx), sw_);                           // box_regression.py:59-62
      tgt[1] = __fdiv_rn(wy * (ty - sy), sh_);
      tgt[2] = ww * logf(__fdiv_rn(tw_, sw_));
      tgt[3] = wh * logf(__fdiv_rn(th_, sh_));
      const float* pd = logits + ((long)pv * R + r) * ld + box_col + 4 * gt;
      float dl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { dl[j] = pd[j] - tgt[j]; lb += fabsf(dl[j]); }       // every lane: the same four terms, same order
      const float dmine = lane == 0 ? dl[0] : lane == 1 ? dl[1] : lane == 2 ? dl[2] : dl[3];
      dsum += gs_box * (dmine > 0.f ? 1.f : (dmine < 0.f ? -1.f : 0.f));
    }
    if (lane == 0) {
      row_loss[(long)v * R + r] = ce;
      row_loss[((long)V + v) * R + r] = lb;
    }
  }
  if (DL) {
    // CE gradient is identical for every target view served by this prediction view
    const float sc = gs_cls * w * (float)ntgt;
    for (int j = lane; j < K1; j += 64) {
      const float p = expf(x[j] - m) / z;
      DL[cls_col + j] = (gt >= 0 && sc != 0.f) ? sc * (p - (j == gt ? 1.f : 0.f)) : 0.f;
    }
    const float d0 = __shfl(dsum, 0), d1 = __shfl(dsum, 1), d2 = __shfl(dsum, 2), d3 = __shfl(dsum, 3);
    for (int j = lane; j < 4 * K; j += 64) {                             // one store per column: the gt class's four deltas, else zero
      const int k = j - 4 * gt;
      DL[box_col + j] = (fg && k >= 0 && k < 4) ? (k == 0 ? d0 : k == 1 ? d1 : k == 2 ? d2 : d3) : 0.f;
    }
  }
}

// loss_view[i] = sum_r row_loss[i][r] / R, fixed summation order (i = (round * 2 + term) * V + view)
__global__ __launch_bounds__(256) void refine_reduce_kernel(int R, const float* __restrict__ row_loss,
                                                            float* __restrict__ loss_view) {
  __shared__ float red[32];
  const float* src = row_loss + (long)blockIdx.x * R;
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) s += src[r];
  s = block_reduce_sum(s, red);
  if (threadIdx.x == 0) loss_view[blockIdx.x] = s / (float)R;           // mean over ALL R (:269-273) / sum over R (:351)
}

// ------------------------------------------------------------------------------------------- Stage-3 focal classification loss
// unbias/ubteacher/modeling/roi_heads/fast_rcnn.py:73-105 (FastRCNNFocalLoss.comput_focal_loss + FocalLoss.forward):
//   CE_r = logsumexp(x_r) - x_r[t_r];  p_r = exp(-CE_r);  loss = sum_r (1 - p_r)^gamma * CE_r / N
// One wave per proposal row (C = K + 1 <= a few hundred logits): max / sum-exp by wave shuffles, the per-row term to a scratch
// array (ordered sum by focal_reduce_kernel: deterministic), and the unit gradient
//   dloss/dx_j = scale * ((1 - p)^gamma + gamma * (1 - p)^(gamma - 1) * p * CE) * (softmax_j - [j == t])        (d(1 - p)/dCE = p)
__global__ __launch_bounds__(256) void focal_loss_kernel(int N, int C, const float* __restrict__ logits, long ld,
                                                         const int* __restrict__ target, float gamma, float scale,
                                                         float* __restrict__ row_loss, float* __restrict__ dlogits, long ld_d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= N) return;
  const float* x = logits + (long)r * ld;
  const int t = target[r];
  if ((unsigned)t >= (unsigned)C) {
    // a class outside [0, C): torch's cross_entropy device-asserts here (the reference); this kernel must neither read out of
    // bounds nor return a plausible number — the row's loss and gradient become NaN, which the trainer's finite check reports
    const float nan = __uint_as_float(0x7FC00000u);
    if (lane == 0) row_loss[r] = nan;
    if (dlogits) for (int j = lane; j < C; j += 64) dlogits[(long)r * ld_d + j] = nan;
    return;
  }
  float m = -FLT_MAX;
  for (int j = lane; j < C; j += 64) m = fmaxf(m, x[j]);
  m = wave_reduce_max(m);
  float se = 0.f;
  for (int j = lane; j < C; j += 64) se += expf(x[j] - m);
  se = wave_reduce_sum(se);
  const float lse = m + logf(se);
  const float ce = lse - x[t];
  const float p = expf(-ce);
  const float om = fmaxf(1.f - p, 0.f);
  const float w = powf(om, gamma);
  if (lane == 0) row_loss[r] = w * ce;
  if (dlogits) {
    const float dw = (gamma > 0.f && om > 0.f) ? gamma * powf(om, gamma - 1.f) * p * ce : 0.f;
    const float g = scale * (w + dw);
    float* d = dlogits + (long)r * ld_d;
    for (int j = lane; j < C; j += 64) d[j] = g * (expf(x[j] - lse) - (j == t ? 1.f : 0.f));
  }
}

__global__ __launch_bounds__(1024) void focal_reduce_kernel(int N, float scale, const float* __restrict__ row_loss,
                                                            float* __restrict__ loss) {
  __shared__ float red[32];
  float s = 0.f;
  for (int r = threadIdx.x; r < N; r += blockDim.x) s += row_loss[r];
  s = block_reduce_sum(s, red);
  if (threadIdx.x == 0) loss[0] = s * scale;
}

// ------------------------------------------------------------------------------------------- mining + labelling
__device__ __forceinline__ unsigned int orderable(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// ascending sort of this key == descending score, ascending index (the oracle's tie rule)
__device__ __forceinline__ unsigned long long make_key(float score, unsigned int idx) {
  return ((unsigned long long)(~orderable(score)) << 32) | idx;
}

// WAVE_LOCAL (keys in LDS): a pass whose partner distance j is < 128 stays inside one aligned block of 128 keys.  A wave owns
// such a block (lane = the pair whose lower index has bit j clear) and runs all the remaining passes of the stage on it back to
// back: the LDS serves one wave's requests in issue order, so those passes need no workgroup barrier — 2048 keys take 14
// barriers instead of 66 (the mining kernel sorts one score column per image-level class on the step's critical path).
// (Holding the block in registers and exchanging by __shfl_xor was slower — four ds_bpermute per pass — and so was giving a thread
// four independent pairs per pass: a pass costs ~0.2 us of issue + LDS round trip either way; in-kernel stamps at R = 2000, G = 2:
// column sorts 27 us, threshold + keys 2, NMS sort 7, NMS 12, labels 6.)
// nseg > 1: nseg independent arrays of N keys back to back, all sorted ascending by the same passes (the passes are latency bound,
// so several score columns sort in the time of one).
template <bool WAVE_LOCAL>
__device__ void bitonic_sort(unsigned long long* keys, int N, int nseg = 1) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int T = N * nseg;
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (WAVE_LOCAL && j < 128) {
        for (int base = wave * 128; base < T; base += nwaves * 128) {
          for (int jj = j; jj > 0; jj >>= 1) {
            const int i = base + (((lane & ~(jj - 1)) << 1) | (lane & (jj - 1))), p = i | jj;
            if (p < T) {
              const unsigned long long a = keys[i], b = keys[p];
              const bool asc = ((i & (N - 1)) & k) == 0;
              if ((a > b) == asc) { keys[i] = b; keys[p] = a; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
        __syncthreads();
        break;
      }
      for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          const bool asc = ((i & (N - 1)) & k) == 0;
          if ((a > b) == asc) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int next_pow2(int n) { int p = 64; while (p < n) p <<= 1; return p; }

__device__ __forceinline__ float box_area(const float* b) { return __fmul_rn(b[2] - b[0], b[3] - b[1]); }

// torchvision nms IoU: inter / (a_i + a_j - inter), inter = max(0,dx)*max(0,dy)
__device__ __forceinline__ float iou_nms(const float* a, const float* b) {
  const float w = fmaxf(0.f, fminf(a[2], b[2]) - fmaxf(a[0], b[0]));
  const float h = fmaxf(0.f, fminf(a[3], b[3]) - fmaxf(a[1], b[1]));
  const float inter = __fmul_rn(w, h);
  return __fdiv_rn(inter, __fadd_rn(box_area(a), box_area(b)) - inter);
}
// iou_nms(a, b) > thresh for thresh >= 0, without the division for disjoint boxes: inter == 0 gives 0 / union = 0 (or 0 / 0 = NaN for
// two empty boxes), never > thresh — most pairs of a 2000-candidate list are disjoint
__device__ __forceinline__ bool iou_nms_above(const float* a, const float* b, float thresh) {
  const float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  const float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  if (!(w > 0.f && h > 0.f)) return false;
  const float inter = __fmul_rn(w, h);
  return __fdiv_rn(inter, __fadd_rn(box_area(a), box_area(b)) - inter) > thresh;
}
// detectron2 pairwise_iou (structures/boxes.py:329-361): 0 where inter <= 0
__device__ __forceinline__ float iou_pair(const float* gtb, const float* pb) {
  const float w = fmaxf(fminf(gtb[2], pb[2]) - fmaxf(gtb[0], pb[0]), 0.f);
  const float h = fmaxf(fminf(gtb[3], pb[3]) - fmaxf(gtb[1], pb[1]), 0.f);
  const float inter = __fmul_rn(w, h);
  return inter > 0.f ? __fdiv_rn(inter, __fadd_rn(box_area(gtb), box_area(pb)) - inter) : 0.f;
}

__device__ __forceinline__ float from_orderable(unsigned int o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

// STAGED (the usual case: the sort keys + 25 bytes per (rank, class) slot fit LDS): the slot lists live in LDS, the NMS sort keys
// are built straight from the thresholded slots (slot order = masked_select order, so the slot number is the tie-break), and the
// greedy NMS walks the candidates in SORTED order with their boxes gathered once into that order — the serial part of the kernel
// (one step per kept box, ~25 at NMS 0.01) reads LDS only.  With the lists in the global workspace each kept box cost two
// dependent L2 round trips per thread plus tid 0's three, and the label pass paid the same per pseudo box (116 us for R = 2000,
// G = 2 — and the step waits on this kernel).  The workspace form stays for slot counts beyond LDS.
template <bool STAGED>
__global__ __launch_bounds__(1024) void mine_label_kernel(int R, int ncol, int K, const float* __restrict__ scores,
                                                          const int* __restrict__ gt_classes, int G,
                                                          const float* __restrict__ boxes, int top_k, float score_thresh,
                                                          float nms_thresh, float iou_bg, float iou_fg,
                                                          int* __restrict__ lab_class, float* __restrict__ lab_weight,
                                                          int* __restrict__ lab_index, int* __restrict__ pgt_count,
                                                          int* __restrict__ pgt_index, int* __restrict__ pgt_class,
                                                          float* __restrict__ pgt_score, char* __restrict__ ws,
                                                          long ws_stride, int keys_in_ws, int class_batch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n_slots = top_k * G;
  {                                             // one workgroup per refinement round, rounds laid out back to back
    const int round = blockIdx.x;
    scores += (long)round * R * ncol;
    lab_class += (long)round * R; lab_weight += (long)round * R; lab_index += (long)round * R;
    pgt_count += round; pgt_index += (long)round * n_slots; pgt_class += (long)round * n_slots;
    pgt_score += (long)round * n_slots; ws += (long)round * ws_stride;
  }
  const int NP = next_pow2(R > n_slots ? R : n_slots);
  // sort keys: LDS while NP * 8 B + the suppression bytes fit (R, top_k*G <= 16384: every VOC / COCO recipe with up to 16
  // image-level classes), else a key array at the end of this round's workspace (COCO's 10000 proposals with >= 17 classes:
  // one workgroup sorts through its CU's write-through L1, __syncthreads() orders the passes)
  unsigned long long* keys = (!STAGED && keys_in_ws) ? (unsigned long long*)(ws + ws_stride - (long)NP * 8) : (unsigned long long*)smem;
  char* const lists = STAGED ? smem + (size_t)max(NP, class_batch * next_pow2(R)) * 8 : ws;   // STAGED: [n_slots] float4 first (16-byte aligned)
  float4* const sbox = (float4*)lists;                                     // STAGED only: candidate boxes in sorted order
  float* slot_score = (float*)(lists + (STAGED ? (size_t)n_slots * 16 : 0));   // [n_slots]  rank-major, class-minor
  int* slot_idx = (int*)(slot_score + n_slots);
  float* c_score = (float*)(slot_idx + n_slots);                         // workspace form only: compacted candidates
  int* c_idx = (int*)(c_score + n_slots);
  int* c_cls = c_idx + n_slots;
  unsigned char* sup = STAGED ? (unsigned char*)(slot_idx + n_slots) : (unsigned char*)(smem + (keys_in_ws ? 0 : (size_t)NP * 8));   // [n_slots]
  __shared__ int s_scan[1024];
  __shared__ int s_nk;
  const int tid = threadIdx.x;

  // ---- 1. per gt class: full sort of the score column, take the first top_k (get_pgt_top_k :646-666)
  const int NPR = next_pow2(R);
  for (int g0 = 0; g0 < G; g0 += class_batch) {              // class_batch score columns sort side by side (STAGED: as LDS allows)
    const int nb = min(class_batch, G - g0);
    for (int i = tid; i < NPR * nb; i += blockDim.x) {
      const int b = i / NPR, r = i - b * NPR;
      keys[i] = r < R ? make_key(scores[(long)r * ncol + gt_classes[g0 + b]], (unsigned)r) : ~0ull;
    }
    __syncthreads();
    if (!STAGED && keys_in_ws) bitonic_sort<false>(keys, NPR, nb); else bitonic_sort<true>((unsigned long long*)smem, NPR, nb);
    for (int i = tid; i < top_k * nb; i += blockDim.x) {
      const int b = i / top_k, rank = i - b * top_k;
      const unsigned long long key = keys[b * NPR + rank];
      slot_idx[rank * G + g0 + b] = (int)(unsigned int)(key & 0xFFFFFFFFu);
      slot_score[rank * G + g0 + b] = from_orderable(~(unsigned int)(key >> 32));     // the key holds the score bit for bit
    }
    __syncthreads();
  }
  if (!STAGED) __threadfence_block();
  // ---- 2. threshold mask with rank 0 always kept (:698-704), masked_select order = slot order
  const int per = (n_slots + 1023) / 1024;
  int cnt = 0;
  for (int s = tid * per; s < min(n_slots, (tid + 1) * per); ++s)
    cnt += (s < G || slot_score[s] >= score_thresh) ? 1 : 0;
  s_scan[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int add = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  const int n = s_scan[1023];
  const int NPN = next_pow2(n);
  int pos = s_scan[tid] - cnt;
  // ---- 3. class-agnostic greedy NMS (get_pgt_mist :576-581; torchvision nms): sort by score desc, position asc
  if (STAGED) {
    for (int s = tid * per; s < min(n_slots, (tid + 1) * per); ++s)
      if (s < G || slot_score[s] >= score_thresh) keys[pos++] = make_key(slot_score[s], (unsigned)s);
    for (int i = n + tid; i < NPN; i += blockDim.x) keys[i] = ~0ull;
  } else {
    for (int s = tid * per; s < min(n_slots, (tid + 1) * per); ++s) {
      if (s < G || slot_score[s] >= score_thresh) {
        c_score[pos] = slot_score[s]; c_idx[pos] = slot_idx[s]; c_cls[pos] = gt_classes[s % G];
        ++pos;
      }
    }
    __syncthreads();
    for (int i = tid; i < NPN; i += blockDim.x) keys[i] = i < n ? make_key(c_score[i], (unsigned)i) : ~0ull;
    for (int i = tid; i < n; i += blockDim.x) sup[i] = 0;
  }
  if (tid == 0) s_nk = 0;
  __syncthreads();
  if (!STAGED && keys_in_ws) bitonic_sort<false>(keys, NPN); else bitonic_sort<true>((unsigned long long*)smem, NPN);
  int nk = 0;
  if (STAGED) {
    for (int t = tid; t < n; t += blockDim.x) {
      const int row = slot_idx[(int)(keys[t] & 0xFFFFFFFFu)];
      sbox[t] = *(const float4*)(boxes + (long)row * 4);
      sup[t] = 0;
    }
    __syncthreads();
    const int lane = tid & 63;
    int t = 0;
    while (true) {
      // next candidate not yet suppressed: every wave scans the same flags 64 at a time (uniform result, no barrier)
      while (t < n) {
        const int tt = t + lane;
        const unsigned long long alive = __ballot(tt < n && !sup[tt]);
        if (alive) { t += __ffsll((long long)alive) - 1; break; }
        t += 64;
      }
      if (t >= n) break;
      const float4 bp = sbox[t];
      for (int u = t + 1 + tid; u < n; u += blockDim.x) {
        const float4 bq = sbox[u];
        if (!sup[u] && iou_nms((const float*)&bp, (const float*)&bq) > nms_thresh) sup[u] = 1;
      }
      __syncthreads();                         // flags final; everybody holds box t in registers
      if (tid == 0) {                          // the kept boxes are the prefix of the sorted list (nk <= t: slots already consumed)
        const unsigned long long key = keys[t];
        const int slot = (int)(key & 0xFFFFFFFFu);
        sbox[nk] = bp;
        pgt_index[nk] = slot_idx[slot]; pgt_class[nk] = gt_classes[slot % G];
        pgt_score[nk] = from_orderable(~(unsigned int)(key >> 32));
      }
      ++nk; ++t;
    }
  } else {
    for (int t = 0; t < n; ++t) {
      const int p = (int)(keys[t] & 0xFFFFFFFFu);
      if (sup[p]) continue;                                               // uniform across the workgroup
      if (tid == 0) { pgt_index[nk] = c_idx[p]; pgt_class[nk] = c_cls[p]; pgt_score[nk] = c_score[p]; }
      ++nk;
      const float* bp = boxes + (long)c_idx[p] * 4;
      for (int u = t + 1 + tid; u < n; u += blockDim.x) {
        const int q = (int)(keys[u] & 0xFFFFFFFFu);
        if (!sup[q] && iou_nms(bp, boxes + (long)c_idx[q] * 4) > nms_thresh) sup[q] = 1;
      }
      __syncthreads();
    }
  }
  if (tid == 0) { pgt_count[0] = nk; __threadfence(); }
  __syncthreads();
  // ---- 4. IoU matching + labels (pairwise_iou, Matcher [iou_bg, iou_fg] -> {0,-1,1}, roi_heads.py:225-257,266-375)
  for (int r = tid; r < R; r += blockDim.x) {
    const float4 pbv = *(const float4*)(boxes + (long)r * 4);
    const float* pb = (const float*)&pbv;
    float best = -1.f; int bj = 0;
    for (int j = 0; j < nk; ++j) {
      float v;
      if (STAGED) { const float4 gb = sbox[j]; v = iou_pair((const float*)&gb, pb); }
      else v = iou_pair(boxes + (long)pgt_index[j] * 4, pb);
      if (v > best) { best = v; bj = j; }                               // first max = lowest pgt index
    }
    int cls; float w = 0.f; int gi = 0;
    if (nk == 0) { cls = K; }
    else {
      const int lab = best >= iou_fg ? 1 : (best >= iou_bg ? -1 : 0);
      cls = lab == 1 ? pgt_class[bj] : (lab == 0 ? K : -1);
      w = pgt_score[bj]; gi = pgt_index[bj];
    }
    lab_class[r] = cls; lab_weight[r] = w; lab_index[r] = gi;
  }
}

// ------------------------------------------------------------------------------------------- inference (a23)
// all_scores[r][j] = mean_k softmax(cls_score_k logits)[j]; all_boxes[r][4c..] = apply_deltas(mean_k deltas_k, proposal)
// (predict_probs_K / predict_boxes_K fast_rcnn_oicr.py:674-735, Box2BoxTransform.apply_deltas box_regression.py:73-110)
__global__ __launch_bounds__(256) void predict_kernel(int R, int K, int RK, const float* __restrict__ logits, long ld,
                                                      int base_col, int round_stride, const float* __restrict__ boxes,
                                                      float wx, float wy, float ww, float wh, float scale_clamp,
                                                      float* __restrict__ all_scores, float* __restrict__ all_boxes) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int K1 = K + 1;
  const float* L = logits + (long)r * ld;
  float* S = all_scores + (long)r * K1;
  for (int j = 0; j < K1; ++j) S[j] = 0.f;
  for (int k = 0; k < RK; ++k) {                       // probs += softmax(scores_k)  (:727-729)
    const float* x = L + base_col + k * round_stride;
    float m = -FLT_MAX;
    for (int j = 0; j < K1; ++j) m = fmaxf(m, x[j]);
    float z = 0.f;
    for (int j = 0; j < K1; ++j) z += expf(x[j] - m);
    for (int j = 0; j < K1; ++j) S[j] += expf(x[j] - m) / z;
  }
  for (int j = 0; j < K1; ++j) S[j] = __fdiv_rn(S[j], (float)RK);
  const float* b = boxes + (long)r * 4;
  const float w = b[2] - b[0], h = b[3] - b[1];
  const float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
  float* O = all_boxes + (long)r * 4 * K;
  for (int c = 0; c < K; ++c) {
    float d[4];
    for (int q = 0; q < 4; ++q) {
      float acc = 0.f;
      for (int k = 0; k < RK; ++k) acc += L[base_col + k * round_stride + K1 + 4 * c + q];   // sum in branch order (:691-693)
      d[q] = __fdiv_rn(acc, (float)RK);
    }
    const float dx = __fdiv_rn(d[0], wx), dy = __fdiv_rn(d[1], wy);
    const float dw = fminf(__fdiv_rn(d[2], ww), scale_clamp), dh = fminf(__fdiv_rn(d[3], wh), scale_clamp);
    const float px = __fadd_rn(__fmul_rn(dx, w), cx), py = __fadd_rn(__fmul_rn(dy, h), cy);
    const float pw = __fmul_rn(expf(dw), w), ph = __fmul_rn(expf(dh), h);
    O[4 * c + 0] = px - __fmul_rn(0.5f, pw); O[4 * c + 1] = py - __fmul_rn(0.5f, ph);
    O[4 * c + 2] = px + __fmul_rn(0.5f, pw); O[4 * c + 3] = py + __fmul_rn(0.5f, ph);
  }
}

__device__ __forceinline__ float clipf(float v, float hi) { return fminf(fmaxf(v, 0.f), hi); }

// max coordinate over the clipped boxes that pass the score filter (boxes.max() inside batched_nms)
__global__ void det_maxcoord_kernel(int R, int K, float thresh, float imw, float imh, const float* __restrict__ scores,
                                    const float* __restrict__ boxes, float* __restrict__ out) {
  float m = 0.f;
  const long total = (long)R * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / K; const int c = (int)(i - r * K);
    if (scores[r * (K + 1) + c] > thresh) {
      const float* b = boxes + r * 4 * K + 4 * c;
      m = fmaxf(m, fmaxf(fmaxf(clipf(b[0], imw), clipf(b[2], imw)), fmaxf(clipf(b[1], imh), clipf(b[3], imh))));
    }
  }
  m = wave_reduce_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)out, __float_as_uint(m));
}

// one workgroup per class: score filter, sort (score desc, proposal asc), greedy NMS on the class-offset boxes exactly as
// torchvision batched_nms forms them (box + class * (max_coord + 1)), keep the first `topk` (fast_rcnn_oicr.py:124-140).
// The candidates (score > thresh) are compacted before the sort (a class's candidates are a fraction of the R rows: the sort runs on
// next_pow2(candidates) keys, not next_pow2(R)).  When the unused tail of the key array can hold their boxes (n <= NP/3: the RPN's
// per-level lists, a class's detections) the suppression runs on LDS boxes in chunks of 64 sorted candidates: the 64 x 64 IoU matrix
// of a chunk by ballots (wave w: rows 4w .. 4w+3), one wave resolves the chunk serially on 64-bit masks, then every thread tests
// its later candidates against the chunk's kept boxes — the same keep set as the one-box-at-a-time greedy loop (a candidate falls
// iff an earlier KEPT box overlaps it), in nv/64 rounds instead of one round per kept box.
__global__ __launch_bounds__(1024) void det_class_nms_kernel(int R, int K, float thresh, float nms_thresh, int topk,
                                                             float imw, float imh, const float* __restrict__ scores,
                                                             const float* __restrict__ boxes,
                                                             const float* __restrict__ maxcoord,
                                                             int* __restrict__ cls_count, int* __restrict__ cls_rows,
                                                             float* __restrict__ cls_scores, int* __restrict__ nvalid,
                                                             unsigned long long* __restrict__ skeys, float4* __restrict__ sboxes,
                                                             int mask_min) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NP = next_pow2(R);
  unsigned long long* keys = (unsigned long long*)smem;           // [NP]
  unsigned char* sup = (unsigned char*)(smem + (size_t)NP * 8);     // [R]
  __shared__ int s_valid, s_nk;
  __shared__ unsigned long long s_row[64], s_kept;
  const float off = __fmul_rn((float)c, __fadd_rn(maxcoord[0], 1.0f));
  const bool pos_thresh = nms_thresh >= 0.f;
#ifdef SW_NMS_TIMING
  long long tt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long t0 = wall_clock64();
#define SW_T(k) { const long long t1 = wall_clock64(); tt[k] += t1 - t0; t0 = t1; }
#else
#define SW_T(k)
#endif
  if (tid == 0) { s_valid = 0; s_nk = 0; }
  __syncthreads();
  for (int i0 = 0; i0 < R; i0 += blockDim.x) {                      // compaction (any order: the keys are unique, the sort orders them)
    const int i = i0 + tid;
    const bool ok = i < R && scores[(long)i * (K + 1) + c] > thresh;
    const unsigned long long m = __ballot(ok);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&s_valid, __popcll(m));
    base = __shfl(base, 0);
    if (ok) keys[base + __popcll(m & ((1ull << lane) - 1))] = make_key(scores[(long)i * (K + 1) + c], (unsigned)i);
    if (i < R) sup[i] = 0;
  }
  __syncthreads();
  const int nv = s_valid;
  const int NPV = next_pow2(nv);
  for (int i = nv + tid; i < NPV; i += blockDim.x) keys[i] = ~0ull;
  __syncthreads();
  SW_T(0)
  bitonic_sort<true>(keys, NPV);
  SW_T(1)
  if (nvalid) {
    // mask form on offer (sw_detect_postprocess2): a class with mask_min or more candidates hands its sorted list to det_mask_kernel /
    // det_resolve_kernel (nvalid = count); a smaller one is finished right here (nvalid = -1: the two kernels skip it)
    if (nv >= mask_min) {
      for (int u = tid; u < nv; u += blockDim.x) {
        const unsigned long long key = keys[u];
        const int q = (int)(key & 0xFFFFFFFFu);
        const float* bq = boxes + (long)q * 4 * K + 4 * c;
        skeys[(long)c * R + u] = key;
        sboxes[(long)c * R + u] = make_float4(__fadd_rn(clipf(bq[0], imw), off), __fadd_rn(clipf(bq[1], imh), off),
                                              __fadd_rn(clipf(bq[2], imw), off), __fadd_rn(clipf(bq[3], imh), off));
      }
      if (tid == 0) nvalid[c] = nv;
      return;
    }
    if (tid == 0) nvalid[c] = -1;
  }
  if ((long)nv * 16 + 16 <= (long)(NP - nv) * 8) {
    float4* sb = (float4*)(((uintptr_t)(keys + nv) + 15) & ~(uintptr_t)15);
    for (int u = tid; u < nv; u += blockDim.x) {
      const int q = (int)(keys[u] & 0xFFFFFFFFu);
      const float* bq = boxes + (long)q * 4 * K + 4 * c;
      sb[u] = make_float4(__fadd_rn(clipf(bq[0], imw), off), __fadd_rn(clipf(bq[1], imh), off), __fadd_rn(clipf(bq[2], imw), off),
                          __fadd_rn(clipf(bq[3], imh), off));
      sup[u] = 0;                                                   // here the flags go by sorted position
    }
    __syncthreads();
    SW_T(2)
    for (int c0 = 0; c0 < nv; c0 += 64) {
      if (s_nk >= topk) break;                                      // uniform (s_nk is written before the barriers below)
      const int cn = min(64, nv - c0);
      // chunk matrix: bit j of s_row[i] = IoU(box c0+i, box c0+j) > nms_thresh
      {
        const bool in = lane < cn;
        const float4 bv = sb[c0 + (in ? lane : 0)];
        float b[4] = {bv.x, bv.y, bv.z, bv.w};
        for (int r = 0; r < 4; ++r) {
          const int i = wave * 4 + r;
          if (i < cn) {                                             // wave-uniform
            const float4 av = sb[c0 + i];
            float a[4] = {av.x, av.y, av.z, av.w};
            const unsigned long long m = __ballot(in && lane > i && (pos_thresh ? iou_nms_above(a, b, nms_thresh) : iou_nms(a, b) > nms_thresh));
            if (lane == 0) s_row[i] = m;
          }
        }
      }
      __syncthreads();
      SW_T(3)
      if (wave == 0) {
        // serial walk over the chunk on wave-uniform 64-bit masks (scalar registers): take the first live candidate, drop what
        // its row suppresses; row i lives in lane i and is fetched by v_readlane
        unsigned long long alive = __ballot(lane < cn && !sup[c0 + lane]);
        unsigned long long kept = 0;
        const int nk0 = __builtin_amdgcn_readfirstlane(s_nk);
        int nk = nk0;
        const unsigned long long myrow = lane < cn ? s_row[lane] : 0ull;
        const int row_lo = (int)(unsigned int)myrow, row_hi = (int)(unsigned int)(myrow >> 32);
        while (alive && nk < topk) {
          const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));
          const unsigned long long ri = (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(row_lo, i) |
                                        ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(row_hi, i) << 32);
          kept |= 1ull << i; ++nk;
          alive &= ~(ri | (1ull << i));
        }
        if ((kept >> lane) & 1ull) {
          const int o = nk0 + __popcll(kept & ((1ull << lane) - 1));
          const unsigned long long key = keys[c0 + lane];
          cls_rows[c * topk + o] = (int)(key & 0xFFFFFFFFu);
          cls_scores[c * topk + o] = from_orderable(~(unsigned int)(key >> 32));            // the key holds the score bit for bit
        }
        if (lane == 0) { s_kept = kept; s_nk = nk; }
      }
      __syncthreads();
      SW_T(4)
      const unsigned long long kept = s_kept;
      if (kept && s_nk < topk) {
        // later candidates against the chunk's kept boxes: a lane holds one candidate (64 per wave and pass), the kept boxes
        // come out of the registers of the lanes that hold the chunk (v_readlane): no LDS access and no divergence inside the
        // loop over the kept boxes.  (A wave-per-candidate form — one broadcast LDS read, 64 tests, one ballot — was bound by
        // the LDS latency of each step: 23 us per chunk against 1 us for the chunk's own 64 x 64 matrix.)
        const unsigned int klo = __builtin_amdgcn_readfirstlane((unsigned int)kept), khi = __builtin_amdgcn_readfirstlane((unsigned int)(kept >> 32));
        const float4 kv = sb[c0 + (lane < cn ? lane : 0)];
        for (int u0 = c0 + 64 + wave * 64; u0 < nv; u0 += 16 * 64) {
          const int u = u0 + lane;
          const bool live = u < nv && !sup[u];
          if (!__ballot(live)) continue;
          const float4 bv = sb[live ? u : c0];
          float b[4] = {bv.x, bv.y, bv.z, bv.w};
          bool hit = false;
#pragma unroll
          for (int half = 0; half < 2; ++half)
            for (unsigned int m = half ? khi : klo; m; m &= m - 1) {
              const int k = __builtin_amdgcn_readfirstlane(__builtin_ctz(m)) + 32 * half;
              float a[4] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(kv.x), k)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kv.y), k)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kv.z), k)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kv.w), k))};
              hit = hit || (pos_thresh ? iou_nms_above(a, b, nms_thresh) : iou_nms(a, b) > nms_thresh);
            }
          if (live && hit) sup[u] = 1;
        }
      }
      __syncthreads();
      SW_T(5)
    }
#ifdef SW_NMS_TIMING
    if (tid == 0 && c == 0) printf("nms level 0 (nv %d, kept %d): compaction %lld sort %lld stage %lld matrix %lld resolve %lld apply %lld  (100 MHz ticks)\n", nv, s_nk, tt[0], tt[1], tt[2], tt[3], tt[4], tt[5]);
#endif
    if (tid == 0) cls_count[c] = s_nk;
    return;
  }
  int nk = 0;
  for (int t = 0; t < nv && nk < topk; ++t) {
    const int p = (int)(keys[t] & 0xFFFFFFFFu);
    if (sup[p]) continue;
    if (tid == 0) { cls_rows[c * topk + nk] = p; cls_scores[c * topk + nk] = scores[(long)p * (K + 1) + c]; }
    ++nk;
    const float* bp = boxes + (long)p * 4 * K + 4 * c;
    float a[4] = {__fadd_rn(clipf(bp[0], imw), off), __fadd_rn(clipf(bp[1], imh), off), __fadd_rn(clipf(bp[2], imw), off),
                  __fadd_rn(clipf(bp[3], imh), off)};
    for (int u = t + 1 + tid; u < nv; u += blockDim.x) {
      const int q = (int)(keys[u] & 0xFFFFFFFFu);
      if (sup[q]) continue;
      const float* bq = boxes + (long)q * 4 * K + 4 * c;
      float b[4] = {__fadd_rn(clipf(bq[0], imw), off), __fadd_rn(clipf(bq[1], imh), off), __fadd_rn(clipf(bq[2], imw), off),
                    __fadd_rn(clipf(bq[3], imh), off)};
      if (iou_nms(a, b) > nms_thresh) sup[q] = 1;
    }
    __syncthreads();
  }
  if (tid == 0) cls_count[c] = nk;
}

// ---------------------------------------------------------------- the same per-class NMS spread over the chip (mask form)
// One workgroup per class is bound by that one CU once a class has ~2000 candidates (N^2 / 2 IoU tests: the RPN's per-level lists;
// measured 23 us per 64-candidate chunk, 0.4-0.7 ms per call).  Mask form, the arithmetic and the keep set unchanged:
//   det_class_nms_kernel (above) per class: candidates compacted + sorted; >= mask_min of them: sorted keys and clipped,
//                       class-offset boxes -> global; fewer: finished there, in LDS
//   det_mask_kernel     tiles of 64 x 64 sorted candidates over ALL CUs: bit j of MT[class][col block][row i] = IoU(i, j) > thresh, j > i
//   det_resolve_kernel  per class one wave walks the chunks of 64 in order: suppressed = OR over the kept rows before the chunk of
//                       their mask words for this chunk (read coalesced: the matrix is stored column-block major), the chunk itself
//                       resolved on wave-uniform 64-bit masks as above; three loader waves stage the next chunk's column meanwhile
__global__ __launch_bounds__(64) void det_mask_kernel(int R, int K, float nms_thresh, const int* __restrict__ nvalid,
                                                      const float4* __restrict__ sboxes, unsigned long long* __restrict__ MT, int NB) {
  __shared__ float4 s_col[64];
  const int lane = threadIdx.x;
  const bool pos_thresh = nms_thresh >= 0.f;
  for (long t = blockIdx.x;; t += gridDim.x) {                     // tile list: per class the (row block <= column block) pairs
    long rem = t;
    int c = 0, nv = 0;
    for (; c < K; ++c) {
      nv = max(nvalid[c], 0);
      const long nb = (nv + 63) >> 6, tri = nb * (nb + 1) / 2;
      if (rem < tri) break;
      rem -= tri;
    }
    if (c >= K) return;                                            // uniform
    int cb = (int)((sqrtf(8.f * (float)rem + 1.f) - 1.f) * 0.5f);
    while ((long)cb * (cb + 1) / 2 > rem) --cb;
    while ((long)(cb + 1) * (cb + 2) / 2 <= rem) ++cb;
    const int rb = (int)(rem - (long)cb * (cb + 1) / 2);
    const int j0 = cb * 64, i = rb * 64 + lane;
    const float4* sb = sboxes + (long)c * R;
    __syncthreads();
    s_col[lane] = j0 + lane < nv ? sb[j0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (i < nv) {
      const float4 av = sb[i];
      float a[4] = {av.x, av.y, av.z, av.w};
      const int jn = min(64, nv - j0);
      unsigned long long bits = 0;
      for (int jj = max(0, i + 1 - j0); jj < jn; ++jj) {
        const float4 bv = s_col[jj];
        float b[4] = {bv.x, bv.y, bv.z, bv.w};
        if (pos_thresh ? iou_nms_above(a, b, nms_thresh) : iou_nms(a, b) > nms_thresh) bits |= 1ull << jj;
      }
      MT[((long)c * NB + cb) * R + i] = bits;
    }
  }
}

constexpr int RES_CAP = 4096;                                       // rows of a chunk's mask column staged in LDS
__global__ __launch_bounds__(256) void det_resolve_kernel(int R, int K, int topk, int NB, const int* __restrict__ nvalid,
                                                          const unsigned long long* __restrict__ skeys,
                                                          const unsigned long long* __restrict__ MT, int* __restrict__ cls_count,
                                                          int* __restrict__ cls_rows, float* __restrict__ cls_scores) {
  __shared__ unsigned long long s_buf[2][RES_CAP];
  __shared__ unsigned long long s_keptm[256];
  __shared__ int s_stop, s_nk;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (nvalid[c] < 0) return;                                     // finished by det_class_nms_kernel (uniform)
  const int nv = nvalid[c], nb = (nv + 63) >> 6;
  const unsigned long long* col = MT + (long)c * NB * R;          // col[w * R + i]: mask word of row i for column block w
  auto stage = [&](int w, int first, int step) {                   // rows [0, (w + 1) * 64) of column block w
    const int nrows = min(min(nv, (w + 1) * 64), RES_CAP);
    for (int i = first; i < nrows; i += step) s_buf[w & 1][i] = col[(long)w * R + i];
  };
  if (tid == 0) { s_stop = 0; s_nk = 0; }
  if (nb > 0) stage(0, tid, 256);
  __syncthreads();
  for (int w = 0; w < nb; ++w) {
    if (wave != 0) {
      if (w + 1 < nb) stage(w + 1, tid - 64, 192);
    } else {
      const unsigned long long* buf = s_buf[w & 1];
      unsigned long long acc = 0;
      for (int p = 0; p < w; ++p) {                                 // kept rows before this chunk: their words for this chunk
        const int i = p * 64 + lane;
        const unsigned long long v = i < RES_CAP ? buf[i] : col[(long)w * R + i];
        if ((s_keptm[p] >> lane) & 1ull) acc |= v;
      }
      unsigned int lo = (unsigned int)acc, hi = (unsigned int)(acc >> 32);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { lo |= (unsigned int)__shfl_xor((int)lo, o); hi |= (unsigned int)__shfl_xor((int)hi, o); }
      const unsigned long long removed = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)hi) << 32) |
                                         (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)lo);
      const int i = w * 64 + lane;
      const unsigned long long myrow = i < nv ? (i < RES_CAP ? buf[i] : col[(long)w * R + i]) : 0ull;
      unsigned long long alive = __ballot(i < nv) & ~removed;
      unsigned long long kept = 0;
      const int nk0 = __builtin_amdgcn_readfirstlane(s_nk);
      int nk = nk0;
      const int row_lo = (int)(unsigned int)myrow, row_hi = (int)(unsigned int)(myrow >> 32);
      while (alive && nk < topk) {
        const int b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));
        const unsigned long long ri = (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(row_lo, b) |
                                      ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(row_hi, b) << 32);
        kept |= 1ull << b; ++nk;
        alive &= ~(ri | (1ull << b));
      }
      if ((kept >> lane) & 1ull) {
        const int o = nk0 + __popcll(kept & ((1ull << lane) - 1));
        const unsigned long long key = skeys[(long)c * R + i];
        cls_rows[c * topk + o] = (int)(key & 0xFFFFFFFFu);
        cls_scores[c * topk + o] = from_orderable(~(unsigned int)(key >> 32));
      }
      if (lane == 0) { s_keptm[w & 255] = kept; s_nk = nk; s_stop = nk >= topk ? 1 : 0; }
    }
    __syncthreads();
    if (s_stop) break;
  }
  if (tid == 0) cls_count[c] = s_nk;
}

// merge the per-class lists: global order = score desc, ties by filtered position (r*K + c) asc; first topk
__global__ __launch_bounds__(1024) void det_merge_kernel(int K, int topk, float imw, float imh, const float* __restrict__ boxes,
                                                         const int* __restrict__ cls_count, const int* __restrict__ cls_rows,
                                                         const float* __restrict__ cls_scores, int* __restrict__ det_count,
                                                         float* __restrict__ det_boxes, float* __restrict__ det_scores,
                                                         int* __restrict__ det_classes, int* __restrict__ det_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n_slots = K * topk, NP = next_pow2(n_slots), tid = threadIdx.x;
  unsigned long long* keys = (unsigned long long*)smem;
  __shared__ int s_n;
  if (tid == 0) { int n = 0; for (int c = 0; c < K; ++c) n += cls_count[c]; s_n = n; }
  for (int i = tid; i < NP; i += blockDim.x) {
    unsigned long long key = ~0ull;
    if (i < n_slots) {
      const int c = i / topk, j = i - c * topk;
      if (j < cls_count[c]) key = make_key(cls_scores[i], (unsigned)(cls_rows[i] * K + c));
    }
    keys[i] = key;
  }
  __syncthreads();
  bitonic_sort<true>(keys, NP);
  const int n = min(s_n, topk);
  for (int t = tid; t < n; t += blockDim.x) {
    const unsigned int pos = (unsigned int)(keys[t] & 0xFFFFFFFFu);
    const int r = pos / K, c = pos - r * K;
    const float* b = boxes + (long)r * 4 * K + 4 * c;
    det_boxes[4 * t + 0] = clipf(b[0], imw); det_boxes[4 * t + 1] = clipf(b[1], imh);
    det_boxes[4 * t + 2] = clipf(b[2], imw); det_boxes[4 * t + 3] = clipf(b[3], imh);
    // score: recover from the sorted key (exact bits)
    const unsigned int ob = ~(unsigned int)(keys[t] >> 32);
    det_scores[t] = __uint_as_float((ob & 0x80000000u) ? (ob & 0x7FFFFFFFu) : ~ob);
    det_classes[t] = c; det_rows[t] = r;
  }
  if (tid == 0) det_count[0] = n;
}

}  // namespace

extern "C" long sw_wsddn_workspace_floats(int V, int R, int K) {
  const long nchunk = (R + WS_CH - 1) / WS_CH;
  return 3L * V * nchunk * K + 2L * V * K;
}

extern "C" int sw_wsddn_mil(int V, int R, int K, const float* logits, long ld, int cls_col, int det_col,
                            const float* gt_onehot, float* scores, float* loss_view, float* dlogits, long ld_d,
                            const float* grad_scale, float* mean_scores, long ld_mean, float* workspace,
                            hipStream_t stream) {
  SW_ENTER();
  if (K > KMAX || V > WS_VMAX || V < 1 || R < 1) return -6;
  const int nchunk = (R + WS_CH - 1) / WS_CH;
  float* pm = workspace;
  float* ps = pm + (long)V * nchunk * K;
  float* pS = ps + (long)V * nchunk * K;
  float* cm = pS + (long)V * nchunk * K;
  float* cz = cm + (long)V * K;
  const size_t lds_sc = (size_t)WS_CH * K * sizeof(float);
  if (lds_sc > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)wsddn_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(wsddn_stats_kernel, dim3(nchunk, V), dim3(WS_CH), lds_sc, stream, R, K, logits, ld, det_col, pm, ps);
  SW_CHECK_LAUNCH();
  if (lds_sc > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)wsddn_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(wsddn_scores_kernel, dim3(nchunk, V), dim3(WS_CH), lds_sc, stream, V, R, K, logits, ld, cls_col, det_col,
                     pm, ps, cm, cz, pS, scores);
  SW_CHECK_LAUNCH();
  if (mean_scores) {
    long blocks = ((long)R * K + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(mean_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, V, R, K, scores, mean_scores, ld_mean);
    SW_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(wsddn_grad_kernel, dim3(dlogits ? nchunk : 1, V), dim3(WS_CH), 0, stream, V, R, K, logits, ld, cls_col,
                     det_col, gt_onehot, cm, cz, pS, nchunk, scores, loss_view, dlogits, ld_d, grad_scale);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_oicr_mean_probs(int V, int R, int K, int n_rounds, const float* logits, long ld, int cls_col0,
                                  int col_stride, float* out, hipStream_t stream) {
  SW_ENTER();
  if (R <= 0 || n_rounds <= 0) return 0;
  if (V > WS_VMAX || V < 1) return -6;
  const size_t lds = (size_t)V * MP_ROWS * (K + 1) * sizeof(float);
  if (lds > 150 * 1024) return -6;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)mean_probs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(mean_probs_kernel, dim3((R + MP_ROWS - 1) / MP_ROWS, n_rounds), dim3(MP_ROWS * V), lds, stream, V, R, K,
                     logits, ld, cls_col0, col_stride, out);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_oicr_refine_loss(int V, int R, int K, int n_rounds, const float* logits, long ld, int cls_col,
                                   int box_col, int col_stride, const float* boxes, const int32_t* lab_class,
                                   const float* lab_weight, const int32_t* lab_index, const int32_t* pred_view,
                                   const float* reg_weights4, float* loss_view, float* dlogits, long ld_d,
                                   const float* grad_scale, float* workspace, hipStream_t stream) {
  SW_ENTER();
  // reg_weights4: HOST pointer (configuration constants BBOX_REG_WEIGHTS); workspace: n_rounds*2*V*R floats
  if (R <= 0 || n_rounds <= 0) return 0;
  hipLaunchKernelGGL(refine_loss_kernel, dim3((R + 3) / 4, V, n_rounds), dim3(256), 0, stream, V, R, K, logits, ld,
                     cls_col, box_col, boxes, lab_class, lab_weight, lab_index, pred_view, reg_weights4[0], reg_weights4[1],
                     reg_weights4[2], reg_weights4[3], workspace, dlogits, ld_d, grad_scale, col_stride);
  SW_CHECK_LAUNCH();
  hipLaunchKernelGGL(refine_reduce_kernel, dim3(2 * V * n_rounds), dim3(256), 0, stream, R, workspace, loss_view);
  SW_CHECK_LAUNCH();
  return 0;
}

static int mine_np(int R, int top_k, int G) {
  const long need = R > (long)top_k * G ? R : (long)top_k * G;
  int np = 64;
  while (np < need) np <<= 1;
  return np;
}
static int mine_npr(int R) { int np = 64; while (np < R) np <<= 1; return np; }
static size_t mine_staged_lds(int R, int top_k, int G, int batch = 1) {        // keys + 25 bytes per slot (mine_label_kernel<true>)
  const size_t nkeys = std::max((size_t)mine_np(R, top_k, G), (size_t)batch * mine_npr(R));
  return nkeys * 8 + (size_t)top_k * G * 25 + 16;
}
static bool mine_staged(int R, int top_k, int G) { return mine_staged_lds(R, top_k, G) <= 144 * 1024; }
static int mine_class_batch(int R, int top_k, int G) {          // score columns sorted side by side: as many as LDS holds, <= 8
  int b = 1;
  while (b < G && b < 8 && mine_staged_lds(R, top_k, G, b + 1) <= 144 * 1024) ++b;
  return b;
}
static bool mine_keys_in_ws(int R, int top_k, int G) {
  return !mine_staged(R, top_k, G) && (size_t)mine_np(R, top_k, G) * 8 + (size_t)top_k * G > 144 * 1024;
}
extern "C" long sw_mine_workspace_bytes(int R, int top_k, int G) {
  long b = (((long)top_k * G * 20 + 64) + 15) / 16 * 16;
  if (mine_keys_in_ws(R, top_k, G)) b += (long)mine_np(R, top_k, G) * 8;
  return b;
}

extern "C" int sw_oicr_mine_label(int R, int ncol, int K, int n_rounds, const float* scores, const int32_t* gt_classes,
                                  int G, const float* boxes, int top_k, float score_thresh, float nms_thresh,
                                  float iou_bg, float iou_fg, int32_t* lab_class, float* lab_weight,
                                  int32_t* lab_index, int32_t* pgt_count, int32_t* pgt_index, int32_t* pgt_class,
                                  float* pgt_score, void* workspace, hipStream_t stream) {
  SW_ENTER();
  if (R > (1 << 22) || (long)top_k * G > (1 << 22) || top_k > R || G < 1 || n_rounds < 1) return -6;
  if (((uintptr_t)boxes) & 15) return -4;                    // proposal boxes are read as float4
  const int np = mine_np(R, top_k, G);
  static const bool no_stage = getenv("SW_MINE_NO_STAGE") != nullptr;      // development switch
  const bool staged = !no_stage && mine_staged(R, top_k, G);
  const int in_ws = (!staged && (size_t)np * 8 + (size_t)top_k * G > 144 * 1024) ? 1 : 0;
  if (in_ws && !mine_keys_in_ws(R, top_k, G)) return -6;    // (only with the development switch: the workspace has no key array)
  const int batch = staged ? mine_class_batch(R, top_k, G) : 1;
  const size_t lds = staged ? mine_staged_lds(R, top_k, G, batch) : (in_ws ? 0 : (size_t)np * 8) + (size_t)top_k * G;
  if (lds > 144 * 1024) return -6;
  auto kern = staged ? mine_label_kernel<true> : mine_label_kernel<false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(n_rounds), dim3(1024), lds, stream, R, ncol, K, scores, gt_classes, G, boxes,
                     top_k, score_thresh, nms_thresh, iou_bg, iou_fg, lab_class, lab_weight, lab_index, pgt_count,
                     pgt_index, pgt_class, pgt_score, (char*)workspace, sw_mine_workspace_bytes(R, top_k, G), in_ws, batch);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_focal_loss(int N, int C, const float* logits, long ld, const int32_t* targets, float gamma, float* loss,
                             float* dlogits, long ld_d, float* workspace, hipStream_t stream) {
  SW_ENTER();
  if (N <= 0 || C <= 0) return -1;           // the reference returns 0 * logits.sum() for an empty batch: the caller's branch
  const float scale = 1.0f / (float)N;       // total_loss / gt_classes.shape[0]
  hipLaunchKernelGGL(focal_loss_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, N, C, logits, ld, targets, gamma, scale, workspace,
                     dlogits, ld_d);
  SW_CHECK_LAUNCH();
  hipLaunchKernelGGL(focal_reduce_kernel, dim3(1), dim3(1024), 0, stream, N, scale, workspace, loss);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_oicr_predict(int R, int K, int refine_k, const float* logits, long ld, int base_col, int round_stride,
                               const float* boxes, const float* reg_weights4, float scale_clamp, float* all_scores,
                               float* all_boxes, hipStream_t stream) {
  SW_ENTER();
  if (R <= 0) return 0;
  hipLaunchKernelGGL(predict_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, R, K, refine_k, logits, ld, base_col,
                     round_stride, boxes, reg_weights4[0], reg_weights4[1], reg_weights4[2], reg_weights4[3], scale_clamp,
                     all_scores, all_boxes);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" long sw_detect_workspace_bytes(int K, int topk) { return (long)K * topk * 8 + (long)K * 4 + 64; }

namespace {
inline long align256(long v) { return (v + 255) / 256 * 256; }
// extra workspace of the mask form: nvalid[K] | sorted keys [K][R] | sorted boxes [K][R] float4 | MT [K][NB][R] u64; 0 = not offered
inline long detect_mask_bytes(int R, int K) {
  if (R < 1 || K < 1 || R > 16384) return 0;
  const long NB = (R + 63) / 64;
  if (NB > 256) return 0;
  const long b = align256((long)K * 4) + align256((long)K * R * 8) + align256((long)K * R * 16) + (long)K * NB * R * 8;
  return b <= (96L << 20) ? b : 0;
}
}  // namespace

extern "C" long sw_detect_workspace_bytes2(int R, int K, int topk) {
  return align256(sw_detect_workspace_bytes(K, topk)) + detect_mask_bytes(R, K);
}

static int detect_postprocess_impl(int R, int K, const float* all_scores, const float* all_boxes, int img_h, int img_w,
                                   float score_thresh, float nms_thresh, int topk, int32_t* det_count, float* det_boxes,
                                   float* det_scores, int32_t* det_classes, int32_t* det_rows, void* workspace, long workspace_bytes,
                                   hipStream_t stream) {
  if (R > 16384 || (long)K * topk > 16384 || topk < 1) return -6;
  char* ws = (char*)workspace;
  float* maxcoord = (float*)ws;                                    // [1] (+pad)
  int* cls_count = (int*)(ws + 64);                                // [K]
  int* cls_rows = cls_count + K;                                   // [K][topk]
  float* cls_scores = (float*)(cls_rows + (long)K * topk);         // [K][topk]
  hipError_t e = hipMemsetAsync(maxcoord, 0, 64, stream);
  if (e != hipSuccess) return (int)e;
  if (R <= 0) return (int)hipMemsetAsync(det_count, 0, 4, stream);
  const long total = (long)R * K;
  long blocks = (total + 255) / 256; if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(det_maxcoord_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, R, K, score_thresh, (float)img_w,
                     (float)img_h, all_scores, all_boxes, maxcoord);
  SW_CHECK_LAUNCH();
  int np = 64; while (np < R) np <<= 1;
  static const bool no_mask = getenv("SW_NMS_NO_MASK") != nullptr;            // development switch
  static const int mask_min = getenv("SW_NMS_MASK_MIN") ? atoi(getenv("SW_NMS_MASK_MIN")) : 256;     // development switch
  const long base = align256(sw_detect_workspace_bytes(K, topk)), mask_need = detect_mask_bytes(R, K);
  const bool mask = !no_mask && mask_need > 0 && workspace_bytes >= base + mask_need && R >= mask_min;
  char* m = ws + base;
  int* nvalid = mask ? (int*)m : nullptr; m += align256((long)K * 4);
  unsigned long long* skeys = (unsigned long long*)m; m += align256((long)K * R * 8);
  float4* sboxes = (float4*)m; m += align256((long)K * R * 16);
  unsigned long long* MT = (unsigned long long*)m;
  const size_t lds1 = (size_t)np * 8 + (size_t)R;
  e = hipFuncSetAttribute((const void*)det_class_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(det_class_nms_kernel, dim3(K), dim3(1024), lds1, stream, R, K, score_thresh, nms_thresh, topk,
                     (float)img_w, (float)img_h, all_scores, all_boxes, maxcoord, cls_count, cls_rows, cls_scores, nvalid, skeys, sboxes,
                     mask_min);
  SW_CHECK_LAUNCH();
  if (mask) {
    // classes of >= mask_min candidates: the 64 x 64 IoU tiles over the chip, then one resolving wave per class
    const int NB = (R + 63) / 64;
    const long max_tiles = (long)NB * (NB + 1) / 2;               // every candidate in one class
    const int mgrid = (int)(max_tiles < 4096 ? max_tiles : 4096);
    hipLaunchKernelGGL(det_mask_kernel, dim3(mgrid), dim3(64), 0, stream, R, K, nms_thresh, nvalid, sboxes, MT, NB);
    SW_CHECK_LAUNCH();
    hipLaunchKernelGGL(det_resolve_kernel, dim3(K), dim3(256), 0, stream, R, K, topk, NB, nvalid, skeys, MT, cls_count, cls_rows,
                       cls_scores);
    SW_CHECK_LAUNCH();
  }
  int np2 = 64; while (np2 < K * topk) np2 <<= 1;
  const size_t lds2 = (size_t)np2 * 8;
  e = hipFuncSetAttribute((const void*)det_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(det_merge_kernel, dim3(1), dim3(1024), lds2, stream, K, topk, (float)img_w, (float)img_h, all_boxes,
                     cls_count, cls_rows, cls_scores, det_count, det_boxes, det_scores, det_classes, det_rows);
  SW_CHECK_LAUNCH();
  return 0;
}

extern "C" int sw_detect_postprocess(int R, int K, const float* all_scores, const float* all_boxes, int img_h, int img_w,
                                     float score_thresh, float nms_thresh, int topk, int32_t* det_count, float* det_boxes,
                                     float* det_scores, int32_t* det_classes, int32_t* det_rows, void* workspace,
                                     hipStream_t stream) {
  SW_ENTER();
  return detect_postprocess_impl(R, K, all_scores, all_boxes, img_h, img_w, score_thresh, nms_thresh, topk, det_count, det_boxes,
                                 det_scores, det_classes, det_rows, workspace, 0, stream);
}

extern "C" int sw_detect_postprocess2(int R, int K, const float* all_scores, const float* all_boxes, int img_h, int img_w,
                                      float score_thresh, float nms_thresh, int topk, int32_t* det_count, float* det_boxes,
                                      float* det_scores, int32_t* det_classes, int32_t* det_rows, void* workspace,
                                      long workspace_bytes, hipStream_t stream) {
  SW_ENTER();
  return detect_postprocess_impl(R, K, all_scores, all_boxes, img_h, img_w, score_thresh, nms_thresh, topk, det_count, det_boxes,
                                 det_scores, det_classes, det_rows, workspace, workspace_bytes, stream);
}
