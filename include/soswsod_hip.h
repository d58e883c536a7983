/*
 * soswsod_hip.h — C ABI of the MI355X-native (gfx950) kernels behind SoS-WSOD's Stage-1 OICR+ hot path.
 *
 * The reference has no C FFI on this path: its operator boundary is a pybind11 module called from
 * torch.autograd.Function (uwsod/projects/WSL/wsl/layers/roi_loop_pool.py:9-35 <->
 * uwsod/projects/WSL/wsl/layers/csrc/vision.cpp:21-23 <-> .../ROILoopPool/ROILoopPool.h:48-107), plus the
 * third-party ops it calls (torchvision RoIPool / nms, cuDNN conv, cuBLAS GEMM).  This header is the plain-C
 * replacement of that operator tier; each entry cites the reference code it stands in for.
 *
 * Conventions (all entry points):
 *   - extern "C", return 0 on success, a positive hipError_t from the launch, or a negative argument error
 *     (-1 bad dtype, -2 split-K without atomic accumulate, -3 unsupported operand layout, -4 pointer not
 *     16-byte aligned, -5 leading dimension / extent not a multiple of the 16-byte chunk, -6 size limit).
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated, freed or synchronised
 *     inside; work is enqueued on `stream`; no global state => thread-compatible.
 *   - dtype: SW_F32 (0) or SW_BF16 (1; raw 16-bit words).  Activations are NHWC.
 */
#ifndef SOSWSOD_HIP_H
#define SOSWSOD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef SW_F32
#define SW_F32 0
#define SW_BF16 1
#endif

typedef struct ihipStream_t* sw_stream_t; /* == hipStream_t */

/* Fused GEMM / implicit-GEMM epilogue:  v = acc (+bias[n]); relu; dropout; relu-backward mask; store. */
typedef struct sw_epilogue {
  const float* bias;          /* [N] or NULL */
  int relu;                   /* v = max(v, 0) */
  const uint8_t* drop_mask;   /* [M][ld_drop] keep mask or NULL: v = keep ? v*drop_scale : 0   (F.dropout, box_head.py:90) */
  long ld_drop;
  float drop_scale;
  const void* relu_ref;       /* [M][ld_ref] or NULL: v = ref>0 ? v*ref_scale : 0   (ReLU(+dropout) backward) */
  long ld_ref;
  float ref_scale;
  int ref_dtype;
  int out_dtype;              /* dtype of C */
  int accumulate_atomic;      /* C is f32 and is atomically accumulated (required when splitk > 1) */
  float* absmax_out;          /* optional device scalar (caller zero-fills): atomicMax of |stored value| (sw_gemm only) */
  /* in-epilogue dropout (drop_mask == NULL, drop_hash_p > 0): element (m, n) is kept iff the keep mask that
   * sw_dropout_mask(seed, offset, p) writes at index m*N + n is 1 — the same Bernoulli stream without the mask tensor */
  uint64_t drop_seed, drop_offset;
  float drop_hash_p;
  /* optional DEVICE counter added to drop_offset when the kernel runs: the stream position then lives on the device
   * (sw_counter_add advances it in-stream), so a captured hipGraph replays with a fresh dropout mask every time */
  const uint64_t* drop_offset_dev;
  /* deterministic split-K (sw_gemm, f32 C, no other epilogue option): every K-split stores its partial tile into its own
   * slab of this workspace (sw_gemm_splitk_workspace_floats floats), a second kernel adds the slabs in fixed order into C
   * (overwritten) — no atomics.  Also used, when given, for the tail peel of the large f32-output GEMMs (see sw_gemm).
   * With OTHER epilogue options set (bias, residual, relu, relu_ref, out_dtype bf16; not the dropout / atomic / absmax ones) and
   * effective splits > 1 the slabs stay plain f32 and the fold applies the epilogue in the GEMM's own order:
   * x = fold_row_scale[m] * sum + bias[n] + residual[m][n]; ReLU; ReLU-mask; convert.  An f32 residual may be C itself (C += A B: a
   * gradient added to one that exists).  One slab is legal when fold_row_scale meets a residual.  Requires N % 4 == 0, ldc % 4 == 0. */
  float* splitk_workspace;
  /* residual [M][ld_res] or NULL (dtype res_dtype): v = v + bias + residual before the ReLU — the shortcut add of a ResNet
   * bottleneck (detectron2/modeling/backbone/resnet.py:205-212 `out += shortcut; out = F.relu_(out)`) inside the 1x1 conv3 GEMM */
  const void* residual;
  long ld_res;
  int res_dtype;
  /* deterministic split-K only (splitk_workspace given, effective splits > 1): the ordered fold writes C[m][n] = fold_row_scale[m] *
   * sum of the slabs (DEVICE [M] or NULL) — the FrozenBN fold of a 1x1 convolution's weight gradient, dW = scale * dW_eff
   * (detectron2/layers/batch_norm.py:52-58), without a pass of its own.  Without a fold (one split, plain f32 C) it runs as a
   * row-scaling pass after the GEMM; sw_gemm returns -5 for any other combination that does not end in a fold. */
  const float* fold_row_scale;
  /* Fused SGD update (round 6; sw_gemm only, bf16 operands, f32 "output", no other epilogue option, the 256x256 ping-pong tile: K-contiguous
   * A, M % 32 == 0, N % 64 == 0 — sw_gemm_sgd_fused_supported answers for a shape): C = A B is the GRADIENT of the [M][N] float32 parameter
   * sgd_fused->param (row pitch = ldc; stage_kind 3: row-major copy stage0 [M][ld0] and transposed copy stage1 [N][ld1], bf16).  The
   * epilogue applies torch.optim.SGD's momentum update (sw_sgd_multi's arithmetic, bit for bit) to the parameter, its momentum buffer and
   * both copies; the gradient is NOT written, except the peeled tail columns (see sw_gemm), which go through C and the tiled update kernel.
   * What it removes: the 411 MB write + read of fc1.weight's gradient and a serialized HBM-bound optimizer launch (the reference:
   * optimizer.step() after the backward, train_net_multi.py:157-164; single GPU, ITER_SIZE 1 only — a data-parallel step needs the
   * all-reduced gradient).  sgd_fused->grad, ->n, ->d1, ->d2 are ignored; ->d0 must equal N. */
  const struct sw_sgd_tensor_s* sgd_fused;
  float sgd_momentum, sgd_grad_scale;
} sw_epilogue;

/* ---- dense contractions (reference: cuBLAS via torch Linear — box_head.py:88-90,
 *      fast_rcnn_wsddn.py:558-559, fast_rcnn_oicr.py:517-519, and their autograd backward) ----------------
 * C[m][n] = sum_k A(m,k) B(k,n).  a_kstrided=0: A[m*lda+k], 1: A[k*lda+m];  b_kstrided=0: B[n*ldb+k], 1: B[k*ldb+n].
 * Supported (a,b): (0,0) forward, (0,1) data gradient, (1,1) weight gradient. */
int sw_gemm(int dtype, int a_kstrided, int b_kstrided, int M, int N, int K, const void* A, long lda, const void* B,
            long ldb, void* C, long ldc, const sw_epilogue* ep, int splitk, sw_stream_t stream);
/* floats of sw_epilogue.splitk_workspace that sw_gemm(M, N, K, splitk) may use (its own tail peel included) */
long sw_gemm_splitk_workspace_floats(int M, int N, int K, int splitk);

/* ---- 3x3 convolution, stride 1, padding = dilation (reference: torch Conv2d/cuDNN, wsl/modeling/backbone/vgg.py:44-97,104-122)
 * in  [nimg][H][W][Cin], wk [Cout][3*3][Cin] (see sw_conv_weight_prep), out [nimg][H][W][Cout].
 * Used for forward (ep: bias+relu) and for the data gradient (wk = flipped/transposed weights, ep: relu_ref). */
int sw_conv3x3_igemm(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* in,
                     const void* wk, void* out, const sw_epilogue* ep, sw_stream_t stream);
/* 3x3 convolution (stride 1, dilation 1, bf16, Cin % 32 == 0 and >= 64, Cout % 64 == 0) + bias + ReLU + 2x2 / stride-2 max pool as ONE
 * launch; out_pooled [nimg][(H-2)/2+1][(W-2)/2+1][Cout].  For layers nothing differentiates through (vgg.py:104-122 below FREEZE_AT):
 * the unpooled map is never written.  Returns 1 = launched, 0 = not covered (run sw_conv3x3_igemm + sw_maxpool2x2_fwd), < 0 error;
 * bit-identical to that pair. */
int sw_conv3x3_relu_pool2(int dtype, int nimg, int H, int W, int Cin, int Cout, const void* in, const void* wk, const float* bias,
                          void* out_pooled, sw_stream_t stream);
/* n stride-1, dilation-1 3x3 convolutions of different maps (and possibly different weights) in ONE launch of the direct kernel:
 * the FPN levels of a detector (reference: the per-level loops of detectron2/modeling/proposal_generator/rpn.py:118-133 and
 * detectron2/modeling/backbone/fpn.py:131-160).  `probs` is a HOST array, n <= 8; every epilogue like sw_conv3x3_igemm's (bias,
 * relu, relu_ref; bf16 only).  Returns 1 = launched, 0 = a problem is not covered (launch them one by one), < 0 = error. */
typedef struct sw_conv_problem {
  int32_t nimg, H, W, Cin, Cout;
  const void* in; const void* wk; void* out;
  const sw_epilogue* ep;
} sw_conv_problem;
int sw_conv3x3_multi(int dtype, int n, const sw_conv_problem* probs, sw_stream_t stream);
/* The same convolution (reference: uwsod/projects/WSL/wsl/modeling/backbone/vgg.py:104-122 forward, autograd's data gradient) in
 * Winograd F(2x2, 3x3) form — 16 multiplications per 2x2 outputs instead of 36 — for bf16 layers with Cin % 32 == 0, dilation 1
 * or 2 (dilation 2 = the four parity classes of the pixel grid as dilation-1 problems).  U: the transformed filters
 * [16][Cout][Cin] bf16 written by sw_winograd_weight_prep.  Epilogue as sw_conv3x3_igemm's (bias, relu | relu_ref).
 * Returns 1 = launched, 0 = shape / epilogue not covered (call sw_conv3x3_igemm), < 0 = error. */
int sw_conv3x3_winograd(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* in, const void* U,
                        void* out, const sw_epilogue* ep, sw_stream_t stream);
/* Transformed filters U = G g G^T of n layers in ONE launch, from the f32 OIHW masters (w: Cout x Cin x 3 x 3), rounded to bf16 once.
 * mode 0: forward filters, U [16][Cout][Cin]; mode 1: the data gradient's filters g'[ci][co][ky][kx] = g[co][ci][2-ky][2-kx],
 * U [16][Cin][Cout].  `descs` is a HOST array. */
typedef struct sw_winograd_prep { const float* w; void* U; int32_t Cout, Cin, mode; } sw_winograd_prep;
int sw_winograd_weight_prep(int n, const sw_winograd_prep* descs, sw_stream_t stream);
/* dW (OIHW f32, overwritten) from x [nimg][H][W][Cin] and dy [nimg][H][W][Cout].  workspace: at least
 * sw_conv3x3_wgrad_workspace_floats(...) floats: every K-split stores its partial [co][tap][ci] tile into its own
 * slab (plain stores), a second kernel adds the slabs in fixed order and permutes to OIHW (deterministic). */
long sw_conv3x3_wgrad_workspace_floats(int dtype, int nimg, int H, int W, int Cin, int Cout, int splitk);
int sw_conv3x3_wgrad(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x,
                     const void* dy, float* dw_oihw, float* workspace, int splitk, sw_stream_t stream);
/* The two halves of sw_conv3x3_wgrad on their own: `_slabs` writes this problem's workspace_floats / (Cout*9*Cin) partial
 * slabs at `workspace`; `_fold` adds `nslab` consecutive slabs in fixed order into dW (OIHW).  Several problems with the same
 * weight (the views of one iteration, possibly running on different streams) put their slabs back to back and share ONE
 * fold: the sum over views that autograd would otherwise form with one add per parameter. */
/* sw_conv3x3_wgrad with dW[co] multiplied by cout_scale[co] (DEVICE [Cout] or NULL) inside the slab fold: the FrozenBN fold of a 3x3
 * convolution's weight gradient */
int sw_conv3x3_wgrad_scaled(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x, const void* dy,
                            float* dw_oihw, float* workspace, int splitk, const float* cout_scale, sw_stream_t stream);
/* sw_conv3x3_wgrad_scaled that ADDS to dw_oihw when accumulate != 0 (dw += cout_scale * fold): the second use of a convolution inside
 * one backward pass (the Stage-3 student runs two forward passes per iteration; unbias/ubteacher/engine/trainer.py:527-538 leaves
 * that sum to autograd). */
int sw_conv3x3_wgrad_acc(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x, const void* dy,
                         float* dw_oihw, float* workspace, int splitk, const float* cout_scale, int accumulate, sw_stream_t stream);
/* the same weight gradient (dilation 1) for maps of a few pixels (nimg * H * W <= 4096: FPN p5 / p6 of small images, below the tile
 * geometry of the MFMA loader: sw_conv3x3_wgrad returns -6 there): one thread per (co, ci), plain f32 sums in pixel order. */
int sw_conv3x3_wgrad_small(int dtype, int nimg, int H, int W, int Cin, int Cout, const void* x, const void* dy,
                           const float* cout_scale, float* dw_oihw, sw_stream_t stream);
/* sw_conv3x3_wgrad_small that adds to dw_oihw when accumulate != 0 (see sw_conv3x3_wgrad_acc) */
int sw_conv3x3_wgrad_small_acc(int dtype, int nimg, int H, int W, int Cin, int Cout, const void* x, const void* dy,
                               const float* cout_scale, float* dw_oihw, int accumulate, sw_stream_t stream);
int sw_conv3x3_wgrad_slabs(int dtype, int nimg, int H, int W, int Cin, int Cout, int dilation, const void* x,
                           const void* dy, float* workspace, int splitk, sw_stream_t stream);
int sw_conv3x3_wgrad_fold(int Cin, int Cout, int nslab, const float* workspace, float* dw_oihw, sw_stream_t stream);
/* sw_conv3x3_wgrad_fold with dW[co] multiplied by cout_scale[co] (DEVICE [Cout] or NULL) and, when accumulate != 0, added to
 * dw_oihw: the ONE fold over the slabs of several (x, dy) pairs of the same weight written by sw_conv3x3_wgrad_grouped — the RPN
 * head's convolution runs on 5 FPN levels in each of the student's two passes (rpn.py:118-133) */
int sw_conv3x3_wgrad_fold_acc(int Cin, int Cout, int nslab, const float* workspace, float* dw_oihw, const float* cout_scale,
                              int accumulate, sw_stream_t stream);
/* ALL weight gradients of a backward pass in ONE launch: problem i writes the slabs sw_conv3x3_wgrad_slabs(..., splitk =
 * nsplit) would write (sw_conv3x3_wgrad_workspace_floats(...) floats at `slabs`), computed by resident workgroups walking the
 * (problem, split, tile) list — instead of one launch of 128x128 tiles per layer and view, each cut into many K-splits to fill the
 * chip.  bf16 lists whose every problem has Cin % 64 == 0, Cout % 64 == 0, dilation 1 or 2, H >= 8 and >= 8 strip rows per split run
 * the direct weight-gradient kernel (conv_wgrad_direct.hip: an input row staged once for all nine taps; the splits are ranges of
 * (image, 32-pixel strip, row) steps — same slab count and layout, another order of additions inside a slab); every other list the
 * implicit GEMM on 256x256 tiles.  `problems` is a HOST array.  Fold with sw_conv3x3_wgrad_fold. */
typedef struct {
  int nimg, H, W, Cin, Cout, dilation, nsplit;
  const void* x;      /* [nimg][H][W][Cin] */
  const void* dy;     /* [nimg][H][W][Cout] */
  float* slabs;
} sw_wgrad_problem;
int sw_conv3x3_wgrad_grouped(int dtype, int n_problems, const sw_wgrad_problem* problems, sw_stream_t stream);
/* The same resident-grid launch for a list of PLAIN weight-gradient GEMMs, slabs[z][M][N] = sum over the z-th K range of
 * A[k][m] * B[k][n] (A [K][lda], B [K][ldb], both K-strided; M, N, lda, ldb multiples of 16 bytes): the 1x1 convolutions of a ResNet
 * bottleneck (reference: autograd's conv2d backward, detectron2/modeling/backbone/resnet.py:155-213) — several weights, each used
 * by every forward pass of an iteration.  `problems` is a HOST array.  sw_gemm_kk_grouped_slabs(dtype, K, nsplit) = slabs a problem
 * writes.  Fold with sw_splitk_fold_multi. */
typedef struct sw_gemm_kk_problem {
  const void* A; const void* B; float* slabs;
  int32_t M, N, K, nsplit;
  long lda, ldb;
} sw_gemm_kk_problem;
int sw_gemm_kk_grouped(int dtype, int n_problems, const sw_gemm_kk_problem* problems, sw_stream_t stream);
long sw_gemm_kk_grouped_slabs(int dtype, int K, int nsplit);
/* n ordered slab folds in ONE launch: C[m][n] = (accumulate ? C[m][n] : 0) + row_scale[m] * sum_z workspace[z][m][n] (row_scale
 * DEVICE [M] or NULL; N % 4 == 0, ldc % 4 == 0, 16-byte aligned).  `folds` is a HOST array. */
typedef struct sw_splitk_fold { int32_t M, N, nslab, accumulate; const float* workspace; float* C; long ldc; const float* row_scale; } sw_splitk_fold;
int sw_splitk_fold_multi(int n, const sw_splitk_fold* folds, sw_stream_t stream);
/* n folds (sw_conv3x3_wgrad_fold) in ONE launch; `folds` is a HOST array */
typedef struct { int Cin, Cout, nslab; const float* workspace; float* dw_oihw; } sw_wgrad_fold;
int sw_conv3x3_wgrad_fold_multi(int n, const sw_wgrad_fold* folds, sw_stream_t stream);
/* OIHW f32 master weights -> kernel layout.  mode 0: wk[co][tap][ci_pad] (forward, ci zero padded to cin_pad);
 * mode 1: wk[ci][8-tap][co] (data gradient: taps flipped, in/out swapped). */
int sw_conv_weight_prep(int dtype, int mode, int Cout, int Cin, int cin_pad, const float* w_oihw, void* wk,
                        sw_stream_t stream);

/* ---- 2x2 max pooling, stride 1 or 2, no padding (reference: nn.MaxPool2d, vgg.py:99-100,119-120) --------- */
int sw_maxpool2x2_fwd(int dtype, int nimg, int H, int W, int C, int stride, const void* in, void* out,
                      sw_stream_t stream);
/* din = route(dout) to the first max of each window (scan order, strict >), times (in > 0) when relu_mask. */
int sw_maxpool2x2_bwd(int dtype, int nimg, int H, int W, int C, int stride, const void* in, const void* dout,
                      void* din, int relu_mask, sw_stream_t stream);

/* ---- image normalisation (reference: MultiInputRCNN.preprocess_image, rcnn_multi.py:256-269) -------------
 * img u8 [3][H][W] -> out [H][W][cpad] = (img - mean) / std, channels >= 3 zero.
 * mean3/std3 are HOST float[3] (configuration constants, passed by value into the launch). */
int sw_preprocess(int dtype, int H, int W, int cpad, const uint8_t* img_chw, const float* mean3,
                  const float* std3, void* out_nhwc, sw_stream_t stream);
/* The same for n images of one size in ONE launch (a view batch): imgs_chw = HOST array of n device pointers,
 * out_nhwc [n][H][W][cpad]. */
int sw_preprocess_multi(int dtype, int n, int H, int W, int cpad, const uint8_t* const* imgs_chw, const float* mean3,
                        const float* std3, void* out_nhwc, sw_stream_t stream);

/* ---- max ROI pooling (reference: torchvision.ops.RoIPool called at wsl/modeling/poolers.py:183-186,267-270;
 *      arithmetic as stated in wsl/layers/csrc/ROILoopPool/ROILoopPool_cpu.cpp:26-79 fwd, :98-123 bwd) -------
 * feat [nimg][H][W][C]; rois [R][5] f32 (batch, x1,y1,x2,y2); out [R][C][PH][PW] (reference flatten order),
 * argmax [R][C][PH][PW]: argmax_bits == 32 -> int32 (h*W+w or -1, the reference's tensor); argmax_bits == 16 ->
 * uint16 (h*W+w or 0xFFFF; maps of < 65535 pixels, else -6): a third less output traffic, half the backward's index
 * traffic.  row_scale (may be NULL): out *= (row_scale[r] + row_scale_add) = the objectness prior of
 * roi_heads_oicrplus.py:200-221.  A +NaN map value never wins a bin (reference: 'val > maxval').
 * ld_out: row pitch (elements) of out AND argmax, 0 = C*PH*PW.  The fc6 GEMM reads `out` fastest when the pitch is not a
 * multiple of 1 KiB (HBM channel / L2 set spread): the hot path passes C*PH*PW + 64. */
int sw_roi_pool_fwd(int dtype, int nimg, int H, int W, int C, int PH, int PW, float spatial_scale, const void* feat,
                    const float* rois, int R, const float* row_scale, float row_scale_add, void* out,
                    void* argmax, int argmax_bits, long ld_out, sw_stream_t stream);
/* The same with a device workspace of >= sw_roi_pool_fwd_workspace_bytes(nimg, R, PH, PW) bytes (16-byte aligned; contents need not
 * survive the call).  With it, on bf16 maps of < 65535 pixels (H, W <= 255; PH, PW <= 8; C % 8 == 0) the ROI geometry — bin ranges,
 * size classes, the (image, row band, class) task lists — is computed ONCE per call by a small kernel instead of once per channel
 * slab inside the pooling kernel (large maps: half of that kernel's cycles).  Results are identical to sw_roi_pool_fwd's;
 * workspace == NULL is sw_roi_pool_fwd. */
long sw_roi_pool_fwd_workspace_bytes(int nimg, int R, int PH, int PW);
int sw_roi_pool_fwd_ws(int dtype, int nimg, int H, int W, int C, int PH, int PW, float spatial_scale, const void* feat,
                       const float* rois, int R, const float* row_scale, float row_scale_add, void* out,
                       void* argmax, int argmax_bits, long ld_out, void* workspace, long workspace_bytes,
                       sw_stream_t stream);
/* dfeat [nimg][H][W][C] (dtype, fully overwritten) = scatter-add of dout by argmax, times the same row scale,
 * times (relu_ref > 0) when relu_ref != NULL (relu_ref has feat's layout/dtype).
 * dout_absmax (device scalar >= max|dout|, e.g. from sw_absmax or a GEMM epilogue; may be NULL): selects the fixed-point LDS
 * accumulation (one 64-bit word per pixel-channel in f32 mode, a hi / lo pair of 32-bit words in bf16 mode: >= 25 bits per term;
 * bitwise reproducible, ~3x faster than LDS float atomics); NULL => f32 atomics.
 * ld_in: row pitch (elements) of dout AND argmax, 0 = C*PH*PW.
 * spatial_scale: the forward's scale, or 0.  With it, on maps whose accumulators need several pixel ranges per plane, a
 * workgroup streams only the ROIs whose rows reach its range (any value yields the same result; 0 streams all ROIs). */
int sw_roi_pool_bwd(int dtype, int nimg, int H, int W, int C, int PH, int PW, const void* dout,
                    const void* argmax, int argmax_bits, long ld_in, const float* rois, int R, const float* row_scale,
                    float row_scale_add, const void* relu_ref, const float* dout_absmax, void* dfeat, float spatial_scale,
                    sw_stream_t stream);
/* out[0] = max |x[i]| (f32 device scalar, overwritten). */
int sw_absmax(int dtype, long n, const void* x, float* out, sw_stream_t stream);

/* ---- WSDDN MIL head: scores, image-level BCE, gradient (reference: fast_rcnn_wsddn.py:556-567 forward,
 *      :340-375 loss; autograd backward) -------------------------------------------------------------------
 * logits f32 [V*R][ld]: cls logits at columns [cls_col, cls_col+K), det logits at [det_col, det_col+K);
 * per view v: scores[v][r][k] = softmax_k(cls) * softmax_r(det); loss_view[v] = BCE(clamp(sum_r scores), onehot)/K.
 * If dlogits != NULL (f32): dlogits[(v*R+r)*ld_d + col] = grad_scale[0]/V * d loss_view[v] / d logit, where
 * grad_scale is a DEVICE scalar (the cotangent of the mean-over-views loss) so that no host sync is needed.
 * mean_scores (may be NULL): [R] rows of pitch ld_mean, columns [0,K) = mean over views of scores, summed in view
 * order (roi_heads_oicrplus.py:290-294) — the mining input of refinement round 0.
 * workspace: >= sw_wsddn_workspace_floats(V, R, K) floats (chunk statistics; the softmax over proposals is cut into
 * 256-row chunks that run on separate CUs).  Limits: V <= 8, K <= 128. */
long sw_wsddn_workspace_floats(int V, int R, int K);
int sw_wsddn_mil(int V, int R, int K, const float* logits, long ld, int cls_col, int det_col, const float* gt_onehot,
                 float* scores, float* loss_view, float* dlogits, long ld_d, const float* grad_scale,
                 float* mean_scores, long ld_mean, float* workspace, sw_stream_t stream);

/* General backward of the WSDDN scores (fast_rcnn_wsddn.py:564-567: scores = softmax(C, dim=1) * softmax(D, dim=0), ONE image's R
 * proposals): logits [R][ld] = [C (K) | D (K) | ...], g_scores [R][ld_g] the cotangent of the scores -> dlogits [R][ld_d] columns
 * 0..2K-1 (others untouched): dC = A (gB - sum_k A g B), dD = B (gA - sum_r B g A).  For callers of the stand-alone predictor API
 * that build their own loss on the scores; the training path's loss gradient comes from sw_wsddn_mil.  Sums in double, fixed order:
 * deterministic.  K <= 159 (64-bit row products in LDS); workspace 8-byte aligned. */
long sw_wsddn_scores_bwd_workspace_floats(int R, int K);
int sw_wsddn_scores_bwd(int R, int K, const float* logits, long ld, const float* g_scores, long ld_g, float* dlogits, long ld_d,
                        float* workspace, sw_stream_t stream);
/* ---- mean over V score matrices (reference: roi_heads_oicrplus.py:290-294,390-395) ----------------------- */
int sw_mean_views(int V, long n, const float* in, float* out, sw_stream_t stream);

/* ---- view-mean softmax scores of the refinement heads (reference: predict_probs fast_rcnn_oicr.py:702-716 +
 *      the view average roi_heads_oicrplus.py:390-395) -------------------------------------------------------
 * out[k][r][j] = mean_v softmax_j(logits[(v*R+r)*ld + cls_col0 + k*col_stride + j]), j in [0,K], k in [0,n_rounds).
 * These are round k+1's mining scores; they depend on the logits only, so every round comes from one launch. */
int sw_oicr_mean_probs(int V, int R, int K, int n_rounds, const float* logits, long ld, int cls_col0, int col_stride,
                       float* out, sw_stream_t stream);

/* ---- pseudo-GT mining + proposal labelling, all device side (reference: get_pgt_top_k
 *      roi_heads_oicrplus.py:607-757, get_pgt_mist :560-605 incl. torchvision batched_nms, pairwise_iou
 *      structures/boxes.py:329-361, Matcher matcher.py:63-111, label_and_sample_proposals roi_heads.py:266-375)
 * n_rounds independent problems (one workgroup each, concurrently), laid out back to back:
 * scores [n_rounds][R][ncol] f32; gt_classes [G] int32 sorted unique; boxes [R][4] f32 (shared).
 * Outputs per round: lab_class (class | K background | -1 ignore), lab_weight, lab_index [n_rounds][R]; the kept
 * pseudo-GT list (score-descending): pgt_count [n_rounds], pgt_index/pgt_class/pgt_score [n_rounds][top_k*G].
 * top_k = max(int(R * MIST_P), 1) is computed by the host exactly as the reference does (:659-660).
 * workspace: >= n_rounds * sw_mine_workspace_bytes(R, top_k, G) bytes.  The sort keys live in LDS while R and top_k*G
 * are <= 16384; beyond that (COCO: PRECOMPUTED_PROPOSAL_TOPK_TRAIN 10000 with >= 17 image-level classes,
 * coco_oicr_plus.yaml:67) the workspace carries them.  Limits: R, top_k*G <= 2^22, top_k*G bytes of LDS (<= 144 KiB). */
long sw_mine_workspace_bytes(int R, int top_k, int G);
int sw_oicr_mine_label(int R, int ncol, int K, int n_rounds, const float* scores, const int32_t* gt_classes, int G,
                       const float* boxes, int top_k, float score_thresh, float nms_thresh, float iou_bg,
                       float iou_fg, int32_t* lab_class, float* lab_weight, int32_t* lab_index, int32_t* pgt_count,
                       int32_t* pgt_index, int32_t* pgt_class, float* pgt_score, void* workspace,
                       sw_stream_t stream);

/* ---- OICR refinement losses + gradient (reference: OICROutputs.__init__/softmax_cross_entropy_loss/
 *      box_reg_loss fast_rcnn_oicr.py:157-226,258-273,276-352; get_deltas box_regression.py:38-71; cross-view
 *      targets and the predictions_k2 quirk roi_heads_oicrplus.py:327-381) -----------------------------------
 * n_rounds refinement heads in one launch.  logits f32 [V*R][ld]: round k has its class logits at
 * [cls_col + k*col_stride, +K+1) and its box deltas at [box_col + k*col_stride, +4K).  boxes [V][R][4].
 * lab_class / lab_weight / lab_index [n_rounds][R] (sw_oicr_mine_label's outputs).
 * pred_view[v] = which view's predictions are paired with view v's targets ({0,1,2,2}).
 * loss_view [n_rounds][2][V] (cls then box, already divided by R).
 * dlogits (f32, may be NULL): the columns of every round are overwritten with
 * (grad_scale[2k]*dcls + grad_scale[2k+1]*dbox)/V; grad_scale is a DEVICE float[2*n_rounds]; reg_weights4 is a HOST
 * float[4] (BBOX_REG_WEIGHTS, a configuration constant passed by value into the launch); workspace:
 * n_rounds*2*V*R floats (per-row loss terms, summed in fixed order => deterministic losses). */
int sw_oicr_refine_loss(int V, int R, int K, int n_rounds, const float* logits, long ld, int cls_col, int box_col,
                        int col_stride, const float* boxes, const int32_t* lab_class, const float* lab_weight,
                        const int32_t* lab_index, const int32_t* pred_view, const float* reg_weights4,
                        float* loss_view, float* dlogits, long ld_d, const float* grad_scale, float* workspace,
                        sw_stream_t stream);

/* Stage-3 (Unbiased-Teacher) focal classification loss of the ROI heads, replaces FastRCNNFocalLoss.comput_focal_loss +
 * FocalLoss.forward (unbias/ubteacher/modeling/roi_heads/fast_rcnn.py:73-105):
 *   loss[0] = sum_r (1 - p_r)^gamma * CE_r / N,  CE_r = cross_entropy(logits[r], targets[r]),  p_r = exp(-CE_r);
 * dlogits (nullable, N x C, pitch ld_d) receives dloss/dlogits.  workspace: N floats.  Rows are summed in a fixed order. */
int sw_focal_loss(int N, int C, const float* logits, long ld, const int32_t* targets, float gamma, float* loss,
                  float* dlogits, long ld_d, float* workspace, sw_stream_t stream);
/* ---- inference (reference: OICRPlusHeads._forward_box_test roi_heads_oicrplus.py:432-475, predict_probs_K /
 *      predict_boxes_K fast_rcnn_oicr.py:674-735, Box2BoxTransform.apply_deltas box_regression.py:73-110,
 *      fast_rcnn_inference_single_image fast_rcnn_oicr.py:86-148 incl. torchvision batched_nms) ----------------
 * sw_oicr_predict: logits f32 [R][ld] with refine_k blocks (cls_score K+1 | bbox_pred 4K) starting at base_col, one
 * every round_stride columns; boxes [R][4].  all_scores [R][K+1] = mean softmax, all_boxes [R][4K] = decoded mean
 * deltas (dw, dh clamped to scale_clamp), NOT clipped.  reg_weights4: HOST float[4]. */
int sw_oicr_predict(int R, int K, int refine_k, const float* logits, long ld, int base_col, int round_stride,
                    const float* boxes, const float* reg_weights4, float scale_clamp, float* all_scores,
                    float* all_boxes, sw_stream_t stream);
/* sw_detect_postprocess: clip boxes to the image, keep score > score_thresh (background column excluded), per-class
 * greedy NMS (IoU > nms_thresh on boxes offset by class*(max_coord+1), as batched_nms forms them), first topk by
 * score.  Outputs det_count[1], det_boxes [topk][4], det_scores, det_classes, det_rows (proposal index).
 * workspace >= sw_detect_workspace_bytes(K, topk).  Limits: R <= 16384, K*topk <= 16384. */
long sw_detect_workspace_bytes(int K, int topk);
int sw_detect_postprocess(int R, int K, const float* all_scores, const float* all_boxes, int img_h, int img_w,
                          float score_thresh, float nms_thresh, int topk, int32_t* det_count, float* det_boxes,
                          float* det_scores, int32_t* det_classes, int32_t* det_rows, void* workspace,
                          sw_stream_t stream);
/* The same, told how large the workspace is: with workspace_bytes >= sw_detect_workspace_bytes2(R, K, topk) the per-class NMS of
 * every class that holds >= 256 candidates runs in its mask form — candidates sorted per class, the 64 x 64 IoU tiles of every class computed over all CUs,
 * one wave per class resolving the chunks in order — instead of one workgroup per class doing its N^2 / 2 tests alone (the RPN's
 * per-level lists of detectron2's find_top_rpn_proposals, proposal_utils.py:20-130: 2000 candidates per level; 20-80 classes of
 * fast_rcnn_inference_single_image).  Identical outputs.  sw_detect_workspace_bytes2 returns the plain size when the mask
 * matrices would exceed 96 MiB (then this call runs the single-workgroup form). */
long sw_detect_workspace_bytes2(int R, int K, int topk);
int sw_detect_postprocess2(int R, int K, const float* all_scores, const float* all_boxes, int img_h, int img_w,
                           float score_thresh, float nms_thresh, int topk, int32_t* det_count, float* det_boxes,
                           float* det_scores, int32_t* det_classes, int32_t* det_rows, void* workspace,
                           long workspace_bytes, sw_stream_t stream);

/* ---- small utilities ------------------------------------------------------------------------------------ */
/* dst[c][r] = src[r][c] (rows x cols elements of `dtype`, row pitches in elements, multiples of 16 bytes).  The weight-gradient
 * GEMMs of the box head read dZ^T through it: with A K-contiguous they run the forward-style kernel (1.2-1.3 PFLOP/s) instead of
 * transposing both operands on the fly inside LDS (1.0). */
int sw_transpose_2d(int dtype, int rows, int cols, const void* src, long ld_src, void* dst, long ld_dst, sw_stream_t stream);
/* ---- Stage-3 (Unbiased-Teacher semi-supervised step, unbias/ubteacher/engine/trainer.py:436-604) building blocks ------------
 * sw_ema_multi: teacher[i] = student[i] * (1 - keep_rate) + teacher[i] * keep_rate for n_tensors f32 tensors in as few
 * launches as 48-tensor batches allow (_update_teacher_model :588-604; keep_rate 0 = the copy at the end of burn-in).
 * teacher / student / numel are HOST arrays (of device pointers / element counts). */
int sw_ema_multi(int n_tensors, float* const* teacher, const float* const* student, const long* numel, double keep_rate,
                 sw_stream_t stream);
/* Weighted sum of n <= 32 scalar losses living anywhere on the device (unbias/ubteacher/engine/trainer.py:520-540: every loss of the
 * record times its weight — 0 for the two pseudo box-regression losses, UNSUP_LOSS_WEIGHT for the other pseudo losses, 1 otherwise —
 * then `sum(loss_dict.values())`): out[i] = values[i][0] * weights[i], out[n] = ((out[0] + out[1]) + out[2]) + ... in that order
 * (the f32 additions of Python's sum over the dict).  values / weights: host arrays.  sw_scale_scalars is its backward:
 * out[i] = g[0] * weights[i].  One launch each instead of 2 n (+ n backward) framework launches per iteration. */
int sw_weighted_sum(int n, const float* const* values, const float* weights, float* out, sw_stream_t stream);
int sw_scale_scalars(int n, const float* g, const float* weights, float* out, sw_stream_t stream);

/* sw_threshold_select: pseudo-label thresholding (threshold_bbox :361-400): keeps detection i iff scores[i] > thres (and, when
 * allowed_classes != NULL, classes[i] is one of the n_allowed image-level labels: the "multi_label" filter), compacted in input
 * order.  out_count[1]; out_boxes [n][4], out_classes [n] (optional: the "rpn" branch has none), out_scores [n], out_index [n]
 * (optional: source positions).  One workgroup; n is a few hundred teacher detections per image. */
int sw_threshold_select(int n, const float* scores, const int32_t* classes, const float* boxes, float thres,
                        const int32_t* allowed_classes, int n_allowed, int32_t* out_count, float* out_boxes,
                        int32_t* out_classes, float* out_scores, int32_t* out_index, sw_stream_t stream);
/* sw_stage_weights_multi: the compute-dtype copies of EVERY weight of a detector in one launch (replaces, per layer and per
 * forward call, the reference's implicit `conv.weight` reads + FrozenBatchNorm2d.forward's scale / bias arithmetic,
 * detectron2/layers/batch_norm.py:52-60, wrappers.py Conv2d.forward).  `descs_dev` is a DEVICE array of n entries ordered by
 * block_start (entry i owns workgroups [block_start_i, block_start_i + sw_stage_blocks(kind, rows, cols))); total_blocks = their sum.
 * kind 0: (rows, cols) f32 -> dst rows of pitch cols; 1: OIHW 3x3 -> [co][tap][ci]; 2: OIHW 3x3 -> [ci][8 - tap][co] (data
 * gradient); 3: f32 copy.  rows = Cout, cols = Cin.  bn_* (all four or none): the folded weight w * scale[co] is staged and
 * scale / shift (optional outputs, rows floats) are written: scale = bn_weight * rsqrt(bn_var + eps), shift = bn_bias - bn_mean * scale. */
typedef struct sw_stage_desc {
  const float* w;
  const float* bn_weight; const float* bn_bias; const float* bn_mean; const float* bn_var;
  float* scale; float* shift;
  void* dst;
  int32_t kind, rows, cols, block_start;
} sw_stage_desc;
int sw_stage_blocks(int kind, int rows, int cols);
int sw_stage_weights_multi(int dtype, int n, const sw_stage_desc* descs_dev, int total_blocks, float eps, sw_stream_t stream);
/* *counter += increment, in stream order (one thread): the dropout stream position of sw_epilogue.drop_offset_dev */
int sw_counter_add(uint64_t* counter, uint64_t increment, sw_stream_t stream);
/* n device-to-device byte copies in one launch (input staging into a captured step's static buffers; replaces n
 * `Tensor.copy_` calls of DatasetMapperMultiInput-shaped batches, dataset_mapper.py:272-439).  Regions must not overlap. */
typedef struct sw_copy_desc { const void* src; void* dst; long bytes; } sw_copy_desc;
int sw_copy_multi(int n, const sw_copy_desc* copies, sw_stream_t stream);
/* out[n] = sum_m X[m][ld..] (column sums; the bias gradients of the reference's conv / Linear backward).  out f32,
 * overwritten.  With `workspace` (sw_colsum_workspace_floats floats) the sum is deterministic: partial rows per row chunk,
 * then an ordered fold.  workspace NULL (or N / ld not a multiple of 16 bytes): zero fill + one f32 atomic per column and
 * row chunk. */
long sw_colsum_workspace_floats(int dtype, int M, int N);
int sw_colsum(int dtype, int M, int N, const void* X, long ld, float* out, float* workspace, sw_stream_t stream);
/* sw_colsum that adds to out when accumulate != 0 (a bias gradient that exists already inside one backward pass; see
 * sw_conv3x3_wgrad_acc) */
int sw_colsum_acc(int dtype, int M, int N, const void* X, long ld, float* out, float* workspace, int accumulate, sw_stream_t stream);
/* halves of the workspace form: `_partial` writes sw_colsum_workspace_floats(dtype, M, N) / N partial rows at `workspace`,
 * `_fold` adds n_partial_rows consecutive rows (of one or several matrices) in fixed order */
int sw_colsum_partial(int dtype, int M, int N, const void* X, long ld, float* workspace, sw_stream_t stream);
int sw_colsum_fold(int N, int n_partial_rows, const float* workspace, float* out, sw_stream_t stream);
/* n `_partial` calls in ONE launch (every bias gradient of a backward pass: conv layers x view batches); `parts` is a HOST array;
 * problem i writes sw_colsum_workspace_floats(dtype, M, N) / N partial rows at its workspace */
typedef struct { int M, N; const void* X; long ld; float* workspace; } sw_colsum_part_desc;
int sw_colsum_partial_multi(int dtype, int n, const sw_colsum_part_desc* parts, sw_stream_t stream);
/* n folds in ONE launch; `folds` is a HOST array */
typedef struct { int N, n_partial_rows; const float* workspace; float* out; } sw_colsum_fold_desc;
int sw_colsum_fold_multi(int n, const sw_colsum_fold_desc* folds, sw_stream_t stream);
/* rows x cols copy/convert f32 -> dtype with independent leading dimensions (weight staging). */
int sw_convert_2d(int dtype, int rows, int cols, const float* src, long ld_src, void* dst, long ld_dst,
                  sw_stream_t stream);
/* Reference-precision GEMMs on the bf16 MFMA (the fp32 mode's fc layers, W/roi_heads/box_head.py:82-91 and their autograd backward):
 * an f32 matrix as three bf16 pieces a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)), laid out as one
 * operand of a six-product GEMM  a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)  = ONE bf16 sw_gemm over K' = 6 K with f32
 * accumulation (the dropped products are below 2^-24 |a||b|).  side 0: blocks [a1|a1|a2|a1|a2|a3]; side 1: [b1|b2|b1|b3|b2|b1].
 * along_rows 0: the K blocks follow each other along the columns (dst: rows x 6 cols); 1: along the rows (dst: 6 rows x cols). */
int sw_split_bf16x3(int rows, int cols, const float* src, long ld_src, void* dst, long ld_dst, int side, int along_rows,
                    sw_stream_t stream);
/* dst[c][r] = src[r][c] converted to dtype (rows x cols f32 source; dst has cols rows of pitch ld_dst): the eager form of
 * sw_sgd_multi's stage_kind 3 transposed copy (first step / after a checkpoint load). */
int sw_convert_2d_t(int dtype, int rows, int cols, const float* src, long ld_src, void* dst, long ld_dst,
                    sw_stream_t stream);
/* f32 NCHW -> dtype NHWC, channels zero padded to cpad (generic backbone entry, vgg.py:216-223 takes NCHW). */
int sw_nchw_to_nhwc(int dtype, int N, int C, int H, int W, int cpad, const float* in_nchw, void* out_nhwc,
                    sw_stream_t stream);
/* ReLU backward in place: grad = ref > 0 ? grad : 0  (F.relu_ backward, vgg.py:105-116). */
int sw_relu_bwd(int dtype, long n, const void* ref, void* grad, sw_stream_t stream);
/* The same into a separate buffer: out = ref > 0 ? grad : 0 (the backbone's entry gradient belongs to autograd and is not
 * modified; out == grad is allowed). */
int sw_relu_bwd_out(int dtype, long n, const void* ref, const void* grad, void* out, sw_stream_t stream);
/* out[m][n] = in[m][n] * colscale[n] (f32 -> dtype): applies each loss term's cotangent to its logit columns. */
int sw_scale_cols(int dtype, int M, int N, const float* in, long ld_in, const float* colscale, void* out,
                  long ld_out, sw_stream_t stream);
/* dst(f32) = src(dtype) */
int sw_to_f32(int dtype, long n, const void* src, float* dst, sw_stream_t stream);
int sw_fill_zero(void* p, long bytes, sw_stream_t stream);
/* keep[i] = hash(seed, offset+i) >= p   (Bernoulli keep mask for F.dropout, box_head.py:90) */
int sw_dropout_mask(uint8_t* keep, long n, uint64_t seed, uint64_t offset, float p, sw_stream_t stream);
/* SGD with momentum + weight decay, torch.optim.SGD semantics (reference: solver/build.py:191-215,
 * train_net_multi.py:157-164):  g += wd*p;  buf = first ? g : mom*buf + g;  p -= lr*buf. */
int sw_sgd_momentum_step(float* param, const float* grad, float* momentum_buf, long n, float lr, float momentum,
                         float weight_decay, int first_step, float grad_scale, sw_stream_t stream);
/* The whole optimizer step in one launch per <= SW_SGD_MAX_TENSORS parameter tensors, with the compute-dtype kernel-layout
 * copies the next forward needs ("weight staging") written from the freshly updated values in the same pass — the f32
 * master is read once instead of once by the optimizer and once more by every staging kernel.
 * tensors: HOST array (copied into the launch).  stage_kind 0: none.  1: the parameter is a (n/d0) x d0 matrix, stage0
 * receives it in stage_dtype with row pitch ld0 (fc / predictor weights).  2: the parameter is an OIHW 3x3 conv weight
 * (Cout=d0, Cin=d1): stage0 (may be NULL) = forward layout [co][tap][d2=cin_pad] and stage1 (may be NULL) = data-gradient
 * layout [ci][8-tap][co], exactly what sw_conv_weight_prep modes 0 / 1 produce.  3: as 1, plus stage1 = the TRANSPOSED
 * matrix (d0 rows of pitch ld1) — fc6's weight is read K-contiguous by the forward AND by the data-gradient GEMM this
 * way (a K-strided operand of 49 KiB pitch costs that GEMM 25 %); n/d0 and d0 must be multiples of 64. */
#define SW_SGD_MAX_TENSORS 24
typedef struct sw_sgd_tensor_s {
  float* param;
  const float* grad;
  float* momentum_buf;
  long n;
  float lr, weight_decay;
  int first_step;
  int stage_kind, stage_dtype;
  void* stage0;
  void* stage1;
  int d0, d1, d2;
  long ld0, ld1;
  const float* hyper_dev;   /* optional DEVICE pointer to {lr, weight_decay}: read by the kernel instead of the two fields above, so
                             * that a captured hipGraph of the step survives learning-rate schedule changes */
} sw_sgd_tensor;
int sw_sgd_multi(int n_tensors, const sw_sgd_tensor* tensors, float momentum, float grad_scale, sw_stream_t stream);
/* 1 if sw_gemm(dtype, a_kstrided, b_kstrided, M, N, K, ...) accepts sw_epilogue.sgd_fused for this shape, else 0 */
int sw_gemm_sgd_fused_supported(int dtype, int a_kstrided, int b_kstrided, int M, int N, int K);
/* out[i] = mean over the n_images images of (sum_v loss_view[b][i][v] / V)   (loss assembly, roi_heads_oicrplus.py:283-288,
 * 384-388; the reference runs one image per GPU, n_images > 1 is the mean DDP would form over as many ranks);
 * total (optional, 2 floats): [0] = sum_i out[i] in index order (train_net_multi.py:129 sum(loss_dict.values())),
 * [1] = 1.0 if that sum is finite else 0.0 (train_loop.py:253-259 _detect_anomaly without a host round trip). */
int sw_loss_finalize(int n_losses, int V, int n_images, const float* loss_view, float* out, float* total,
                     sw_stream_t stream);
/* sw_scale_cols with the column scale formed from the cotangents: ((g_losses ? g_losses[col_to_loss[n]] : 0) +
 * (g_total ? g_total[0] : 0)) * mul for n < n_valid, 0 for the padding columns (never read from `in`). */
int sw_scale_cols_loss(int dtype, int M, int N, int n_valid, const float* in, long ld_in, const float* g_losses,
                       const float* g_total, const int32_t* col_to_loss, float mul, void* out, long ld_out,
                       sw_stream_t stream);
/* One pass of Pillow's 8-bit separable resize over a planar (C, H, W) u8 image — the arithmetic of the reference's
 * ResizeTransform.apply_image (PIL Image.resize(..., BILINEAR), used by DatasetMapperMultiInput dataset_mapper.py:303-352 and
 * DatasetMapperTTAAVG test_time_augmentation_avg.py:199-310): out = clip8((2^21 + sum_t in[first + t] * kk[o][t]) >> 22) along
 * x (horizontal != 0: out (C, H, out_size)) or y (out (C, out_size, W)).  bounds int32 [out_size][2] = (first input index, tap
 * count), kk int32 [out_size][ksize] = 22-bit fixed-point weights, both DEVICE arrays computed by the caller as Pillow's
 * precompute_coeffs does.  in_ld / in_plane: row and plane strides of `in` in bytes (a crop window of a larger image — the mapper's
 * CropTransform, dataset_mapper.py:278-285 — is `in` offset to its first pixel with the parent's strides).  out_flip (optional): the
 * same rows mirrored in x.  Bit exact with Pillow (tests/golden/resize_*.npz). */
int sw_resize_pass_u8(int C, int H, int W, long in_ld, long in_plane, int out_size, int horizontal, const uint8_t* in,
                      const int32_t* bounds, const int32_t* kk, int ksize, uint8_t* out, uint8_t* out_flip, sw_stream_t stream);
/* RandomBrightness + RandomSaturation of the 4-view training mapper (uwsod/detectron2/data/transforms/augmentation_impl.py:403-455
 * through fvcore's BlendTransform on a uint8 image): mode bit 0 = brightness `u8(clip(f32(w_bright) * px))`, bit 1 = saturation
 * `u8(clip(src_w_sat * grey + f64(f32(w_sat) * px)))` with grey = px0 * 0.299 + px1 * 0.587 + px2 * 0.114 in double on the
 * brightened pixels, src_w_sat = 1 - w (host double).  in / out planar u8 [3][H][W]; out_flip (optional): rows mirrored in x. */
int sw_color_jitter_u8(int H, int W, int mode, const uint8_t* in, float w_bright, double src_w_sat, float w_sat, uint8_t* out,
                       uint8_t* out_flip, sw_stream_t stream);
/* The four views' (R,4) proposal boxes and (R,) objectness logits of one image (box_ptrs4 / obj_ptrs4: HOST arrays of 4
 * device pointers) -> boxes [4][R][4], obj [4][R], rois [2][2R][5] = (batch index 0 | 1, box) per scale
 * (poolers.py:81-108 convert_boxes_to_pooler_format for the view / flipped-view pair of a scale). */
int sw_pack_views(int R, const float* const* box_ptrs4, const float* const* obj_ptrs4, float* boxes, float* obj,
                  float* rois, sw_stream_t stream);

/* ==== Stage-3 detector: the ResNet-50-FPN Faster R-CNN of the Unbiased-Teacher step =====================
 * Reference: unbias/ubteacher/modeling/ over the second tree's detectron2/detectron2/modeling/ (v0.4).  Dense layers
 * reuse sw_gemm (1x1 convolutions on NHWC pixels, fc) and sw_conv3x3_igemm / _wgrad; the entry points below are what that
 * model needs on top (csrc/detector.hip).  All activations NHWC. */
/* img u8 [3][h][w] -> out [H][W][4] = (img - mean) / std inside the image, 0 in the padding (ImageList.from_tensors pads at the
 * bottom / right to the size divisibility) and in channel 3 (meta_arch/rcnn.py:220-228, structures/image_list.py:60-124). */
int sw_preprocess_pad(int dtype, int h, int w, int H, int W, const uint8_t* img_chw, const float* mean3, const float* std3,
                      void* out_nhwc4, sw_stream_t stream);
/* BasicStem.conv1 (backbone/resnet.py:334-359): 7x7, stride 2, padding 3, 3 -> 64 channels, then y * scale[c] + bias[c] (the
 * FrozenBatchNorm2d fold, layers/batch_norm.py:52-58) and ReLU.  in [N][H][W][4], w f32 OIHW [64][3][7][7], out [N][OH][OW][64]
 * with OH = (H - 1) / 2 + 1.  Forward only: the stem is frozen (FREEZE_AT 2). */
int sw_stem_conv7x7(int dtype, int N, int H, int W, const void* in_nhwc4, const float* w_oihw, const float* scale,
                    const float* bias, void* out_nhwc64, sw_stream_t stream);
/* F.max_pool2d(kernel 3, stride 2, padding 1) (resnet.py:358): out [N][(H - 1) / 2 + 1][(W - 1) / 2 + 1][C] */
int sw_maxpool3x3s2(int dtype, int N, int H, int W, int C, const void* in, void* out, sw_stream_t stream);
/* out[n][y][x][:] = in[n][2y][2x][:], out [N][(H + 1) / 2][(W + 1) / 2][C]: the pixels a 1x1 stride-2 convolution reads (bottleneck
 * conv1 / shortcut with STRIDE_IN_1X1, resnet.py:146-166) and FPN's p6 = max_pool2d(p5, kernel 1, stride 2) (fpn.py:188-189);
 * sw_scatter2x is its backward: out [N][H][W][C] = g at the even pixels, 0 elsewhere (fully written). */
int sw_subsample2x(int dtype, int N, int H, int W, int C, const void* in, void* out, sw_stream_t stream);
int sw_scatter2x(int dtype, int N, int H, int W, int C, const void* g, void* out, sw_stream_t stream);
/* out = a + b, with ReLU when relu != 0 (the residual join of a bottleneck block, resnet.py:209-211) */
int sw_add_relu(int dtype, long n, const void* a, const void* b, void* out, int relu, sw_stream_t stream);
/* FPN top-down pathway (fpn.py:142-144): out [N][2h][2w][C] = lateral + nearest-neighbour 2x upsampling of top [N][h][w][C];
 * sw_downsample2x_sum is the upsampling's backward: out [N][h][w][C] = sum over each 2x2 block of g [N][2h][2w][C]. */
int sw_upsample2x_add(int dtype, int N, int h, int w, int C, const void* lateral, const void* top, void* out, sw_stream_t stream);
int sw_downsample2x_sum(int dtype, int N, int h, int w, int C, const void* g, void* out, sw_stream_t stream);
/* ROIAlign, aligned = True (poolers.py:204-213 "ROIAlignV2" -> layers/roi_align.py:7-74 -> torchvision.ops.roi_align; the
 * arithmetic as stated in-tree at uwsod/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:20-400).  feat [N][H][W][C] of ONE
 * FPN level; rois [R][5] (batch, x1, y1, x2, y2); sel [n_sel] int32 = the rows of `rois` / `out` assigned to this level
 * (poolers.py:232-250 loops over the levels); out [R][C][PH][PW] with row pitch ld_out (rows not in `sel` are not touched).
 * n_sel_dev (may be NULL): DEVICE int holding the real length of `sel` (sw_roi_assign_levels' sel_cnt); n_sel is then its host bound.
 * Backward: dfeat_f32 [N][H][W][C] float32, zero-filled by the caller, accumulated with f32 atomics. */
int sw_roi_align_fwd(int dtype, int H, int W, int C, int PH, int PW, float spatial_scale, int sampling_ratio, const void* feat,
                     const float* rois, const int32_t* sel, int n_sel, const int32_t* n_sel_dev, void* out, long ld_out,
                     sw_stream_t stream);
int sw_roi_align_bwd(int dtype, int H, int W, int C, int PH, int PW, float spatial_scale, int sampling_ratio, const void* gout,
                     long ld, const float* rois, const int32_t* sel, int n_sel, const int32_t* n_sel_dev, float* dfeat_f32,
                     sw_stream_t stream);
/* The deterministic backward of ROIAlign (the reference's CPU path, ROIAlign_cpu.cpp:286-400, is a sequential loop: reproducible; f32
 * atomics are not): the same contributions as sw_roi_align_bwd accumulated as 64-bit fixed-point integers, llrint(v * 2^40 / max|gout|)
 * — integer addition commutes, two runs give the same bits.  gout_absmax: DEVICE float holding max|gout| over the whole gradient
 * (sw_absmax); acc_i64 [N][H][W][C] int64, zero-filled by the caller.  sw_fx_to_float: out[i] = acc[i] * absmax / 2^40 (f32 / bf16;
 * every element NaN when absmax is NaN / Inf). */
int sw_roi_align_bwd_fx(int dtype, int H, int W, int C, int PH, int PW, float spatial_scale, int sampling_ratio, const void* gout,
                        long ld, const float* rois, const int32_t* sel, int n_sel, const int32_t* n_sel_dev, const float* gout_absmax,
                        long long* acc_i64, sw_stream_t stream);
int sw_fx_to_float(int out_dtype, long n, const long long* acc_i64, const float* absmax, void* out, sw_stream_t stream);
/* Box2BoxTransform.apply_deltas (box_regression.py:76-116): out[i] = decode(deltas[i], boxes[i % n_boxes]); deltas row pitch
 * ld_deltas floats; weights4: HOST float[4]; dw, dh clamped to scale_clamp. */
int sw_decode_boxes(long n, long n_boxes, const float* deltas, long ld_deltas, const float* boxes, const float* weights4,
                    float scale_clamp, float* out, sw_stream_t stream);
/* ---- Stage-3 detector, index side (csrc/proposals.hip): the reference's torch sort / nonzero / randperm logic as device code.
 * RPN proposal selection (detectron2/detectron2/modeling/proposal_generator/proposal_utils.py:22-130 find_top_rpn_proposals, up to its
 * batched_nms): for every image and level the pre_topk highest objectness logits (= sort(descending, stable)[:k]: ties -> ascending
 * anchor index), their boxes decoded (box_regression.py:88-116, weights4, scale_clamp), and the candidate rows written in the form
 * sw_detect_postprocess2 takes with "class" = level: image `i` owns rows [i * L * pre_topk, (i + 1) * L * pre_topk), level l the
 * pre_topk rows from l * pre_topk, in descending score order; cand_scores [rows][L + 1] = -inf except column l (-inf there too for
 * unused rows and for boxes that are empty after clipping to img_hw_dev[i] = (h, w): proposal_utils.py:96-106), cand_boxes
 * [rows][4 L] the box repeated.  finite_dev[i] = 0 if a selected box / logit is not finite (:86-94), else nonzero.
 * logits / deltas / anchors: HOST arrays of L device pointers, level l: logits [N][n_l], deltas [N][n_l][4], anchors [n_l][4];
 * img_stride != 0: image i's row of level l starts img_stride anchors after image i-1's (the levels are column ranges of ONE
 * [N][sum n_l] array, sw_rpn_unpack's output), 0: every level is dense on its own.
 * sel_idx: int32 [N * L][pre_topk] scratch.  pre_topk <= 16384, N * L <= 40. */
long sw_rpn_select_workspace_bytes(int N, int L, const int* n_per_level);
int sw_rpn_select_pack(int N, int L, const float* const* logits, const float* const* deltas, const float* const* anchors,
                       const int* n_per_level, long img_stride, int pre_topk, const float* weights4, float scale_clamp,
                       const int* img_hw_dev, float* cand_scores, float* cand_boxes, int* finite_dev, int* sel_idx, void* workspace,
                       long workspace_bytes, sw_stream_t stream);
/* RPN anchor labels (detectron2/.../proposal_generator/rpn.py:305-360 label_and_sample_anchors): IoU of every anchor [A][4] with the
 * image's ground-truth boxes (gt_boxes: the images' boxes back to back, gt_count_per_image HOST array), Matcher thresholds
 * [thr_lo, thr_hi] with labels [0, -1, 1] and low-quality matches (matcher.py:60-126), then subsample_labels (sampling.py:8-54):
 * up to max_pos = int(batch_size * positive_fraction) positives and batch_size - that many negatives per image, drawn as the
 * candidates with the smallest random keys  key(position in the ascending-index candidate list) = splitmix64(seed + position) >> 40
 * (= argsort(keys, stable)[:num], the fixtures' closed-form stand-in for torch.randperm); seeds: HOST u64 [2 N] = (positives,
 * negatives) per image.  labels int8 [N][A] in {-1, 0, 1}; matched f32 [N][A][4] = the box of the arg-max-IoU gt (0 without gt). */
long sw_rpn_label_workspace_bytes(int N, long A, int total_gt);
int sw_rpn_label_anchors(int N, long A, const float* anchors, const float* gt_boxes, const int* gt_count_per_image, float thr_lo,
                         float thr_hi, int batch_size, int max_pos, const uint64_t* seeds, int8_t* labels, float* matched,
                         void* workspace, long workspace_bytes, sw_stream_t stream);
/* ROI-head label + sample (unbias/ubteacher/modeling/roi_heads/roi_heads.py:324-375 label_and_sample_proposals): image i's
 * proposals are the first p_cnt_dev[i] rows (a DEVICE count: it comes from the RPN's NMS) of proposals [n_img][p_stride][4] (+ its
 * ground-truth boxes appended when append_gt); IoU >= iou_thresh -> foreground with the matched box's class, else background
 * `num_classes`; up to max_pos foreground and batch_size - that many background rows by the random-key rule of sw_rpn_label_anchors
 * (seeds: HOST u64 [2 n_img]), foreground list first, each list in key order.  gt_boxes / gt_classes: the images' ground truth back to
 * back, g_off / g_cnt HOST int arrays.  p_stride + g_cnt[i] <= 4096, n_img <= 8.  Outputs, image i at rows [i * out_stride, ...):
 * out_count [n_img], out_index (row in the image's candidate list), out_classes, out_boxes [..][4], out_gt_boxes [..][4]. */
int sw_roi_label_sample(int n_img, const int* p_cnt_dev, int p_stride, const float* proposals, const int* g_off, const int* g_cnt,
                        const float* gt_boxes, const int32_t* gt_classes, int append_gt, float iou_thresh, int num_classes,
                        int batch_size, int max_pos, const uint64_t* seeds, int out_stride, int32_t* out_count, int32_t* out_index,
                        int32_t* out_classes, float* out_boxes, float* out_gt_boxes, sw_stream_t stream);
/* The RPN head's output in anchor order.  y [rows][ld] f32 is ONE GEMM over the pixels of all levels (row = level offset + image *
 * hw_per_level[l] + pixel; columns [objectness a | delta 4 a + b], A anchors per location); logits [N][At], deltas [N][At][4] with
 * At = A * sum hw: level-major, location-major, anchor-minor — the order of rpn.py:230-260 / anchor_generator.py.  _bwd: dy from the
 * two gradients (either may be NULL), each times its DEVICE scalar g_*_dev (NULL = 1: the losses' cotangents), every column written
 * (padding 0). */
int sw_rpn_unpack(int N, int L, int A, const int* hw_per_level, const float* y, long ld, float* logits, float* deltas, sw_stream_t stream);
int sw_rpn_unpack_bwd(int N, int L, int A, const int* hw_per_level, const float* dlogits, const float* ddeltas,
                      const float* g_logits_dev, const float* g_deltas_dev, float* dy, long ld, sw_stream_t stream);
/* FPN level of every ROI (poolers.py:17-50,196-250): image i contributes row_cnt[i] boxes starting at boxes + box_off_floats[i]
 * (HOST arrays); rois [R][5] = (image, box) dense, level_of [R] in 0..3 (= level 2..5), sel [4][R] the rows of each level in ascending
 * order, sel_cnt [4] their counts (DEVICE: sw_roi_align_* read them).  R <= 8192. */
int sw_roi_assign_levels(int n_img, const int* row_cnt, const long* box_off_floats, const float* boxes, float* rois, int32_t* level_of,
                         int32_t* sel, int32_t* sel_cnt, sw_stream_t stream);
/* out[m][n] = in[m][n] * (n < split ? g0_dev[0] : g1_dev[0]) for n < N, 0 for the padding columns N <= n < ld (in and out share the row
 * pitch ld): the cotangents of the ROI heads' two losses (fast_rcnn.py:73-105) applied to the unit gradient of the packed logits. */
int sw_scale_col_blocks(long M, int N, int split, const float* in, long ld, const float* g0_dev, const float* g1_dev, float* out,
                        sw_stream_t stream);
/* RPN losses (proposal_generator/rpn.py:362-420, box_regression.py:229-260) over n = N * A anchors: losses2[0] = sum over
 * label >= 0 of BCE-with-logits(logit, label) * inv_norm, losses2[1] = sum over label == 1 of |delta - get_deltas(anchor, gt)|_1
 * * inv_norm (smooth-L1 with beta 0), and their unit gradients dlogits [n], ddeltas [n][4] (either may be NULL).  anchors
 * [n_anchors][4] repeat over the images (row i uses anchor i % n_anchors), matched_gt_boxes [n][4], labels int8 in {-1, 0, 1}.
 * Ordered two-stage reduction: deterministic.  workspace: sw_rpn_loss_workspace_floats() floats. */
long sw_rpn_loss_workspace_floats(void);
int sw_rpn_loss(long n, long n_anchors, const float* logits, const float* deltas, const int8_t* labels, const float* anchors,
                const float* matched_gt_boxes, const float* weights4, float inv_norm, float* losses2, float* dlogits,
                float* ddeltas, float* workspace, sw_stream_t stream);

const char* sw_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SOSWSOD_HIP_H */
