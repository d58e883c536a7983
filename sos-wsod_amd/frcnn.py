"""Stage-3 detector on gfx950 (BASELINE config #5): the ResNet-50-FPN Faster R-CNN that the
Unbiased-Teacher step trains — `TwoStagePseudoLabGeneralizedRCNN` (unbias/ubteacher/modeling/meta_arch/rcnn.py:8-107) over
`build_resnet_fpn_backbone` (detectron2/detectron2/modeling/backbone/{resnet,fpn}.py, second tree, v0.4), `PseudoLabRPN`
(unbias/ubteacher/modeling/proposal_generator/rpn.py over detectron2/.../proposal_generator/{rpn,proposal_utils}.py,
anchor_generator.py) and `StandardROIHeadsPseudoLab` with the focal classification loss (unbias/.../roi_heads/{roi_heads,
fast_rcnn}.py over detectron2/.../{poolers,roi_heads/box_head,roi_heads/fast_rcnn}.py).  Same module tree / state-dict names
as the reference model (backbone.bottom_up.res3.0.conv1.{weight,norm.*}, backbone.fpn_lateral2, proposal_generator.rpn_head.*,
roi_heads.box_head.fc1, roi_heads.box_predictor.{cls_score,bbox_pred}), same branch interface
(`forward(batched_inputs, branch=...)`), so semisup.SemiSupStep drives it like the reference's trainer drives its model.

Execution plan (MI355X-first): NHWC activations end to end; a 1x1 convolution is sw_gemm over the pixels (stride 2 = a pixel
subsample in front, STRIDE_IN_1X1), a 3x3 convolution the implicit-GEMM / direct MFMA kernel with bias + ReLU fused, FrozenBN is
folded into the weight and bias; the frozen stem + res2 run without autograd; ROIAlign, the FPN joins, the RPN losses, the focal
/ L1 losses, box decoding and NMS are kernels of csrc/detector.hip / heads.hip.  Every dense layer is a torch.autograd.Function
around those kernels with an explicit backward; autograd only links the nodes.  The index side — per-level top-k of the RPN,
anchor <-> ground-truth matching, the label sampling of RPN and ROI heads, FPN level assignment — is device code as well
(csrc/proposals.hip); one host read per RPN call (the proposal counts after NMS) remains, where the reference has a dozen.

Random sampling (detectron2/modeling/sampling.py:49-50 draws torch.randperm per candidate list): `sampler.next_seed()` supplies one
64-bit seed per list, the kernels give the candidate at position i of the list the key splitmix64(seed + i) >> 40 and the `num`
smallest keys win; tests inject the seeds of the closed-form keys the fixtures were generated with."""
import math
import os
from collections import OrderedDict
from typing import Dict, List

import torch
import torch.nn as nn

from . import ops
from .registry import META_ARCH_REGISTRY
from .structures import Boxes, Instances

R50_STAGES = [("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2)]
FPN_STAGES = (2, 3, 4, 5)
STRIDES = (4, 8, 16, 32, 64)
SCALE_CLAMP = math.log(1000.0 / 16)
GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))        # proposal_utils.py:170


# layer flags (the `relu` argument of _LinearFn / _Conv3x3Fn; True / False keep their meaning).  A ReLU mask in the backward pass is a
# kernel of its own (relu_bwd: read the output and the gradient, write the masked gradient) unless the PRODUCER of that gradient — the
# data-gradient kernel of the layer's only consumer — applies it in its epilogue:
#   _MASK_INPUT_GRAD  on the consumer: its input is a ReLU output nobody else reads; d/d(input) leaves the layer masked by input > 0;
#   _GRAD_PREMASKED   on the producer of that input: the gradient it receives is already masked, its own relu_bwd is skipped.
# The two are set in pairs by the module that owns both layers (bottleneck chain of a stage, RPN head, box head -> predictor).
_RELU, _MASK_INPUT_GRAD, _GRAD_PREMASKED = 1, 2, 4
MASKS_IN_PRODUCERS = os.environ.get("SW_S3_MASKS_IN_PRODUCERS", "1") != "0"    # off: every layer masks its own incoming gradient (relu_bwd), the flag pairs are ignored
GROUP_LINEAR_WGRADS = True   # _LinearFn layers used several times per backward queue their weight gradients for one grouped launch (ops.grad_scope)
FUSED_BLOCKS = True      # BottleneckBlock as one autograd node (_BottleneckFn); False: layer by layer (the form the fused one is tested against)


def _epc(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def _pad8(n):
    return (n + 7) // 8 * 8


_ALIGN_BWD_FX = os.environ.get("SW_ROI_ALIGN_BWD_FX", "1") == "1"     # development switch: "0" = the f32-atomic ROIAlign backward

class _Staged:
    """what one layer reads from the model's stage plan: w (+ wd: 3x3 data-gradient layout), scale / shift (FrozenBN) or packed bias"""
    __slots__ = ("w", "wd", "scale", "shift", "bias")

    def __init__(self, w=None, wd=None, scale=None, shift=None, bias=None):
        self.w, self.wd, self.scale, self.shift, self.bias = w, wd, scale, shift, bias


def _staged_of(module):
    st = module.__dict__.get("_st")
    if st is None:
        raise RuntimeError(f"{type(module).__name__}: no staged weights — the detector's layers run under TwoStagePseudoLabGeneralizedRCNN, "
                           "whose forward refreshes the stage plan (sw_stage_weights_multi)")
    return st


class WeightStage:
    """Compute-dtype copies of every weight of one detector, rewritten by ONE launch (ops.StagePlan / sw_stage_weights_multi) when a
    parameter or a FrozenBN buffer changed: optimizer step (version counters / ops.PARAM_EPOCH), teacher EMA (ops.PARAM_EPOCH and
    ops.BUFFER_EPOCH), load_state_dict.  Replaces, per layer and per forward call, the fold `weight * scale`, the dtype conversion
    and the layout change (and their autograd nodes: the layers' backward multiplies by `scale` itself)."""

    def __init__(self, model, compute_dtype):
        self.layers = [m for m in model.modules() if hasattr(m, "_stage_entries")]
        entries = []
        for m in self.layers:
            entries += m._stage_entries(compute_dtype)
        self.plan = ops.StagePlan(entries, compute_dtype)
        self.sources = []
        for e in entries:
            self.sources.append(e["w"])
            if e.get("bn") is not None:
                self.sources += list(e["bn"])
        self.key = None

    def valid_for(self, compute_dtype):
        return self.plan.dtype == compute_dtype and all(e[0].data_ptr() == p for e, p in zip(self.plan._keep, self.plan.ptrs))

    def refresh(self):
        key = (ops.PARAM_EPOCH, ops.BUFFER_EPOCH, tuple(t._version for t in self.sources))
        if key != self.key:
            self.plan.run()
            self.key = key


# ====================================================================================================== autograd nodes
def _pad_scale(scale, ld):
    """the FrozenBN scale (out,) as a row-scale vector of the padded (ld, in) weight gradient (pad rows: 1); out % 8 == 0 for every
    ConvBN of the ResNet, so this is the tensor itself"""
    if scale.shape[0] == ld:
        return scale
    return torch.cat([scale, torch.ones(ld - scale.shape[0], device=scale.device, dtype=scale.dtype)])


class _LinearFn(ops.CountedFunction):
    """y (P, out) = x (P, in) @ W_eff^T (+ bias) (ReLU): sw_gemm with the epilogue fused; explicit backward (dgrad, wgrad GEMMs, column
    sums).  `staged` (ld, in) is the compute-dtype copy of the effective weight (rows beyond `out` zero), written by the model's
    stage plan; `bias` f32 or None; `scale` (out,) f32 or None = the FrozenBN fold (W_eff = W * scale: dW = scale * dW_eff).
    `splits`: the row counts of the parameters packed into `staged`; `params`: those weight parameters in packing order, then their
    bias parameters if they have any — autograd routes the gradient pieces to them.  y compute dtype or f32 (out_f32).
    `residual` (P, out) or None: added before the ReLU inside the GEMM epilogue (the bottleneck's shortcut); its gradient is the
    masked output gradient."""

    @staticmethod
    def forward(ctx, x, staged, bias, scale, relu, out_f32, splits, residual, *params):
        P, D = x.shape
        out_f = sum(splits)
        cd = x.dtype
        ld = staged.shape[0]
        flags = int(relu) if MASKS_IN_PRODUCERS else (int(relu) & _RELU)
        relu = bool(flags & _RELU)
        ydt = torch.float32 if out_f32 else cd
        assert not (relu and out_f32) and ld == (out_f + 7) // 8 * 8 and staged.dtype == cd
        ybuf = torch.empty(P, ld, device=x.device, dtype=ydt)
        if ld != out_f and P > 0:
            ops.fill_zero(ybuf)                                      # columns beyond out_f stay 0
        y = ybuf[:, :out_f]
        if P > 0:
            ops.gemm(x, staged, y, P, out_f, D, ep=ops.make_epilogue(bias=bias, relu=relu, out_dtype=ydt,
                                                                residual=None if residual is None else residual.detach()),
                     splitk=_few_tile_splits(P, out_f, D) if ld == out_f else 1, tag="s3_gemm_fwd")
            if ops.TIMER is not None:            # bench.py's roofline_stage3: algorithmic FLOP and bytes (operands once, result once) of this launch
                es = 2 if cd == torch.bfloat16 else 4
                ops.TIMER.note("s3_gemm_fwd", 2.0 * P * out_f * D)
                ops.TIMER.note("s3_gemm_fwd_bytes", float(es) * (P * D + out_f * D) + (4.0 if out_f32 else es) * P * out_f)
        ctx.save_for_backward(x, staged, ybuf if relu else None, scale)
        assert residual is None or (residual.shape == (P, out_f) and residual.dtype == ydt and residual.is_contiguous())
        ctx.relu, ctx.out_f, ctx.splits = relu, out_f, tuple(splits)
        ctx.mask_in, ctx.premasked = bool(flags & _MASK_INPUT_GRAD), bool(flags & _GRAD_PREMASKED)
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.wkey = id(params[0]) if params else None
        if GROUP_LINEAR_WGRADS and params and any(ctx.needs_input_grad[8:8 + len(splits)]):
            ops.count_use(("lin", ctx.wkey))
        ctx.bkey = id(params[len(splits)]) if len(params) > len(splits) else None
        return y

    @staticmethod
    def backward(ctx, g):
        x, ws, ybuf, scale = ctx.saved_tensors
        P, D = x.shape
        out_f, ld, cd, nw = ctx.out_f, ws.shape[0], x.dtype, len(ctx.splits)
        need = ctx.needs_input_grad
        need_w = any(need[8:8 + nw]); need_b = any(need[8 + nw:])
        if ld == out_f:
            if ctx.relu and not ctx.premasked:
                gs = ops.relu_bwd(ybuf, g.contiguous(), out=torch.empty(P, ld, device=g.device, dtype=cd)) if P > 0 else g.contiguous()
            else:
                gs = g.contiguous() if g.dtype == cd else g.to(cd)
        else:
            gs = torch.empty(P, ld, device=g.device, dtype=cd)
            if P > 0:
                ops.fill_zero(gs)
                gs[:, :out_f] = g
                if ctx.relu:
                    ops.relu_bwd(ybuf, gs)                              # in place on the padded buffers (pad columns: 0 stays 0)
        dx = None
        if need[0]:
            dx = torch.empty(P, D, device=g.device, dtype=cd)
            if P > 0:
                ops.gemm(gs, ws, dx, P, D, ld, b_kstrided=True, ep=ops.make_epilogue(out_dtype=cd, relu_ref=x if ctx.mask_in else None),
                         splitk=_few_tile_splits(P, D, ld))
            else:
                dx.zero_()
        dws, dbs = [None] * nw, [None] * (len(ctx.shapes) - nw)
        if need_w:
            # K = pixels (10^4 .. 10^5 for a 1x1 convolution), M x N = a handful of 128x128 tiles: K-split slabs + ordered fold inside
            # sw_gemm (deterministic; unsplit, 4 workgroups walked 60 000 pixels).  FrozenBN: dW = scale * dW_eff, applied to the rows
            # inside the fold; `scale` covers the out_f real rows, the pad rows of dwp are never handed out.  None: added to the
            # buffer an earlier node of this backward pass returned (ops.grad_scope)
            scope = ops.GRAD_SCOPE
            lkey = ("lin", ctx.wkey)
            if scope is not None and ctx.wkey is not None and scope.uses.get(lkey, 0) > 1:
                # every use of this weight queues its (gs, x) pair; the last one runs them as one grouped launch + one fold
                q = scope.queued.get(lkey)
                first = q is None
                if first:
                    q = scope.queued[lkey] = dict(bufs=[torch.empty(ld, D, device=g.device, dtype=torch.float32)], probs=[],
                                                  scales=[None if scale is None else _pad_scale(scale, ld)], left=scope.uses[lkey])
                if P > 0:
                    q["probs"].append([(gs, x)])
                q["left"] -= 1
                if q["left"] == 0:
                    if q["probs"]:
                        _flush_wgrad_1x1(q)
                    else:
                        q["bufs"][0].zero_()
                dwp = q["bufs"][0] if first else None
            else:
                dwp = (_wgrad_1x1(gs, x, None if scale is None else _pad_scale(scale, ld), ctx.wkey) if P > 0
                       else torch.zeros(ld, D, device=g.device, dtype=torch.float32))
            r0 = 0
            for i, n in enumerate(ctx.splits):
                if need[8 + i] and dwp is not None:
                    dws[i] = dwp[r0:r0 + n].view(ctx.shapes[i])
                r0 += n
        if need_b:
            dbp = _bias_grad(gs, ld, ctx.bkey)                          # (packed biases: one buffer, registered under the first one)
            r0 = 0
            for i, n in enumerate(ctx.splits):
                if i < len(dbs) and need[8 + nw + i] and dbp is not None:
                    dbs[i] = dbp[r0:r0 + n]
                r0 += n
        dres = (gs if ld == out_f else gs[:, :out_f].contiguous()) if need[7] else None       # d(residual) = the masked output gradient
        return (dx, None, None, None, None, None, None, dres) + tuple(dws) + tuple(dbs)


class _Conv3x3Fn(ops.CountedFunction):
    """3x3, stride 1, padding 1 on NHWC: sw_conv3x3_igemm (+ bias, ReLU fused) on the staged [co][tap][ci] copy of the effective
    weight; backward = weight gradient (split-K slabs, ordered fold; x FrozenBN scale), column sums, data gradient through the
    staged flipped-tap layout `staged_d` [ci][tap][co].  `w` / `b`: the parameters the gradients go to (b may be None)."""

    @staticmethod
    def forward(ctx, x, staged, staged_d, bias, scale, relu, w, b):
        n, H, W, cin = x.shape
        cout = w.shape[0]
        cd = x.dtype
        flags = int(relu) if MASKS_IN_PRODUCERS else (int(relu) & _RELU)
        relu = bool(flags & _RELU)
        ctx.premasked = bool(flags & _GRAD_PREMASKED)
        out = torch.empty(n, H, W, cout, device=x.device, dtype=cd)
        ops.conv3x3(x, staged, out, 1, ops.make_epilogue(bias=bias, relu=relu, out_dtype=cd))
        ctx.save_for_backward(x, staged_d, out if relu else None, scale)
        ctx.relu, ctx.cout, ctx.wkey, ctx.bkey = relu, cout, id(w), (None if b is None else id(b))
        if ctx.needs_input_grad[6]:
            ops.count_use(id(w))
        return out

    @staticmethod
    def backward(ctx, g):
        x, wkd, out, scale = ctx.saved_tensors
        n, H, W, cin = x.shape
        cout = ctx.cout
        cd = x.dtype
        g = g.contiguous()
        dz = ops.relu_bwd(out, g, out=torch.empty_like(g)) if (ctx.relu and not ctx.premasked) else g
        dx = dw = db = None
        if ctx.needs_input_grad[6]:
            dw = _wgrad_3x3(x, dz, scale, ctx.wkey)
        if ctx.needs_input_grad[7]:
            db = _bias_grad(dz.view(n * H * W, cout), cout, ctx.bkey)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, H, W, cin, device=g.device, dtype=cd)
            ops.conv3x3(dz, wkd, dx, 1, ops.make_epilogue(out_dtype=cd))
        return dx, None, None, None, None, None, dw, db


def _few_tile_splits(P, N, K):
    """K-splits of a bf16 / f32 GEMM with an epilogue whose tile grid fills a fraction of the chip (res5's 1x1 convolutions, the box
    head's fc6 on 512-1024 ROIs): slabs + the epilogue in the fold (sw_gemm).  Measured (tools/probes/gemm_splitk_ep_probe.py): pays from
    K = 2048 on — K = 12544, 64 tiles: 103 -> 46 us; K = 2048, 32-64 tiles: 28 -> 22 us; K = 1024: 19 -> 22 us (not split)."""
    if K < 2048:
        return 1
    tiles = ((P + 127) // 128) * ((N + 127) // 128)
    return max(1, min(8, 256 // max(tiles, 1), K // 512))


def _view4(dw, cout, cin):
    return None if dw is None else dw.view(cout, cin, 1, 1)


def _eff_splits(K, sk, bf16):
    """the K-split count sw_gemm will really use (gemm.hip effective_splits)"""
    bk = 64 if bf16 else 32
    kps = -(-K // max(1, sk))
    kps = -(-kps // bk) * bk
    return -(-K // kps)


def _wgrad_1x1(gs, x, scale, key=None):
    """dW (out, in) f32 = scale[:, None] * gs^T x over the pixels: K-split slabs + ordered fold (deterministic).  Inside
    ops.grad_scope a second use of the same weight (`key`) adds to the first use's buffer in the fold / epilogue and returns None
    (ALWAYS, once a buffer is registered: autograd may already have replaced the registered tensor by a sum of its own if a later
    use handed it a gradient too)."""
    P, ld = gs.shape
    D = x.shape[1]
    tiles = ((ld + 127) // 128) * ((D + 127) // 128)
    sk = max(1, min(64, 512 // tiles, P // 512))
    prev = ops.pending_grad(key, (ld, D))
    if prev is not None:
        ws = None
        if scale is not None and _eff_splits(P, sk, gs.dtype == torch.bfloat16) == 1:
            ws = torch.empty(ld * D, device=gs.device, dtype=torch.float32)        # one slab: row scale + residual run in the fold
        ops.gemm(gs, x, prev, ld, D, P, a_kstrided=True, b_kstrided=True, splitk=sk,
                 ep=ops.make_epilogue(out_dtype=torch.float32, row_scale=scale, residual=prev, splitk_workspace=ws))
        return None
    dw = torch.empty(ld, D, device=gs.device, dtype=torch.float32)
    ep = None if scale is None else ops.make_epilogue(out_dtype=torch.float32, row_scale=scale)
    ops.gemm(gs, x, dw, ld, D, P, a_kstrided=True, b_kstrided=True, splitk=sk, ep=ep)
    ops.note_grad(key, dw)
    return dw


def _bias_grad(gs2d, n, key=None):
    """db (n,) f32 = column sums of the (masked) output gradient; ops.grad_scope as _wgrad_1x1 (key: the bias parameter)"""
    P = gs2d.shape[0]
    prev = ops.pending_grad(key, (n,))
    if prev is not None:
        if P > 0:
            ops.colsum(gs2d, P, n, prev, accumulate=True)
        return None
    db = torch.empty(n, device=gs2d.device, dtype=torch.float32)
    if P > 0:
        ops.colsum(gs2d, P, n, db)
    else:
        db.zero_()
    ops.note_grad(key, db)
    return db.view(n)


_GROUP_TARGETS_1X1 = (4, 6, 8, 12, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160)


def _flush_wgrad_1x1(q):
    """the queued 1x1 weight-gradient problems of one bottleneck block — per use (dh1, x), (gs, h2)[, (gs, x)] — as ONE
    sw_gemm_kk_grouped launch and ONE sw_splitk_fold_multi (x FrozenBN scale) into the buffers autograd already holds"""
    from .backbone_vgg import _wgrad_grouped_splits, _wgrad_grouped_target
    uses, bufs, scales = q["probs"], q["bufs"], q["scales"]
    dtype = uses[0][0][0].dtype
    bk = 64 if dtype == torch.bfloat16 else 32
    shapes = [(a.shape[0], a.shape[1], b.shape[1]) for use in uses for a, b in use]
    target = _wgrad_grouped_target(shapes, bk, candidates=_GROUP_TARGETS_1X1)
    probs, folds = [], []
    for w, (buf, scale) in enumerate(zip(bufs, scales)):
        ns = [_wgrad_grouped_splits(use[w][0].shape[0], bk, target) for use in uses]
        nsl = [ops.gemm_kk_nslab(dtype, use[w][0].shape[0], s_) for use, s_ in zip(uses, ns)]
        ws = torch.empty(sum(nsl), buf.numel(), device=buf.device, dtype=torch.float32)
        off = 0
        for use, s_, k in zip(uses, ns, nsl):
            probs.append((use[w][0], use[w][1], ws[off:], s_))
            off += k
        folds.append((ws, sum(nsl), buf, scale, False))
    ops.gemm_kk_grouped(probs)
    ops.splitk_fold_multi(folds)
    q["probs"] = []


_GROUP_TARGETS = (8, 12, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160)      # K-tiles per work item tried for a grouped launch


def _small_map(H, W):
    """maps of a few pixels (p5 / p6 of small images: 4x4, 2x2) are below the gathering loader's tile geometry (sw_conv3x3_wgrad
    returns -6): a direct kernel, one thread per (co, ci), takes them"""
    return (64 // W) + 1 > 2 * H


def _flush_wgrad_3x3(q):
    """all queued (x, dy) pairs of one 3x3 weight: ONE grouped 256x256-tile launch (sw_conv3x3_wgrad_grouped: every pair's K-splits
    as work items of one resident grid) + ONE fold over all slabs (x FrozenBN scale) into the buffer autograd already holds.
    Measured (tools/probes/stage3_grouped_wgrad_probe.py): the RPN head's 10 uses 641 -> 383 us, an FPN output convolution's two 330 -> 256
    (p2) / 71 -> 46 (p4), res5 conv2 83 -> 48, res3 conv2 74 -> 52."""
    from .backbone_vgg import _wgrad_grouped_splits, _wgrad_grouped_target
    dw, scale, probs = q["buf"], q["scale"], q["probs"]
    cout, cin = dw.shape[:2]
    big = [(x, dz) for x, dz in probs if not _small_map(x.shape[1], x.shape[2])]
    small = [(x, dz) for x, dz in probs if _small_map(x.shape[1], x.shape[2])]
    wrote = False
    if big:
        bk = 64 if big[0][0].dtype == torch.bfloat16 else 32
        shapes = [(x.shape[0] * x.shape[1] * x.shape[2], cout, 9 * cin) for x, _ in big]
        target = _wgrad_grouped_target(shapes, bk, candidates=_GROUP_TARGETS)
        splits = [_wgrad_grouped_splits(sh[0], bk, target) for sh in shapes]
        nsl = [ops.conv3x3_wgrad_nslab(x, cout, sp) for (x, _), sp in zip(big, splits)]
        ws = torch.empty(sum(nsl), cout * 9 * cin, device=dw.device, dtype=torch.float32)
        off, items = 0, []
        for (x, dz), sp, n in zip(big, splits, nsl):
            items.append((x, dz, ws[off:], 1, sp))
            off += n
        ops.conv3x3_wgrad_grouped(items)
        ops.conv3x3_wgrad_fold(ws, sum(nsl), dw, cout_scale=scale)
        wrote = True
    for x, dz in small:
        ops.conv3x3_wgrad_small(x, dz, dw, cout_scale=scale, accumulate=wrote)
        wrote = True
    q["probs"] = []


def _wgrad_3x3(x4, dz4, scale, key=None):
    """dW (cout, cin, 3, 3) f32 of a 3x3 convolution (sw_conv3x3_wgrad: slabs + fold, x FrozenBN scale).  Inside ops.grad_scope: a
    weight whose uses were counted in the forward passes (ops.count_use) queues its (x, dy) pairs and the LAST use computes all of
    them at once (_flush_wgrad_3x3) into the buffer the first use handed to autograd; an uncounted weight adds to the first use's
    buffer in the fold, as _wgrad_1x1 does."""
    n, H, W, cin = x4.shape
    cout = dz4.shape[3]
    sc = ops.GRAD_SCOPE
    if sc is not None and key is not None and sc.uses.get(key, 0) > 1:
        q = sc.queued.get(key)
        first = q is None
        if first:
            q = sc.queued[key] = dict(buf=torch.empty(cout, cin, 3, 3, device=x4.device, dtype=torch.float32), scale=scale, probs=[],
                                      left=sc.uses[key])
        q["probs"].append((x4, dz4))
        q["left"] -= 1
        if q["left"] == 0:
            _flush_wgrad_3x3(q)
        return q["buf"].view(cout, cin, 3, 3) if first else None
    prev = ops.pending_grad(key, (cout, cin, 3, 3))
    dw = prev if prev is not None else torch.empty(cout, cin, 3, 3, device=x4.device, dtype=torch.float32)
    if _small_map(H, W):
        ops.conv3x3_wgrad_small(x4, dz4, dw, cout_scale=scale, accumulate=prev is not None)
    else:
        tiles = ((cout + 127) // 128) * ((9 * cin + 127) // 128)
        sk = max(1, min(32, 512 // tiles, max(1, n * H * W // 1024)))
        ops.conv3x3_wgrad(x4, dz4, dw, 1, splitk=sk, cout_scale=scale, accumulate=prev is not None)
    if prev is not None:
        return None
    ops.note_grad(key, dw)
    # (a view: the registered alias keeps `dw` itself referenced as its base, and autograd's accumulator copies a gradient whose
    # tensor object somebody else holds instead of adopting it — 18 device-to-device copies per iteration)
    return dw.view(cout, cin, 3, 3)


class _BottleneckFn(ops.CountedFunction):
    """One bottleneck block (conv1 1x1 [stride s] -> ReLU -> conv2 3x3 -> ReLU -> conv3 1x1, + shortcut, ReLU) as ONE autograd node
    with an explicit backward.  Layer by layer (`_LinearFn` / `_Conv3x3Fn` nodes) every ReLU mask was a kernel of its own in front of
    the layer's gradients and autograd added the two gradients of the block input with a torch kernel; here the masks of conv1's and
    conv2's outputs run in the epilogue of the data-gradient kernel that produces the gradient (`relu_ref`), the shortcut branch's
    gradient is the `residual` of conv1's data-gradient GEMM, and a stride-2 block subsamples its input once for conv1 and the
    shortcut (one scatter in the backward).  Per block 2 mask kernels, 1 add (stride 2: + 1 subsample, 1 scatter, 1 add) fewer.
    Against the layer-by-layer form (tests/test_gpu_stage3.py): same losses; fp32 gradients equal up to the association of the
    three-term sums at the stage outputs (1e-6), bf16: the block-input gradient is rounded once instead of twice.
    args: x NHWC, `blk` the BottleneckBlock (staged operands, strides), then conv1/conv2/conv3[/shortcut] weights for autograd."""

    @staticmethod
    def forward(ctx, x, blk, *weights):
        """x (n, H, W, cin) -> (n, H', W', cout); or, blk = (block, maps): x (rows, cin) = the rows of SEVERAL maps [(n, H, W), ...] back
        to back (the student's passes in lockstep: every 1x1 convolution is one GEMM over all rows, the 3x3 one multi-problem launch)
        -> (rows', cout) in the same map order"""
        single = not isinstance(blk, tuple)
        if single:
            maps = [tuple(x.shape[:3])]
            cin = x.shape[3]
            x = x.reshape(-1, cin)
        else:
            blk, maps = blk
            maps = [tuple(m) for m in maps]
            cin = x.shape[1]
        c1, c2, c3, sc = blk.conv1, blk.conv2, blk.conv3, blk.shortcut
        s1, s2, s3 = _staged_of(c1), _staged_of(c2), _staged_of(c3)
        cd = x.dtype
        in_maps = maps
        if c1.stride == 2:
            maps = [(n, (H + 1) // 2, (W + 1) // 2) for n, H, W in in_maps]
            xs = torch.empty(sum(n * H * W for n, H, W in maps), cin, device=x.device, dtype=cd)
            r0 = q0 = 0
            for (n, H, W), (_, h, w) in zip(in_maps, maps):
                ops.subsample2x(x[r0:r0 + n * H * W].view(n, H, W, cin), xs[q0:q0 + n * h * w].view(n, h, w, cin))
                r0 += n * H * W; q0 += n * h * w
            x = xs
        P = x.shape[0]
        offs = [0]
        for n, H, W in maps:
            offs.append(offs[-1] + n * H * W)
        assert offs[-1] == P
        mid, cout = c1.weight.shape[0], c3.weight.shape[0]
        x2 = x

        def lin(a, st, relu, out_f, residual=None):
            y = torch.empty(P, out_f, device=a.device, dtype=cd)
            D = a.shape[1]
            ops.gemm(a, st.w, y, P, out_f, D, ep=ops.make_epilogue(bias=st.shift, relu=relu, out_dtype=cd, residual=residual),
                     splitk=_few_tile_splits(P, out_f, D), tag="s3_gemm_fwd")
            if ops.TIMER is not None:
                es = 2 if cd == torch.bfloat16 else 4
                ops.TIMER.note("s3_gemm_fwd", 2.0 * P * out_f * D)
                ops.TIMER.note("s3_gemm_fwd_bytes", float(es) * (P * D + out_f * D + P * out_f))
            return y
        h1 = lin(x2, s1, True, mid)
        h2 = torch.empty(P, mid, device=x.device, dtype=cd)
        ops.conv3x3_multi([(h1[offs[i]:offs[i + 1]].view(n, H, W, mid), s2.w, h2[offs[i]:offs[i + 1]].view(n, H, W, mid),
                            ops.make_epilogue(bias=s2.shift, relu=True, out_dtype=cd)) for i, (n, H, W) in enumerate(maps)])
        short = x2 if sc is None else lin(x2, _staged_of(sc), False, cout)
        out = lin(h2, s3, True, cout, residual=short)
        ctx.save_for_backward(x2, h1, h2, out, s1.w, s2.wd, s3.w, None if sc is None else _staged_of(sc).w,
                              s1.scale, s2.scale, s3.scale, None if sc is None else _staged_of(sc).scale)
        ctx.geom = (maps, in_maps, offs, cin, mid, cout, c1.stride, single)
        ctx.flags = (bool(blk.mask_input_grad) and c1.stride == 1, bool(blk.grad_premasked)) if MASKS_IN_PRODUCERS else (False, False)
        ctx.keys = tuple(id(w) for w in weights) + (None,) * (4 - len(weights))
        if ctx.needs_input_grad[3]:
            for _ in maps:
                ops.count_use(id(weights[1]))                            # conv2: every map's weight gradient joins one grouped launch
        if all(ctx.needs_input_grad[2:]):
            ops.count_use(("1x1", id(weights[0])))                      # the block's 1x1 weights: one grouped launch for all passes
        n, H, W = maps[0]
        return out.view(n, H, W, cout) if single else out

    @staticmethod
    def backward(ctx, g):
        x2, h1, h2, out, w1, w2d, w3, wsc, sc1, sc2, sc3, scs = ctx.saved_tensors
        maps, in_maps, offs, cin, mid, cout, stride, single = ctx.geom
        P = offs[-1]
        cd = x2.dtype
        need = ctx.needs_input_grad                                   # (x, blk, w1, w2, w3[, wsc])
        mask_in, premasked = ctx.flags
        g2 = g.contiguous().view(P, cout)
        # (premasked: this block's output is read by the next block only, whose conv1 data gradient left masked by it)
        gs = g2 if premasked else ops.relu_bwd(out, g2, out=torch.empty(P, cout, device=g.device, dtype=cd))

        def dgrad(a, w, D, **ep):
            d = torch.empty(P, D, device=a.device, dtype=cd)
            K = a.shape[1]
            ops.gemm(a, w, d, P, D, K, b_kstrided=True, ep=ops.make_epilogue(out_dtype=cd, **ep), splitk=_few_tile_splits(P, D, K))
            return d

        def m4(t, i, c):                                              # map i of a row-concatenated (P, c) tensor as NHWC
            n, H, W = maps[i]
            return t[offs[i]:offs[i + 1]].view(n, H, W, c)
        keys = ctx.keys
        dh2 = dgrad(gs, w3, mid, relu_ref=h2)                         # masked by conv2's ReLU
        dw2 = None
        if need[3]:
            for i in range(len(maps)):
                r = _wgrad_3x3(m4(h1, i, mid), m4(dh2, i, mid), sc2, keys[1])
                dw2 = r if dw2 is None else dw2                       # (inside ops.grad_scope only the first use hands out a buffer)
                if r is not None and r is not dw2:
                    dw2 = dw2 + r                                     # outside a scope: plain sums
        dh1 = torch.empty(P, mid, device=g.device, dtype=cd)
        ops.conv3x3_multi([(m4(dh2, i, mid), w2d, m4(dh1, i, mid), ops.make_epilogue(out_dtype=cd, relu_ref=h1[offs[i]:offs[i + 1]]))
                           for i in range(len(maps))])                # masked by conv1's ReLU (reference as (pixels, mid) rows)
        dx = None
        if need[0]:
            side = gs if wsc is None else dgrad(gs, wsc, cin)         # the shortcut branch's gradient of the (subsampled) block input
            dx = dgrad(dh1, w1, cin, residual=side, **({"relu_ref": x2} if mask_in else {}))
            if stride == 2:
                full = torch.empty(sum(n * H * W for n, H, W in in_maps), cin, device=g.device, dtype=cd)
                r0 = 0
                for i, (n, H, W) in enumerate(in_maps):
                    ops.scatter2x(m4(dx, i, cin), full[r0:r0 + n * H * W].view(n, H, W, cin))
                    r0 += n * H * W
                dx = full
            if single:
                n, H, W = in_maps[0]
                dx = dx.view(n, H, W, cin)
        # ---- the 1x1 weight gradients: conv1 (dh1^T x), conv3 (gs^T h2), shortcut (gs^T x)
        scope = ops.GRAD_SCOPE
        bkey = ("1x1", keys[0])
        dw1 = dw3 = dwsc = None
        if scope is not None and all(need[2:]) and (scope.uses.get(bkey, 0) > 1 or (len(maps) > 1 and scope.uses.get(bkey, 0) == 1)):
            # every pass's three problems are queued; the last use runs all of them as ONE grouped launch + ONE multi-fold
            # (several maps in one call: the rows of all of them are one problem per weight — still one grouped launch)
            q = scope.queued.get(bkey)
            first = q is None
            if first:
                shp = [(mid, cin), (cout, mid)] + ([(cout, cin)] if wsc is not None else [])
                q = scope.queued[bkey] = dict(bufs=[torch.empty(a, b, device=g.device, dtype=torch.float32) for a, b in shp],
                                              scales=[sc1, sc3] + ([scs] if wsc is not None else []), probs=[], left=scope.uses[bkey])
            q["probs"].append([(dh1, x2), (gs, h2)] + ([(gs, x2)] if wsc is not None else []))
            q["left"] -= 1
            if q["left"] == 0:
                _flush_wgrad_1x1(q)
            if first:
                dw1, dw3 = q["bufs"][0].view(mid, cin, 1, 1), q["bufs"][1].view(cout, mid, 1, 1)
                dwsc = q["bufs"][2].view(cout, cin, 1, 1) if wsc is not None else None
        else:
            dw3 = _view4(_wgrad_1x1(gs, h2, sc3, keys[2]), cout, mid) if need[4] else None
            dw1 = _view4(_wgrad_1x1(dh1, x2, sc1, keys[0]), mid, cin) if need[2] else None
            if wsc is not None and len(need) > 5 and need[5]:
                dwsc = _view4(_wgrad_1x1(gs, x2, scs, keys[3]), cout, cin)
        return (dx, None, dw1, dw2, dw3) + ((dwsc,) if wsc is not None else ())


class _Conv3x3LevelsFn(ops.CountedFunction):
    """L independent 3x3 convolutions (stride 1, padding 1, + bias, optional ReLU) of L maps as ONE launch each way (ops.conv3x3_multi:
    the FPN levels — the RPN head's shared convolution on p2..p6, the four FPN output convolutions).  args: L, flags (the `relu`
    argument of _Conv3x3Fn), then L inputs, L staged forward weights, L staged data-gradient weights, L bias values, L weight parameters,
    L bias parameters (a shared layer repeats its tensors).  Weight / bias gradients go through the same helpers as _Conv3x3Fn."""

    @staticmethod
    def forward(ctx, L, relu, *ts):
        xs, sws, swds, bvs, ws, bs = (ts[i * L:(i + 1) * L] for i in range(6))
        flags = int(relu) if MASKS_IN_PRODUCERS else (int(relu) & _RELU)
        relu = bool(flags & _RELU)
        cd = xs[0].dtype
        outs, probs = [], []
        global _LEVELS_CAT
        cat, _LEVELS_CAT = _LEVELS_CAT, None          # (rows, cout) buffer the caller wants the L outputs in, maps back to back (RPN head)
        r0 = 0
        for x, sw, bv, w in zip(xs, sws, bvs, ws):
            n, H, W, _ = x.shape
            if cat is not None:
                out = cat[r0:r0 + n * H * W].view(n, H, W, w.shape[0]); r0 += n * H * W
            else:
                out = torch.empty(n, H, W, w.shape[0], device=x.device, dtype=cd)
            probs.append((x, sw, out, ops.make_epilogue(bias=bv, relu=relu, out_dtype=cd)))
            outs.append(out)
        ops.conv3x3_multi(probs)
        ctx.save_for_backward(*xs, *swds, *(outs if relu else ()))
        ctx.L, ctx.relu, ctx.premasked = L, relu, bool(flags & _GRAD_PREMASKED)
        ctx.wkeys, ctx.bkeys, ctx.couts = [id(w) for w in ws], [id(b) for b in bs], [int(w.shape[0]) for w in ws]
        need = ctx.needs_input_grad
        for i, w in enumerate(ws):
            if need[2 + 4 * L + i]:
                ops.count_use(id(w))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        L = ctx.L
        sv = ctx.saved_tensors
        xs, swds, outs = sv[:L], sv[L:2 * L], sv[2 * L:]
        need = ctx.needs_input_grad
        cd = xs[0].dtype
        dzs = []
        for i, g in enumerate(gs):
            if g is None:                                             # an output nobody used in the loss
                n, H, W, _ = xs[i].shape
                g = ops.fill_zero(torch.empty(n, H, W, ctx.couts[i], device=xs[i].device, dtype=cd))
            g = g.contiguous()
            dzs.append(ops.relu_bwd(outs[i], g, out=torch.empty_like(g)) if (ctx.relu and not ctx.premasked) else g)
        dxs, dws, dbs = [None] * L, [None] * L, [None] * L
        for i in range(L):
            n, H, W, _ = xs[i].shape
            cout = dzs[i].shape[3]
            if need[2 + 4 * L + i]:
                dws[i] = _wgrad_3x3(xs[i], dzs[i], None, ctx.wkeys[i])
            if need[2 + 5 * L + i]:
                dbs[i] = _bias_grad(dzs[i].view(n * H * W, cout), cout, ctx.bkeys[i])
        probs = []
        for i in range(L):
            if need[2 + i]:
                dxs[i] = torch.empty_like(xs[i])
                probs.append((dzs[i], swds[i], dxs[i], ops.make_epilogue(out_dtype=cd)))
        ops.conv3x3_multi(probs)
        return (None, None) + tuple(dxs) + (None,) * (3 * L) + tuple(dws) + tuple(dbs)


_LEVELS_CAT = None


def _conv3x3_levels(convs, xs, relu=False, cat=None):
    """the 3x3 `Conv` modules `convs[i]` applied to `xs[i]` (convs may repeat one module: a layer shared by the levels).  cat: a
    (sum of the maps' pixels, cout) buffer — the outputs are then row blocks of it, written there by the convolution itself"""
    global _LEVELS_CAT
    L = len(xs)
    sts = [_staged_of(c) for c in convs]
    _LEVELS_CAT = cat
    return _Conv3x3LevelsFn.apply(L, relu, *xs, *[st.w for st in sts], *[st.wd for st in sts], *[c.bias.detach() for c in convs],
                                  *[c.weight for c in convs], *[c.bias for c in convs])


class _Subsample2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, H, W, C = x.shape
        ctx.shape = tuple(x.shape)
        return ops.subsample2x(x, torch.empty(n, (H + 1) // 2, (W + 1) // 2, C, device=x.device, dtype=x.dtype))

    @staticmethod
    def backward(ctx, g):
        return ops.scatter2x(g.contiguous(), torch.empty(ctx.shape, device=g.device, dtype=g.dtype))


class _UpsampleAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lateral, top):
        ctx.top_shape = tuple(top.shape)
        return ops.upsample2x_add(lateral, top, torch.empty_like(lateral))

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return g, ops.downsample2x_sum(g, torch.empty(ctx.top_shape, device=g.device, dtype=g.dtype))


class _RoIAlignFn(torch.autograd.Function):
    """(rois (R,5), per-level row lists sel (4, R) + their DEVICE lengths, scales, *level features NHWC) -> pooled (R, C*7*7) in the
    reference's (C,7,7) order.  Every row belongs to exactly one level, so the four launches together write every row."""

    @staticmethod
    def forward(ctx, rois, sel, sel_cnt, scales, *feats):
        R = rois.shape[0]
        C = feats[0].shape[3]
        out = torch.empty(R, C * 49, device=feats[0].device, dtype=feats[0].dtype)
        if R:
            for l, (f, sc) in enumerate(zip(feats, scales)):
                ops.roi_align_fwd(f, rois, sel[l], out, sc, n_sel_dev=sel_cnt[l:l + 1])
        ctx.rois, ctx.sel, ctx.sel_cnt, ctx.scales = rois, sel, sel_cnt, scales
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.dtype = feats[0].dtype
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        grads = []
        amax = None
        R = ctx.rois.shape[0]
        for l, (shp, sc, need) in enumerate(zip(ctx.shapes, ctx.scales, ctx.needs_input_grad[4:])):
            if not need:
                grads.append(None)
                continue
            if _ALIGN_BWD_FX and R:
                # deterministic (round 6): 64-bit fixed-point accumulation, one conversion into the map's dtype (csrc/detector.hip
                # roi_align_bwd_fx_kernel) — the float-atomic form below left the iteration irreproducible in the last bits
                if amax is None:
                    amax = ops.absmax(g)
                acc = ops.fill_zero(torch.empty(shp, device=g.device, dtype=torch.int64))
                ops.roi_align_bwd_fx(g, ctx.rois, ctx.sel[l], acc, sc, amax, n_sel_dev=ctx.sel_cnt[l:l + 1])
                grads.append(ops.fx_to_float(acc, amax, torch.empty(shp, device=g.device, dtype=ctx.dtype)))
                continue
            d = ops.fill_zero(torch.empty(shp, device=g.device, dtype=torch.float32))
            if R:
                ops.roi_align_bwd(g, ctx.rois, ctx.sel[l], d, sc, n_sel_dev=ctx.sel_cnt[l:l + 1])
            grads.append(d if ctx.dtype == torch.float32 else ops.convert_flat(d, ctx.dtype))
        return (None, None, None, None) + tuple(grads)


class _RpnLossFn(torch.autograd.Function):
    """y (rows, 5A) = the RPN head's packed output; logits (N, At) / deltas (N, At, 4) = its anchor-order form (ops.rpn_unpack, computed
    once outside autograd: the proposal selection reads it too) -> (loss_rpn_cls, loss_rpn_loc): one kernel for both values and both
    unit gradients; the backward maps them back onto y's layout, times the two cotangents, in one kernel"""

    @staticmethod
    def forward(ctx, y, logits, deltas, labels_i8, anchors, matched, weights, inv_norm, meta):
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        dl = torch.empty_like(logits); dd = torch.empty_like(deltas)
        ops.rpn_loss(logits.reshape(-1), deltas.reshape(-1, 4), labels_i8, anchors, matched, weights, inv_norm, out, dl.reshape(-1),
                     dd.reshape(-1, 4))
        ctx.save_for_backward(dl, dd)
        ctx.meta = meta + (y.shape[0], y.shape[1], y.stride(0))
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_loc):
        dl, dd = ctx.saved_tensors
        N, A, hw, rows, cols, ld = ctx.meta
        dy = ops.rpn_unpack_bwd(dl, dd, N, A, hw, rows, ld, dl.device, g_logits=g_cls.contiguous(), g_deltas=g_loc.contiguous())
        return dy[:, :cols], None, None, None, None, None, None, None, None


class _RoiLossFn(torch.autograd.Function):
    """packed logits (R, >= 5K+1) f32 = [cls_score K+1 | bbox_pred 4K] -> (focal loss_cls, L1 loss_box_reg), both / R
    (unbias/ubteacher/modeling/roi_heads/fast_rcnn.py:73-105; detectron2/modeling/roi_heads/fast_rcnn.py:245-317).
    boxes2 (2R, 4) = the sampled proposal boxes followed by their matched gt boxes (row R + i is row i's target)."""

    @staticmethod
    def forward(ctx, logits, K, gt_classes_i32, boxes2, reg_weights, gamma):
        R = logits.shape[0]
        dev = logits.device
        c = _consts(dev, R)
        loss_cls = ops.fill_zero(torch.empty(1, device=dev))
        unit = ops.fill_zero(torch.empty_like(logits))
        lg = logits.detach()
        # box term through the refinement-loss kernel: targets boxes2[R + i], class-specific columns, L1 / R (CE weights 0); it writes the
        # box columns of `unit` (and zeros into the class columns, which the focal kernel then overwrites)
        lv = torch.empty(1, 2, 1, device=dev)
        ops.oicr_refine_loss(lg, 1, R, K, 0, K + 1, boxes2, gt_classes_i32.view(1, R), c["zero_w"].view(1, R), c["idx"].view(1, R),
                             c["zero_i"], reg_weights, lv, unit, c["ones2"])
        ops.focal_loss(lg[:, :K + 1], gt_classes_i32, gamma, loss_cls, unit[:, :K + 1])
        ctx.save_for_backward(unit)
        ctx.K = K
        return loss_cls[0], lv[0, 1, 0]

    @staticmethod
    def backward(ctx, g_cls, g_box):
        (unit,) = ctx.saved_tensors
        K = ctx.K
        d = torch.empty_like(unit)
        ops.scale_col_blocks(unit, d, K + 1, g_cls.contiguous(), g_box.contiguous())
        return d, None, None, None, None, None


_CONSTS = {}


def _consts(dev, R):
    """small constant device tensors of the ROI loss (zero CE weights, target row indices R + i, ...), built once per (device, R)"""
    key = (str(dev), R)
    hit = _CONSTS.get(key)
    if hit is None:
        if len(_CONSTS) > 64:
            _CONSTS.clear()
        hit = _CONSTS[key] = dict(zero_w=torch.zeros(R, device=dev), idx=(torch.arange(R, device=dev, dtype=torch.int32) + R).contiguous(),
                                  zero_i=torch.zeros(1, dtype=torch.int32, device=dev), ones2=torch.ones(2, device=dev))
    return hit


# ====================================================================================================== modules
class FrozenBatchNorm2d(nn.Module):
    """layers/batch_norm.py:15-100: buffers only; folded into the convolution in front of it"""

    def __init__(self, c):
        super().__init__()
        self.register_buffer("weight", torch.ones(c)); self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c)); self.register_buffer("running_var", torch.ones(c) - 1e-5)

    def fold(self):
        """(scale, shift) of y = x * scale + shift; cached: the statistics are frozen buffers (rebuilt when one of them was written,
        e.g. by load_state_dict or the teacher's EMA, which bump the tensors' version counters)"""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version, self.weight.device,
               ops.BUFFER_EPOCH)           # (BUFFER_EPOCH: the teacher EMA writes the buffers behind torch's version counters)
        hit = self.__dict__.get("_fold")
        if hit is None or hit[0] != key:
            scale = self.weight * torch.rsqrt(self.running_var + 1e-5)
            shift = self.bias - self.running_mean * scale
            if hit is not None and hit[1][0].device == scale.device and hit[1][0].shape == scale.shape:
                # the SAME two tensors for the module's lifetime, rewritten in place: a captured launch that read them (the no-grad
                # backbone graph of TwoStagePseudoLabGeneralizedRCNN._features) sees the new values
                hit[1][0].copy_(scale); hit[1][1].copy_(shift)
                hit = (key, hit[1])
            else:
                hit = (key, (scale.contiguous(), shift.contiguous()))
            self.__dict__["_fold"] = hit
        return hit[1]


class ConvBN(nn.Module):
    """Conv2d(bias=False, norm=FrozenBN) of the reference: parameter `weight` (OIHW) + submodule `norm`.  The FrozenBN fold lives in
    the staged weight (W * scale) and the epilogue bias (shift); d/dweight = scale * d/dW_eff in the layer's backward."""

    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.norm = FrozenBatchNorm2d(cout)
        self.k, self.stride = k, stride

    def _stage_entries(self, cd):
        if self.k == 7:
            return []                                               # the stem kernel reads the f32 weight and the fold directly
        cout, cin = self.weight.shape[:2]
        dev = self.weight.device
        n = self.norm
        bn = (n.weight, n.bias, n.running_mean, n.running_var)
        st = _Staged(scale=torch.empty(cout, device=dev), shift=torch.empty(cout, device=dev))
        if self.k == 3:
            st.w = torch.empty(cout, 9, cin, device=dev, dtype=cd)
            ent = [dict(kind=1, w=self.weight, dst=st.w, bn=bn, scale=st.scale, shift=st.shift)]
            if self.weight.requires_grad:
                st.wd = torch.empty(cin, 9, cout, device=dev, dtype=cd)
                ent.append(dict(kind=2, w=self.weight, dst=st.wd, bn=bn))
        else:
            st.w = torch.zeros(_pad8(cout), cin, device=dev, dtype=cd)
            ent = [dict(kind=0, w=self.weight, dst=st.w, bn=bn, scale=st.scale, shift=st.shift)]
        self.__dict__["_st"] = st
        return ent

    def forward(self, x, relu, residual=None):
        st = _staged_of(self)
        if self.k == 3:
            assert residual is None
            return _Conv3x3Fn.apply(x, st.w, st.wd, st.shift, st.scale, relu, self.weight, None)
        if self.stride == 2:
            x = _Subsample2Fn.apply(x)
        n, H, W, C = x.shape
        cout = self.weight.shape[0]
        res = None if residual is None else residual.reshape(n * H * W, cout)
        y = _LinearFn.apply(x.reshape(n * H * W, C), st.w, st.shift, st.scale, relu, False, (cout,), res, self.weight)
        return y.reshape(n, H, W, cout)


class Conv(nn.Module):
    """plain Conv2d with bias (FPN laterals / outputs, RPN head).  packed=True: staged by the owner as part of a packed GEMM."""

    def __init__(self, cin, cout, k, packed=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k)); self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=1)
        self.k, self.packed = k, packed

    def _stage_entries(self, cd):
        if self.packed:
            return []
        cout, cin = self.weight.shape[:2]
        dev = self.weight.device
        st = _Staged()
        if self.k == 3:
            st.w = torch.empty(cout, 9, cin, device=dev, dtype=cd); st.wd = torch.empty(cin, 9, cout, device=dev, dtype=cd)
            ent = [dict(kind=1, w=self.weight, dst=st.w), dict(kind=2, w=self.weight, dst=st.wd)]
        else:
            st.w = torch.zeros(_pad8(cout), cin, device=dev, dtype=cd)
            ent = [dict(kind=0, w=self.weight, dst=st.w)]
        self.__dict__["_st"] = st
        return ent

    def forward(self, x, relu=False):
        """relu: False / True or layer flags (_RELU | _GRAD_PREMASKED ...)"""
        st = _staged_of(self)
        if self.k == 3:
            return _Conv3x3Fn.apply(x, st.w, st.wd, self.bias.detach(), None, relu, self.weight, self.bias)
        n, H, W, C = x.shape
        cout = self.weight.shape[0]
        y = _LinearFn.apply(x.reshape(n * H * W, C), st.w, self.bias.detach(), None, relu, False, (cout,), None, self.weight, self.bias)
        return y.reshape(n, H, W, cout)


def _packed_linear_entries(weights, biases, cd):
    """stage entries of several (out_i, in) weights stacked into one (pad8(sum out_i), in) GEMM operand + their biases in one f32 row"""
    rows = [w.shape[0] for w in weights]
    cin = weights[0].numel() // rows[0]
    dev = weights[0].device
    st = _Staged(w=torch.zeros(_pad8(sum(rows)), cin, device=dev, dtype=cd), bias=torch.zeros(_pad8(sum(rows)), device=dev))
    ent, r0 = [], 0
    for w, b, n in zip(weights, biases, rows):
        ent.append(dict(kind=0, w=w, dst=st.w[r0:r0 + n]))
        ent.append(dict(kind=3, w=b, dst=st.bias[r0:r0 + n]))
        r0 += n
    return st, ent


class BottleneckBlock(nn.Module):
    """backbone/resnet.py:100-213 with STRIDE_IN_1X1 (the stride sits in conv1 and in the shortcut)"""

    def __init__(self, cin, cout, mid, stride):
        super().__init__()
        self.shortcut = ConvBN(cin, cout, 1, stride) if cin != cout else None
        self.conv1 = ConvBN(cin, mid, 1, stride); self.conv2 = ConvBN(mid, mid, 3); self.conv3 = ConvBN(mid, cout, 1)
        # set by the owner of a chain of blocks (ResNet): this block's input is the previous block's output and nobody else reads it /
        # this block's output is read by the next block only (see _MASK_INPUT_GRAD / _GRAD_PREMASKED; fused form only)
        self.mask_input_grad = self.grad_premasked = False

    def forward(self, x):
        if FUSED_BLOCKS:
            ws = (self.conv1.weight, self.conv2.weight, self.conv3.weight) + (() if self.shortcut is None else (self.shortcut.weight,))
            return _BottleneckFn.apply(x.contiguous(), self, *ws)
        sc = self.shortcut(x, False) if self.shortcut is not None else x
        # out = relu(conv3(...) + shortcut): the add and the ReLU run in conv3's GEMM epilogue (the stand-alone add + ReLU kernel
        # was 45 launches / 1.1 ms of HBM traffic per Stage-3 iteration)
        return self.conv3(self.conv2(self.conv1(x, True), True), True, residual=sc.contiguous())


class BasicStem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = ConvBN(3, 64, 7, 2)

    @staticmethod
    def out_hw(H, W):
        h2, w2 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        return (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1

    def forward(self, x4, out=None):                                # (N, H, W, 4) normalised + padded -> (N, H/4, W/4, 64) [written to `out`]
        n, H, W, _ = x4.shape
        scale, shift = self.conv1.norm.fold()
        y = ops.stem_conv7x7(x4, self.conv1.weight.detach().contiguous(), scale.contiguous(), shift.contiguous(),
                             torch.empty(n, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64, device=x4.device, dtype=x4.dtype))
        if out is None:
            out = torch.empty(n, (y.shape[1] - 1) // 2 + 1, (y.shape[2] - 1) // 2 + 1, 64, device=y.device, dtype=y.dtype)
        return ops.maxpool3x3s2(y, out)


class ResNet(nn.Module):
    def __init__(self, freeze_at=2):
        super().__init__()
        self.stem = BasicStem()
        cin = 64
        self.stage_names = []
        for name, nblk, mid, cout, stride in R50_STAGES:
            blocks = [BottleneckBlock(cin if b == 0 else cout, cout, mid, stride if b == 0 else 1) for b in range(nblk)]
            for b, blk in enumerate(blocks):         # inside a stage block b's output feeds block b + 1 and nothing else (the LAST
                blk.mask_input_grad = b > 0          # block's is the stage output: the next stage and an FPN lateral read it)
                blk.grad_premasked = b < nblk - 1
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
            cin = cout
        self.freeze_at = freeze_at
        if freeze_at >= 1:
            for p in self.stem.parameters():
                p.requires_grad = False
        for i, name in enumerate(self.stage_names, start=2):
            if freeze_at >= i:
                for p in getattr(self, name).parameters():
                    p.requires_grad = False

    def forward(self, x4):
        feats = {}
        with torch.no_grad():
            x = self.stem(x4)
        for i, name in enumerate(self.stage_names, start=2):
            if self.freeze_at >= i:
                with torch.no_grad():
                    x = getattr(self, name)(x)
            else:
                x = getattr(self, name)(x)
            feats[name] = x
        return feats

    def forward_lockstep(self, x4s):
        """several independent batches (different padded sizes allowed) through the residual stages IN LOCKSTEP: after the stem of
        every batch, the batches' rows sit back to back in one matrix, every 1x1 convolution of res2..res5 is ONE GEMM over all rows
        and every 3x3 convolution one multi-problem launch over the batches' maps (_BottleneckFn with several maps).
        Rows are independent in every layer, so each image's features are what forward() gives for its own batch.
        -> one feature dict per batch.  Needs the fused block form and freeze_at == 2; anything else runs the batches one by one."""
        if not (FUSED_BLOCKS and self.freeze_at == 2 and len(x4s) > 1):
            return [self.forward(x) for x in x4s]
        feats = [{} for _ in x4s]
        # the stems' max-pools write their outputs back to back into ONE matrix: from there on — res2 included (round 5; it ran per batch
        # and its outputs were then copied together: 93 MB, 123 us) — every block is one pass over all batches' rows
        maps = [(int(x4.shape[0]),) + self.stem.out_hw(int(x4.shape[1]), int(x4.shape[2])) for x4 in x4s]
        joint = torch.empty(sum(n * H * W for n, H, W in maps), 64, device=x4s[0].device, dtype=x4s[0].dtype)
        r0 = 0
        with torch.no_grad():
            for x4, (n, H, W) in zip(x4s, maps):
                self.stem(x4, out=joint[r0:r0 + n * H * W].view(n, H, W, 64)); r0 += n * H * W
        for name in self.stage_names:
            for blk in getattr(self, name):
                ws = (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight) + (() if blk.shortcut is None else (blk.shortcut.weight,))
                joint = _BottleneckFn.apply(joint, (blk, maps), *ws)
                if blk.conv1.stride == 2:
                    maps = [(n, (H + 1) // 2, (W + 1) // 2) for n, H, W in maps]
            for f, v in zip(feats, _SplitRowsFn.apply(joint, maps)):
                f[name] = v
        return feats


class FPN(nn.Module):
    """backbone/fpn.py:18-188: laterals 1x1, top-down nearest upsampling + add, outputs 3x3, p6 = p5 subsampled (LastLevelMaxPool)"""
    size_divisibility = 32

    def __init__(self, bottom_up):
        super().__init__()
        self.bottom_up = bottom_up
        for s, c in zip(FPN_STAGES, (256, 512, 1024, 2048)):
            self.add_module(f"fpn_lateral{s}", Conv(c, 256, 1)); self.add_module(f"fpn_output{s}", Conv(256, 256, 3))

    def forward(self, x4):
        return self.forward_top(self.bottom_up(x4))

    def forward_top(self, c):
        """the pyramid from the bottom-up features {res2..res5}"""
        prevs, prev = [], None
        for s in reversed(FPN_STAGES):
            lat = getattr(self, f"fpn_lateral{s}")(c[f"res{s}"])
            prev = lat if prev is None else _UpsampleAddFn.apply(lat, prev)
            prevs.insert(0, prev)
        # the four output convolutions are independent once the top-down sums exist: one launch (each way)
        outs = list(_conv3x3_levels([getattr(self, f"fpn_output{s}") for s in FPN_STAGES], [p.contiguous() for p in prevs]))
        outs.append(_Subsample2Fn.apply(outs[-1]))
        return outs                                                  # [p2, p3, p4, p5, p6] NHWC


class StandardRPNHead(nn.Module):
    def __init__(self, num_anchors=3):
        super().__init__()
        self.conv = Conv(256, 256, 3)
        self.objectness_logits = Conv(256, num_anchors, 1, packed=True); self.anchor_deltas = Conv(256, 4 * num_anchors, 1, packed=True)
        for m in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(m.weight, std=0.01)
        self.A = num_anchors

    def _stage_entries(self, cd):
        st, ent = _packed_linear_entries([self.objectness_logits.weight, self.anchor_deltas.weight],
                                         [self.objectness_logits.bias, self.anchor_deltas.bias], cd)
        self.__dict__["_st"] = st
        return ent

    def forward(self, feats):
        """-> (y (rows, 5A) f32, N, pixels per level): the two 1x1 convolutions of ALL levels as ONE GEMM of 5A (+ pad) columns over the
        concatenated pixels (row = level offset + image * Hi*Wi + pixel, columns [objectness a | delta 4a + b]); ops.rpn_unpack re-orders
        it into the reference's anchor order for the losses and the proposal selection"""
        A = self.A
        st = _staged_of(self)
        # (each conv output is read by the packed 1x1 GEMM only: that GEMM's data gradient leaves masked by the conv's ReLU)
        # the five conv outputs ARE the row blocks of the GEMM's operand: the convolution writes them there (the copy that packed them
        # was 34 + 84 + 34 us per iteration)
        N, C = feats[0].shape[0], self.conv.weight.shape[0]
        hw = [f.shape[1] * f.shape[2] for f in feats]
        rows = sum(N * v for v in hw)
        x = torch.empty(rows, C, device=feats[0].device, dtype=feats[0].dtype)
        ts = _conv3x3_levels([self.conv] * len(feats), [f.contiguous() for f in feats], relu=_RELU | _GRAD_PREMASKED, cat=x)
        y = _LinearFn.apply(_CatRowsFn.apply(x, *[t.reshape(N * v, C) for t, v in zip(ts, hw)]), st.w, st.bias, None, _MASK_INPUT_GRAD, True,
                            (A, 4 * A), None, self.objectness_logits.weight, self.anchor_deltas.weight, self.objectness_logits.bias,
                            self.anchor_deltas.bias)
        return y, N, tuple(hw)


class _CatRowsFn(torch.autograd.Function):
    """rows of several (r_i, C) matrices back to back in `buf` (one copy launch, sw_copy_multi); backward: row slices of the gradient"""

    @staticmethod
    def forward(ctx, buf, *parts):
        r0, pairs = 0, []
        for p in parts:
            dst = buf[r0:r0 + p.shape[0]]
            if not (p.is_contiguous() and p.data_ptr() == dst.data_ptr() and p.shape == dst.shape):      # (already written in place)
                pairs.append((p.contiguous(), dst))
            r0 += p.shape[0]
        if pairs:
            ops.copy_multi(pairs)
        ctx.rows = [p.shape[0] for p in parts]
        return buf

    @staticmethod
    def backward(ctx, g):
        out, r0 = [], 0
        for r in ctx.rows:
            out.append(g[r0:r0 + r]); r0 += r
        return (None,) + tuple(out)


class _SplitRowsFn(torch.autograd.Function):
    """(rows, C) holding several maps back to back -> one (n, H, W, C) view per map (no copy); backward: the maps' gradients copied
    back to back into one (rows, C) gradient (one sw_copy_multi launch; a map nobody used: zeros)"""

    @staticmethod
    def forward(ctx, buf, maps):
        ctx.maps, ctx.C = [tuple(m) for m in maps], buf.shape[1]
        out, r0 = [], 0
        for n, H, W in ctx.maps:
            out.append(buf[r0:r0 + n * H * W].view(n, H, W, ctx.C))
            r0 += n * H * W
        ctx.rows = r0
        return tuple(out)

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        full = torch.empty(ctx.rows, ctx.C, device=ref.device, dtype=ref.dtype)
        pairs, r0 = [], 0
        for (n, H, W), g in zip(ctx.maps, gs):
            dst = full[r0:r0 + n * H * W]
            if g is None:
                ops.fill_zero(dst)
            else:
                pairs.append((g.contiguous().view(n * H * W, ctx.C), dst))
            r0 += n * H * W
        ops.copy_multi(pairs)
        return full, None


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


class Sampler:
    """seeds of the label sampling (sampling.py:49-50 draws one torch.randperm per candidate list): one 64-bit seed per list; the
    kernels derive every candidate's random key from (seed, position in the list)"""

    def __init__(self, seed=0):
        self.seed, self.k = int(seed), 0

    def next_seed(self):
        self.k += 1
        return _splitmix64(((self.seed & 0xFFFFFFFF) << 32) ^ self.k)


class Speculation:
    """Counts the training path would otherwise read back from the device, ASSUMED and checked once per iteration.

    A semi-supervised iteration read eight counts back (per model call the proposals left by the RPN's NMS and its finite flags, per
    student batch the sampled rows, the teacher's detections, two pseudo-label counts): every read drains the queue and the GPU then
    idles until new launches arrive — 16.0 -> 14.2 ms with the reads stubbed out (and 13.1 with the teacher's pass beside the
    student's, which only pays once the reads are gone).  In training the first three kinds are constants in all but degenerate cases:
    the NMS leaves its cap (post_nms_topk) whenever the image has that many distinct candidates, the sampler fills its batch whenever
    there are that many proposals.  While a ledger is open (`with Speculation() as sp:` sets frcnn.SPECULATE) those sites assume the
    constant, keep the device tensor and go on — their padded rows are defined (zero boxes, class -1, score 0) so that a wrong
    assumption computes garbage, never touches memory it should not.  `sp.holds()` compares everything in ONE read (all ranks agree
    through a MAX all-reduce when a process group is up); the caller (semisup.SemiSupStep) then either steps the optimizer or throws
    the attempt away and repeats the iteration with the exact, reading code."""

    def __init__(self):
        self.items, self._prev = [], None

    def __enter__(self):
        global SPECULATE
        self._prev, SPECULATE = SPECULATE, self
        return self

    def __exit__(self, *exc):
        global SPECULATE
        SPECULATE = self._prev
        return False

    def expect(self, dev_ints, values):
        """dev_ints (int32, any shape) is assumed to equal `values` (flat list of ints)"""
        self.items.append((dev_ints.reshape(-1), [int(v) for v in values]))

    def seal(self):
        """Call when the forward pass is queued (every assumption is on the ledger, the backward is not yet issued): the comparison —
        and the ranks' agreement on it — runs on a stream of its own behind the forward only, its one flag lands in pinned host memory;
        holds() then waits for THAT, which has long happened while the backward still runs: the optimizer step is queued without the
        GPU ever draining.  (Read on the main stream instead, the flag waited for the whole backward: 0.5 ms of empty queue per iteration.)"""
        if not self.items or not self.items[0][0].is_cuda:
            return
        import torch.distributed as dist
        dev = self.items[0][0].device
        main = torch.cuda.current_stream(dev)
        chk = _CHECK_STREAMS.get(dev.index)
        if chk is None:
            chk = _CHECK_STREAMS[dev.index] = ops.worker_stream("chk", dev)
        chk.wait_stream(main)
        want_l = tuple(v for _, vs in self.items for v in vs)
        const = _CHECK_CONST.get(dev.index)
        if const is None or const[0] != want_l:                          # the expectation is the same list every iteration: uploaded once
            const = _CHECK_CONST[dev.index] = (want_l, torch.tensor(want_l, dtype=torch.int32).to(dev), torch.empty(1, dtype=torch.int32).pin_memory())
            torch.cuda.current_stream(dev).synchronize()
        want, host = const[1], const[2]
        with torch.cuda.stream(chk):
            got = torch.cat([t for t, _ in self.items]) if len(self.items) > 1 else self.items[0][0]
            flag = (got != want).any().to(torch.int32).reshape(1)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl":
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)               # (RCCL waits for `chk`, the current stream, only)
                self._agreed = True
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(chk)
        main.wait_stream(chk)                                             # (the ledger's tensors are not freed under the comparison)
        self._sealed = (ev, host, got, want, flag)

    def holds(self):
        import torch.distributed as dist
        sealed, agreed = getattr(self, "_sealed", None), getattr(self, "_agreed", False)
        if sealed is not None:
            sealed[0].synchronize()
            ok = int(sealed[1][0]) == 0
            if not ok and os.environ.get("SW_S3_SPEC_DEBUG"):
                print("speculation miss: have", sealed[2].tolist(), "want", sealed[3].tolist(), flush=True)
        elif not self.items:
            ok = True
        else:
            got = torch.cat([t for t, _ in self.items]) if len(self.items) > 1 else self.items[0][0]
            want = [v for _, vs in self.items for v in vs]
            have = got.tolist()
            ok = have == want
            if not ok and os.environ.get("SW_S3_SPEC_DEBUG"):
                print("speculation miss: have", have, "want", want, flush=True)
        if not agreed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = self.items[0][0].device if (self.items and dist.get_backend() == "nccl") else "cpu"
            flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            ok = int(flag.item()) == 0
        return ok


SPECULATE = None
_CHECK_STREAMS = {}
_CHECK_CONST = {}


class PseudoLabRPN(nn.Module):
    """unbias/ubteacher/modeling/proposal_generator/rpn.py:11-57 over detectron2's RPN"""

    def __init__(self, sampler, batch_size_per_image=256, positive_fraction=0.25, pre_nms_topk=(2000, 1000), post_nms_topk=(1000, 1000),
                 nms_thresh=0.7, anchor_sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0)):
        super().__init__()
        self.rpn_head = StandardRPNHead(len(aspect_ratios))
        self.sampler = sampler
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self.pre_nms_topk, self.post_nms_topk, self.nms_thresh = pre_nms_topk, post_nms_topk, nms_thresh
        self.anchor_sizes, self.aspect_ratios = anchor_sizes, aspect_ratios
        self.bbox_weights = (1.0, 1.0, 1.0, 1.0)
        self._anchor_cache = {}

    def anchors(self, grid_sizes, device):
        """anchor_generator.py:17-31,134-199 (offset 0): per level (Hi*Wi*A, 4), location-major"""
        key = (tuple(grid_sizes), str(device))
        hit = self._anchor_cache.get(key)
        if hit is None:
            res = []
            for (gh, gw), stride, size in zip(grid_sizes, STRIDES, self.anchor_sizes):
                cell = []
                for r in self.aspect_ratios:
                    w = math.sqrt(size ** 2.0 / r); h = r * w
                    cell.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
                cell = torch.tensor(cell, dtype=torch.float32, device=device)
                sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32, device=device)
                sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32, device=device)
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
                res.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4).contiguous())
            hit = self._anchor_cache[key] = (res, torch.cat(res, 0).contiguous())
        return hit

    @torch.no_grad()
    def label_and_sample_anchors(self, anchors_all, gt_boxes_list):
        """detectron2 rpn.py:305-360: IoU thresholds [0.3, 0.7], labels [0, -1, 1], low-quality matches; 256 per image, <= 25 % positive.
        One launch sequence for all images (sw_rpn_label_anchors): -> labels int8 (N, A), matched gt boxes (N, A, 4); no host sync"""
        counts = [int(g.shape[0]) for g in gt_boxes_list]
        gts = [g for g in gt_boxes_list if g.shape[0]]
        gt_cat = gts[0].contiguous() if len(gts) == 1 else (torch.cat(gts, 0).contiguous() if gts else None)
        seeds = [self.sampler.next_seed() for _ in range(2 * len(counts))]          # (positives, negatives) per image, in that order
        return ops.rpn_label_anchors(anchors_all, gt_cat, counts, seeds, self.batch_size_per_image,
                                     int(self.batch_size_per_image * self.positive_fraction))

    def _image_hw(self, image_sizes, device):
        # (a dict: one iteration of the semi-supervised step calls this with three different batches — a one-entry cache rebuilt the
        # tensor and copied it to the device every call)
        key = (tuple((int(h), int(w)) for h, w in image_sizes), str(device))
        cache = self.__dict__.setdefault("_hw_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 64:
                cache.clear()
            hit = cache[key] = torch.tensor([list(k) for k in key[0]], dtype=torch.int32).to(device)
        return hit

    @torch.no_grad()
    def predict_proposals(self, anchors, logits, deltas, image_sizes):
        """detectron2 rpn.py:478-533 + proposal_utils.py:20-130: per-level top-k (sort, stable: ties -> ascending index), decode, clip, drop
        empty boxes (sw_rpn_select_pack), NMS per level (sw_detect_postprocess2 with level = class), the best post_nms_topk per image.
        logits (N, At), deltas (N, At, 4) in anchor order.  ONE host read per call: the images' proposal counts and finite flags."""
        N = logits.shape[0]
        dev = logits.device
        pre, post = self.pre_nms_topk[0 if self.training else 1], self.post_nms_topk[0 if self.training else 1]
        L = len(anchors)
        ints = torch.empty(2 * N, device=dev, dtype=torch.int32)                          # proposal counts | finite flags
        sc, bx, _ = ops.rpn_select_pack(logits, deltas, anchors, pre, self.bbox_weights, SCALE_CLAMP, self._image_hw(image_sizes, dev),
                                        ints_out=ints[N:])
        spec = SPECULATE if self.training else None
        if spec is not None:                                             # rows beyond an image's count stay zero boxes (see Speculation)
            dboxes = torch.zeros(N, post, 4, device=dev); dscores = torch.zeros(N, post, device=dev)
        else:
            dboxes = torch.empty(N, post, 4, device=dev); dscores = torch.empty(N, post, device=dev)
        scratch = torch.empty(2, post, device=dev, dtype=torch.int32)
        for n, (h, w) in enumerate(image_sizes):
            ops.detect_postprocess(sc[n], bx[n], int(h), int(w), -3.0e38, self.nms_thresh, post,
                                   out=(ints[n:n + 1], dboxes[n], dscores[n], scratch[0], scratch[1]))
        if spec is not None:
            host = [post] * N + [-1] * N                                 # every image at the cap, everything finite (sw_rpn_select_pack
            spec.expect(ints, host)                                      # leaves the flag at 0xFFFFFFFF): checked by spec.holds()
        else:
            host = ints.tolist()
        out = []
        for n, (h, w) in enumerate(image_sizes):
            k, fin = host[n], host[N + n]
            if self.training and not fin:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
            p = Instances((int(h), int(w)))
            p.proposal_boxes = Boxes(dboxes[n, :k]); p.objectness_logits = dscores[n, :k]
            p._sw_src = (dboxes, ints, n, k)              # the (N, post, 4) block + device counts: the ROI heads sample from it directly
            out.append(p)
        return out

    def forward(self, image_sizes, feats, gt_instances=None, compute_loss=True, compute_val_loss=False):
        dev = feats[0].device
        anchors, anchors_all = self.anchors([tuple(f.shape[1:3]) for f in feats], dev)
        y, N, hw = self.rpn_head(feats)
        A = self.rpn_head.A
        logits, deltas = ops.rpn_unpack(y.detach(), N, A, hw)                              # (N, At), (N, At, 4), anchor order
        losses = {}
        if (self.training and compute_loss) or compute_val_loss:
            labels, matched = self.label_and_sample_anchors(anchors_all, [g.gt_boxes.tensor.to(dev).float() for g in gt_instances])
            l_cls, l_loc = _RpnLossFn.apply(y, logits, deltas, labels.reshape(-1), anchors_all, matched.reshape(-1, 4), self.bbox_weights,
                                            1.0 / (self.batch_size_per_image * N), (N, A, hw))
            losses = {"loss_rpn_cls": l_cls, "loss_rpn_loc": l_loc}
            self.last_labels = labels
        proposals = self.predict_proposals(anchors, logits, deltas, image_sizes)
        return proposals, losses


class FastRCNNConvFCHead(nn.Module):
    def __init__(self, d_in=256 * 7 * 7, fc_dim=1024):
        super().__init__()
        self.fc1 = nn.Linear(d_in, fc_dim); self.fc2 = nn.Linear(fc_dim, fc_dim)
        for m in (self.fc1, self.fc2):
            nn.init.kaiming_uniform_(m.weight, a=1); nn.init.constant_(m.bias, 0)

    def _stage_entries(self, cd):
        st = []
        ent = []
        for m in (self.fc1, self.fc2):
            o, i = m.weight.shape
            st.append(_Staged(w=torch.zeros(_pad8(o), i, device=m.weight.device, dtype=cd)))
            ent.append(dict(kind=0, w=m.weight, dst=st[-1].w))
        self.__dict__["_st"] = st
        return ent

    def forward(self, x, out_grad_premasked=False):
        """out_grad_premasked: the caller's only consumer of the result masks its data gradient by result > 0 (_MASK_INPUT_GRAD)"""
        s1, s2 = _staged_of(self)
        x = _LinearFn.apply(x, s1.w, self.fc1.bias.detach(), None, _RELU | _GRAD_PREMASKED, False, (self.fc1.out_features,), None,
                            self.fc1.weight, self.fc1.bias)
        f2 = _RELU | _MASK_INPUT_GRAD | (_GRAD_PREMASKED if out_grad_premasked else 0)
        return _LinearFn.apply(x, s2.w, self.fc2.bias.detach(), None, f2, False, (self.fc2.out_features,), None, self.fc2.weight, self.fc2.bias)


class FastRCNNFocaltLossOutputLayers(nn.Module):
    """unbias/ubteacher/modeling/roi_heads/fast_rcnn.py:12-40 over detectron2's FastRCNNOutputLayers"""

    def __init__(self, d_in, num_classes, test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100):
        super().__init__()
        self.cls_score = nn.Linear(d_in, num_classes + 1); self.bbox_pred = nn.Linear(d_in, 4 * num_classes)
        nn.init.normal_(self.cls_score.weight, std=0.01); nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.cls_score.bias, 0); nn.init.constant_(self.bbox_pred.bias, 0)
        self.num_classes = num_classes
        self.bbox_weights = (10.0, 10.0, 5.0, 5.0)
        self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image = test_score_thresh, test_nms_thresh, test_topk_per_image

    def _stage_entries(self, cd):
        st, ent = _packed_linear_entries([self.cls_score.weight, self.bbox_pred.weight], [self.cls_score.bias, self.bbox_pred.bias], cd)
        self.__dict__["_st"] = st
        return ent

    def forward(self, x, mask_input_grad=False):
        """-> packed f32 logits (R, 5K+1) = [cls_score | bbox_pred]: one GEMM.  mask_input_grad: x is a ReLU output that only this layer
        reads (the box head's): d/dx leaves masked by x > 0"""
        st = _staged_of(self)
        K = self.num_classes
        return _LinearFn.apply(x, st.w, st.bias, None, _MASK_INPUT_GRAD if mask_input_grad else 0, True, (K + 1, 4 * K), None, self.cls_score.weight, self.bbox_pred.weight,
                               self.cls_score.bias, self.bbox_pred.bias)


class StandardROIHeadsPseudoLab(nn.Module):
    """unbias/ubteacher/modeling/roi_heads/roi_heads.py:377-546"""

    def __init__(self, num_classes, sampler, batch_size_per_image=512, positive_fraction=0.25, proposal_append_gt=True):
        super().__init__()
        self.box_head = FastRCNNConvFCHead()
        self.box_predictor = FastRCNNFocaltLossOutputLayers(1024, num_classes)
        self.num_classes, self.sampler = num_classes, sampler
        self.batch_size_per_image, self.positive_fraction, self.proposal_append_gt = batch_size_per_image, positive_fraction, proposal_append_gt

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets, append_gt):
        """roi_heads.py:324-375 for all images in ONE launch (sw_roi_label_sample): append the gt boxes, IoU-match at 0.5 (labels [0, 1], no
        low-quality matches), sample batch_size_per_image rows with at most positive_fraction foreground by the random-key rule; no host
        round trip (the proposal counts are read on the device where the RPN's NMS left them)"""
        K, B = self.num_classes, self.batch_size_per_image
        N = len(proposals)
        dev = proposals[0].proposal_boxes.tensor.device
        src = [getattr(p, "_sw_src", None) for p in proposals]
        if all(s_ is not None for s_ in src) and all(s_[0] is src[0][0] and s_[2] == i for i, s_ in enumerate(src)) and src[0][0].shape[0] == N:
            buf, cnt_dev = src[0][0], src[0][1][:N]
        else:                                                     # proposals from somewhere else: pack them into one block
            ps = max(max(len(p) for p in proposals), 1)
            buf = torch.empty(N, ps, 4, device=dev)
            pairs = [(p.proposal_boxes.tensor.float().contiguous(), buf[i, :len(p)]) for i, p in enumerate(proposals) if len(p)]
            if pairs:
                ops.copy_multi(pairs)
            cnt_dev = torch.tensor([len(p) for p in proposals], dtype=torch.int32).to(dev)
        counts = [len(t) for t in targets]
        gts = [(t.gt_boxes.tensor.to(dev).float(), t.gt_classes.to(dev)) for t in targets if len(t)]
        if gts:
            gt_b = gts[0][0].contiguous() if len(gts) == 1 else torch.cat([g[0] for g in gts], 0).contiguous()
            gt_c = (gts[0][1] if len(gts) == 1 else torch.cat([g[1] for g in gts], 0)).to(torch.int32).contiguous()
        else:
            gt_b = gt_c = None
        seeds = [self.sampler.next_seed() for _ in range(2 * N)]                            # (foreground, background) per image
        cnt, idx, cls, both = ops.roi_label_sample(cnt_dev, buf, gt_b, gt_c, counts, seeds, append_gt, 0.5, K, B, int(B * self.positive_fraction))
        # Rows per image = num_pos + num_neg of sampling.py:36-47 = min(n_pos, max_pos) + min(n_neg, B - num_pos): LESS than min(B, candidates)
        # when an image has more than max_pos foreground candidates and fewer than B - max_pos background ones (200 fg + 100 bg -> 228,
        # not 300).  Only the kernel knows n_pos, so its count is read (one host read per call; the rows beyond it are class -1 / empty
        # boxes, never stale memory).  Without ground truth every candidate is background: the count is min(B, candidates), no read.
        need = [i for i in range(N) if counts[i] > 0]
        if SPECULATE is not None and self.training and all(len(p) >= B for p in proposals):
            got = [B] * N                                                # full batches (padded rows: class -1, zero boxes): checked later
            SPECULATE.expect(cnt, got)
        else:
            got = cnt.tolist() if need else None
        out = []
        for i, p in enumerate(proposals):
            n = int(got[i]) if got is not None else min(B, len(p))
            s_ = Instances(p.image_size)
            s_.proposal_boxes = Boxes(both[0, i, :n]); s_.gt_classes = cls[i, :n]; s_.gt_boxes = Boxes(both[1, i, :n])
            s_._sw_dense = (both, cls, i, n)
            out.append(s_)
        return out

    def _pool(self, feats, boxes_per_image):
        """poolers.py:17-50,196-250: level = floor(4 + log2(sqrt(area) / 224 + 1e-8)) clamped to [2, 5]; ROIAlign 7x7 per level.  The dense
        (image, box) rows, the levels and the per-level row lists come from one kernel (sw_roi_assign_levels), the list lengths stay on the
        device (sw_roi_align_* read them): no host round trip"""
        bs = [b if (b.dtype == torch.float32 and b.dim() == 2 and (b.shape[0] == 0 or (b.stride(1) == 1 and b.stride(0) == 4))) else
              b.float().contiguous() for b in boxes_per_image]
        base = min((b for b in bs if b.shape[0]), key=lambda t: t.data_ptr(), default=bs[0])
        rois, lv, sel, cnt = ops.roi_assign_levels(base, [b.shape[0] for b in bs], [(b.data_ptr() - base.data_ptr()) // 4 for b in bs])
        self.last_levels = lv
        return _RoIAlignFn.apply(rois, sel, cnt, [1.0 / s for s in STRIDES[:4]], *feats[:4])

    def forward(self, feats, proposals, targets=None, compute_loss=True, branch="", compute_val_loss=False):
        if self.training and compute_loss:
            proposals = self.label_and_sample_proposals(proposals, targets, self.proposal_append_gt)
        elif compute_val_loss:
            proposals = self.label_and_sample_proposals(proposals, targets, False)
        pooled = self._pool(feats, [p.proposal_boxes.tensor for p in proposals])
        logits = self.box_predictor(self.box_head(pooled, out_grad_premasked=True), mask_input_grad=True)
        K = self.num_classes
        if (self.training and compute_loss) or compute_val_loss:
            self.last_sampled, self.last_logits = proposals, logits
            R = logits.shape[0]
            if R == 0:
                z = 0.0 * logits.sum()
                return proposals, {"loss_cls": z, "loss_box_reg": z}
            dense = [getattr(p, "_sw_dense", None) for p in proposals]
            B = self.batch_size_per_image
            if all(d is not None and d[0] is dense[0][0] and d[3] == B for d in dense) and dense[0][0].shape[1] == len(proposals):
                both, cls = dense[0][0], dense[0][1]             # every image sampled a full batch: the sampler's blocks ARE the dense rows
                boxes2, gtc = both.view(2 * R, 4), cls.view(R)
            else:
                boxes2 = torch.cat([p.proposal_boxes.tensor for p in proposals] + [p.gt_boxes.tensor for p in proposals]).contiguous()
                gtc = torch.cat([p.gt_classes for p in proposals]).to(torch.int32).contiguous()
            l_cls, l_box = _RoiLossFn.apply(logits, K, gtc, boxes2, self.box_predictor.bbox_weights, 1.5)
            return proposals, {"loss_cls": l_cls, "loss_box_reg": l_box}
        # inference form (fast_rcnn.py:44-160): softmax, decode, clip, score > 0.05, per-class NMS 0.5, top 100
        pred, off = [], 0
        lg = logits.detach()
        bp = self.box_predictor
        for p in proposals:
            n = len(p)
            h, w = p.image_size
            sc = torch.empty(n, K + 1, device=lg.device); bx = torch.empty(n, 4 * K, device=lg.device)
            r = Instances((h, w))
            if n:
                ops.oicr_predict(lg[off:off + n], n, K, 1, 0, 5 * K + 1, p.proposal_boxes.tensor.float().contiguous(), bp.bbox_weights,
                                 SCALE_CLAMP, sc, bx)
                cnt, b, s, c, _ = ops.detect_postprocess(sc, bx, int(h), int(w), bp.test_score_thresh, bp.test_nms_thresh, bp.test_topk_per_image)
                if SPECULATE is not None and self.training:
                    # the teacher's weak pass inside a speculative iteration: all topk rows, the ones beyond the count with score 0 /
                    # zero box / class 0 (ops.detect_postprocess zero-fills) — the only consumer is the pseudo-label threshold
                    # (score > thres >= 0: semisup.threshold_bbox), which compacts on the device; no count is read here
                    r.pred_boxes = Boxes(b); r.scores = s; r.pred_classes = c.to(torch.int64)
                    r._sw_count = cnt
                else:
                    k = int(cnt.item())
                    r.pred_boxes = Boxes(b[:k].clone()); r.scores = s[:k].clone(); r.pred_classes = c[:k].to(torch.int64)
            else:
                r.pred_boxes = Boxes(torch.zeros(0, 4, device=lg.device)); r.scores = torch.zeros(0, device=lg.device)
                r.pred_classes = torch.zeros(0, dtype=torch.int64, device=lg.device)
            pred.append(r)
            off += n
        return pred, logits


@META_ARCH_REGISTRY.register()
class TwoStagePseudoLabGeneralizedRCNN(nn.Module):
    """unbias/ubteacher/modeling/meta_arch/rcnn.py:8-107.  `TwoStagePseudoLabGeneralizedRCNN(cfg)` (the registry call of
    rcnn_multi.build_model, selected by MODEL.META_ARCHITECTURE of unbias/configs/code_release/voc_ssod.yaml) or explicit arguments."""

    def __init__(self, cfg=None, *, num_classes=20, compute_dtype=torch.float32, freeze_at=2, sampler=None,
                 pixel_mean=(103.530, 116.280, 123.675), pixel_std=(1.0, 1.0, 1.0)):
        super().__init__()
        if cfg is not None:
            M = cfg.MODEL
            # the configuration this class implements (Base-RCNN-FPN.yaml + voc_ssod.yaml); anything else is refused, not ignored
            # (the RPN / ROI-head / test keys: _cfg_kwargs)
            assert M.BACKBONE.get("NAME", "build_resnet_fpn_backbone") == "build_resnet_fpn_backbone"
            assert M.get("PROPOSAL_GENERATOR", {}).get("NAME", "PseudoLabRPN") == "PseudoLabRPN"
            assert M.ROI_HEADS.get("NAME", "StandardROIHeadsPseudoLab") == "StandardROIHeadsPseudoLab"
            assert M.ROI_HEADS.get("LOSS", "FocalLoss") == "FocalLoss" and M.get("RPN", {}).get("LOSS", "CrossEntropy") == "CrossEntropy"
            num_classes, freeze_at = M.ROI_HEADS.NUM_CLASSES, M.BACKBONE.get("FREEZE_AT", 2)
            pixel_mean, pixel_std = M.PIXEL_MEAN, M.PIXEL_STD
            dt_name = M.get("AMD", {}).get("COMPUTE_DTYPE", "fp32")
            compute_dtype = torch.bfloat16 if dt_name == "bf16" else torch.float32
            sampler = Sampler(int(cfg.get("SEED", 0)) if int(cfg.get("SEED", 0)) >= 0 else 0)
            rpn_kw, roi_kw, pred_kw = self._cfg_kwargs(M, cfg.get("TEST", {}))
        else:
            rpn_kw, roi_kw, pred_kw = {}, {}, {}
        sampler = sampler if sampler is not None else Sampler()
        self.backbone = FPN(ResNet(freeze_at))
        self.proposal_generator = PseudoLabRPN(sampler, **rpn_kw)
        self.roi_heads = StandardROIHeadsPseudoLab(num_classes, sampler, **roi_kw)
        bp = self.roi_heads.box_predictor
        for k, v in pred_kw.items():
            setattr(bp, k, v)
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)
        self.compute_dtype = compute_dtype
        self.sampler = sampler
        self.graph_nograd_backbone = False                       # _features

    @staticmethod
    def _cfg_kwargs(M, T):
        """The detectron2 keys this detector takes from a config (detectron2/detectron2/config/defaults.py values as defaults):
        passed on where the implementation is parametric, REFUSED where it is fixed — a config that changes one of them must not
        silently train or evaluate something else."""
        def get(node, key, default):
            return node.get(key, default) if hasattr(node, "get") else default
        R, H, B, A = get(M, "RPN", {}), get(M, "ROI_HEADS", {}), get(M, "ROI_BOX_HEAD", {}), get(M, "ANCHOR_GENERATOR", {})
        rpn_kw = dict(batch_size_per_image=get(R, "BATCH_SIZE_PER_IMAGE", 256), positive_fraction=get(R, "POSITIVE_FRACTION", 0.25),
                      pre_nms_topk=(get(R, "PRE_NMS_TOPK_TRAIN", 2000), get(R, "PRE_NMS_TOPK_TEST", 1000)),
                      post_nms_topk=(get(R, "POST_NMS_TOPK_TRAIN", 1000), get(R, "POST_NMS_TOPK_TEST", 1000)),
                      nms_thresh=get(R, "NMS_THRESH", 0.7),
                      anchor_sizes=tuple(s_[0] for s_ in get(A, "SIZES", [[32], [64], [128], [256], [512]])),
                      aspect_ratios=tuple(get(A, "ASPECT_RATIOS", [[0.5, 1.0, 2.0]])[0]))
        roi_kw = dict(batch_size_per_image=get(H, "BATCH_SIZE_PER_IMAGE", 512), positive_fraction=get(H, "POSITIVE_FRACTION", 0.25),
                      proposal_append_gt=get(H, "PROPOSAL_APPEND_GT", True))
        pred_kw = dict(test_score_thresh=get(H, "SCORE_THRESH_TEST", 0.05), test_nms_thresh=get(H, "NMS_THRESH_TEST", 0.5),
                       test_topk_per_image=get(T, "DETECTIONS_PER_IMAGE", 100))
        fixed = [(get(R, "IOU_THRESHOLDS", [0.3, 0.7]), [0.3, 0.7], "RPN.IOU_THRESHOLDS"), (get(R, "IOU_LABELS", [0, -1, 1]), [0, -1, 1], "RPN.IOU_LABELS"),
                 (get(R, "SMOOTH_L1_BETA", 0.0), 0.0, "RPN.SMOOTH_L1_BETA"), (get(R, "BBOX_REG_LOSS_TYPE", "smooth_l1"), "smooth_l1", "RPN.BBOX_REG_LOSS_TYPE"),
                 (tuple(get(R, "BBOX_REG_WEIGHTS", (1.0, 1.0, 1.0, 1.0))), (1.0, 1.0, 1.0, 1.0), "RPN.BBOX_REG_WEIGHTS"),
                 (get(R, "LOSS_WEIGHT", 1.0), 1.0, "RPN.LOSS_WEIGHT"), (get(R, "BBOX_REG_LOSS_WEIGHT", 1.0), 1.0, "RPN.BBOX_REG_LOSS_WEIGHT"),
                 (get(R, "BOUNDARY_THRESH", -1), -1, "RPN.BOUNDARY_THRESH"),
                 (list(get(H, "IOU_THRESHOLDS", [0.5])), [0.5], "ROI_HEADS.IOU_THRESHOLDS"), (list(get(H, "IOU_LABELS", [0, 1])), [0, 1], "ROI_HEADS.IOU_LABELS"),
                 (get(B, "FC_DIM", 1024), 1024, "ROI_BOX_HEAD.FC_DIM"), (get(B, "NUM_FC", 2), 2, "ROI_BOX_HEAD.NUM_FC"), (get(B, "NUM_CONV", 0), 0, "ROI_BOX_HEAD.NUM_CONV"),
                 (get(B, "POOLER_RESOLUTION", 7), 7, "ROI_BOX_HEAD.POOLER_RESOLUTION"), (get(B, "POOLER_SAMPLING_RATIO", 0), 0, "ROI_BOX_HEAD.POOLER_SAMPLING_RATIO"),
                 (get(B, "POOLER_TYPE", "ROIAlignV2"), "ROIAlignV2", "ROI_BOX_HEAD.POOLER_TYPE"), (get(B, "SMOOTH_L1_BETA", 0.0), 0.0, "ROI_BOX_HEAD.SMOOTH_L1_BETA"),
                 (get(B, "BBOX_REG_LOSS_TYPE", "smooth_l1"), "smooth_l1", "ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE"),
                 (tuple(get(B, "BBOX_REG_WEIGHTS", (10.0, 10.0, 5.0, 5.0))), (10.0, 10.0, 5.0, 5.0), "ROI_BOX_HEAD.BBOX_REG_WEIGHTS"),
                 (get(B, "CLS_AGNOSTIC_BBOX_REG", False), False, "ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG"), (get(B, "TRAIN_ON_PRED_BOXES", False), False, "ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES")]
        for got, want, name in fixed:
            if isinstance(want, float):
                ok = abs(float(got) - want) < 1e-12
            else:
                ok = (list(got) == list(want)) if isinstance(want, (list, tuple)) else got == want
            assert ok, f"MODEL.{name} = {got!r} is not implemented by this detector (implemented: {want!r})"
        return rpn_kw, roi_kw, pred_kw

    def refresh_staged_weights(self):
        """the layers' compute-dtype weight copies follow the parameters: one launch when anything changed since the last call"""
        ws = self.__dict__.get("_weight_stage")
        if ws is None or not ws.valid_for(self.compute_dtype):
            ws = self.__dict__["_weight_stage"] = WeightStage(self, self.compute_dtype)
            # the captured no-grad backbone graphs (_features) replay against the OLD stage's buffers: a re-homed parameter (.to(),
            # load_state_dict(assign=True), a flat-master re-pack, another compute dtype) would leave the teacher emitting features of
            # stale weights with no error — they go with the stage they were captured on
            self.__dict__.pop("_bb_graphs", None)
        ws.refresh()

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        """-> (N, H, W, 4) normalised, padded to the size divisibility, and the image sizes"""
        ims = [x["image"].to(self.device) for x in batched_inputs]
        sizes = [tuple(int(v) for v in im.shape[1:]) for im in ims]
        d = self.backbone.size_divisibility
        H = (max(s[0] for s in sizes) + d - 1) // d * d; W = (max(s[1] for s in sizes) + d - 1) // d * d
        out = torch.empty(len(ims), H, W, 4, device=self.device, dtype=self.compute_dtype)
        key = (self.pixel_mean._version, self.pixel_std._version, self.pixel_mean.data_ptr())
        hit = self.__dict__.get("_mean_std")
        if hit is None or hit[0] != key:                                 # (a device -> host read: once, not per call)
            hit = self.__dict__["_mean_std"] = (key, self.pixel_mean.flatten().tolist(), self.pixel_std.flatten().tolist())
        mean, std = hit[1], hit[2]
        for i, im in enumerate(ims):
            ops.preprocess_pad(im.contiguous(), out[i], mean, std)
        return out, sizes

    BACKBONE_GRAPHS = 6          # input shapes whose no-grad backbone pass is kept as a hipGraph (least recently used goes)

    def _features(self, x4, allow_graph=False):
        """FPN features of a preprocessed batch.  Under torch.no_grad() (the teacher's weak pass, inference) the ~105 launches of the
        ResNet-50 + FPN are shape-static and leave no autograd state: the second time an input shape is seen they are captured, from
        then on ONE hipGraph replay into the graph's own buffers (valid until the next call with this shape; every consumer runs on
        the same stream before that).  The staged weights live in persistent buffers the stage plan rewrites in place (WeightStage), so
        a replay reads the current teacher.  Off unless `graph_nograd_backbone` is set (semisup.SemiSupStep sets it on its teacher when
        it speculates; SW_S3_BACKBONE_GRAPH=0/1 overrides): bit-equal to plain launches
        (test_teacher_on_a_side_stream_and_its_backbone_as_a_graph_give_the_same_step); it holds one activation set per input shape.
        Alone it gained nothing (15.7-16.1 vs 15.8-16.0 ms: the iteration waited at its count read-backs, not for launches) — see
        semisup.SemiSupStep for what it does once those are gone."""
        env = os.environ.get("SW_S3_BACKBONE_GRAPH")
        on = self.graph_nograd_backbone if env is None else env == "1"
        # allow_graph: only the training iteration's weak pass (forward(branch="unsup_data_weak")) — evaluation (inference()) sees many
        # padded shapes once each; with 6 graphs kept it would capture, evict and re-capture (a synchronize + an activation pool each)
        if (not allow_graph or torch.is_grad_enabled() or not on or not x4.is_cuda or ops.TIMER is not None
                or torch.cuda.is_current_stream_capturing()):
            return self.backbone(x4)
        # What the backbone reads besides the input and the staged weights (refresh_staged_weights, done by the caller) must be brought
        # up to date HERE, outside the graph, into buffers that never move: the stem's FrozenBN fold is cached per buffer epoch — a
        # capture made while the cache was warm holds no launch that recomputes it (found by a replay that followed two calls inside
        # one EMA epoch: stale fold, 2 % off in the pseudo losses; tools/diag/s3_graph_debug.py)
        self.backbone.bottom_up.stem.conv1.norm.fold()
        cache = self.__dict__.setdefault("_bb_graphs", OrderedDict())
        # what a capture bakes in besides the shape: the stage's staged buffers and the stem's weight / fold buffers
        stem = self.backbone.bottom_up.stem.conv1
        key = (tuple(x4.shape), x4.dtype, x4.device.index, id(self.__dict__.get("_weight_stage")), stem.weight.data_ptr())
        hit = cache.get(key)
        if hit is None:
            cache[key] = 1                                               # seen once: eager (caches and workspaces warm up)
            while len(cache) > 4 * self.BACKBONE_GRAPHS:
                cache.popitem(last=False)
            return self.backbone(x4)
        cache.move_to_end(key)
        if hit == 1:
            n_graphs = sum(1 for v in cache.values() if v != 1)
            for k in [k for k, v in cache.items() if v != 1][:max(0, n_graphs + 1 - self.BACKBONE_GRAPHS)]:
                del cache[k]
            static_in = torch.empty_like(x4)
            static_in.copy_(x4)
            g = torch.cuda.CUDAGraph()
            with ops.capture_guard(), torch.cuda.graph(g):
                outs = self.backbone(static_in)
            hit = cache[key] = (g, static_in, outs)
        g, static_in, outs = hit
        static_in.copy_(x4)
        g.replay()
        return outs

    def forward(self, batched_inputs, branch="supervised", given_proposals=None, val_mode=False, second=None, second_targets=None):
        """second (branch "supervised" only): a SECOND, independent batch of the same branch — the semi-supervised step's pseudo-labelled
        views next to its labelled ones (unbias/ubteacher/engine/trainer.py:527-538 calls the model twice).  Both batches' backbones run
        in lockstep (ResNet.forward_lockstep), everything image- and loss-specific per batch, first batch first (the order of the two
        calls: the label samplers draw their keys in it) -> (result of the first batch, result of the second batch).
        second_targets: a callable returning the second batch's list of Instances, called only when its heads need them — after both
        backbones and the first batch's heads are queued: the teacher that produces the pseudo labels can run beside all of that on
        another stream (semisup.SemiSupStep)."""
        if (not self.training) and (not val_mode):
            return self.inference(batched_inputs)
        self.refresh_staged_weights()
        if second is not None:
            if branch != "supervised":
                raise ValueError("second batch: branch 'supervised' only")
            results = []
            pre = [self.preprocess_image(b) for b in (batched_inputs, second)]
            cs = self.backbone.bottom_up.forward_lockstep([x4 for x4, _ in pre])
            for bi, (b, (_, sizes), c) in enumerate(zip((batched_inputs, second), pre, cs)):
                if bi == 1 and second_targets is not None:
                    gt = second_targets()
                else:
                    gt = [x["instances"] for x in b] if "instances" in b[0] else None
                feats = self.backbone.forward_top(c)
                proposals, rpn_losses = self.proposal_generator(sizes, feats, gt)
                _, det_losses = self.roi_heads(feats, proposals, gt, branch=branch)
                losses = dict(det_losses); losses.update(rpn_losses)
                results.append((losses, [], [], None))
            return tuple(results)
        x4, sizes = self.preprocess_image(batched_inputs)
        gt = [x["instances"] for x in batched_inputs] if "instances" in batched_inputs[0] else None
        feats = self._features(x4, allow_graph=(branch == "unsup_data_weak"))
        if branch == "supervised":
            proposals, rpn_losses = self.proposal_generator(sizes, feats, gt)
            _, det_losses = self.roi_heads(feats, proposals, gt, branch=branch)
            losses = dict(det_losses); losses.update(rpn_losses)
            return losses, [], [], None
        if branch == "unsup_data_weak":
            proposals, _ = self.proposal_generator(sizes, feats, None, compute_loss=False)
            dets, preds = self.roi_heads(feats, proposals, targets=None, compute_loss=False, branch=branch)
            return {}, proposals, dets, preds
        if branch == "val_loss":
            proposals, rpn_losses = self.proposal_generator(sizes, feats, gt, compute_val_loss=True)
            _, det_losses = self.roi_heads(feats, proposals, gt, branch=branch, compute_val_loss=True)
            losses = dict(det_losses); losses.update(rpn_losses)
            return losses, [], [], None
        raise ValueError(branch)

    @torch.no_grad()
    def inference(self, batched_inputs, do_postprocess=True):
        """GeneralizedRCNN.inference (detectron2/detectron2/modeling/meta_arch/rcnn.py:177-219): detections of the ROI heads, then
        `_postprocess` (:243-259) -> `detector_postprocess` (modeling/postprocessing.py:9-59): boxes rescaled from the network's
        input resolution to each input's "height" / "width" (the dataset image: MIN_SIZE_TEST resizes every test image), clipped,
        empty ones dropped.  do_postprocess=False returns the raw list[Instances] in network-input coordinates."""
        from .inference import detector_postprocess
        self.refresh_staged_weights()
        x4, sizes = self.preprocess_image(batched_inputs)
        feats = self._features(x4)
        proposals, _ = self.proposal_generator(sizes, feats, None, compute_loss=False)
        dets, _ = self.roi_heads(feats, proposals, targets=None, compute_loss=False)
        if not do_postprocess:
            return dets
        return [{"instances": detector_postprocess(d, inp.get("height", s[0]), inp.get("width", s[1]))}
                for d, inp, s in zip(dets, batched_inputs, sizes)]
