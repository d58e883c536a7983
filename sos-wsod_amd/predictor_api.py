"""Stand-alone call API of the predictor modules — `forward` / `losses` / `predict_probs` / `inference` as the reference's
`WSDDNOutputLayers` (fast_rcnn_wsddn.py:542-589,658-681) and `OICROutputLayers` (fast_rcnn_oicr.py:504-528,530-614,702-716)
expose them — for callers that drive a predictor directly instead of through OICRPlusHeads' fused training function.

Same kernels as the fused path: `_HipLinear` is an autograd node around sw_gemm (forward, data gradient, weight gradient) +
sw_colsum; the WSDDN image-level loss and the OICR weighted-CE / L1 losses are the fused loss kernels (sw_wsddn_mil,
sw_oicr_refine_loss), which emit the loss AND its logit gradient in one sweep — their autograd nodes only scale that gradient
by the incoming cotangent.  `forward` returns plain tensors like the reference.  Every returned tensor is a real autograd output
of the f32 logits (the OICR pair are column slices of them; the WSDDN scores leave `_WsddnScores`, whose backward is the analytic
gradient of softmax(dim=1) * softmax(dim=0)), so caller-side losses built from them train the layer.  The tensors also remember
their logits (attribute `_sw_logits`), which lets `losses` / `inference` run the FUSED kernels; a caller that dropped the
attribute on the way (`.detach()`, slicing, `torch.cat` over images) still gets the same values: the OICR functions re-join
`scores | deltas`, the WSDDN loss falls back to the reference's own expression on the scores.
The compute-dtype copy of a layer's weights is cached on the layer until a parameter changes (ops.param_key)."""
import torch

from . import ops


class _HipLinear(torch.autograd.Function):
    """y (N, out) f32 = x (N, in) @ W^T + b through the MFMA GEMM; explicit backward"""

    @staticmethod
    def forward(ctx, x, w, b, compute_dtype, staged=None):
        ops._need_gpu(x, w)
        N, D = x.shape
        out_f = w.shape[0]
        xs = x.detach().contiguous()
        if xs.dtype != compute_dtype:
            xs = xs.to(compute_dtype)
        ld = (out_f + 7) // 8 * 8                                            # 16-byte K pieces for the data-gradient GEMM
        if staged is not None and staged[0] is not None:
            ws = staged[0]                                                   # the layer's cached copy (still current)
        else:
            ws = torch.zeros(ld, D, device=x.device, dtype=compute_dtype)    # rows beyond out_f stay zero
            ops.convert_2d(w.detach().float().contiguous(), ws, out_f, D)
            if staged is not None:
                staged[0] = ws
        y = torch.empty(N, ld, device=x.device, dtype=torch.float32)[:, :out_f]
        ops.gemm(xs, ws, y, N, out_f, D, ep=ops.make_epilogue(bias=None if b is None else b.detach().float().contiguous(),
                                                              out_dtype=torch.float32))
        ctx.save_for_backward(xs, ws)
        ctx.has_bias, ctx.x_dtype, ctx.needs = b is not None, x.dtype, ctx.needs_input_grad
        return y

    @staticmethod
    def backward(ctx, g):
        xs, ws = ctx.saved_tensors
        N, D = xs.shape
        out_f = g.shape[1]
        cd = xs.dtype
        ld = ws.shape[0]
        gs = torch.zeros(N, ld, device=g.device, dtype=cd)                   # 16-byte row pitch for the K-strided operand
        gs[:, :out_f] = g
        dx = dw = db = None
        if ctx.needs[0]:
            dx = torch.empty(N, D, device=g.device, dtype=torch.float32)
            ops.gemm(gs, ws, dx, N, D, ld, b_kstrided=True)
            if ctx.x_dtype != torch.float32:
                dx = dx.to(ctx.x_dtype)
        if ctx.needs[1]:
            dwp = torch.empty(ld, D, device=g.device, dtype=torch.float32)
            ops.gemm(gs, xs, dwp, ld, D, N, a_kstrided=True, b_kstrided=True)
            dw = dwp[:out_f]
        if ctx.has_bias and ctx.needs[2]:
            dbp = torch.empty(ld, device=g.device, dtype=torch.float32)
            ops.colsum(gs, N, ld, dbp)
            db = dbp[:out_f]
        return dx, dw, db, None, None


class _LossFromUnitGrad(torch.autograd.Function):
    """loss whose logit gradient was produced by the fused loss kernel in the forward sweep"""

    @staticmethod
    def forward(ctx, logits, loss, unit_grad):
        ctx.save_for_backward(unit_grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (ug,) = ctx.saved_tensors
        return ug * g, None, None


def _splits(proposals, n):
    return [n] if proposals is None else [len(p) for p in proposals]


def _linear_cached(layer, names, x, compute_dtype):
    """x @ cat(weights)^T + cat(biases) with the compute-dtype operand cached on the layer while no parameter changed"""
    ws_ = [getattr(layer, n).weight for n in names]
    bs_ = [getattr(layer, n).bias for n in names]
    key = (tuple(ops.param_key(p) for p in ws_), compute_dtype, x.device)
    hit = layer.__dict__.get("_api_stage")
    box = [hit[1] if (hit is not None and hit[0] == key) else None]
    y = _HipLinear.apply(x, torch.cat(ws_, 0), torch.cat(bs_, 0), compute_dtype, box)
    layer.__dict__["_api_stage"] = (key, box[0])
    return y


class _WsddnScores(torch.autograd.Function):
    """scores = softmax(C, dim=1) * softmax(D, dim=0) per image (fast_rcnn_wsddn.py:564-567): forward by the MIL kernel, backward the
    analytic gradient — with A = softmax_k(C), B = softmax_r(D) and g the cotangent:
    dC = A * (gB - sum_k A gB),  dD = B * (gA - sum_r B gA)."""

    @staticmethod
    def forward(ctx, logits, K, sizes):
        N = logits.shape[0]
        scores = torch.empty(N, K, device=logits.device, dtype=torch.float32)
        zeros_oh = torch.zeros(K, device=logits.device)
        lg, off = logits.detach(), 0
        for n in sizes:
            lv = torch.empty(1, device=logits.device)
            ops.wsddn_mil(lg[off:off + n], 1, n, K, 0, K, zeros_oh, scores[off:off + n].view(1, n, K), lv)
            off += n
        ctx.save_for_backward(lg)
        ctx.K, ctx.sizes = K, tuple(sizes)
        return scores

    @staticmethod
    def backward(ctx, g):
        (lg,) = ctx.saved_tensors
        K = ctx.K
        d = ops.fill_zero(torch.empty_like(lg))
        g = g.contiguous().float()
        off = 0
        for n in ctx.sizes:                                  # per image: the softmax over proposals spans one image's rows
            if n:
                ops.wsddn_scores_bwd(lg[off:off + n], K, g[off:off + n], d[off:off + n])
            off += n
        return d, None, None


# ------------------------------------------------------------------------------------------------ WSDDN predictor
def wsddn_forward(layer, x, proposals=None, compute_dtype=torch.float32):
    """fast_rcnn_wsddn.py:542-589: scores = softmax(cls(x), dim=1) * softmax(det(x), dim=0) per image; zero deltas"""
    if x.dim() > 2:
        x = torch.flatten(x, start_dim=1)
    K = layer.num_classes
    logits = _linear_cached(layer, ("cls", "det"), x, compute_dtype)          # (N, 2K): [cls | det]
    N = logits.shape[0]
    scores = _WsddnScores.apply(logits, K, _splits(proposals, N))
    scores._sw_logits = logits
    deltas = torch.zeros(N, 4 * K, device=x.device, dtype=torch.float32)      # :579-585
    return scores, deltas


def wsddn_losses(layer, predictions, proposals, gt_classes_img_oh):
    """fast_rcnn_wsddn.py:658-681 -> {"loss_cls": BCE(clamp(sum_r scores), onehot, mean over K) / N_img} (MEAN_LOSS True)"""
    scores, _ = predictions
    logits = getattr(scores, "_sw_logits", None)
    K = layer.num_classes
    if logits is None:
        # the scores reached us without their logits (detached / sliced / re-concatenated by the caller): the reference's own
        # expression (:340-375) on the scores — differentiable through _WsddnScores when they still carry a graph
        total, off = 0.0, 0
        sizes = _splits(proposals, scores.shape[0])
        for i, n in enumerate(sizes):
            p = scores[off:off + n].sum(0).clamp(1e-6, 1 - 1e-6)
            total = total + torch.nn.functional.binary_cross_entropy(p, gt_classes_img_oh[i].float().to(p.device), reduction="mean")
            off += n
        return {"loss_cls": total / len(sizes) * layer.loss_weight.get("loss_cls", 1.0)}
    N = logits.shape[0]
    sizes = _splits(proposals, N)
    dev = logits.device
    ones = torch.ones(2, device=dev)
    unit = torch.zeros(N, logits.shape[1], device=dev)
    total = torch.zeros((), device=dev)
    off = 0
    sc = torch.empty(N, K, device=dev)
    for i, n in enumerate(sizes):
        lv = torch.empty(1, device=dev)
        ops.wsddn_mil(logits.detach()[off:off + n], 1, n, K, 0, K, gt_classes_img_oh[i].float().contiguous(), sc[off:off + n].view(1, n, K), lv,
                      unit[off:off + n], ones)
        total = total + lv[0]
        off += n
    n_img = len(sizes)
    loss = _LossFromUnitGrad.apply(logits, total / n_img, unit / n_img)
    return {"loss_cls": loss * layer.loss_weight.get("loss_cls", 1.0)}


# ------------------------------------------------------------------------------------------------ OICR refinement predictor
def oicr_forward(layer, x, compute_dtype=torch.float32):
    """fast_rcnn_oicr.py:504-528 -> (scores (N, K+1) logits, proposal_deltas (N, 4K))"""
    if x.dim() > 2:
        x = torch.flatten(x, start_dim=1)
    K = layer.num_classes
    logits = _linear_cached(layer, ("cls_score", "bbox_pred"), x, compute_dtype)    # (N, 5K+1): [cls_score | bbox_pred]
    scores, deltas = logits[:, :K + 1], logits[:, K + 1:]
    scores._sw_logits = logits
    return scores, deltas


def _oicr_logits(predictions):
    """the (N, 5K+1) logits behind a (scores, deltas) pair: remembered by forward, or re-joined (differentiably) if a caller
    handed in tensors that lost the attribute"""
    scores, deltas = predictions
    logits = getattr(scores, "_sw_logits", None)
    return logits if logits is not None else torch.cat([scores, deltas], 1).float().contiguous()


def oicr_losses(layer, predictions, proposals):
    """fast_rcnn_oicr.py:530-554 (OICROutputs :157-352): proposals carry proposal_boxes, gt_boxes, gt_classes, gt_weights.
    loss_cls = mean_r(CE(ignore -1) * w), loss_box_reg = sum_fg L1(deltas[gt class] - get_deltas(proposal, gt_box)) / R"""
    logits = _oicr_logits(predictions)
    K = layer.num_classes
    dev = logits.device
    N = logits.shape[0]
    pb = torch.cat([p.proposal_boxes.tensor for p in proposals], 0).float()
    gb = torch.cat([p.gt_boxes.tensor for p in proposals], 0).float().to(dev)
    cls = torch.cat([p.gt_classes for p in proposals], 0).to(device=dev, dtype=torch.int32).contiguous()
    wts = torch.cat([p.gt_weights for p in proposals], 0).to(device=dev, dtype=torch.float32).contiguous()
    # the kernel takes its regression target from boxes[lab_index]: put the gt boxes behind the proposals
    boxes = torch.cat([pb.to(dev), gb], 0).contiguous()
    idx = (torch.arange(N, device=dev, dtype=torch.int32) + N).contiguous()
    lv = torch.empty(1, 2, 1, device=dev)
    unit = torch.zeros(N, logits.shape[1], device=dev)
    ops.oicr_refine_loss(logits.detach(), 1, N, K, 0, K + 1, boxes, cls.view(1, N), wts.view(1, N), idx.view(1, N),
                         torch.zeros(1, dtype=torch.int32, device=dev), layer.bbox_reg_weights, lv, unit, torch.ones(2, device=dev))
    # one kernel, two losses: the class columns carry d loss_cls, the box columns d loss_box_reg
    m_cls = torch.zeros(logits.shape[1], device=dev); m_cls[:K + 1] = 1
    loss_cls = _LossFromUnitGrad.apply(logits, lv[0, 0, 0], unit * m_cls)
    loss_box = _LossFromUnitGrad.apply(logits, lv[0, 1, 0], unit * (1 - m_cls))
    return {"loss_cls": loss_cls * layer.loss_weight.get("loss_cls", 1.0),
            "loss_box_reg": loss_box * layer.loss_weight.get("loss_box_reg", 1.0)}


def oicr_predict_probs(layer, predictions, proposals):
    """fast_rcnn_oicr.py:702-716: row softmax, split per image"""
    scores, _ = predictions
    K = layer.num_classes
    N = scores.shape[0]
    logits = getattr(scores, "_sw_logits", None)
    src = (logits if logits is not None else scores).detach().float()
    out = torch.empty(1, N, K + 1, device=scores.device)
    ops.oicr_mean_probs(src, 1, N, K, 1, 0, 0, out)
    return out[0].split(_splits(proposals, N), dim=0)


def oicr_inference(layer, predictions, proposals):
    """fast_rcnn_oicr.py:584-614 -> (list[Instances], list[kept proposal rows]) : softmax scores, decoded + clipped boxes,
    score threshold, per-class NMS, top-k (fast_rcnn_inference :46-148)"""
    from .structures import Boxes, Instances
    K = layer.num_classes
    logits = _oicr_logits(predictions).detach()
    results, kept = [], []
    off = 0
    import math
    for p in proposals:
        n = len(p)
        lg = logits[off:off + n]
        all_scores = torch.empty(n, K + 1, device=lg.device); all_boxes = torch.empty(n, 4 * K, device=lg.device)
        ops.oicr_predict(lg, n, K, 1, 0, 5 * K + 1, p.proposal_boxes.tensor.float().contiguous(), layer.bbox_reg_weights,
                         math.log(1000.0 / 16), all_scores, all_boxes)
        h, w = p.image_size
        cnt, b, s, c, rows = ops.detect_postprocess(all_scores, all_boxes, h, w, layer.test_score_thresh, layer.test_nms_thresh,
                                                    layer.test_topk_per_image)
        k = int(cnt.item())
        r = Instances((h, w)); r.pred_boxes = Boxes(b[:k]); r.scores = s[:k]; r.pred_classes = c[:k].to(torch.int64)
        results.append(r); kept.append(rows[:k].to(torch.int64))
        off += n
    return results, kept
