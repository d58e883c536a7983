"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the Stage-3 detector (BASELINE config #5): the ResNet-50-FPN Faster R-CNN of the
Unbiased-Teacher step — `TwoStagePseudoLabGeneralizedRCNN` over `build_resnet_fpn_backbone`, `PseudoLabRPN`,
`StandardROIHeadsPseudoLab` with the focal classification loss.  Only tests/ may import this file, as the checker.

Paths: U/ = /root/reference/unbias/ubteacher/,  D2/ = /root/reference/detectron2/detectron2/ (the second, v0.4 tree).
Dense contractions (conv, linear) use torch-CPU ops as the reference itself does; everything with integer outputs (anchor
matching, sampling, top-k order, NMS keep lists, level assignment) is numpy; ROIAlign is roialign_oracle.c.

PARITY PINNING: pinned against tests/golden/stage3_*.npz, written by tests/golden/make_stage3_golden.py, which loads the
reference's own modeling files from both trees through a shim and RUNS them.  Third-party arithmetic absent from
/root/reference is restated: torchvision.ops.roi_align (-> roialign_oracle.c, the arithmetic stated in-tree at
uwsod/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp), torchvision nms / batched_nms, fvcore smooth_l1_loss.
torch.randperm (D2/modeling/sampling.py:49-50) is replaced, on both sides, by a closed-form permutation (`perm`).

Tie rule where torch leaves it open (`sort(descending=True)` of the RPN logits): equal scores -> ascending index."""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import detgen
from . import oicr_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
PIXEL_MEAN = (103.530, 116.280, 123.675)       # D2/config/defaults.py:38 (BGR)
PIXEL_STD = (1.0, 1.0, 1.0)
SIZE_DIVISIBILITY = 32                         # D2/modeling/backbone/fpn.py:112 (the last stride)
R50 = [("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2)]
FPN_STAGES = (2, 3, 4, 5)
ANCHOR_SIZES = (32, 64, 128, 256, 512)         # U/../configs/Base-RCNN-FPN.yaml:9-10, one per level p2..p6
ASPECT_RATIOS = (0.5, 1.0, 2.0)
STRIDES = (4, 8, 16, 32, 64)
RPN_BBOX_WEIGHTS = (1.0, 1.0, 1.0, 1.0)        # D2/config/defaults.py:220
ROI_BBOX_WEIGHTS = (10.0, 10.0, 5.0, 5.0)      # :294
BN_EPS = 1e-5                                  # D2/layers/batch_norm.py:40


# --------------------------------------------------------------------------- C ROIAlign
_lib = None


def _roialign_lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_roialign.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ci = ctypes.c_int
        _lib.oracle_roi_align_fwd.argtypes = [fp, ctypes.c_float, ci, ci, ci, ci, ci, ci, fp, ci, fp]
        _lib.oracle_roi_align_bwd.argtypes = [fp, ctypes.c_float, ci, ci, ci, ci, ci, ci, fp, ci, fp]
    return _lib


def roi_align_fwd(feat, rois, scale, ph=7, pw=7, sampling_ratio=0):
    """feat (N,C,H,W) f32, rois (R,5) -> (R,C,ph,pw).  roialign_oracle.c"""
    feat = np.ascontiguousarray(feat, np.float32); rois = np.ascontiguousarray(rois, np.float32)
    n, c, h, w = feat.shape
    out = np.zeros((rois.shape[0], c, ph, pw), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    _roialign_lib().oracle_roi_align_fwd(feat.ctypes.data_as(fp), ctypes.c_float(scale), c, h, w, ph, pw, sampling_ratio,
                                         rois.ctypes.data_as(fp), rois.shape[0], out.ctypes.data_as(fp))
    return out


def roi_align_bwd(gout, rois, scale, feat_shape, sampling_ratio=0):
    gout = np.ascontiguousarray(gout, np.float32); rois = np.ascontiguousarray(rois, np.float32)
    n, c, h, w = feat_shape
    gi = np.zeros(feat_shape, np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    _roialign_lib().oracle_roi_align_bwd(gout.ctypes.data_as(fp), ctypes.c_float(scale), c, h, w, gout.shape[2], gout.shape[3],
                                         sampling_ratio, rois.ctypes.data_as(fp), rois.shape[0], gi.ctypes.data_as(fp))
    return gi


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, scale):
        ctx.save_for_backward(rois)
        ctx.scale, ctx.shape = scale, tuple(feat.shape)
        return torch.from_numpy(roi_align_fwd(feat.detach().numpy(), rois.numpy(), scale))

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        return torch.from_numpy(roi_align_bwd(g.numpy(), rois.numpy(), ctx.scale, ctx.shape)), None, None


# --------------------------------------------------------------------------- closed-form parameters / inputs / permutations
def make_params(K=20, tag="fr", head_scale=1.0, fc_dim=1024):
    """state-dict names and shapes of the reference model (D2 v0.4 GeneralizedRCNN + FPN + StandardRPNHead +
    FastRCNNConvFCHead + FastRCNNOutputLayers), values from detgen with the reference's init statistics; the FrozenBN
    statistics are spread so that the affine fold matters (weight 0.5..1.5, var 0.5..1.5)."""
    p = {}

    def conv_bn(name, cout, cin, k):
        std = math.sqrt(2.0 / (cout * k * k))                       # c2_msra_fill (fan_out)
        p[name + ".weight"] = detgen.normal(tag + name + ".weight", (cout, cin, k, k), std=std)
        # gains that keep the activations O(1) through 16 residual blocks with random weights (a trained net's BN does that):
        # the stem sees +-128 pixel values, a block's last conv and its shortcut add up
        gain = 0.02 if "stem" in name else 0.35 if name.endswith("conv3") else 0.6 if name.endswith("shortcut") else 1.0
        p[name + ".norm.weight"] = detgen.uniform(tag + name + ".norm.weight", (cout,), 0.5, 1.5) * np.float32(gain)
        p[name + ".norm.bias"] = detgen.normal(tag + name + ".norm.bias", (cout,), std=0.1)
        p[name + ".norm.running_mean"] = detgen.normal(tag + name + ".norm.running_mean", (cout,), std=0.1)
        p[name + ".norm.running_var"] = detgen.uniform(tag + name + ".norm.running_var", (cout,), 0.5, 1.5)
    conv_bn("backbone.bottom_up.stem.conv1", 64, 3, 7)
    cin = 64
    for stage, nblk, mid, cout, _ in R50:
        for b in range(nblk):
            pre = f"backbone.bottom_up.{stage}.{b}"
            if b == 0:
                conv_bn(pre + ".shortcut", cout, cin, 1)
            conv_bn(pre + ".conv1", mid, cin, 1)
            conv_bn(pre + ".conv2", mid, mid, 3)
            conv_bn(pre + ".conv3", cout, mid, 1)
            cin = cout
    for s, c in zip(FPN_STAGES, (256, 512, 1024, 2048)):             # c2_xavier_fill: kaiming_uniform(a=1)
        for nm, k, ci in ((f"backbone.fpn_lateral{s}", 1, c), (f"backbone.fpn_output{s}", 3, 256)):
            lim = math.sqrt(3.0 / (ci * k * k))
            p[nm + ".weight"] = detgen.uniform(tag + nm + ".weight", (256, ci, k, k), -lim, lim)
            p[nm + ".bias"] = detgen.normal(tag + nm + ".bias", (256,), std=0.01)
    A = len(ASPECT_RATIOS)
    for nm, co, k in (("conv", 256, 3), ("objectness_logits", A, 1), ("anchor_deltas", 4 * A, 1)):
        name = "proposal_generator.rpn_head." + nm
        p[name + ".weight"] = detgen.normal(tag + name + ".weight", (co, 256, k, k), std=0.01 * (head_scale if nm != "conv" else 1.0))
        p[name + ".bias"] = detgen.normal(tag + name + ".bias", (co,), std=0.01)
    d_in = 256 * 7 * 7
    for i in (1, 2):
        name = f"roi_heads.box_head.fc{i}"
        lim = math.sqrt(3.0 / d_in)
        p[name + ".weight"] = detgen.uniform(tag + name + ".weight", (fc_dim, d_in), -lim, lim)
        p[name + ".bias"] = detgen.normal(tag + name + ".bias", (fc_dim,), std=0.01)
        d_in = fc_dim
    name = "roi_heads.box_predictor"
    p[name + ".cls_score.weight"] = detgen.normal(tag + name + ".cls_score.weight", (K + 1, d_in), std=0.01 * head_scale)
    p[name + ".cls_score.bias"] = detgen.normal(tag + name + ".cls_score.bias", (K + 1,), std=0.01)
    p[name + ".bbox_pred.weight"] = detgen.normal(tag + name + ".bbox_pred.weight", (4 * K, d_in), std=0.001)
    p[name + ".bbox_pred.bias"] = detgen.normal(tag + name + ".bbox_pred.bias", (4 * K,), std=0.001)
    return p


def make_image(h, w, tag):
    return np.floor(detgen.uniform(tag + "img", (3, h, w), 0, 256)).clip(0, 255).astype(np.uint8)


def make_gt(h, w, n, K, tag):
    """n ground-truth boxes inside an (h, w) image, sides 24 .. ~2/3 of the image, integer classes"""
    x1 = detgen.uniform(tag + "x1", (n,), 0, max(w - 40, 1)); y1 = detgen.uniform(tag + "y1", (n,), 0, max(h - 40, 1))
    bw = 24 + detgen.uniform(tag + "bw", (n,)) * np.maximum(0.66 * w - 24, 0); bh = 24 + detgen.uniform(tag + "bh", (n,)) * np.maximum(0.66 * h - 24, 0)
    boxes = np.stack([x1, y1, np.minimum(x1 + bw, w), np.minimum(y1 + bh, h)], 1).astype(np.float32)
    return boxes, detgen.randint(tag + "cls", (n,), 0, K)


class Perm:
    """closed-form replacement of torch.randperm(n) (D2/modeling/sampling.py:49-50): the k-th call returns the argsort (stable,
    ascending index on ties) of detgen.uniform(f"{tag}perm{k}", (n,))"""

    def __init__(self, tag):
        self.tag, self.k = tag, 0

    def priorities(self, n):
        u = detgen.uniform(f"{self.tag}perm{self.k}", (max(n, 1),))[:n]
        self.k += 1
        return u

    def __call__(self, n):
        return np.argsort(self.priorities(n), kind="stable")


# --------------------------------------------------------------------------- backbone
def _bn_fold(P, name):
    scale = P[name + ".norm.weight"] * torch.rsqrt(P[name + ".norm.running_var"] + BN_EPS)     # D2/layers/batch_norm.py:52-58
    return scale, P[name + ".norm.bias"] - P[name + ".norm.running_mean"] * scale


def _conv_bn(x, P, name, stride=1, padding=0):
    y = F.conv2d(x, P[name + ".weight"], None, stride=stride, padding=padding)
    s, b = _bn_fold(P, name)
    return y * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def resnet_fpn_forward(x, P, collect=None):
    """D2/modeling/backbone/resnet.py:355-359 (BasicStem), :195-213 (BottleneckBlock, STRIDE_IN_1X1 True: the stride sits in
    conv1 and the shortcut), :445-466 (ResNet.forward); D2/modeling/backbone/fpn.py:115-155 (top-down pathway, nearest
    upsampling, fuse sum), :188-189 (LastLevelMaxPool: p6 = p5 subsampled by 2).  -> [p2, p3, p4, p5, p6]"""
    x = F.relu(_conv_bn(x, P, "backbone.bottom_up.stem.conv1", stride=2, padding=3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = {}
    for stage, nblk, mid, cout, stride in R50:
        for b in range(nblk):
            pre = f"backbone.bottom_up.{stage}.{b}"
            s = stride if b == 0 else 1
            out = F.relu(_conv_bn(x, P, pre + ".conv1", stride=s))
            out = F.relu(_conv_bn(out, P, pre + ".conv2", padding=1))
            out = _conv_bn(out, P, pre + ".conv3")
            sc = _conv_bn(x, P, pre + ".shortcut", stride=s) if b == 0 else x
            x = F.relu(out + sc)
        feats[stage] = x
        if collect is not None:
            collect[stage] = x
    results, prev = [], None
    for s in reversed(FPN_STAGES):
        lat = F.conv2d(feats[f"res{s}"], P[f"backbone.fpn_lateral{s}.weight"], P[f"backbone.fpn_lateral{s}.bias"])
        prev = lat if prev is None else lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        results.insert(0, F.conv2d(prev, P[f"backbone.fpn_output{s}.weight"], P[f"backbone.fpn_output{s}.bias"], padding=1))
    results.append(F.max_pool2d(results[-1], kernel_size=1, stride=2, padding=0))
    return results


def preprocess(images_u8):
    """D2/modeling/meta_arch/rcnn.py:220-228 + D2/structures/image_list.py:60-124: (x - mean) / std per image, zero padded at the
    bottom / right to the batch maximum rounded up to the size divisibility.  -> (tensor (N,3,H,W), [(h, w)])"""
    mean = torch.tensor(PIXEL_MEAN, dtype=torch.float32).view(3, 1, 1); std = torch.tensor(PIXEL_STD, dtype=torch.float32).view(3, 1, 1)
    ims = [(torch.from_numpy(np.asarray(im)).to(torch.float32) - mean) / std for im in images_u8]
    sizes = [tuple(im.shape[1:]) for im in ims]
    H = (max(s[0] for s in sizes) + SIZE_DIVISIBILITY - 1) // SIZE_DIVISIBILITY * SIZE_DIVISIBILITY
    W = (max(s[1] for s in sizes) + SIZE_DIVISIBILITY - 1) // SIZE_DIVISIBILITY * SIZE_DIVISIBILITY
    out = torch.zeros(len(ims), 3, H, W)
    for i, im in enumerate(ims):
        out[i, :, :im.shape[1], :im.shape[2]] = im
    return out, sizes


# --------------------------------------------------------------------------- anchors, RPN
def cell_anchors(size):
    """D2/modeling/anchor_generator.py:164-199 (XYXY, centred at 0)"""
    out = []
    area = size ** 2.0
    for r in ASPECT_RATIOS:
        w = math.sqrt(area / r); h = r * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return np.asarray(out, np.float32)            # torch.tensor(list of python floats) -> float32


def grid_anchors(grid_sizes):
    """D2/modeling/anchor_generator.py:17-31,134-148 (offset 0): per level (Hi*Wi*A, 4), location-major, anchor-minor"""
    res = []
    for (gh, gw), stride, size in zip(grid_sizes, STRIDES, ANCHOR_SIZES):
        sx = np.arange(0, gw * stride, stride, dtype=np.float32); sy = np.arange(0, gh * stride, stride, dtype=np.float32)
        yy, xx = np.meshgrid(sy, sx, indexing="ij")
        shifts = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], 1)
        res.append((shifts[:, None, :] + cell_anchors(size)[None, :, :]).reshape(-1, 4).astype(np.float32))
    return res


def rpn_head(feats, P):
    """D2/modeling/proposal_generator/rpn.py:130-177 + U/modeling/proposal_generator/rpn.py:28-43: shared 3x3 conv + ReLU, 1x1
    objectness (A) and deltas (4A); -> per level logits (N, Hi*Wi*A), deltas (N, Hi*Wi*A, 4)"""
    logits, deltas = [], []
    pre = "proposal_generator.rpn_head."
    for f in feats:
        t = F.relu(F.conv2d(f, P[pre + "conv.weight"], P[pre + "conv.bias"], padding=1))
        lg = F.conv2d(t, P[pre + "objectness_logits.weight"], P[pre + "objectness_logits.bias"])
        dl = F.conv2d(t, P[pre + "anchor_deltas.weight"], P[pre + "anchor_deltas.bias"])
        n, a, h, w = lg.shape
        logits.append(lg.permute(0, 2, 3, 1).flatten(1))
        deltas.append(dl.view(n, a, 4, h, w).permute(0, 3, 4, 1, 2).flatten(1, -2))
    return logits, deltas


def matcher(iou, thresholds, labels, allow_low_quality_matches):
    """D2/modeling/matcher.py:60-126 (thresholds given WITHOUT the -inf / +inf ends).  iou (M, N) -> matches (N,), labels (N,)"""
    if iou.size == 0:
        return np.zeros(iou.shape[1], np.int64), np.full(iou.shape[1], labels[0], np.int8)
    return O.matcher(iou, thresholds=tuple(thresholds), labels=tuple(labels), allow_low_quality_matches=allow_low_quality_matches)


def subsample_labels(labels, num_samples, positive_fraction, bg_label, perm):
    """D2/modeling/sampling.py:8-54 with torch.randperm -> perm"""
    positive = np.nonzero((labels != -1) & (labels != bg_label))[0]
    negative = np.nonzero(labels == bg_label)[0]
    num_pos = min(positive.size, int(num_samples * positive_fraction))
    num_neg = min(negative.size, num_samples - num_pos)
    p1 = perm(positive.size)[:num_pos]
    p2 = perm(negative.size)[:num_neg]
    return positive[p1], negative[p2]


def rpn_label_and_sample(anchors_all, gt_boxes_list, perm, batch_size=256, positive_fraction=0.25):
    """D2/modeling/proposal_generator/rpn.py:305-360 (IOU_THRESHOLDS [0.3, 0.7], labels [0, -1, 1], low-quality matches on;
    U/../configs/code_release/voc_ssod.yaml:11 POSITIVE_FRACTION 0.25)"""
    gt_labels, matched = [], []
    for gtb in gt_boxes_list:
        iou = O.pairwise_iou(gtb, anchors_all)
        m, lab = matcher(iou, (0.3, 0.7), (0, -1, 1), True)
        lab = lab.astype(np.int64)
        pos, neg = subsample_labels(lab, batch_size, positive_fraction, 0, perm)
        out = np.full(lab.shape, -1, np.int64); out[pos] = 1; out[neg] = 0
        gt_labels.append(out)
        matched.append(gtb[m] if len(gtb) else np.zeros_like(anchors_all))
    return gt_labels, matched


def rpn_losses(anchors_all, logits, deltas, gt_labels, matched_gt, batch_size=256):
    """D2/modeling/proposal_generator/rpn.py:362-420, D2/modeling/box_regression.py:229-260: L1 (beta 0) on the positives, BCE with
    logits on the sampled anchors, both summed and divided by batch_size_per_image * num_images"""
    n_img = len(gt_labels)
    lab = torch.from_numpy(np.stack(gt_labels))
    pos = lab == 1
    lg = torch.cat(logits, 1); dl = torch.cat(deltas, 1)
    tgt = torch.stack([O.get_deltas(torch.from_numpy(anchors_all), torch.from_numpy(np.ascontiguousarray(m)), RPN_BBOX_WEIGHTS) for m in matched_gt])
    loc = torch.abs(dl[pos] - tgt[pos]).sum()
    valid = lab >= 0
    obj = F.binary_cross_entropy_with_logits(lg[valid], lab[valid].to(torch.float32), reduction="sum")
    norm = batch_size * n_img
    return {"loss_rpn_cls": obj / norm, "loss_rpn_loc": loc / norm}


def _sort_desc_stable(v):
    return np.argsort(-v.astype(np.float64), kind="stable")


def batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.boxes.batched_nms (third party, restated): boxes of different groups are moved apart by idx * (max + 1), then
    plain greedy NMS in descending score order; returns kept indices, score-descending"""
    if len(boxes) == 0:
        return np.zeros(0, np.int64)
    off = idxs.astype(np.float32) * np.float32(boxes.max() + np.float32(1))
    return np.asarray(O.nms_keep((boxes + off[:, None]).astype(np.float32), scores, thr), np.int64)


def find_top_rpn_proposals(proposals, logits, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk, min_box_size=0.0):
    """D2/modeling/proposal_generator/proposal_utils.py:20-130.  proposals / logits: per level (N, Hi*Wi*A, 4) / (N, Hi*Wi*A) numpy"""
    topk_scores, topk_props, level_ids = [], [], []
    n_img = len(image_sizes)
    for lvl, (p, lg) in enumerate(zip(proposals, logits)):
        k = min(lg.shape[1], pre_nms_topk)
        idx = np.stack([_sort_desc_stable(lg[n])[:k] for n in range(n_img)])
        topk_scores.append(np.take_along_axis(lg, idx, 1)); topk_props.append(np.take_along_axis(p, idx[:, :, None], 1))
        level_ids.append(np.full(k, lvl, np.int64))
    topk_scores = np.concatenate(topk_scores, 1); topk_props = np.concatenate(topk_props, 1); level_ids = np.concatenate(level_ids)
    out = []
    for n, (h, w) in enumerate(image_sizes):
        boxes, sc, lvl = topk_props[n].copy(), topk_scores[n], level_ids
        valid = np.isfinite(boxes).all(1) & np.isfinite(sc)
        boxes, sc, lvl = boxes[valid], sc[valid], lvl[valid]
        boxes[:, 0::2] = boxes[:, 0::2].clip(0, w); boxes[:, 1::2] = boxes[:, 1::2].clip(0, h)
        keep = ((boxes[:, 2] - boxes[:, 0]) > min_box_size) & ((boxes[:, 3] - boxes[:, 1]) > min_box_size)
        boxes, sc, lvl = boxes[keep], sc[keep], lvl[keep]
        k = batched_nms(boxes, sc, lvl, nms_thresh)[:post_nms_topk]
        out.append(dict(boxes=boxes[k], logits=sc[k]))
    return out


def rpn_proposals(anchors, logits, deltas, image_sizes, training):
    """D2/modeling/proposal_generator/rpn.py:478-533: decode with the RPN weights (scale clamp log(1000/16)), top-k per level
    (Base-RCNN-FPN.yaml:13-19: 2000 train / 1000 test), NMS 0.7 per level, 1000 kept per image"""
    props = []
    for a, d in zip(anchors, deltas):
        n = d.shape[0]
        at = torch.from_numpy(a)[None].expand(n, -1, -1).reshape(-1, 4)
        props.append(O.apply_deltas(d.detach().reshape(-1, 4), at, RPN_BBOX_WEIGHTS).view(n, -1, 4).numpy())
    return find_top_rpn_proposals(props, [l.detach().numpy() for l in logits], image_sizes, 0.7, 2000 if training else 1000, 1000)


# --------------------------------------------------------------------------- ROI heads
GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))       # D2/modeling/proposal_generator/proposal_utils.py:170


def roi_label_and_sample(proposals, gts, K, perm, batch_size=512, positive_fraction=0.25, append_gt=True):
    """U/modeling/roi_heads/roi_heads.py:496-546 + D2/modeling/roi_heads/roi_heads.py:175-215 (_sample_proposals): append the gt
    boxes, IoU-match at 0.5 (no low-quality matches), sample 512 with at most 25 % foreground.  gts: [(boxes, classes)]"""
    out = []
    for prop, (gtb, gtc) in zip(proposals, gts):
        boxes = np.concatenate([prop["boxes"], gtb], 0) if append_gt else prop["boxes"]
        has_gt = len(gtb) > 0
        iou = O.pairwise_iou(gtb, boxes) if has_gt else np.zeros((0, len(boxes)), np.float32)
        m, lab = matcher(iou, (0.5,), (0, 1), False)
        if has_gt:
            cls = np.asarray(gtc, np.int64)[m].copy()
            cls[lab == 0] = K
            cls[lab == -1] = -1
        else:
            cls = np.full(len(boxes), K, np.int64)
        fg, bg = subsample_labels(cls, batch_size, positive_fraction, K, perm)
        idx = np.concatenate([fg, bg])
        out.append(dict(boxes=boxes[idx], gt_classes=cls[idx], gt_boxes=(gtb[m[idx]] if has_gt else np.zeros((len(idx), 4), np.float32)),
                        sampled_idx=idx))
    return out


def assign_levels(boxes, min_level=2, max_level=5, canonical_box_size=224, canonical_level=4):
    """D2/modeling/poolers.py:17-50 (float32 arithmetic as torch computes it)"""
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).astype(np.float32)
    sizes = np.sqrt(area)
    lv = np.floor(np.float32(canonical_level) + np.log2(sizes / np.float32(canonical_box_size) + np.float32(1e-8)))
    return np.clip(lv, min_level, max_level).astype(np.int64) - min_level


def roi_pool_multi(feats, boxes_per_image):
    """D2/modeling/poolers.py:196-250: level by Eqn. (1) of the FPN paper, ROIAlign(7, scale 1/stride, sampling 0, aligned)"""
    rois = np.concatenate([np.concatenate([np.full((len(b), 1), i, np.float32), b], 1) for i, b in enumerate(boxes_per_image)], 0)
    lv = assign_levels(rois[:, 1:])
    out = torch.zeros(len(rois), feats[0].shape[1], 7, 7)
    for l in range(4):
        inds = np.nonzero(lv == l)[0]
        if len(inds):
            out = out.index_put((torch.from_numpy(inds),), _RoIAlignFn.apply(feats[l], torch.from_numpy(rois[inds]), 1.0 / STRIDES[l]))
    return out, lv


def box_head(x, P):
    """D2/modeling/roi_heads/box_head.py:82-95 (FastRCNNConvFCHead, NUM_FC 2): flatten, fc1 + ReLU, fc2 + ReLU;
    D2/modeling/roi_heads/fast_rcnn.py:441-455 (cls_score K+1, bbox_pred 4K)"""
    x = x.flatten(1)
    for i in (1, 2):
        x = F.relu(F.linear(x, P[f"roi_heads.box_head.fc{i}.weight"], P[f"roi_heads.box_head.fc{i}.bias"]))
    pre = "roi_heads.box_predictor."
    return F.linear(x, P[pre + "cls_score.weight"], P[pre + "cls_score.bias"]), F.linear(x, P[pre + "bbox_pred.weight"], P[pre + "bbox_pred.bias"])


def roi_losses(scores, deltas, sampled, K, gamma=1.5):
    """U/modeling/roi_heads/fast_rcnn.py:73-105 (focal loss, gamma 1.5, sum / number of sampled proposals) and
    D2/modeling/roi_heads/fast_rcnn.py:245-317 (L1 on the gt class's 4 columns of the foreground rows, / number of sampled)"""
    gtc = torch.from_numpy(np.concatenate([s["gt_classes"] for s in sampled]))
    if gtc.numel() == 0:
        return {"loss_cls": 0.0 * scores.sum(), "loss_box_reg": 0.0 * deltas.sum()}
    ce = F.cross_entropy(scores, gtc, reduction="none")
    p = torch.exp(-ce)
    loss_cls = ((1 - p) ** gamma * ce).sum() / gtc.shape[0]
    props = torch.from_numpy(np.concatenate([s["boxes"] for s in sampled])); gtb = torch.from_numpy(np.concatenate([s["gt_boxes"] for s in sampled]))
    fg = torch.nonzero((gtc >= 0) & (gtc < K)).flatten()
    cols = 4 * gtc[fg][:, None] + torch.arange(4)
    tgt = O.get_deltas(props, gtb, ROI_BBOX_WEIGHTS)
    loss_box = torch.abs(deltas[fg[:, None], cols] - tgt[fg]).sum() / gtc.numel()
    return {"loss_cls": loss_cls, "loss_box_reg": loss_box}


def fast_rcnn_inference(scores, deltas, proposals, image_sizes, K, score_thresh=0.05, nms_thresh=0.5, topk=100):
    """D2/modeling/roi_heads/fast_rcnn.py:44-160: softmax, decode (10,10,5,5), clip, score > thresh, per-class NMS, top-k"""
    probs = F.softmax(scores.detach(), -1)
    out, off = [], 0
    for prop, (h, w) in zip(proposals, image_sizes):
        n = len(prop["boxes"])
        pb = O.apply_deltas(deltas.detach()[off:off + n], torch.from_numpy(prop["boxes"]), ROI_BBOX_WEIGHTS).numpy().reshape(n, K, 4).copy()
        sc = probs[off:off + n, :-1].numpy()
        off += n
        pb[..., 0::2] = pb[..., 0::2].clip(0, w); pb[..., 1::2] = pb[..., 1::2].clip(0, h)
        r, c = np.nonzero(sc > np.float32(score_thresh))
        b, s = pb[r, c], sc[r, c]
        keep = batched_nms(b, s, c, nms_thresh)[:topk]
        out.append(dict(pred_boxes=b[keep], scores=s[keep], pred_classes=c[keep]))
    return out


# --------------------------------------------------------------------------- whole branches
def _tensors(P, want_grads):
    return {k: torch.from_numpy(np.asarray(v, np.float32)).clone().requires_grad_(want_grads and ".norm." not in k) for k, v in P.items()}


def supervised_forward(P, images_u8, gts, K, perm, want_grads=False):
    """branch == "supervised" of U/modeling/meta_arch/rcnn.py:8-40: backbone, RPN (losses + proposals), ROI heads (losses).
    gts: [(boxes (G,4) f32, classes (G,) int)] per image.  -> (losses dict of floats, aux, grads or None)"""
    Pt = _tensors(P, want_grads)
    x, sizes = preprocess(images_u8)
    feats = resnet_fpn_forward(x, Pt)
    anchors = grid_anchors([tuple(f.shape[-2:]) for f in feats])
    logits, deltas = rpn_head(feats, Pt)
    anchors_all = np.concatenate(anchors, 0)
    gt_labels, matched = rpn_label_and_sample(anchors_all, [g[0] for g in gts], perm)
    losses = rpn_losses(anchors_all, logits, deltas, gt_labels, matched)
    proposals = rpn_proposals(anchors, logits, deltas, sizes, training=True)
    sampled = roi_label_and_sample(proposals, gts, K, perm)
    pooled, levels = roi_pool_multi(feats[:4], [s["boxes"] for s in sampled])
    scores, bdeltas = box_head(pooled, Pt)
    losses.update(roi_losses(scores, bdeltas, sampled, K))
    aux = dict(feats=[f.detach().numpy() for f in feats], rpn_logits=[l.detach().numpy() for l in logits],
               rpn_deltas=[d.detach().numpy() for d in deltas], anchors=anchors, rpn_labels=gt_labels, proposals=proposals,
               sampled=sampled, levels=levels, pooled=pooled.detach().numpy(), scores=scores.detach().numpy(),
               box_deltas=bdeltas.detach().numpy(), image_sizes=sizes)
    grads = None
    if want_grads:
        sum(losses.values()).backward()
        grads = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in Pt.items()}
    return {k: float(v.detach()) for k, v in losses.items()}, aux, grads


def weak_forward(P, images_u8, K):
    """branch == "unsup_data_weak" (U/modeling/meta_arch/rcnn.py:43-86): RPN proposals without losses (the module stays in
    TRAINING mode in the reference, so the train top-k counts apply), ROI heads in inference form -> detections"""
    Pt = _tensors(P, False)
    with torch.no_grad():
        x, sizes = preprocess(images_u8)
        feats = resnet_fpn_forward(x, Pt)
        anchors = grid_anchors([tuple(f.shape[-2:]) for f in feats])
        logits, deltas = rpn_head(feats, Pt)
        proposals = rpn_proposals(anchors, logits, deltas, sizes, training=True)
        pooled, _ = roi_pool_multi(feats[:4], [p["boxes"] for p in proposals])
        scores, bdeltas = box_head(pooled, Pt)
        dets = fast_rcnn_inference(scores, bdeltas, proposals, sizes, K)
    return proposals, dets
