"""Inference of the OICR+ heads: the reference's `OICRPlusHeads._forward_box_test`
(roi_heads_oicrplus.py:432-475) -> `OICROutputLayers.inference` with the K refinement branches averaged
(fast_rcnn_oicr.py:584-614, 674-735) -> `fast_rcnn_inference_single_image` (:86-148).
ROIPool -> x(objectness+1) -> fc6/fc7 (no dropout) -> the packed predictor GEMM -> sw_oicr_predict -> sw_detect_postprocess;
the only host synchronisation is reading the detection count to size the returned Instances."""
import math

import torch

from . import ops
from .structures import Boxes, Instances

SCALE_CLAMP = math.log(1000.0 / 16)      # box_regression.py:12


@torch.no_grad()
def oicr_inference(heads, features, proposals):
    dt_ = heads.compute_dtype
    f = features[heads.box_in_features[-1]]
    dev = f.device
    feat = f.permute(0, 2, 3, 1).contiguous()
    if feat.dtype != dt_:
        feat = feat.to(dt_)
    assert len(proposals) == feat.shape[0] == 1, "one image per call (rcnn_multi.py:148)"
    prop = proposals[0]
    K = heads.num_classes
    boxes = prop.proposal_boxes.tensor.to(device=dev, dtype=torch.float32).contiguous()
    obj = prop.objectness_logits.to(device=dev, dtype=torch.float32).contiguous()
    R = boxes.shape[0]
    P = heads.box_pooler.output_size
    C = feat.shape[3]
    rois = torch.cat([torch.zeros(R, 1, device=dev), boxes], 1).contiguous()
    pooled = torch.empty(R, C * P * P, device=dev, dtype=dt_)
    argmax = torch.empty(R, C * P * P, device=dev, dtype=ops.roi_argmax_dtype(feat.shape[1], feat.shape[2]))
    ops.roi_pool_fwd(feat, rois, pooled, argmax, heads.box_pooler.scale, P, P, row_scale=obj, row_scale_add=1.0)
    h = heads.box_head(pooled)                                        # eval: no dropout
    params = [p.detach() for p in heads._flat_params()]
    Wh, bh = heads._pack_head_weights(params, dev)
    LD = heads.ld_head
    logits = torch.empty(R, LD, device=dev, dtype=torch.float32)
    ops.gemm(h, Wh, logits, R, LD, h.shape[1], ep=ops.make_epilogue(bias=bh, out_dtype=torch.float32))
    all_scores = torch.empty(R, K + 1, device=dev, dtype=torch.float32)
    all_boxes = torch.empty(R, 4 * K, device=dev, dtype=torch.float32)
    ops.oicr_predict(logits, R, K, heads.refine_K, 2 * K, 5 * K + 1, boxes, heads.bbox_reg_weights, SCALE_CLAMP, all_scores,
                     all_boxes)
    if getattr(heads, "test_scores_only", False):       # TTA: the per-view scores / boxes are averaged, per-view detections unused
        return [None], all_scores.unsqueeze(0), all_boxes.unsqueeze(0)
    H, W = prop.image_size
    cnt, dboxes, dscores, dclasses, drows = ops.detect_postprocess(all_scores, all_boxes, H, W, heads.test_score_thresh,
                                                                   heads.test_nms_thresh, heads.test_topk_per_image)
    n = int(cnt.item())
    result = Instances((H, W))
    result.pred_boxes = Boxes(dboxes[:n])
    result.scores = dscores[:n]
    result.pred_classes = dclasses[:n].to(torch.int64)
    return [result], all_scores.unsqueeze(0), all_boxes.unsqueeze(0)


@torch.no_grad()
def oicr_view_scores(heads, feat_nhwc, proposals):
    """The scores-only inference of several views of ONE size in one pass (the TTA wrapper's flip pairs): feat_nhwc (N, H, W, C),
    proposals: N Instances.  Same arithmetic per view as oicr_inference — ROIPool with the objectness prior, fc6 / fc7, the packed
    predictor GEMM, sw_oicr_predict — on stacked rows; -> [(all_scores (R_i, K+1), all_boxes (R_i, 4K))] per view."""
    dt_ = heads.compute_dtype
    dev = feat_nhwc.device
    feat = feat_nhwc.contiguous()
    if feat.dtype != dt_:
        feat = feat.to(dt_)
    assert len(proposals) == feat.shape[0]
    K = heads.num_classes
    boxes = [p.proposal_boxes.tensor.to(device=dev, dtype=torch.float32) for p in proposals]
    obj = torch.cat([p.objectness_logits.to(device=dev, dtype=torch.float32) for p in proposals]).contiguous()
    counts = [b.shape[0] for b in boxes]
    allb = torch.cat(boxes, 0).contiguous()
    R = allb.shape[0]
    P = heads.box_pooler.output_size
    C = feat.shape[3]
    idx = torch.cat([torch.full((n, 1), float(i), device=dev) for i, n in enumerate(counts)], 0)
    rois = torch.cat([idx, allb], 1).contiguous()
    pooled = torch.empty(R, C * P * P, device=dev, dtype=dt_)
    argmax = torch.empty(R, C * P * P, device=dev, dtype=ops.roi_argmax_dtype(feat.shape[1], feat.shape[2]))
    ops.roi_pool_fwd(feat, rois, pooled, argmax, heads.box_pooler.scale, P, P, row_scale=obj, row_scale_add=1.0)
    h = heads.box_head(pooled)
    params = [p.detach() for p in heads._flat_params()]
    Wh, bh = heads._pack_head_weights(params, dev)
    LD = heads.ld_head
    logits = torch.empty(R, LD, device=dev, dtype=torch.float32)
    ops.gemm(h, Wh, logits, R, LD, h.shape[1], ep=ops.make_epilogue(bias=bh, out_dtype=torch.float32))
    all_scores = torch.empty(R, K + 1, device=dev, dtype=torch.float32)
    all_boxes = torch.empty(R, 4 * K, device=dev, dtype=torch.float32)
    ops.oicr_predict(logits, R, K, heads.refine_K, 2 * K, 5 * K + 1, allb, heads.bbox_reg_weights, SCALE_CLAMP, all_scores, all_boxes)
    out, r0 = [], 0
    for n in counts:
        out.append((all_scores[r0:r0 + n], all_boxes[r0:r0 + n])); r0 += n
    return out


def detector_postprocess(results, output_height, output_width):
    """modeling/postprocessing.py:9-44 (boxes only): scale the detections from the network's input resolution to the
    requested output resolution, clip, drop empty boxes."""
    from .structures import Instances
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((int(output_height), int(output_width)), **results.get_fields())
    if out.has("pred_boxes"):
        boxes = out.pred_boxes.clone()
        boxes.scale(scale_x, scale_y)
        boxes.clip(out.image_size)
        out.pred_boxes = boxes
        out = out[boxes.nonempty()]
    return out


class VOCDetectionWriter:
    """The detection-result wire format of the reference's evaluator (evaluation/pascal_voc_evaluation.py:57-118), which
    Stage 2 (`tools/pgf.py`) consumes: per detection the text line `image_id score xmin+1 ymin+1 xmax ymax` (score %.3f,
    coordinates %.1f) grouped by class, and the JSON list of
    {"image_id": int, "category_id": class+1, "score": float, "bbox": [x1, y1, x2, y2]} parsed back from those lines."""

    def __init__(self, num_classes):
        self.num_classes = num_classes
        self.reset()

    def reset(self):
        self._predictions = {c: [] for c in range(self.num_classes)}

    def process(self, inputs, outputs):
        for inp, outp in zip(inputs, outputs):
            image_id = inp["image_id"]
            inst = outp["instances"].to("cpu")
            boxes = inst.pred_boxes.tensor.numpy()
            for box, score, cls in zip(boxes, inst.scores.tolist(), inst.pred_classes.tolist()):
                xmin, ymin, xmax, ymax = box
                xmin += 1                                       # the inverse of the VOC loader's 0-based shift
                ymin += 1
                self._predictions[cls].append(f"{image_id} {score:.3f} {xmin:.1f} {ymin:.1f} {xmax:.1f} {ymax:.1f}")

    def lines(self):
        return self._predictions

    def records(self):
        out = []
        for cls_id in range(self.num_classes):
            for line in self._predictions[cls_id]:
                m = line.split(" ")
                out.append({"image_id": int(m[0]), "category_id": cls_id + 1, "score": float(m[1]),
                            "bbox": [float(m[2]), float(m[3]), float(m[4]), float(m[5])]})
        return out

    def dump(self, path):
        import json
        with open(path, "w") as f:
            json.dump(self.records(), f)
