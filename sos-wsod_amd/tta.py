"""Test-time augmentation with score / box averaging (SURVEY §8f row 2;
uwsod/projects/WSL/wsl/modeling/test_time_augmentation_avg.py:199-427, boxes only — the path has no mask head).

For every augmented view (resize to a TEST.AUG min size, optional horizontal flip) the detector runs WITHOUT its
post-processing and returns the per-proposal score matrix (R, K+1) and class-wise boxes (R, 4K) in the view's
coordinates; the boxes are mapped back through the inverse transforms (un-flip in the resized frame, then un-resize), both
are averaged over the views (the proposals are index aligned across views), and the averaged set goes through the usual
threshold / per-class NMS / top-k (`sw_detect_postprocess`).

The reference builds the views on the CPU (`DatasetMapperTTAAVG`: PIL resize, numpy flip, `transform_proposals`).
`DeviceTTAMapper` builds them on the device: the resize is the HIP restatement of Pillow's 8-bit bilinear resampling
(`resize.resize_bilinear_u8`, bit-identical pixels, the flipped view from the same launch) and the proposal boxes go through
the same affine maps (fvcore's ResizeTransform / HFlipTransform `apply_box`, third-party, restated in `ViewTransform`).
"""
from typing import List, Tuple

import torch

from .resize import resize_bilinear_u8
from .structures import Boxes, Instances


class ViewTransform:
    """[crop: shift by -(x0, y0) into a window of size orig_hw (fvcore CropTransform.apply_coords),] resize (orig h,w -> new h,w),
    then an optional horizontal flip in the resized frame; every step in the boxes' own dtype, as numpy does it"""

    def __init__(self, orig_hw: Tuple[int, int], new_hw: Tuple[int, int], flip: bool, crop_xy=None):
        self.orig_hw, self.new_hw, self.flip = tuple(orig_hw), tuple(new_hw), bool(flip)
        self.crop_xy = None if crop_xy is None else (int(crop_xy[0]), int(crop_xy[1]))

    def apply_box(self, boxes: torch.Tensor) -> torch.Tensor:
        sx, sy = self.new_hw[1] / self.orig_hw[1], self.new_hw[0] / self.orig_hw[0]
        if self.crop_xy is not None:
            x0, y0 = self.crop_xy
            boxes = _offset_xyxy(boxes, -float(x0), -float(y0))
        b = _scale_xyxy(boxes, sx, sy)
        if self.flip:                                        # HFlipTransform.apply_box: x -> W - x, corners re-sorted
            w = float(self.new_hw[1])
            b = torch.stack([w - b[:, 2], b[:, 1], w - b[:, 0], b[:, 3]], 1)
        return b

    def inverse_box(self, boxes: torch.Tensor) -> torch.Tensor:
        assert self.crop_xy is None, "the test-time views are not cropped"
        b = boxes
        if self.flip:
            w = float(self.new_hw[1])
            b = torch.stack([w - b[:, 2], b[:, 1], w - b[:, 0], b[:, 3]], 1)
        sx, sy = self.orig_hw[1] / self.new_hw[1], self.orig_hw[0] / self.new_hw[0]
        return _scale_xyxy(b, sx, sy)


def _scale_xyxy(b: torch.Tensor, sx: float, sy: float) -> torch.Tensor:
    """b * [sx, sy, sx, sy] without building that 4-vector on the device: `b.new_tensor([...])` is a pageable host-to-device copy that
    waits for the stream — 24 of them were 13.8 of the 20 ms of a 12-view TTA call (tools/diag/infer_host_profile.py).  The scalar
    form multiplies by the same float32 values: identical bits."""
    out = b.clone()
    out[:, 0::2].mul_(sx)
    out[:, 1::2].mul_(sy)
    return out


def _offset_xyxy(b: torch.Tensor, dx: float, dy: float) -> torch.Tensor:
    out = b.clone()
    out[:, 0::2].add_(dx)
    out[:, 1::2].add_(dy)
    return out


class DeviceTTAMapper:
    """dataset dict -> list of (augmented dict, ViewTransform) for TEST.AUG.{MIN_SIZES, MAX_SIZE, FLIP}"""

    def __init__(self, min_sizes=(480, 576, 688, 864, 1000, 1200), max_size=4000, flip=True, proposal_topk=None, min_box_size=0):
        self.min_sizes, self.max_size, self.flip = tuple(min_sizes), max_size, flip
        self.proposal_topk, self.min_box_size = proposal_topk, min_box_size

    @staticmethod
    def _shortest_edge(h, w, size, max_size):                # ResizeShortestEdge.get_output_shape
        scale = size * 1.0 / min(h, w)
        nh, nw = (size, scale * w) if h < w else (scale * h, size)
        if max(nh, nw) > max_size:
            s = max_size * 1.0 / max(nh, nw)
            nh, nw = nh * s, nw * s
        return int(nh + 0.5), int(nw + 0.5)

    def __call__(self, d):
        img = d["image"]                                     # (3, h, w) uint8
        h, w = img.shape[-2:]
        prop = d["proposals"]
        out = []
        for size in self.min_sizes:
            nh, nw = self._shortest_edge(h, w, size, self.max_size)
            r, r_flip = resize_bilinear_u8(img, (nh, nw), with_flip=True)
            for flip in ((False, True) if self.flip else (False,)):
                t = ViewTransform((h, w), (nh, nw), flip)
                # transform_proposals (test_time_augmentation_avg.py:29-71): map, clip to the view, drop empty boxes, top-k.
                # (The reference filters per view and later stacks the views' (R, 4K) outputs: views that drop different
                # proposals fail there; here too the merge requires equal counts.)
                b = t.apply_box(prop.proposal_boxes.tensor.float())
                b = torch.stack([b[:, 0].clamp(0, nw), b[:, 1].clamp(0, nh), b[:, 2].clamp(0, nw), b[:, 3].clamp(0, nh)], 1)
                keep = ((b[:, 2] - b[:, 0]) > self.min_box_size) & ((b[:, 3] - b[:, 1]) > self.min_box_size)
                logits = prop.objectness_logits
                if not bool(keep.all()):
                    b, logits = b[keep], logits[keep]
                p = Instances((nh, nw))
                p.proposal_boxes = Boxes(b[:self.proposal_topk])
                p.objectness_logits = logits[:self.proposal_topk]
                view = {"image": r_flip if flip else r, "proposals": p, "height": d.get("height", h),
                        "width": d.get("width", w)}
                out.append((view, t))
        return out


class GeneralizedRCNNWithTTAAVG(torch.nn.Module):
    """same call interface as the model's inference forward; `tta_mapper(dict) -> [(view dict, ViewTransform)]`"""

    def __init__(self, model, tta_mapper=None, batch_views=True):
        super().__init__()
        if isinstance(model, torch.nn.parallel.DistributedDataParallel):
            model = model.module
        self.model = model
        self.tta_mapper = tta_mapper if tta_mapper is not None else DeviceTTAMapper()
        self.batch_views = bool(batch_views)

    @torch.no_grad()
    def __call__(self, batched_inputs: List[dict]):
        return [self._inference_one_image(x) for x in batched_inputs]

    def _inference_one_image(self, inp):
        from . import ops
        assert not self.model.training
        h, w = inp["image"].shape[-2:]
        orig = (inp.get("height", h), inp.get("width", w))
        views = self.tta_mapper(dict(inp, height=orig[0], width=orig[1]))
        heads = self.model.roi_heads
        K = heads.num_classes
        sum_scores = sum_boxes = None
        heads.test_scores_only = True             # per-view NMS / top-k (and their host sync) would be thrown away
        try:
            # consecutive views of one size (a scale and its flip) run as one batch: the batch-1 launches of a 375 x 500 image's
            # views fill half the chip (12 views: 26.6 -> see tools/infer_bench.py); the per-view results enter the sums in the
            # mapper's order either way
            per_view = []
            i = 0
            while i < len(views):
                j = i + 1
                while (self.batch_views and j < len(views) and j - i < 2
                       and views[j][0]["image"].shape == views[i][0]["image"].shape):
                    j += 1
                if j - i > 1 and hasattr(self.model, "view_scores"):
                    per_view += self.model.view_scores([v for v, _ in views[i:j]])
                else:
                    for v, _ in views[i:j]:
                        _, sc, bx = self.model.inference([v], do_postprocess=False)       # (1, R, K+1), (1, R, 4K), view coordinates
                        per_view.append((sc[0], bx[0]))
                i = j
            for (view, tfm), (scores, boxes) in zip(views, per_view):
                R = boxes.shape[0]
                back = tfm.inverse_box(boxes.reshape(R * K, 4)).reshape(R, 4 * K)
                if (tfm.orig_hw != orig):                        # the mapper resized from the tensor's size, not the dataset's
                    sx, sy = orig[1] / tfm.orig_hw[1], orig[0] / tfm.orig_hw[0]
                    back = _scale_xyxy(back.reshape(R * K, 4), sx, sy).reshape(R, 4 * K)
                sum_scores = scores.clone() if sum_scores is None else sum_scores + scores
                sum_boxes = back if sum_boxes is None else sum_boxes + back
        finally:
            heads.test_scores_only = False
        n = float(len(views))
        all_scores, all_boxes = (sum_scores / n).contiguous(), (sum_boxes / n).contiguous()
        self.last_avg = (all_scores, all_boxes)                    # tests: the view-averaged matrices that enter the merge
        cnt, dboxes, dscores, dclasses, _ = ops.detect_postprocess(all_scores, all_boxes, int(orig[0]), int(orig[1]),
                                                                   heads.test_score_thresh, heads.test_nms_thresh,
                                                                   heads.test_topk_per_image)
        n_det = int(cnt.item())
        res = Instances((int(orig[0]), int(orig[1])))
        res.pred_boxes = Boxes(dboxes[:n_det])
        res.scores = dscores[:n_det]
        res.pred_classes = dclasses[:n_det].to(torch.int64)
        return {"instances": res}
