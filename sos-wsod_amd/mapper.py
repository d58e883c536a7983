"""Device-side multi-view training mapper (the reference's `DatasetMapperMultiInput`,
uwsod/detectron2/data/dataset_mapper.py:272-425, for the recipe without RandomCrop).

One dataset dict (decoded image + precomputed proposals + image-level annotations) becomes the four index-aligned views the
detector trains on: two scales drawn from INPUT.MIN_SIZE_TRAIN (the second forced to differ from the first), each with its
horizontal flip — `image1`, `image1_flip`, `image2`, `image2_flip`, the matching `proposals*` and `instances*`.

The box side is the reference's rule for rule (`transform_proposals_multi`, detection_utils.py:208-260): every view maps the
SAME proposal list through resize (+ flip), clips it, and computes a keep mask (first occurrence of each rounded-corner hash,
`Boxes.unique_boxes` boxes.py:214-226, AND non-empty); nothing is filtered per view — the four masks are ANDed and applied to
all four sets, so row i is the same proposal in every view (the consistency losses need that).  All of it runs as tensor ops
on the image's device, no host round trip.

The pixel side: the reference resizes with PIL bilinear on the CPU (`ResizeTransform.apply_image`); here the same 8-bit
two-pass arithmetic runs as a HIP kernel (`resize.resize_bilinear_u8`, `sw_resize_pass_u8`), bit-identical to Pillow
(tests/golden/resize_*.npz), the flipped view written by the same launch.  There is no CPU pixel path: a host-resident image
raises unless the mapper was built with `resize_pixels=False` (box / label side only: host logic tests).  Decoding the image
file and reading the proposal pickle stay with the caller.
"""
import sys
from typing import Optional, Sequence

import numpy as np
import torch

from .resize import resize_bilinear_u8
from .structures import Boxes, Instances
from .tta import DeviceTTAMapper, ViewTransform


def unique_boxes_mask(boxes: torch.Tensor) -> torch.Tensor:
    """bool mask of the rows `Boxes.unique_boxes()` would return: hash = sum(round(corner) * (1, 1e3, 1e6, 1e9)), the first
    row of every distinct hash"""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.bool, device=boxes.device)
    r = torch.round(boxes.float()).to(torch.int64)                      # half-to-even like np.round
    h = r[:, 0] + r[:, 1] * 1000 + r[:, 2] * 1000000 + r[:, 3] * 1000000000
    u, inv = torch.unique(h, return_inverse=True)
    first = torch.full((u.shape[0],), n, dtype=torch.int64, device=boxes.device)
    first.scatter_reduce_(0, inv, torch.arange(n, device=boxes.device), reduce="amin")
    mask = torch.zeros(n, dtype=torch.bool, device=boxes.device)
    mask[first] = True
    return mask


def transform_proposals_multi(boxes: torch.Tensor, tfm: ViewTransform, min_box_size: float = 0.0):
    """-> (clipped boxes in the view's frame, keep mask) — detection_utils.py:208-260 without the top-k slice"""
    b = tfm.apply_box(boxes)
    h, w = tfm.new_hw
    b = torch.stack([b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)], 1)
    keep = unique_boxes_mask(b) & ((b[:, 2] - b[:, 0]) > min_box_size) & ((b[:, 3] - b[:, 1]) > min_box_size)
    return b, keep


class DeviceMultiInputMapper:
    """`mapper(dataset_dict) -> dict` with the keys the detector's training forward reads (rcnn_multi.py).

    dataset_dict: "image" (3, h, w) uint8 tensor (any device), "proposal_boxes" (N, 4) in "proposal_bbox_mode" (XYXY_ABS = 0
    default, XYWH_ABS = 1; `proposals.load_proposals_into_dataset` fills all three), "proposal_objectness_logits" (N,), optional
    "annotations" = list of {"bbox": XYXY_ABS, "category_id": int, "iscrowd": 0/1}.
    """

    def __init__(self, min_sizes: Sequence[int] = (480, 576, 688, 864, 1000, 1200), max_size: int = 2000,
                 proposal_topk: Optional[int] = 2000, min_box_size: float = 0.0, seed: Optional[int] = None,
                 resize_pixels: bool = True):
        assert len(min_sizes) >= 2, "two different scales are drawn per image (dataset_mapper.py:305-321)"
        self.min_sizes, self.max_size = tuple(min_sizes), max_size
        self.proposal_topk, self.min_box_size = proposal_topk, min_box_size
        self.rng = np.random.RandomState(seed)
        self.resize_pixels = resize_pixels

    def _draw_shapes(self, h, w):
        s1 = int(self.rng.choice(self.min_sizes))
        hw1 = DeviceTTAMapper._shortest_edge(h, w, s1, self.max_size)
        rest = [s for s in self.min_sizes if s != min(hw1)]
        for _ in range(64):                                 # the reference loops until the shapes differ
            # the second ResizeShortestEdge is rebuilt WITHOUT a max size (dataset_mapper.py:311-313: sys.maxsize default)
            hw2 = DeviceTTAMapper._shortest_edge(h, w, int(self.rng.choice(rest)), sys.maxsize)
            if hw2 != hw1:
                return hw1, hw2
        raise RuntimeError(f"no second scale of {self.min_sizes} gives a shape different from {hw1} "
                           f"(the reference would loop forever here)")

    def _resize_with_flip(self, img, hw):
        """-> (resized view, its horizontal flip); PIL-bilinear pixels from the HIP kernel"""
        if not self.resize_pixels:                       # box / label side only: shapes are right, pixels are not produced
            z = torch.zeros(img.shape[0], hw[0], hw[1], dtype=torch.uint8, device=img.device)
            return z, z.clone()
        return resize_bilinear_u8(img, hw, with_flip=True)

    def __call__(self, d, shapes=None):
        img = d["image"]
        dev = img.device
        h, w = img.shape[-2:]
        hw1, hw2 = shapes if shapes is not None else self._draw_shapes(h, w)
        out = {k: v for k, v in d.items() if k not in ("image", "proposal_boxes", "proposal_objectness_logits",
                                                       "proposal_bbox_mode", "annotations")}
        out.setdefault("height", h)
        out.setdefault("width", w)
        out["image1"], out["image1_flip"] = self._resize_with_flip(img, hw1)
        out["image2"], out["image2_flip"] = self._resize_with_flip(img, hw2)
        views = (("1", hw1, False), ("2", hw2, False), ("1_flip", hw1, True), ("2_flip", hw2, True))
        tfms = {name: ViewTransform((h, w), hw, flip) for name, hw, flip in views}

        if "proposal_boxes" in d:
            boxes = torch.as_tensor(d["proposal_boxes"], dtype=torch.float32, device=dev).reshape(-1, 4)
            if int(d.get("proposal_bbox_mode", 0)) == 1:                 # XYWH_ABS -> XYXY_ABS (BoxMode.convert)
                boxes = torch.cat([boxes[:, :2], boxes[:, :2] + boxes[:, 2:]], 1)
            logits = torch.as_tensor(d["proposal_objectness_logits"], dtype=torch.float32, device=dev).reshape(-1)
            per_view, keep = {}, None
            for name, t in tfms.items():
                b, k = transform_proposals_multi(boxes, t, self.min_box_size)
                per_view[name] = b
                keep = k if keep is None else keep & k
            topk = self.proposal_topk if self.proposal_topk is not None else boxes.shape[0]
            keep = keep[:topk]          # the reference indexes the top-k slice with the full-length mask; equal lengths in its recipes
            for name, t in tfms.items():
                p = Instances(t.new_hw)
                p.proposal_boxes = Boxes(per_view[name][:topk][keep])
                p.objectness_logits = logits[:topk][keep]
                out["proposals" + name] = p

        if "annotations" in d:
            annos = [a for a in d["annotations"] if a.get("iscrowd", 0) == 0]
            gt = torch.as_tensor(np.asarray([a["bbox"] for a in annos], dtype=np.float64).reshape(-1, 4), device=dev)
            cls = torch.as_tensor([int(a["category_id"]) for a in annos], dtype=torch.int64, device=dev)
            for name, t in tfms.items():
                g = t.apply_box(gt)                            # float64 like the reference's per-annotation numpy path
                hh, ww = t.new_hw
                g = torch.stack([g[:, 0].clamp(0, ww), g[:, 1].clamp(0, hh), g[:, 2].clamp(0, ww), g[:, 3].clamp(0, hh)], 1)
                inst = Instances(t.new_hw)
                inst.gt_boxes = Boxes(g.float())
                inst.gt_classes = cls
                out["instances" + name] = inst
        return out
