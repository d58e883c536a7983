"""Device-side multi-view training mapper (the reference's `DatasetMapperMultiInput`,
uwsod/detectron2/data/dataset_mapper.py:192-439).

One dataset dict (decoded image + precomputed proposals + image-level annotations) becomes the four index-aligned views the
detector trains on: an optional RandomCrop of the image (INPUT.CROP, `from_config` :244-248, applied first :278-285; every
code_release recipe enables it), two scales drawn from INPUT.MIN_SIZE_TRAIN (the second forced to differ from the first), a
RandomBrightness and a RandomSaturation blend per scale (`build_augmentation`, detection_utils.py:621-646), each scale with
its horizontal flip — `image1`, `image1_flip`, `image2`, `image2_flip`, the matching `proposals*` and `instances*`.  The
crop's offset is prepended to every box transform (`transform_randomcrop + transforms_k`, :341-353, :365-390).

Random draws come from one `numpy.random.RandomState` in the reference's order (crop size `rand(2)`, crop offsets `randint` y
then x, then per scale `choice` of the short side, brightness `uniform`, saturation `uniform`, the second scale redrawn — all
three — until its shape differs): seeded like the reference's global numpy generator, the mapper draws the same views
(tests/golden/mapper_*.npz, written by running the reference's mapper).

The box side is the reference's rule for rule (`transform_proposals_multi`, detection_utils.py:208-260): every view maps the
SAME proposal list through resize (+ flip), clips it, and computes a keep mask (first occurrence of each rounded-corner hash,
`Boxes.unique_boxes` boxes.py:214-226, AND non-empty); nothing is filtered per view — the four masks are ANDed and applied to
all four sets, so row i is the same proposal in every view (the consistency losses need that).  All of it runs as tensor ops
on the image's device, no host round trip.

The pixel side: the reference resizes with PIL bilinear on the CPU (`ResizeTransform.apply_image`); here the same 8-bit
two-pass arithmetic runs as a HIP kernel (`resize.resize_bilinear_u8`, `sw_resize_pass_u8`), bit-identical to Pillow
(tests/golden/resize_*.npz); the crop is a window of the source tensor handed to that kernel by pointer and strides (no copy),
the two blends are one launch (`sw_color_jitter_u8`: fvcore BlendTransform's float32 / float64 arithmetic) that also writes the
flipped view.  There is no CPU pixel path: a host-resident image
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


def crop_size_rule(crop_type, crop_size, h, w, rng):
    """RandomCrop.get_crop_size (augmentation_impl.py:252-276) -> (crop h, crop w)"""
    if crop_type == "relative":
        ch, cw = crop_size
        return int(h * ch + 0.5), int(w * cw + 0.5)
    if crop_type == "relative_range":
        cs = np.asarray(crop_size, dtype=np.float32)
        ch, cw = cs + rng.rand(2) * (1 - cs)
        return int(h * ch + 0.5), int(w * cw + 0.5)
    if crop_type == "absolute":
        return min(crop_size[0], h), min(crop_size[1], w)
    if crop_type == "absolute_range":
        assert crop_size[0] <= crop_size[1]
        ch = rng.randint(min(h, crop_size[0]), min(h, crop_size[1]) + 1)
        cw = rng.randint(min(w, crop_size[0]), min(w, crop_size[1]) + 1)
        return ch, cw
    raise ValueError(f"unknown INPUT.CROP.TYPE {crop_type!r}")


class DeviceMultiInputMapper:
    """`mapper(dataset_dict) -> dict` with the keys the detector's training forward reads (rcnn_multi.py).

    dataset_dict: "image" (3, h, w) uint8 tensor (any device), "proposal_boxes" (N, 4) in "proposal_bbox_mode" (XYXY_ABS = 0
    default, XYWH_ABS = 1; `proposals.load_proposals_into_dataset` fills all three), "proposal_objectness_logits" (N,), optional
    "annotations" = list of {"bbox": XYXY_ABS, "category_id": int, "iscrowd": 0/1}.

    crop: None or (INPUT.CROP.TYPE, INPUT.CROP.SIZE); brightness / saturation: None or the (min, max) intensity range.
    `from_config` builds the reference's training recipe; the bare constructor's defaults leave crop and blends off.
    """

    def __init__(self, min_sizes: Sequence[int] = (480, 576, 688, 864, 1000, 1200), max_size: int = 2000,
                 proposal_topk: Optional[int] = 2000, min_box_size: float = 0.0, seed: Optional[int] = None,
                 resize_pixels: bool = True, crop=None, brightness=None, saturation=None, sample_style: str = "choice"):
        assert len(min_sizes) >= 2, "two different scales are drawn per image (dataset_mapper.py:305-321)"
        assert sample_style in ("choice", "range")
        self.min_sizes, self.max_size = tuple(min_sizes), max_size
        self.proposal_topk, self.min_box_size = proposal_topk, min_box_size
        self.rng = np.random.RandomState(seed)
        self.resize_pixels = resize_pixels
        self.crop = None if crop is None else (str(crop[0]), tuple(crop[1]))
        if self.crop is not None:
            assert self.crop[0] in ("relative_range", "relative", "absolute", "absolute_range"), self.crop[0]
        self.brightness = None if brightness is None else tuple(float(v) for v in brightness)
        self.saturation = None if saturation is None else tuple(float(v) for v in saturation)
        self.sample_style = sample_style

    @classmethod
    def from_config(cls, cfg, seed: Optional[int] = None, resize_pixels: bool = True):
        """DatasetMapperMultiInput.from_config (dataset_mapper.py:243-270) + build_augmentation (detection_utils.py:621-646) for
        is_train and META_ARCHITECTURE MultiInputRCNN: [RandomCrop if INPUT.CROP.ENABLED], ResizeShortestEdge, RandomBrightness(1/1.5,
        1.5), RandomSaturation(1/1.5, 1.5); no RandomFlip (the flipped views are built explicitly)"""
        inp = cfg.INPUT
        crop_cfg = inp.get("CROP", {}) if hasattr(inp, "get") else getattr(inp, "CROP", {})
        crop = None
        if crop_cfg and crop_cfg.get("ENABLED", False):
            crop = (crop_cfg.get("TYPE", "relative_range"), crop_cfg.get("SIZE", [0.9, 0.9]))
        topk = cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN if cfg.MODEL.LOAD_PROPOSALS else None
        return cls(min_sizes=tuple(inp.MIN_SIZE_TRAIN), max_size=inp.MAX_SIZE_TRAIN, proposal_topk=topk, seed=seed,
                   resize_pixels=resize_pixels, crop=crop, brightness=(1.0 / 1.5, 1.5), saturation=(1.0 / 1.5, 1.5),
                   sample_style=inp.get("MIN_SIZE_TRAIN_SAMPLING", "choice") if hasattr(inp, "get") else "choice")

    # ---- random draws, in the reference's order
    def _draw_blend(self):
        wb = float(self.rng.uniform(*self.brightness)) if self.brightness is not None else None
        ws = float(self.rng.uniform(*self.saturation)) if self.saturation is not None else None
        return wb, ws

    def _draw_views(self, h, w):
        """-> (hw1, hw2, blend1, blend2) for a (cropped) h x w image: ResizeShortestEdge.get_transform (augmentation_impl.py:155-175)
        then the two blends, per scale; the second scale's list drops min(first shape) and its resize has NO max size
        (dataset_mapper.py:305-321: `ResizeShortestEdge(size_list_, sample_style="choice")`)"""
        if self.sample_style == "range":
            s1 = int(self.rng.randint(self.min_sizes[0], self.min_sizes[1] + 1))
        else:
            s1 = int(self.rng.choice(self.min_sizes))
        hw1 = DeviceTTAMapper._shortest_edge(h, w, s1, self.max_size)
        blend1 = self._draw_blend()
        rest = [s for s in self.min_sizes if s != min(hw1)]
        for _ in range(64):                                 # the reference loops until the shapes differ
            hw2 = DeviceTTAMapper._shortest_edge(h, w, int(self.rng.choice(rest)), sys.maxsize)
            blend2 = self._draw_blend()
            if hw2 != hw1:
                return hw1, hw2, blend1, blend2
        raise RuntimeError(f"no second scale of {self.min_sizes} gives a shape different from {hw1} "
                           f"(the reference would loop forever here)")

    def _draw_shapes(self, h, w):
        hw1, hw2, _, _ = self._draw_views(h, w)
        return hw1, hw2

    def _draw_crop(self, h, w):
        """RandomCrop.get_transform (augmentation_impl.py:243-250) -> (y0, x0, crop h, crop w)"""
        ch, cw = crop_size_rule(self.crop[0], self.crop[1], h, w, self.rng)
        assert h >= ch and w >= cw, "Shape computation in RandomCrop has bugs."
        y0 = int(self.rng.randint(h - ch + 1))
        x0 = int(self.rng.randint(w - cw + 1))
        return y0, x0, int(ch), int(cw)

    def _view_pixels(self, win, hw, blend):
        """-> (view, its horizontal flip): PIL-bilinear resize of the (cropped) window, then the blends; HIP kernels only"""
        if not self.resize_pixels:                       # box / label side only: shapes are right, pixels are not produced
            z = torch.zeros(win.shape[0], hw[0], hw[1], dtype=torch.uint8, device=win.device)
            return z, z.clone()
        wb, ws = blend
        if wb is None and ws is None:
            return resize_bilinear_u8(win, hw, with_flip=True)
        from . import ops
        return ops.color_jitter_u8(resize_bilinear_u8(win, hw), wb, ws, with_flip=True)

    def __call__(self, d, shapes=None, draws=None):
        """shapes = (hw1, hw2) or draws = {"crop": (y0, x0, h, w) | None, "hw1", "hw2", "blend1": (wb, ws), "blend2"} replace
        the random draws (tests)"""
        img = d["image"]
        dev = img.device
        h, w = img.shape[-2:]
        if draws is not None:
            crop = draws.get("crop")
            hw1, hw2 = tuple(draws["hw1"]), tuple(draws["hw2"])
            blend1, blend2 = draws.get("blend1", (None, None)), draws.get("blend2", (None, None))
        else:
            crop = self._draw_crop(h, w) if self.crop is not None else None
            ch, cw = (crop[2], crop[3]) if crop is not None else (h, w)
            if shapes is not None:
                (hw1, hw2), blend1, blend2 = shapes, self._draw_blend(), self._draw_blend()
            else:
                hw1, hw2, blend1, blend2 = self._draw_views(ch, cw)
        self.last_draws = {"crop": crop, "hw1": hw1, "hw2": hw2, "blend1": blend1, "blend2": blend2}
        y0, x0, ch, cw = crop if crop is not None else (0, 0, h, w)
        win = img[:, y0:y0 + ch, x0:x0 + cw] if crop is not None else img
        out = {k: v for k, v in d.items() if k not in ("image", "proposal_boxes", "proposal_objectness_logits",
                                                       "proposal_bbox_mode", "annotations")}
        out.setdefault("height", h)                      # check_image_size: the size of the image as read, before the crop
        out.setdefault("width", w)
        out["image1"], out["image1_flip"] = self._view_pixels(win, hw1, blend1)
        out["image2"], out["image2_flip"] = self._view_pixels(win, hw2, blend2)
        views = (("1", hw1, False), ("2", hw2, False), ("1_flip", hw1, True), ("2_flip", hw2, True))
        tfms = {name: ViewTransform((ch, cw), hw, flip, crop_xy=(x0, y0) if crop is not None else None) for name, hw, flip in views}

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
