"""TEST INFRASTRUCTURE ONLY: numpy restatement of the box side of the reference's multi-view
training mapper.  Only tests/ may import this file.

Follows
  * `DatasetMapperMultiInput.__call__`  uwsod/detectron2/data/dataset_mapper.py:272-425 (view construction, the AND of the
    four `final_keep` masks, the four index-aligned proposal sets),
  * `transform_proposals_multi`         uwsod/detectron2/data/detection_utils.py:208-260 (transform, clip, unique / non-empty
    masks that do NOT filter, top-k slice),
  * `Boxes.unique_boxes`                uwsod/detectron2/structures/boxes.py:214-226 (hash of the rounded corners, first
    occurrence wins),
  * `ResizeShortestEdge.get_transform`  uwsod/detectron2/data/transforms/augmentation_impl.py (output shape rule),
  * fvcore `ResizeTransform.apply_coords` / `HFlipTransform.apply_coords` and `Transform.apply_box` — fvcore is a pip
    dependency that is absent from /root/reference (setup.py pins `fvcore>=0.1.1`); their published rule is restated:
    x *= new_w / w, y *= new_h / h;  x -> W - x;  a box is mapped through its 4 corners and re-boxed by min / max.

  * `RandomCrop` (augmentation_impl.py:232-276), `RandomBrightness` / `RandomSaturation` (:403-455), fvcore `CropTransform` /
    `BlendTransform` (published rule restated), and the ORDER of the random draws.

Pinned by tests/golden/input_a.npz (the clip / unique / non-empty masks of the reference's own `Boxes` class,
tests/golden/make_golden.py::run_input) and by tests/golden/mapper_{a,v,m}.npz, written by RUNNING the reference's
`DatasetMapperMultiInput.__call__` (make_mapper_golden.py): draws, boxes, annotations, pixels.  The PIL resize itself is
oracle/resize_oracle.py (pinned against Pillow).
"""
import numpy as np


def shortest_edge_shape(h, w, size, max_size):
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def crop_size(crop_type, size, h, w, rng):
    """RandomCrop.get_crop_size, augmentation_impl.py:252-276 (rng: a numpy RandomState standing for the global generator)"""
    if crop_type == "relative":
        return int(h * size[0] + 0.5), int(w * size[1] + 0.5)
    if crop_type == "relative_range":
        cs = np.asarray(size, dtype=np.float32)
        ch, cw = cs + rng.rand(2) * (1 - cs)
        return int(h * ch + 0.5), int(w * cw + 0.5)
    if crop_type == "absolute":
        return min(size[0], h), min(size[1], w)
    assert crop_type == "absolute_range"
    return (rng.randint(min(h, size[0]), min(h, size[1]) + 1), rng.randint(min(w, size[0]), min(w, size[1]) + 1))


def draw_views(seed, h, w, min_sizes, max_size, crop=("relative_range", (0.9, 0.9)), intensity=(1.0 / 1.5, 1.5)):
    """the random draws of one `DatasetMapperMultiInput.__call__` (dataset_mapper.py:272-321) in the order the reference makes
    them on numpy's global generator after `np.random.seed(seed)`: RandomCrop (size, y0, x0), then per scale the short side
    (`choice`), the brightness and the saturation weight; the second scale — all three draws — repeated until the shape differs"""
    import sys
    rng = np.random.RandomState(seed)
    ch, cw = crop_size(crop[0], crop[1], h, w, rng)
    y0 = int(rng.randint(h - ch + 1)); x0 = int(rng.randint(w - cw + 1))
    hw1 = shortest_edge_shape(ch, cw, int(rng.choice(min_sizes)), max_size)
    blend1 = (rng.uniform(*intensity), rng.uniform(*intensity))
    rest = [s for s in min_sizes if s != min(hw1)]
    tries = 0
    while True:
        tries += 1
        hw2 = shortest_edge_shape(ch, cw, int(rng.choice(rest)), sys.maxsize)
        blend2 = (rng.uniform(*intensity), rng.uniform(*intensity))
        if hw2 != hw1:
            break
    return {"crop": (y0, x0, int(ch), int(cw)), "hw1": hw1, "hw2": hw2, "blend1": blend1, "blend2": blend2, "tries2": tries}


def blend_u8(img_hwc, w_bright, w_sat):
    """RandomBrightness then RandomSaturation (augmentation_impl.py:403-455) through fvcore's BlendTransform.apply_image on a uint8
    HWC image (fvcore is absent from /root/reference; published rule: float32 image, `src_weight * src_image + dst_weight * img`,
    clip to [0, 255], cast to uint8).  The grey image of the saturation blend is a float64 dot with (0.299, 0.587, 0.114), summed
    left to right here (the reference's BLAS may fuse: pinned by the fixture generator's check that no pixel sits on a rounding edge)"""
    b = np.clip(0.0 + np.float32(w_bright) * img_hwc.astype(np.float32), 0, 255).astype(np.uint8)
    g = (b[..., 0].astype(np.float64) * 0.299 + b[..., 1].astype(np.float64) * 0.587) + b[..., 2].astype(np.float64) * 0.114
    v = (1 - w_sat) * g[..., None] + (np.float32(w_sat) * b.astype(np.float32)).astype(np.float64)
    return np.clip(v, 0, 255).astype(np.uint8)


def apply_box(boxes, orig_hw, new_hw, flip, dtype=np.float32, crop_xy=None):
    """boxes (N,4) XYXY -> [cropped,] resized (+ flipped) frame; 4-corner min/max like fvcore's Transform.apply_box.  The arithmetic
    runs in the dtype the boxes arrive in: float32 for the proposal files, float64 for annotation lists.  crop_xy = (x0, y0) of
    a CropTransform applied first (coords -= (x0, y0)); orig_hw is then the crop's size."""
    b = np.asarray(boxes, dtype=dtype).reshape(-1, 4)
    idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
    c = b[:, idxs].reshape(-1, 2).copy()
    if crop_xy is not None:
        c[:, 0] -= int(crop_xy[0])
        c[:, 1] -= int(crop_xy[1])
    # python-float factors: the product stays in the boxes' dtype (float32 proposals are scaled in float32)
    c[:, 0] = c[:, 0] * (int(new_hw[1]) * 1.0 / int(orig_hw[1]))
    c[:, 1] = c[:, 1] * (int(new_hw[0]) * 1.0 / int(orig_hw[0]))
    if flip:
        c[:, 0] = int(new_hw[1]) - c[:, 0]
    c = c.reshape(-1, 4, 2)
    mn, mx = c.min(axis=1), c.max(axis=1)
    return np.concatenate([mn, mx], axis=1)


def clip(boxes, hw):
    b = boxes.copy()
    b[:, 0::2] = np.clip(b[:, 0::2], 0, hw[1])
    b[:, 1::2] = np.clip(b[:, 1::2], 0, hw[0])
    return b


def unique_mask(boxes):
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(boxes.astype(np.float32) * 1.0).dot(v).astype(np.int64)
    _, index = np.unique(hashes, return_index=True)
    m = np.zeros(len(boxes), dtype=bool)
    m[index] = True
    return m


def nonempty(boxes, thr=0.0):
    return ((boxes[:, 2] - boxes[:, 0]) > thr) & ((boxes[:, 3] - boxes[:, 1]) > thr)


def transform_proposals_multi(boxes, logits, orig_hw, new_hw, flip, topk, min_box_size=0, crop_xy=None):
    b = clip(apply_box(boxes, orig_hw, new_hw, flip, crop_xy=crop_xy), new_hw)
    keep = unique_mask(b) & nonempty(b, min_box_size)
    return b[:topk], np.asarray(logits, dtype=np.float32)[:topk], keep


def multi_input_proposals(boxes, logits, orig_hw, hw1, hw2, topk, min_box_size=0, crop_xy=None):
    """-> dict name -> (boxes, logits) for proposals1, proposals1_flip, proposals2, proposals2_flip, and the joint mask"""
    views = {"proposals1": (hw1, False), "proposals2": (hw2, False), "proposals1_flip": (hw1, True),
             "proposals2_flip": (hw2, True)}
    res, keep = {}, None
    for name, (hw, flip) in views.items():
        b, l, k = transform_proposals_multi(boxes, logits, orig_hw, hw, flip, topk, min_box_size, crop_xy=crop_xy)
        res[name] = (b, l)
        keep = k if keep is None else keep & k
    keep = keep[:topk]          # the reference indexes the top-k slice with the full-length mask (equal lengths in its recipes)
    return {n: (b[keep[:len(b)]], l[keep[:len(b)]]) for n, (b, l) in res.items()}, keep


def transform_annotation_boxes(gt_boxes, orig_hw, new_hw, flip, crop_xy=None):
    """`transform_instance_annotations` detection_utils.py:310-345 (boxes): transform in float64, clip to [0, w] x [0, h];
    `annotations_to_instances` then stores float32"""
    return clip(apply_box(gt_boxes, orig_hw, new_hw, flip, dtype=np.float64, crop_xy=crop_xy), new_hw).astype(np.float32)
