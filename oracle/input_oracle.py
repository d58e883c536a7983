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

Pinned by tests/golden/input_a.npz: the clip / unique / non-empty masks there were produced by the reference's own `Boxes`
class (tests/golden/make_golden.py::run_input).  The pixel side (PIL bilinear resize) is not restated: parity unpinned
for pixels.
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


def apply_box(boxes, orig_hw, new_hw, flip, dtype=np.float32):
    """boxes (N,4) XYXY -> resized (+ flipped) frame; 4-corner min/max like fvcore's Transform.apply_box.  The arithmetic
    runs in the dtype the boxes arrive in: float32 for the proposal files, float64 for annotation lists."""
    b = np.asarray(boxes, dtype=dtype).reshape(-1, 4)
    idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
    c = b[:, idxs].reshape(-1, 2).copy()
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


def transform_proposals_multi(boxes, logits, orig_hw, new_hw, flip, topk, min_box_size=0):
    b = clip(apply_box(boxes, orig_hw, new_hw, flip), new_hw)
    keep = unique_mask(b) & nonempty(b, min_box_size)
    return b[:topk], np.asarray(logits, dtype=np.float32)[:topk], keep


def multi_input_proposals(boxes, logits, orig_hw, hw1, hw2, topk, min_box_size=0):
    """-> dict name -> (boxes, logits) for proposals1, proposals1_flip, proposals2, proposals2_flip, and the joint mask"""
    views = {"proposals1": (hw1, False), "proposals2": (hw2, False), "proposals1_flip": (hw1, True),
             "proposals2_flip": (hw2, True)}
    res, keep = {}, None
    for name, (hw, flip) in views.items():
        b, l, k = transform_proposals_multi(boxes, logits, orig_hw, hw, flip, topk, min_box_size)
        res[name] = (b, l)
        keep = k if keep is None else keep & k
    keep = keep[:topk]          # the reference indexes the top-k slice with the full-length mask (equal lengths in its recipes)
    return {n: (b[keep[:len(b)]], l[keep[:len(b)]]) for n, (b, l) in res.items()}, keep


def transform_annotation_boxes(gt_boxes, orig_hw, new_hw, flip):
    """`transform_instance_annotations` detection_utils.py:310-345 (boxes): transform in float64, clip to [0, w] x [0, h];
    `annotations_to_instances` then stores float32"""
    return clip(apply_box(gt_boxes, orig_hw, new_hw, flip, dtype=np.float64), new_hw).astype(np.float32)
