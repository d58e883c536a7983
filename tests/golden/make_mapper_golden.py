"""Generate tests/golden/mapper_{a,v}.npz by RUNNING the reference's own 4-view training mapper
(/root/reference/uwsod/detectron2/data/dataset_mapper.py:192-439 `DatasetMapperMultiInput.__call__`) — RandomCrop,
two ResizeShortestEdge draws, RandomBrightness / RandomSaturation, the two flips, `transform_proposals_multi`
(detection_utils.py:208-260), `transform_instance_annotations` / `annotations_to_instances` — in the build container only:

    python tests/golden/make_mapper_golden.py

Loaded from the reference by path (no package __init__ runs): `data/dataset_mapper.py`, `data/detection_utils.py`,
`data/transforms/{transform,augmentation,augmentation_impl}.py`, `structures/{boxes,instances}.py`, on top of
ref_shim.install().  The augmentation list is the one `DatasetMapperMultiInput.from_config` + `build_augmentation`
(detection_utils.py:621-646) build for META_ARCHITECTURE MultiInputRCNN with INPUT.CROP.ENABLED: RandomCrop(TYPE, SIZE),
ResizeShortestEdge(MIN_SIZE_TRAIN, MAX_SIZE_TRAIN, "choice"), RandomBrightness(1/1.5, 1.5), RandomSaturation(1/1.5, 1.5).

Third-party code absent from /root/reference, restated from its published algorithm (fvcore/transforms/transform.py; setup.py
pins `fvcore>=0.1.1`): Transform / TransformList / HFlipTransform / NoOpTransform (make_tta_golden.install_fvcore_transforms) and, here,
  * CropTransform(x0, y0, w, h): image[y0:y0+h, x0:x0+w]; coords -= (x0, y0)
  * BlendTransform(src_image, src_weight, dst_weight): uint8 image -> float32, `src_weight * src_image + dst_weight * img`,
    clip to [0, 255], cast to uint8; coordinates unchanged.
pycocotools / the dataset catalog are imported by detection_utils.py but not used on this path: empty stand-in modules.
The image file the mapper reads is a PNG written to a temporary directory from closed-form pixels.

The fixture holds inputs (BGR image, proposal boxes / logits, annotations, the numpy seed) and the reference's outputs: the
four views (pixels for case a, CRC32 + shape for case v), the four proposal sets, the four annotation sets, and the
transforms the reference drew (crop window, the two output shapes, the four blend weights), read back from the
TransformLists the reference built."""
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import make_tta_golden as TG  # noqa: E402

VOC_SIZES = (480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896, 928, 960, 992, 1024, 1056, 1088, 1120, 1152,
             1184, 1216)
CASES = {
    # name: (h, w, n proposals, MIN_SIZE_TRAIN, MAX_SIZE_TRAIN, topk, numpy seed, store pixels)
    "a": (120, 160, 300, (96, 112, 128, 144, 160, 176), 2000, 4000, 7, True),
    "v": (375, 500, 600, VOC_SIZES, 2000, 4000, 11, False),
    "m": (200, 333, 400, (480, 1216), 600, 4000, 3, False),        # MAX_SIZE_TRAIN binds on the first draw only
}


def install_mapper():
    ns = ref_shim.install()
    tt = TG.install_fvcore_transforms()
    Transform = tt.Transform

    class CropTransform(Transform):
        def __init__(self, x0, y0, w, h):
            self.x0, self.y0, self.w, self.h = x0, y0, w, h

        def apply_image(self, img):
            return img[self.y0:self.y0 + self.h, self.x0:self.x0 + self.w]

        def apply_coords(self, coords):
            coords[:, 0] -= self.x0
            coords[:, 1] -= self.y0
            return coords

    class BlendTransform(Transform):
        def __init__(self, src_image, src_weight, dst_weight):
            self.src_image, self.src_weight, self.dst_weight = src_image, src_weight, dst_weight

        def apply_image(self, img, interp=None):
            if img.dtype == np.uint8:
                img = img.astype(np.float32)
                img = self.src_weight * self.src_image + self.dst_weight * img
                return np.clip(img, 0, 255).astype(np.uint8)
            return self.src_weight * self.src_image + self.dst_weight * img

        def apply_coords(self, coords):
            return coords

        def apply_segmentation(self, seg):
            return seg
    for n, c in dict(CropTransform=CropTransform, BlendTransform=BlendTransform).items():
        setattr(tt, n, c)
        setattr(sys.modules["fvcore.transforms"], n, c)

    from PIL import Image
    if not hasattr(Image, "LINEAR"):
        Image.LINEAR = Image.BILINEAR
    REF = ref_shim.REF
    dt = ref_shim._pkg("detectron2.data.transforms", REF + "/detectron2/data/transforms")
    mods = [ref_shim._load("detectron2.data.transforms." + n, REF + f"/detectron2/data/transforms/{n}.py")
            for n in ("transform", "augmentation", "augmentation_impl")]
    for m in mods + [tt]:
        for n in getattr(m, "__all__", [k for k in vars(m) if not k.startswith("_")]):
            setattr(dt, n, getattr(m, n))
    # modules detection_utils.py imports and this path never calls
    pc = ref_shim._pkg("pycocotools"); pm = ref_shim._pkg("pycocotools.mask"); pc.mask = pm
    fio = sys.modules["fvcore.common.file_io"]
    fio.PathManager = type("PathManager", (), {"open": staticmethod(open)})
    st = sys.modules["detectron2.structures"]
    for n in ("BitMasks", "Keypoints", "PolygonMasks", "RotatedBoxes", "polygons_to_bitmask"):
        setattr(st, n, None)
    cat = ref_shim._pkg("detectron2.data.catalog"); cat.MetadataCatalog = None
    if not hasattr(np, "int"):
        np.int = int                                           # boxes.py:224 uses the alias numpy 2 removed
    du = ref_shim._load("detectron2.data.detection_utils", REF + "/detectron2/data/detection_utils.py")
    dm = ref_shim._load("detectron2.data.dataset_mapper", REF + "/detectron2/data/dataset_mapper.py")
    return ns, dt, du, dm


def case_inputs(name):
    h, w, n, sizes, max_size, topk, seed, _ = CASES[name]
    rng = np.random.RandomState(1000 + seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3 + yy * 2 + 40 * c) % 256 for c in range(3)], 0).astype(np.int64)      # smooth ramps + noise, BGR planar
    img = np.clip(img + rng.randint(-30, 31, img.shape), 0, 255).astype(np.uint8)
    img[:, h // 3:h // 2, w // 4:w // 2] = rng.randint(0, 256, (1, h // 2 - h // 3, w // 2 - w // 4))   # a grey patch (B = G = R)
    import make_golden as MG
    boxes, logits = MG.input_case_boxes(seed, n, h, w)
    boxes[20:30, 0] = 0; boxes[20:30, 2] = 3                   # slivers at the left border: a crop offset > 3 empties them
    boxes[30:40, 1] = h - 4; boxes[30:40, 3] = h - 1           # slivers at the bottom border
    annos = [{"bbox": [10.5, 20.25, w * 0.6, h * 0.7], "category_id": 3, "iscrowd": 0},
             {"bbox": [0.0, 0.0, 8.0, 9.0], "category_id": 7, "iscrowd": 0},                # may be cropped away entirely
             {"bbox": [w * 0.5, h * 0.4, w - 1.0, h - 1.0], "category_id": 3, "iscrowd": 1},  # crowd: dropped
             {"bbox": [w * 0.3, h * 0.1, w - 0.5, h - 0.25], "category_id": 12, "iscrowd": 0}]
    return img, boxes, logits, annos


def run_case(name, dt, du, dm):
    from PIL import Image
    h, w, n, sizes, max_size, topk, seed, store_pixels = CASES[name]
    img, boxes, logits, annos = case_inputs(name)
    BoxMode = sys.modules["detectron2.structures"].BoxMode
    augs = [dt.RandomCrop("relative_range", [0.9, 0.9]), dt.ResizeShortestEdge(sizes, max_size, "choice"),
            dt.RandomBrightness(1.0 / 1.5, 1.5), dt.RandomSaturation(1.0 / 1.5, 1.5)]
    cfg = types.SimpleNamespace(INPUT=types.SimpleNamespace(MIN_SIZE_TRAIN=sizes))
    mapper = dm.DatasetMapperMultiInput(True, augmentations=augs, image_format="BGR", precomputed_proposal_topk=topk, cfg=cfg)
    log = []
    orig_apply = dt.StandardAugInput.apply_augmentations

    def logged(self, augmentations):
        tl = orig_apply(self, augmentations)
        log.append(tl)
        return tl
    dt.StandardAugInput.apply_augmentations = logged
    with tempfile.TemporaryDirectory() as tmp:
        fn = os.path.join(tmp, "img.png")
        Image.fromarray(np.ascontiguousarray(img.transpose(1, 2, 0)[:, :, ::-1])).save(fn)     # RGB on disk, BGR after read_image
        d = {"file_name": fn, "height": h, "width": w, "image_id": 17, "proposal_boxes": boxes.copy(),
             "proposal_objectness_logits": logits.copy(), "proposal_bbox_mode": BoxMode.XYXY_ABS,
             "annotations": [dict(a, bbox=list(a["bbox"]), bbox_mode=BoxMode.XYXY_ABS) for a in annos]}
        np.random.seed(seed)
        try:
            res = mapper(d)
        finally:
            dt.StandardAugInput.apply_augmentations = orig_apply
    # the transforms the reference drew: log = [crop, aug1, aug2 tries..., aug3, aug4]
    crop = log[0].transforms[0]
    t1, t2 = log[1], log[-3]
    assert len(log[-2].transforms) == 4 and len(log[-1].transforms) == 4
    out = {"seed": np.array(seed), "image": img, "boxes": boxes, "logits": logits, "min_sizes": np.array(sizes),
           "max_size": np.array(max_size), "topk": np.array(topk), "n_tries2": np.array(len(log) - 4),
           "anno_boxes": np.array([a["bbox"] for a in annos], np.float64), "anno_classes": np.array([a["category_id"] for a in annos]),
           "anno_crowd": np.array([a["iscrowd"] for a in annos]),
           "crop": np.array([crop.y0, crop.x0, crop.h, crop.w]),
           "hw1": np.array([t1.transforms[0].new_h, t1.transforms[0].new_w]),
           "hw2": np.array([t2.transforms[0].new_h, t2.transforms[0].new_w]),
           "blend1": np.array([t1.transforms[1].dst_weight, t1.transforms[2].dst_weight], np.float64),
           "blend2": np.array([t2.transforms[1].dst_weight, t2.transforms[2].dst_weight], np.float64)}
    for key in ("1", "2", "1_flip", "2_flip"):
        im = res["image" + key].numpy()
        out[f"hw_{key}"] = np.array(im.shape[1:])
        out[f"crc_{key}"] = np.array(zlib.crc32(np.ascontiguousarray(im).tobytes()), np.int64)
        if store_pixels:
            out[f"image{key}"] = im
        p = res["proposals" + key]
        out[f"pboxes_{key}"] = p.proposal_boxes.tensor.numpy().copy()
        out[f"plogits_{key}"] = p.objectness_logits.numpy().copy()
        out[f"psize_{key}"] = np.array(p.image_size)
        g = res["instances" + key]
        out[f"gboxes_{key}"] = g.gt_boxes.tensor.numpy().copy()
        out[f"gclasses_{key}"] = g.gt_classes.numpy().copy()
    # the saturation blend runs in float64 with a BLAS dot for the grey value: make sure the fixture does not sit on a rounding
    # edge (the product path fixes one operation order).  Recompute image1 both ways from the cropped, resized pixels.
    y0, x0, ch, cw = out["crop"]
    base = np.asarray(Image.fromarray(np.ascontiguousarray(img.transpose(1, 2, 0)[y0:y0 + ch, x0:x0 + cw]))
                      .resize((int(out["hw1"][1]), int(out["hw1"][0])), Image.BILINEAR))
    wb, ws = out["blend1"]
    b = np.clip(np.float32(wb) * base.astype(np.float32), 0, 255).astype(np.uint8)
    g64 = (b[..., 0].astype(np.float64) * 0.299 + b[..., 1].astype(np.float64) * 0.587) + b[..., 2].astype(np.float64) * 0.114
    val = (1 - ws) * g64[..., None] + (np.float32(ws) * b.astype(np.float32)).astype(np.float64)
    frac = np.abs(val - np.round(val))
    assert frac.min() > 1e-9, f"case {name}: a pixel sits {frac.min():.3e} from an integer; pick another seed"
    mine = np.clip(val, 0, 255).astype(np.uint8).transpose(2, 0, 1)
    assert np.array_equal(mine, res["image1"].numpy()), "restated blend order differs from the reference run"
    np.savez_compressed(os.path.join(HERE, f"mapper_{name}.npz"), **out)
    print(f"[mapper {name}] crop {tuple(out['crop'])} of {(h, w)}; views {tuple(out['hw1'])} {tuple(out['hw2'])} "
          f"({int(out['n_tries2'])} draw(s) for the second); blend {out['blend1']} {out['blend2']}; "
          f"kept {len(out['plogits_1'])} of {n} proposals; gt {out['gboxes_1'].shape[0]}; "
          f"min distance to a rounding edge {frac.min():.2e}")


if __name__ == "__main__":
    ns, dt, du, dm = install_mapper()
    for c in (sys.argv[1:] or list(CASES)):
        run_case(c, dt, du, dm)
