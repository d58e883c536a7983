"""Generate tests/golden/tta_s0.npz by RUNNING the reference's own TTA-avg code
(/root/reference/uwsod/projects/WSL/wsl/modeling/test_time_augmentation_avg.py: DatasetMapperTTAAVG :131-197,
GeneralizedRCNNWithTTAAVG._inference_one_image / _get_augmented_boxes / _merge_detections :311-393) on the reference
model of fixture s0 (closed-form parameters and inputs, oracle/detgen.py), in the build container only:

    python tests/golden/make_tta_golden.py

What is loaded from the reference by path (no package __init__ runs): the TTA file itself, its view builder's
`detectron2/data/transforms/{transform,augmentation,augmentation_impl}.py` (ResizeTransform incl. its PIL resize,
ResizeShortestEdge, RandomFlip, apply_augmentations) and the stock `detectron2/modeling/roi_heads/fast_rcnn.py`
(fast_rcnn_inference_single_image, which the merge calls), on top of ref_shim.install().

Third-party code absent from /root/reference and restated here from its published algorithm (fvcore, setup.py pins
`fvcore>=0.1.1`; fvcore/transforms/transform.py): `Transform._set_attributes`, `Transform.apply_box` (a box goes through
its 4 corners and is re-boxed by min / max), `TransformList` (apply_* in order, `+`, `inverse()` = inverses in reverse
order), `HFlipTransform` (x -> width - x; image flipped along axis 1), `NoOpTransform`.

The fixture holds the reference's outputs only: per-view image CRC32s and transformed proposal boxes (the mapper), the
view-averaged score / box matrices and the merged detections (boxes, scores, classes)."""
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import ref_shim  # noqa: E402
from oracle import oicr_oracle as O  # noqa: E402

MIN_SIZES, MAX_SIZE, FLIP = (96, 144, 80), 4000, True     # identity, up-scale, down-scale of the 96x128 image; x flip


def install_fvcore_transforms():
    tr = ref_shim._pkg("fvcore.transforms")
    tt = ref_shim._pkg("fvcore.transforms.transform")

    class Transform:
        def _set_attributes(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)

        def apply_box(self, box):
            idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
            coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
            coords = self.apply_coords(coords).reshape((-1, 4, 2))
            return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)

        def apply_segmentation(self, seg):
            return self.apply_image(seg)

        def inverse(self):
            raise NotImplementedError

        @classmethod
        def register_type(cls, data_type, func=None):        # fvcore: adds `apply_<data_type>` to the class
            if func is None:
                return lambda f: cls.register_type(data_type, f) or f
            setattr(cls, "apply_" + data_type, lambda self, x: func(self, x))

    class TransformList(Transform):
        def __init__(self, transforms):
            flat = []
            for t in transforms:
                assert isinstance(t, Transform), t
                flat.extend(t.transforms if isinstance(t, TransformList) else [t])
            self.transforms = flat

        def _apply(self, x, meth):
            for t in self.transforms:
                x = getattr(t, meth)(x)
            return x

        def __getattribute__(self, name):
            if name.startswith("apply_"):
                return lambda x: self._apply(x, name)
            return super().__getattribute__(name)

        def __add__(self, other):
            return TransformList(self.transforms + (other.transforms if isinstance(other, TransformList) else [other]))

        def __iadd__(self, other):
            self.transforms.extend(other.transforms if isinstance(other, TransformList) else [other])
            return self

        def __radd__(self, other):
            return TransformList((other.transforms if isinstance(other, TransformList) else [other]) + self.transforms)

        def __len__(self):
            return len(self.transforms)

        def __getitem__(self, i):
            return self.transforms[i]

        def inverse(self):
            return TransformList([t.inverse() for t in self.transforms[::-1]])

    class HFlipTransform(Transform):
        def __init__(self, width):
            self.width = width

        def apply_image(self, img):
            return np.flip(img, axis=1) if img.ndim <= 3 else np.flip(img, axis=-2)

        def apply_coords(self, coords):
            coords[:, 0] = self.width - coords[:, 0]
            return coords

        def inverse(self):
            return self

    class NoOpTransform(Transform):
        def apply_image(self, img):
            return img

        def apply_coords(self, coords):
            return coords

        def inverse(self):
            return self

        def __getattr__(self, name):
            if name.startswith("apply_"):
                return lambda x: x
            raise AttributeError(name)

    class _Unused(Transform):
        def __init__(self, *a, **k):
            raise NotImplementedError
    for n, c in dict(Transform=Transform, TransformList=TransformList, HFlipTransform=HFlipTransform, NoOpTransform=NoOpTransform,
                     VFlipTransform=_Unused, CropTransform=_Unused, BlendTransform=_Unused).items():
        setattr(tt, n, c); setattr(tr, n, c)
    return tt


def install_tta(ns):
    REF, WSL = ref_shim.REF, ref_shim.WSL
    tt = install_fvcore_transforms()
    from PIL import Image
    if not hasattr(Image, "LINEAR"):
        Image.LINEAR = Image.BILINEAR                          # the alias Pillow >= 10 dropped (a default argument of ExtentTransform)
    dt = ref_shim._pkg("detectron2.data.transforms", REF + "/detectron2/data/transforms")
    mods = [ref_shim._load("detectron2.data.transforms." + n, REF + f"/detectron2/data/transforms/{n}.py")
            for n in ("transform", "augmentation", "augmentation_impl")]
    for m in mods + [tt]:
        for n in getattr(m, "__all__", [k for k in vars(m) if not k.startswith("_")]):
            setattr(dt, n, getattr(m, n))
    sys.modules["detectron2.data.detection_utils"].read_image = None
    import fvcore.nn as fvnn
    fvnn.giou_loss = lambda *a, **k: None
    fr = ref_shim._load("detectron2.modeling.roi_heads.fast_rcnn", REF + "/detectron2/modeling/roi_heads/fast_rcnn.py")
    sys.modules["detectron2.modeling.meta_arch"].MultiInputRCNN = ns.multi.MultiInputRCNN
    ma = ref_shim._pkg("wsl.modeling.meta_arch"); ma.GeneralizedRCNNWSL = type("GeneralizedRCNNWSL", (), {})
    pp = ref_shim._pkg("wsl.modeling.postprocessing"); pp.detector_postprocess = lambda r, h, w: r
    tta = ref_shim._load("wsl.modeling.test_time_augmentation_avg", WSL + "/wsl/modeling/test_time_augmentation_avg.py")
    return tta, fr


class _Cfg(types.SimpleNamespace):
    def clone(self):
        return self


def main():
    ns = ref_shim.install()
    tta, _ = install_tta(ns)
    import make_golden as MG                                  # E2E_CASES, load_params (its module-level install() is idempotent)
    H, W, R, n_gt, dan, hs = MG.E2E_CASES["s0"]
    K = 20
    P = O.make_params(K, dan, tag="ps0", head_scale=hs)
    views, gt = O.make_views(H, W, R, n_gt=n_gt, K=K, tag="vs0")
    model = ref_shim.build_reference_model(ns, K, dan)
    MG.load_params(model, P)
    model.eval()
    v = views[0]
    Boxes, Instances = ns.boxes.Boxes, ns.instances.Instances
    p = Instances((H, W)); p.proposal_boxes = Boxes(torch.from_numpy(v["boxes"])); p.objectness_logits = torch.from_numpy(v["obj"])
    cfg = _Cfg(MODEL=_Cfg(KEYPOINT_ON=False, MASK_ON=False, LOAD_PROPOSALS=True,
                          ROI_HEADS=_Cfg(SCORE_THRESH_TEST=1e-6, NMS_THRESH_TEST=0.3)),
               TEST=_Cfg(DETECTIONS_PER_IMAGE=100, AUG=_Cfg(MIN_SIZES=MIN_SIZES, MAX_SIZE=MAX_SIZE, FLIP=FLIP)),
               INPUT=_Cfg(FORMAT="BGR"), DATASETS=_Cfg(PRECOMPUTED_PROPOSAL_TOPK_TEST=4000))
    wrapper = tta.GeneralizedRCNNWithTTAAVG(cfg, model)
    inp = {"image": torch.from_numpy(v["image"]), "proposals": p, "height": H, "width": W}
    out = {"min_sizes": np.array(MIN_SIZES), "max_size": np.array(MAX_SIZE), "flip": np.array(FLIP)}
    with torch.no_grad(), ns.events.EventStorage(0):
        # the mapper's views (recorded), then the reference's own one-image inference
        aug, tfms = wrapper._get_augmented_inputs(dict(inp))
        for i, a in enumerate(aug):
            out[f"view{i}/hw"] = np.array(a["image"].shape[1:])
            out[f"view{i}/crc"] = np.array(zlib.crc32(np.ascontiguousarray(a["image"].numpy()).tobytes()), np.int64)
            out[f"view{i}/boxes"] = a["proposals"].proposal_boxes.tensor.numpy().copy()
        all_boxes, all_scores, _ = wrapper._get_augmented_boxes(aug, tfms)
        out["avg_boxes"], out["avg_scores"] = all_boxes.numpy().copy(), all_scores.numpy().copy()
        res = wrapper([inp])[0]["instances"]
    out["pred_boxes"] = res.pred_boxes.tensor.numpy(); out["scores"] = res.scores.numpy()
    out["pred_classes"] = res.pred_classes.numpy()
    np.savez_compressed(os.path.join(HERE, "tta_s0.npz"), **out)
    print(f"[tta s0] {len(aug)} views {[tuple(int(x) for x in out[f'view{i}/hw']) for i in range(len(aug))]}; "
          f"{len(out['scores'])} merged detections, top score {out['scores'][:3]}")


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
