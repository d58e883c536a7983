"""Import the reference's hot-path Python files (from /root/reference, THIS container only)
under torch 2.10 without running any package __init__, with the absent third-party
dependencies stubbed.  Used only by make_golden.py to generate the committed fixtures;
nothing under tests/ that runs on the GPU box imports this file.

Third-party restatements (dependency, pin, what is restated):
  * torchvision==0.7.0  ops.RoIPool      -> oracle/roipool_oracle.c (same arithmetic as the
    reference's in-tree ROILoopPool_cpu.cpp:26-123)
  * torchvision==0.7.0  ops.nms / ops.boxes.batched_nms -> greedy NMS below (torch ops)
  * fvcore              nn.smooth_l1_loss -> below (beta==0 -> L1)
  * fvcore              nn.weight_init.c2_msra_fill -> kaiming_normal fan_out (init only)
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference/uwsod"
WSL = REF + "/projects/WSL"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import oicr_oracle as O  # noqa: E402


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(sys.modules[parent], child, m)
    return m


def _load(name, file):
    spec = importlib.util.spec_from_file_location(name, file)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(sys.modules[parent], child, m)
    spec.loader.exec_module(m)
    return m


class Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj

    def get(self, name):
        return self[name]


def install():
    # ---- fvcore
    _pkg("fvcore"); fvnn = _pkg("fvcore.nn"); _pkg("fvcore.common")

    def smooth_l1_loss(input, target, beta, reduction="none"):
        if beta < 1e-5:
            loss = torch.abs(input - target)
        else:
            n = torch.abs(input - target)
            loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss

    fvnn.smooth_l1_loss = smooth_l1_loss
    fvnn.giou_loss = None
    wi = _pkg("fvcore.nn.weight_init")

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    wi.c2_msra_fill = c2_msra_fill
    wi.c2_xavier_fill = c2_msra_fill
    fio = _pkg("fvcore.common.file_io"); fio.PathManager = object
    hb = _pkg("fvcore.common.history_buffer")

    class HistoryBuffer:
        def __init__(self, max_length=1000000):
            self._d = []

        def update(self, v, it=None):
            self._d.append((v, it))

        def latest(self):
            return self._d[-1][0]
    hb.HistoryBuffer = HistoryBuffer
    _pkg("cv2")

    # ---- torchvision
    _pkg("torchvision"); tvo = _pkg("torchvision.ops"); tvb = _pkg("torchvision.ops.boxes")

    class RoIPool(nn.Module):
        def __init__(self, output_size, spatial_scale):
            super().__init__()
            self.output_size = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
            self.spatial_scale = spatial_scale

        def forward(self, x, rois):
            return O._RoIPoolFn.apply(x, rois, self.spatial_scale, self.output_size[0], self.output_size[1])
    tvo.RoIPool = RoIPool

    def nms(boxes, scores, thr):
        order = scores.argsort(descending=True)
        keep = []
        sup = torch.zeros(len(boxes), dtype=torch.bool)
        area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        for _i in range(len(order)):
            i = int(order[_i])
            if sup[i]:
                continue
            keep.append(i)
            rest = order[_i + 1:]
            xx1 = torch.maximum(boxes[i, 0], boxes[rest, 0]); yy1 = torch.maximum(boxes[i, 1], boxes[rest, 1])
            xx2 = torch.minimum(boxes[i, 2], boxes[rest, 2]); yy2 = torch.minimum(boxes[i, 3], boxes[rest, 3])
            inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
            ovr = inter / (area[i] + area[rest] - inter)
            sup[rest[ovr > thr]] = True
        return torch.tensor(keep, dtype=torch.int64)

    def batched_nms(boxes, scores, idxs, thr):
        if boxes.numel() == 0:
            return torch.empty((0,), dtype=torch.int64)
        off = idxs.to(boxes) * (boxes.max() + 1)
        return nms(boxes + off[:, None], scores, thr)
    tvb.batched_nms = batched_nms; tvb.nms = nms; tvo.nms = nms; tvo.boxes = tvb

    # ---- reference packages as bare namespaces (no __init__ executed)
    _pkg("detectron2", REF + "/detectron2")
    _pkg("detectron2.utils", REF + "/detectron2/utils")
    _load("detectron2.utils.env", REF + "/detectron2/utils/env.py")
    ev = _load("detectron2.utils.events", REF + "/detectron2/utils/events.py")
    reg = _pkg("detectron2.utils.registry"); reg.Registry = Registry
    lg = _pkg("detectron2.utils.logger"); lg.log_first_n = lambda *a, **k: None
    cfgm = _pkg("detectron2.config")

    def configurable(init_func):
        import functools

        @functools.wraps(init_func)
        def wrapped(self, *a, **k):
            init_func(self, *a, **k)
        return wrapped
    cfgm.configurable = configurable

    lay = _pkg("detectron2.layers", REF + "/detectron2/layers")
    w = _load("detectron2.layers.wrappers", REF + "/detectron2/layers/wrappers.py")
    ss = _load("detectron2.layers.shape_spec", REF + "/detectron2/layers/shape_spec.py")
    for n in ["Conv2d", "Linear", "cat", "nonzero_tuple", "interpolate"]:
        setattr(lay, n, getattr(w, n))
    lay.ShapeSpec = ss.ShapeSpec
    lay.batched_nms = batched_nms
    lay.get_norm = lambda norm, ch: None

    class _NA(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError
    lay.ROIAlign = _NA; lay.ROIAlignRotated = _NA
    lay.FrozenBatchNorm2d = type("FrozenBatchNorm2d", (), {"convert_frozen_batchnorm": staticmethod(lambda m: m)})

    st = _pkg("detectron2.structures", REF + "/detectron2/structures")
    b = _load("detectron2.structures.boxes", REF + "/detectron2/structures/boxes.py")
    il = _load("detectron2.structures.image_list", REF + "/detectron2/structures/image_list.py")
    ins = _load("detectron2.structures.instances", REF + "/detectron2/structures/instances.py")
    st.Boxes = b.Boxes; st.BoxMode = b.BoxMode; st.pairwise_iou = b.pairwise_iou
    st.ImageList = il.ImageList; st.Instances = ins.Instances

    _pkg("detectron2.data"); du = _pkg("detectron2.data.detection_utils"); du.convert_image_to_rgb = None
    _pkg("detectron2.modeling", REF + "/detectron2/modeling")
    mt = _load("detectron2.modeling.matcher", REF + "/detectron2/modeling/matcher.py")
    br = _load("detectron2.modeling.box_regression", REF + "/detectron2/modeling/box_regression.py")
    _load("detectron2.modeling.sampling", REF + "/detectron2/modeling/sampling.py")
    pl = _pkg("detectron2.modeling.poolers"); pl.ROIPooler = None
    bb = _pkg("detectron2.modeling.backbone", REF + "/detectron2/modeling/backbone")
    bbb = _load("detectron2.modeling.backbone.backbone", REF + "/detectron2/modeling/backbone/backbone.py")
    bbuild = _pkg("detectron2.modeling.backbone.build"); bbuild.BACKBONE_REGISTRY = Registry("BACKBONE")
    bb.Backbone = bbb.Backbone; bb.build_backbone = None
    pg = _pkg("detectron2.modeling.proposal_generator"); pg.build_proposal_generator = None
    pgu = _pkg("detectron2.modeling.proposal_generator.proposal_utils"); pgu.add_ground_truth_to_proposals = None
    pp = _pkg("detectron2.modeling.postprocessing"); pp.detector_postprocess = lambda r, h, w: r
    rh = _pkg("detectron2.modeling.roi_heads")
    rh.ROI_HEADS_REGISTRY = Registry("ROI_HEADS"); rh.ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
    rh.build_roi_heads = None
    for sub, names in [("box_head", ["build_box_head"]), ("keypoint_head", ["build_keypoint_head"]),
                       ("mask_head", ["build_mask_head"]), ("fast_rcnn", ["FastRCNNOutputLayers"])]:
        m = _pkg("detectron2.modeling.roi_heads." + sub)
        for n in names:
            setattr(m, n, None)
    _pkg("detectron2.modeling.meta_arch", REF + "/detectron2/modeling/meta_arch")
    mab = _pkg("detectron2.modeling.meta_arch.build"); mab.META_ARCH_REGISTRY = Registry("META_ARCH")

    _pkg("wsl", WSL + "/wsl")
    wl = _pkg("wsl.layers"); wl.ROILoopPool = type("ROILoopPool", (nn.Module,), {}); wl.ROIMerge = None; wl.pcl_loss = None
    _pkg("wsl.modeling", WSL + "/wsl/modeling")
    _pkg("wsl.modeling.backbone", WSL + "/wsl/modeling/backbone")
    rws = _pkg("wsl.modeling.backbone.resnet_ws"); rws.BottleneckBlock = None; rws.make_stage = None
    _pkg("wsl.modeling.roi_heads", WSL + "/wsl/modeling/roi_heads")
    _pkg("wsl.modeling.roi_heads.third_party")
    tpp = _pkg("wsl.modeling.roi_heads.third_party.pcl"); tpp.PCL = None

    ns = types.SimpleNamespace()
    ns.events = ev; ns.shape_spec = ss; ns.boxes = b; ns.instances = ins; ns.matcher = mt; ns.box_regression = br
    ns.poolers = _load("wsl.modeling.poolers", WSL + "/wsl/modeling/poolers.py")
    ns.vgg = _load("wsl.modeling.backbone.vgg", WSL + "/wsl/modeling/backbone/vgg.py")
    ns.roi_heads = _load("wsl.modeling.roi_heads.roi_heads", WSL + "/wsl/modeling/roi_heads/roi_heads.py")
    ns.oicr = _load("wsl.modeling.roi_heads.fast_rcnn_oicr", WSL + "/wsl/modeling/roi_heads/fast_rcnn_oicr.py")
    ns.wsddn = _load("wsl.modeling.roi_heads.fast_rcnn_wsddn", WSL + "/wsl/modeling/roi_heads/fast_rcnn_wsddn.py")
    ns.box_head = _load("wsl.modeling.roi_heads.box_head", WSL + "/wsl/modeling/roi_heads/box_head.py")
    ns.plus = _load("wsl.modeling.roi_heads.roi_heads_oicrplus", WSL + "/wsl/modeling/roi_heads/roi_heads_oicrplus.py")
    ns.multi = _load("detectron2.modeling.meta_arch.rcnn_multi", REF + "/detectron2/modeling/meta_arch/rcnn_multi.py")
    return ns


def build_reference_model(ns, K=20, dan_dim=(4096, 4096)):
    """Construct the reference MultiInputRCNN with explicit kwargs (voc07_oicr_plus.yaml values)."""
    Matcher = ns.matcher.Matcher
    b2b = ns.box_regression.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    backbone = ns.vgg.VGG16(conv5_dilation=2, freeze_at=2, out_features=["plain5"])
    shape = backbone.output_shape()
    pooler = ns.poolers.ROIPooler(output_size=7, scales=(1.0 / shape["plain5"].stride,), sampling_ratio=0,
                                  pooler_type="ROIPool")
    head = ns.box_head.DiscriminativeAdaptionNeck(ns.shape_spec.ShapeSpec(channels=512, height=7, width=7),
                                                  conv_dims=[], fc_dims=list(dan_dim), conv_norm="")
    common = dict(box2box_transform=b2b, num_classes=K, test_score_thresh=1e-6, test_nms_thresh=0.3,
                  test_topk_per_image=100, loss_weight={"loss_box_reg": 1.0}, mean_loss=True)
    pred = ns.wsddn.WSDDNOutputLayers(head.output_shape, **common)
    refs = [ns.oicr.OICROutputLayers(head.output_shape, refine_k=k, refine_reg=[True] * 4, **common) for k in range(4)]
    cfg = types.SimpleNamespace(WSL=types.SimpleNamespace(REFINE_REG=[True] * 4),
                                OICRPLUS=types.SimpleNamespace(BBOX_UPDATE=False))
    heads = ns.plus.OICRPlusHeads(
        box_in_features=["plain5"], box_pooler=pooler, box_head=head, box_predictor=pred, refine_K=4,
        refine_mist=True, mist_p=0.10, mist_thre=0.05, mist_type="nms", refine_reg=[True] * 4, box_refinery=refs,
        cls_agnostic_bbox_reg=False, pooler_type="ROIPool", cfg=cfg, num_classes=K, batch_size_per_image=4096,
        positive_fraction=1.0, proposal_matcher=Matcher([0.5, 0.6], [0, -1, 1], allow_low_quality_matches=False),
        proposal_append_gt=False)
    model = ns.multi.MultiInputRCNN(backbone=backbone, proposal_generator=None, roi_heads=heads,
                                    pixel_mean=[103.939, 116.779, 123.68], pixel_std=[1.0, 1.0, 1.0],
                                    input_format="BGR", vis_period=0)
    return model
