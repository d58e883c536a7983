"""Import the Stage-3 detector's Python from the reference's SECOND tree (/root/reference/detectron2, v0.4) and from
/root/reference/unbias/ubteacher under torch 2.10, file by file (no package __init__ runs), with the absent third-party
dependencies stubbed — THIS container only; used by make_stage3_golden.py to generate the committed fixtures.  Run it in
its own process: the module names collide with ref_shim.py's (the first tree is also called `detectron2`).

Third-party restatements (dependency -> what is restated):
  * torchvision.ops.roi_align        -> oracle/roialign_oracle.c (the arithmetic stated in-tree at
                                        uwsod/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:20-400)
  * torchvision.ops.nms / boxes.batched_nms -> greedy NMS below (torch ops), as in ref_shim.py
  * fvcore.nn.smooth_l1_loss         -> below (beta == 0 -> L1);  fvcore.nn.weight_init.* -> torch inits (init only)
  * fvcore.common.registry.Registry  -> dict-backed stand-in (decorator + get)
  * detectron2.config.configurable   -> restated decorator: a `cfg` first argument goes through the class's own
                                        `from_config`, explicit keyword arguments pass through
"""
import functools
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

D2 = "/root/reference/detectron2/detectron2"
UB = "/root/reference/unbias/ubteacher"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import frcnn_oracle as FO  # noqa: E402


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


def configurable(init_func=None, *, from_config=None):
    assert init_func is not None and from_config is None

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        cfg = args[0] if args else kwargs.get("cfg")
        if cfg is not None and hasattr(cfg, "MODEL"):                      # called with a config: the class's own from_config
            explicit = type(self).from_config(*args, **kwargs)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)
    return wrapped


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
    fvnn.giou_loss = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    wi = _pkg("fvcore.nn.weight_init")

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    def c2_xavier_fill(m):
        nn.init.kaiming_uniform_(m.weight, a=1)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    wi.c2_msra_fill = c2_msra_fill; wi.c2_xavier_fill = c2_xavier_fill
    fvd = _pkg("fvcore.nn.distributed"); fvd.differentiable_all_reduce = None
    reg = _pkg("fvcore.common.registry"); reg.Registry = Registry
    hb = _pkg("fvcore.common.history_buffer")

    class HistoryBuffer:
        def __init__(self, max_length=1000000):
            self._d = []

        def update(self, v, it=None):
            self._d.append((v, it))

        def latest(self):
            return self._d[-1][0]
    hb.HistoryBuffer = HistoryBuffer
    fio = _pkg("fvcore.common.file_io"); fio.PathManager = object
    _pkg("iopath"); _pkg("iopath.common"); iof = _pkg("iopath.common.file_io"); iof.PathManager = object

    # ---- torchvision
    tv = _pkg("torchvision"); tvo = _pkg("torchvision.ops"); tvb = _pkg("torchvision.ops.boxes")
    tv.__version__ = "0.8.2"                                  # the pin of the reference's environment (README); only compared with 0.7

    def roi_align(input, boxes, output_size, spatial_scale, sampling_ratio, aligned):
        assert aligned and sampling_ratio == 0 and tuple(output_size) == (7, 7)
        return FO._RoIAlignFn.apply(input, boxes, spatial_scale)
    tvo.roi_align = roi_align
    tvo.RoIPool = None

    def nms(boxes, scores, thr):
        order = scores.argsort(descending=True, stable=True)
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

    # ---- the second tree as bare namespaces
    _pkg("detectron2", D2)
    _pkg("detectron2.utils", D2 + "/utils")
    env = _load("detectron2.utils.env", D2 + "/utils/env.py")
    ufio = _pkg("detectron2.utils.file_io"); ufio.PathManager = object
    ev = _load("detectron2.utils.events", D2 + "/utils/events.py")
    _load("detectron2.utils.memory", D2 + "/utils/memory.py")
    comm = _pkg("detectron2.utils.comm"); comm.get_world_size = lambda: 1
    r = _pkg("detectron2.utils.registry"); r.Registry = Registry
    lg = _pkg("detectron2.utils.logger"); lg.log_first_n = lambda *a, **k: None
    cfgm = _pkg("detectron2.config"); cfgm.configurable = configurable

    lay = _pkg("detectron2.layers", D2 + "/layers")
    w = _load("detectron2.layers.wrappers", D2 + "/layers/wrappers.py")
    ss = _load("detectron2.layers.shape_spec", D2 + "/layers/shape_spec.py")
    bn = _load("detectron2.layers.batch_norm", D2 + "/layers/batch_norm.py")
    bl = _load("detectron2.layers.blocks", D2 + "/layers/blocks.py")
    ra = _load("detectron2.layers.roi_align", D2 + "/layers/roi_align.py")
    sys.modules["detectron2"]._C = None
    for n in ["Conv2d", "cat", "nonzero_tuple", "cross_entropy"]:
        setattr(lay, n, getattr(w, n))
    lay.ShapeSpec = ss.ShapeSpec
    lay.FrozenBatchNorm2d = bn.FrozenBatchNorm2d; lay.get_norm = bn.get_norm
    lay.CNNBlockBase = bl.CNNBlockBase
    lay.ROIAlign = ra.ROIAlign
    lay.ROIAlignRotated = type("ROIAlignRotated", (nn.Module,), {})
    lay.DeformConv = lay.ModulatedDeformConv = None

    def d2_batched_nms(boxes, scores, idxs, iou_threshold):                 # D2/layers/nms.py:19-38 (the < 40000 boxes branch)
        assert boxes.shape[-1] == 4 and len(boxes) < 40000
        return batched_nms(boxes.float(), scores, idxs, iou_threshold)
    lay.batched_nms = d2_batched_nms

    st = _pkg("detectron2.structures", D2 + "/structures")
    b = _load("detectron2.structures.boxes", D2 + "/structures/boxes.py")
    il = _load("detectron2.structures.image_list", D2 + "/structures/image_list.py")
    ins = _load("detectron2.structures.instances", D2 + "/structures/instances.py")
    st.Boxes = b.Boxes; st.BoxMode = b.BoxMode; st.pairwise_iou = b.pairwise_iou
    st.ImageList = il.ImageList; st.Instances = ins.Instances
    st.RotatedBoxes = type("RotatedBoxes", (), {}); st.ROIMasks = type("ROIMasks", (), {})

    _pkg("detectron2.data"); du = _pkg("detectron2.data.detection_utils"); du.convert_image_to_rgb = None
    _pkg("detectron2.modeling", D2 + "/modeling")
    mt = _load("detectron2.modeling.matcher", D2 + "/modeling/matcher.py")
    br = _load("detectron2.modeling.box_regression", D2 + "/modeling/box_regression.py")
    sp = _load("detectron2.modeling.sampling", D2 + "/modeling/sampling.py")
    ag = _load("detectron2.modeling.anchor_generator", D2 + "/modeling/anchor_generator.py")
    pl = _load("detectron2.modeling.poolers", D2 + "/modeling/poolers.py")
    pp = _load("detectron2.modeling.postprocessing", D2 + "/modeling/postprocessing.py")     # the real detector_postprocess (eval mode)
    bb = _pkg("detectron2.modeling.backbone", D2 + "/modeling/backbone")
    bbb = _load("detectron2.modeling.backbone.backbone", D2 + "/modeling/backbone/backbone.py")
    bbuild = _pkg("detectron2.modeling.backbone.build"); bbuild.BACKBONE_REGISTRY = Registry("BACKBONE")
    bb.Backbone = bbb.Backbone; bb.build_backbone = None
    rn = _load("detectron2.modeling.backbone.resnet", D2 + "/modeling/backbone/resnet.py")
    rg = _pkg("detectron2.modeling.backbone.regnet"); rg.build_regnet_backbone = None
    fpn = _load("detectron2.modeling.backbone.fpn", D2 + "/modeling/backbone/fpn.py")
    pg = _pkg("detectron2.modeling.proposal_generator", D2 + "/modeling/proposal_generator")
    pgb = _pkg("detectron2.modeling.proposal_generator.build"); pgb.PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
    pu = _load("detectron2.modeling.proposal_generator.proposal_utils", D2 + "/modeling/proposal_generator/proposal_utils.py")
    rpn = _load("detectron2.modeling.proposal_generator.rpn", D2 + "/modeling/proposal_generator/rpn.py")
    pg.RPN = rpn.RPN; pg.build_proposal_generator = None
    rh = _pkg("detectron2.modeling.roi_heads", D2 + "/modeling/roi_heads")
    bh = _load("detectron2.modeling.roi_heads.box_head", D2 + "/modeling/roi_heads/box_head.py")
    fr = _load("detectron2.modeling.roi_heads.fast_rcnn", D2 + "/modeling/roi_heads/fast_rcnn.py")
    for sub, names in [("keypoint_head", ["build_keypoint_head"]), ("mask_head", ["build_mask_head"])]:
        m = _pkg("detectron2.modeling.roi_heads." + sub)
        for n in names:
            setattr(m, n, None)
    rhs = _load("detectron2.modeling.roi_heads.roi_heads", D2 + "/modeling/roi_heads/roi_heads.py")
    rh.ROI_HEADS_REGISTRY = rhs.ROI_HEADS_REGISTRY; rh.StandardROIHeads = rhs.StandardROIHeads
    ma = _pkg("detectron2.modeling.meta_arch", D2 + "/modeling/meta_arch")
    mab = _pkg("detectron2.modeling.meta_arch.build"); mab.META_ARCH_REGISTRY = Registry("META_ARCH")
    bb.build_backbone = None
    sys.modules["detectron2.modeling.roi_heads"].build_roi_heads = None
    rcnn = _load("detectron2.modeling.meta_arch.rcnn", D2 + "/modeling/meta_arch/rcnn.py")

    # ---- ubteacher's modeling files
    _pkg("ubteacher", UB); _pkg("ubteacher.modeling", UB + "/modeling")
    _pkg("ubteacher.modeling.meta_arch", UB + "/modeling/meta_arch")
    _pkg("ubteacher.modeling.proposal_generator", UB + "/modeling/proposal_generator")
    _pkg("ubteacher.modeling.roi_heads", UB + "/modeling/roi_heads")
    ufr = _load("ubteacher.modeling.roi_heads.fast_rcnn", UB + "/modeling/roi_heads/fast_rcnn.py")
    urh = _load("ubteacher.modeling.roi_heads.roi_heads", UB + "/modeling/roi_heads/roi_heads.py")
    urpn = _load("ubteacher.modeling.proposal_generator.rpn", UB + "/modeling/proposal_generator/rpn.py")
    urcnn = _load("ubteacher.modeling.meta_arch.rcnn", UB + "/modeling/meta_arch/rcnn.py")

    return types.SimpleNamespace(events=ev, shape_spec=ss, boxes=b, instances=ins, image_list=il, matcher=mt, box_regression=br,
                                 sampling=sp, anchor_generator=ag, poolers=pl, resnet=rn, fpn=fpn, rpn=rpn, proposal_utils=pu,
                                 box_head=bh, fast_rcnn=fr, roi_heads=rhs, rcnn=rcnn, ub_fast_rcnn=ufr, ub_roi_heads=urh,
                                 ub_rpn=urpn, ub_rcnn=urcnn, env=env)


class _Cfg(types.SimpleNamespace):
    pass


def build_reference_model(ns, K=20):
    """TwoStagePseudoLabGeneralizedRCNN with explicit arguments = unbias/configs/code_release/voc_ssod.yaml over
    Base-RCNN-FPN.yaml and the v0.4 defaults."""
    SS = ns.shape_spec.ShapeSpec
    stem = ns.resnet.BasicStem(in_channels=3, out_channels=64, norm="FrozenBN")
    stages = ns.resnet.ResNet.make_default_stages(50, norm="FrozenBN", stride_in_1x1=True)
    bottom_up = ns.resnet.ResNet(stem, stages, out_features=["res2", "res3", "res4", "res5"], freeze_at=2)
    backbone = ns.fpn.FPN(bottom_up=bottom_up, in_features=["res2", "res3", "res4", "res5"], out_channels=256, norm="",
                          top_block=ns.fpn.LastLevelMaxPool(), fuse_type="sum")
    shapes = backbone.output_shape()
    in_feats = ["p2", "p3", "p4", "p5", "p6"]
    anchor_gen = ns.anchor_generator.DefaultAnchorGenerator(sizes=[[32], [64], [128], [256], [512]], aspect_ratios=[[0.5, 1.0, 2.0]],
                                                            strides=[shapes[f].stride for f in in_feats], offset=0.0)
    head = ns.rpn.StandardRPNHead(in_channels=256, num_anchors=3, box_dim=4)
    rpn = ns.ub_rpn.PseudoLabRPN(
        in_features=in_feats, head=head, anchor_generator=anchor_gen,
        anchor_matcher=ns.matcher.Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True),
        box2box_transform=ns.box_regression.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0)), batch_size_per_image=256,
        positive_fraction=0.25, pre_nms_topk=(2000, 1000), post_nms_topk=(1000, 1000), nms_thresh=0.7, min_box_size=0.0,
        anchor_boundary_thresh=-1.0, loss_weight={"loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0}, box_reg_loss_type="smooth_l1",
        smooth_l1_beta=0.0)
    box_in = ["p2", "p3", "p4", "p5"]
    pooler = ns.poolers.ROIPooler(output_size=7, scales=tuple(1.0 / shapes[f].stride for f in box_in), sampling_ratio=0,
                                  pooler_type="ROIAlignV2")
    box_head = ns.box_head.FastRCNNConvFCHead(SS(channels=256, height=7, width=7), conv_dims=[], fc_dims=[1024, 1024])
    cfg = _Cfg(MODEL=_Cfg(ROI_HEADS=_Cfg(NUM_CLASSES=K, SCORE_THRESH_TEST=0.05, NMS_THRESH_TEST=0.5),
                          ROI_BOX_HEAD=_Cfg(BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0), CLS_AGNOSTIC_BBOX_REG=False, SMOOTH_L1_BETA=0.0,
                                            BBOX_REG_LOSS_TYPE="smooth_l1", BBOX_REG_LOSS_WEIGHT=1.0)),
               TEST=_Cfg(DETECTIONS_PER_IMAGE=100))
    predictor = ns.ub_fast_rcnn.FastRCNNFocaltLossOutputLayers(cfg, box_head.output_shape)
    heads = ns.ub_roi_heads.StandardROIHeadsPseudoLab(
        box_in_features=box_in, box_pooler=pooler, box_head=box_head, box_predictor=predictor, num_classes=K,
        batch_size_per_image=512, positive_fraction=0.25,
        proposal_matcher=ns.matcher.Matcher([0.5], [0, 1], allow_low_quality_matches=False), proposal_append_gt=True)
    model = ns.ub_rcnn.TwoStagePseudoLabGeneralizedRCNN(backbone=backbone, proposal_generator=rpn, roi_heads=heads,
                                                        pixel_mean=list(FO.PIXEL_MEAN), pixel_std=list(FO.PIXEL_STD),
                                                        input_format="BGR", vis_period=0)
    return model
