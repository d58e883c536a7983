"""OICR+ ROI heads on gfx950 behind the reference's interface
(uwsod/projects/WSL/wsl/modeling/roi_heads/roi_heads_oicrplus.py:37-757 `OICRPlusHeads`,
helpers from roi_heads.py:144-164,225-375).

Same constructor keys / child-module names (`box_pooler`, `box_head`, `box_predictor`, `box_refinery_{k}`), same
call: `roi_heads(images_list, features_list, proposals_list, targets_list) -> (None, losses)` in training.

What differs is the execution plan (MI355X-first, not a translation):
  * the 4 views are stacked to one (4R)-row problem, so fc6/fc7 and the 10 predictor matrices run as 3 MFMA
    GEMMs (the predictor matrices are packed into one (22K+4) x 4096 operand);
  * ROIPool writes the fc6 input directly ((R,C,7,7) order, objectness prior fused);
  * MIL scoring, pseudo-GT mining (top-p%, threshold, NMS), IoU labelling and the refinement losses are one
    kernel each, entirely device side: no .item()/nonzero host syncs inside the iteration;
  * the forward is ONE block with an explicit backward (no autograd tape over ~200 ops), cut into three autograd
    nodes only so that the gradients are released in the order they are produced (see _HeadsPoolFunction);
    the loss kernels emit unit logit-gradients in the forward sweep, the backward only scales them by the
    incoming cotangents.
Reference quirks kept on purpose: losses_k2_flip pairs predictions_k2 with the flipped targets
(#1), pseudo-GT weights = scores (#2), class-agnostic NMS at 0.01 (#3), int() truncation of R*0.1 and rank-0
always kept (#4), CE mean over all R / box loss / R, L1 (#5), targets are proposals[gt_index] (#6)."""
from typing import Dict, List

import os
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .box_head import DiscriminativeAdaptionNeck  # noqa: F401  (registers the box head the configs name)
from .events import get_event_storage, has_event_storage
from .fast_rcnn_oicr import OICROutputLayers
from .fast_rcnn_wsddn import WSDDNOutputLayers
from .poolers import ROIPooler
from .registry import ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY
from .structures import ShapeSpec

LOSS_NAMES_FMT = ["loss_cls"]


def get_image_level_gt(targets, num_classes):
    """roi_heads.py:144-164 — host side (a handful of ints; done before anything is queued on the GPU)."""
    if targets is None:
        return None, None, None
    cls, ints, ohs = [], [], []
    for t in targets:
        g = t.gt_classes
        u = np.unique(g.detach().cpu().numpy().astype(np.int64))
        oh = np.zeros((1, num_classes), np.float32)
        oh[0, u] = 1.0
        cls.append(torch.from_numpy(u)); ints.append(torch.from_numpy(u)); ohs.append(torch.from_numpy(oh))
    return cls, ints, torch.cat(ohs, dim=0)


def _padded(rows, cols, device, dtype, pad=128):
    """rows x cols matrix whose row pitch is NOT a multiple of 1 KiB.  The 4096- and 25088-wide bf16 matrices of the box
    head have 8 KiB / 49 KiB pitches: the rows of a GEMM tile then start on the same HBM channel / L2 set and the
    K-tile loads queue behind each other (measured: fc7 fwd 267 -> 221 us, fc6 dgrad 1840 -> 1684 us; tools/probes/gemm_ld_sweep*.py)."""
    return torch.empty(rows, cols + pad, device=device, dtype=dtype)[:, :cols]


class LossDict(dict):
    """The reference's dict of named scalar losses; every value is a view of ONE device vector, kept as `.vector`.
    `total()` = sum of all losses (what train_net_multi.py:129 computes with Python's sum()), produced by the same kernel
    that finalises the vector — a second output of the heads' autograd node, so `total().backward(...)` runs no torch
    arithmetic at all."""

    def __init__(self, names, vector, total=None, finite=None):
        super().__init__({n: vector[i] for i, n in enumerate(names)})
        self.vector = vector
        self._total, self._finite = total, finite

    def total(self):
        return self.vector.sum() if self._total is None else self._total.view(())      # a view: its backward is a reshape

    def finite_flag(self):
        """device scalar: 1.0 if the summed loss is finite (written by the kernel that finalises the losses)"""
        return None if self._finite is None else self._finite.view(())


def loss_names(refine_K):
    names = ["loss_cls"]
    for k in range(refine_K):
        names += [f"loss_cls_r{k}", f"loss_box_reg_r{k}"]
    return names


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def _current_rank():
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def derive_dropout_seed(base, rank):
    """per-rank dropout stream from the rank-independent base seed (engine/defaults.py:147 seeds SEED + rank)"""
    return _splitmix64(_splitmix64(base & 0xFFFFFFFFFFFFFFFF) ^ (0x5051 + rank))


class _DropoutStream:
    """Checkpointable dropout stream.  The file is written by rank 0 only (DetectionCheckpointer.save), so what it holds is the
    RANK-INDEPENDENT base seed + the stream position; every rank re-derives its own seed from (base, its rank) on load — a
    resumed multi-rank run keeps one stream per rank (rank 0 continues its stream exactly; the others restart theirs at rank 0's
    position, which differs from their own only when the ranks saw different proposal counts)."""

    def __init__(self, heads):
        self.heads = heads

    def state_dict(self):
        self.heads._dropout_stream_seed()
        return {"base_seed": self.heads._dropout_base, "dropout_seed": self.heads.dropout_seed,
                "drop_counter": int(self.heads._drop_counter)}

    def load_state_dict(self, state):
        if "base_seed" in state:
            self.heads._dropout_base = int(state["base_seed"])
            self.heads.dropout_seed = derive_dropout_seed(self.heads._dropout_base, _current_rank())
        elif "dropout_seed" in state:                  # files written before the base seed was stored: rank 0's seed
            rank = _current_rank()
            self.heads.dropout_seed = state["dropout_seed"] if rank == 0 else derive_dropout_seed(state["dropout_seed"], rank)
        self.heads._drop_counter = int(state.get("drop_counter", self.heads._drop_counter))


# The heads are ONE forward computation and a backward in three stages, each its own autograd node, so that the gradients leave
# in the order they are produced and a data-parallel reducer (DDP hooks fire when a node returns) can start on them while the
# rest of the backward still runs:
#   _HeadsLossFunction  (handle, fc7 + predictor params) -> (loss vector, total)   backward: logits -> ... -> dZ1; releases the
#                                                                                  predictor / fc7 gradients (68 MB)
#   _HeadsFc6Function   (handle, fc6 weight, bias)       -> handle                 backward: fc6 weight gradient; releases the 411 MB
#                                                                                  that dominate the all-reduce
#   _HeadsPoolFunction  (feat_0 .. feat_{n-1})           -> handle                 backward: fc6 data gradient + ROIPool backward
# The 1.6 ms of the last stage and the whole conv backward then overlap fc6's all-reduce; as one node every gradient of the
# heads became ready at the same instant, after the ROIPool backward.  The forward runs entirely inside the first node (the
# parameters reach it in a plain list); the handles are 1-element tensors that only carry the graph edges, the activations
# travel in the shared state `box[0]`.
class _HeadsPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heads, inp, params, box, *feats):
        st = heads._train_forward(inp, feats, params)
        box[0] = st
        ctx.heads, ctx.box = heads, box
        ctx.feat_req = tuple(ctx.needs_input_grad[4:])
        ctx.set_materialize_grads(False)
        return heads._handle(feats[0].device, fresh=True)

    @staticmethod
    def backward(ctx, g_handle):
        st, ctx.box[0] = ctx.box[0], None
        if g_handle is None or st is None or "dz1" not in st:
            return (None,) * (4 + len(ctx.feat_req))
        return (None, None, None, None) + tuple(ctx.heads._train_backward_pool(st, ctx.feat_req))


class _HeadsFc6Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heads, box, handle, W1, b1):
        ctx.heads, ctx.box = heads, box
        ctx.req = tuple(ctx.needs_input_grad[2:5])
        ctx.set_materialize_grads(False)
        return heads._handle(W1.device, fresh=True)

    @staticmethod
    def backward(ctx, g_handle):
        st = ctx.box[0]
        if g_handle is None or st is None or "dz1" not in st:
            return (None,) * 5
        # fused update (Trainer sets heads._fused_opt on the single-GPU, ITER_SIZE 1 path): fc1.weight's SGD step runs in the epilogue of its
        # weight-gradient GEMM and rewrites the bf16 copies the fc6 DATA gradient reads — so that GEMM is deferred until the data gradient
        # has been computed (stage 3), and fc1.weight receives no .grad (the optimizer's step() skips it)
        dW1, db1 = ctx.heads._train_backward_fc6(st, defer_fused=bool(ctx.req[0]))
        if not ctx.req[0]:
            ctx.box[0] = None                      # no feature gradient wanted: this is the last stage
        return (None, None, ctx.heads._handle(db1.device) if ctx.req[0] else None,
                dW1 if ctx.req[1] else None, db1 if ctx.req[2] else None)


class _HeadsLossFunction(torch.autograd.Function):
    """(handle, *fc7 and predictor params) -> (vector of 1 + 2*refine_K losses, their sum)."""

    @staticmethod
    def forward(ctx, heads, box, handle, *params):
        st = box[0]
        ctx.heads, ctx.box, ctx.params = heads, box, params
        ctx.set_materialize_grads(False)
        return st["losses"], st["total"]

    @staticmethod
    def backward(ctx, g_losses, g_total):
        st = ctx.box[0]
        if g_losses is None and g_total is None:
            return (None,) * (3 + len(ctx.params))
        if st is None or "dz1" in st:
            raise RuntimeError("OICRPlusHeads: the activations of this iteration were released by its first backward pass "
                               "(a second backward through the same forward is not supported)")
        dparams = ctx.heads._train_backward_top(st, ctx.params, g_losses, g_total)
        return (None, None, ctx.heads._handle(st["pooled"].device)) + tuple(dparams)


@ROI_HEADS_REGISTRY.register()
class OICRPlusHeads(nn.Module):
    def __init__(self, *, box_in_features: List[str], box_pooler: ROIPooler, box_head: nn.Module,
                 box_predictor: nn.Module, refine_K: int = 4, refine_mist: bool = True, mist_p: float = 0.10,
                 mist_thre: float = 0.05, mist_type: str = "nms", refine_reg=None, box_refinery=None,
                 cls_agnostic_bbox_reg: bool = False, pooler_type: str = "ROIPool", cfg=None, num_classes: int = 20,
                 iou_thresholds=(0.5, 0.6), iou_labels=(0, -1, 1), bbox_reg_weights=(10.0, 10.0, 5.0, 5.0),
                 test_score_thresh=1e-6, test_nms_thresh=0.3, test_topk_per_image=100,
                 compute_dtype=torch.bfloat16, seed=-1, **unused):
        super().__init__()
        assert refine_mist and mist_type == "nms", "WSL.REFINE_MIST True / MIST_TYPE nms is the configured path"
        assert list(iou_labels) == [0, -1, 1] and len(iou_thresholds) == 2
        assert pooler_type == "ROIPool" and not cls_agnostic_bbox_reg
        self.box_in_features = box_in_features
        self.in_features = box_in_features
        self.box_pooler = box_pooler
        self.box_head = box_head
        self.box_predictor = box_predictor
        self.refine_K = refine_K
        self.mist_p, self.mist_thre = mist_p, mist_thre
        self.refine_reg = refine_reg if refine_reg is not None else [True] * refine_K
        self.box_refinery = []
        for k in range(refine_K):
            self.add_module("box_refinery_{}".format(k), box_refinery[k])     # roi_heads_oicrplus.py:76-79
            self.box_refinery.append(box_refinery[k])
        self.num_classes = num_classes
        self.iou_thresholds = tuple(float(t) for t in iou_thresholds)
        self.bbox_reg_weights = tuple(float(w) for w in bbox_reg_weights)
        self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image = test_score_thresh, test_nms_thresh, test_topk_per_image
        self.compute_dtype = compute_dtype
        self.cfg = cfg
        self.iter = 0
        # dropout stream = hash(dropout_seed, element counter): the seed is drawn from SEED and the rank on first use (every
        # rank its own stream, engine/defaults.py:147 seeds SEED + rank), both travel in the state dict (extra state)
        self.seed = int(seed)
        self.train_dropout = True         # tests: switch the fc6 / fc7 dropout off (box_head.py:90 always drops 0.5 in training)
        self.dropout_seed = None
        self._dropout_base = None         # rank-independent seed the per-rank stream seeds derive from (checkpointed)
        self._drop_counter_host = 0       # stream position; lives in a device scalar once the first training forward ran, so that
        self._drop_ctr_dev = None         # a captured hipGraph of the step draws a fresh mask on every replay
        self._prestaged_labels = None     # graph capture / replay: (static device buffer, per-image class counts), see stage_labels
        self._stage_cache = {}            # name -> (key list, persistent compute-dtype weight copy)
        # fp32 mode only: the fc6 / fc7 GEMMs (forward, data and weight gradient: 97 % of the fp32 step's MFMA time) as six-product
        # bf16x3 GEMMs — about f32 accuracy at 16/6 of the exact-f32 MFMA rate (ops.gemm_f32x3).  Off by default: the parity mode
        # stays the exact-f32 MFMA; MODEL.AMD.FP32_GEMM "bf16x3" / SW_FP32X3=1 turn it on.
        self.fp32x3 = compute_dtype == torch.float32 and os.environ.get("SW_FP32X3", "0") == "1"
        self._x3_cache = {}
        self.debug_drop_masks = None      # tests: [[m1, m2] per view] uint8 keep masks (A.2 #9)
        self.last_aux = None              # tests / metrics: device tensors of the last iteration
        K = num_classes
        self.n_head_cols = 2 * K + refine_K * (5 * K + 1)
        self.ld_head = (self.n_head_cols + 7) // 8 * 8

    @property
    def gt_classes_img_int(self):
        """roi_heads_oicrplus.py:206: the image-level classes of the last training forward, int64 per image (built on access:
        the step itself launches no conversion kernel)"""
        v = self.__dict__.get("_gt_int64")
        if v is None and self.__dict__.get("_gt_int32") is not None:
            v = [g.to(torch.int64) for g in self._gt_int32]
        return v

    @gt_classes_img_int.setter
    def gt_classes_img_int(self, value):
        self.__dict__["_gt_int64"] = value
        self.__dict__["_gt_int32"] = None

    # ------------------------------------------------------------------ dropout stream state
    def _dropout_stream_seed(self):
        if self.dropout_seed is None:
            if self._dropout_base is None:
                self._dropout_base = self.seed if self.seed >= 0 else torch.initial_seed()
            self.dropout_seed = derive_dropout_seed(self._dropout_base, _current_rank())
        elif self._dropout_base is None:
            self._dropout_base = self.seed if self.seed >= 0 else torch.initial_seed()
        return self.dropout_seed

    @property
    def _drop_counter(self):
        if self._drop_ctr_dev is not None:
            return int(self._drop_ctr_dev.item())
        return self._drop_counter_host

    @_drop_counter.setter
    def _drop_counter(self, value):
        self._drop_counter_host = int(value)
        if self._drop_ctr_dev is not None:
            self._drop_ctr_dev.fill_(int(value))

    def _drop_counter_device(self, device):
        if self._drop_ctr_dev is None or self._drop_ctr_dev.device != device:
            self._drop_ctr_dev = torch.full((1,), int(self._drop_counter_host), dtype=torch.int64, device=device)
        return self._drop_ctr_dev

    @property
    def dropout_stream(self):
        """checkpointable view of the dropout stream position (`state_dict()` / `load_state_dict()`): DetectionCheckpointer
        stores it next to "model" / "optimizer" / "scheduler", so the model's own state dict keeps exactly the reference's keys"""
        return _DropoutStream(self)

    # ------------------------------------------------------------------ construction from cfg
    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        from .backbone_vgg import _dtype_from_cfg
        dtype = _dtype_from_cfg(cfg)
        # keys the reference acts on that this path does not implement: refuse instead of training with other numerics
        assert not cfg.get("OICRPLUS", {}).get("BBOX_UPDATE", False), "OICRPLUS.BBOX_UPDATE True is not implemented"
        assert float(cfg.MODEL.ROI_BOX_HEAD.get("BBOX_REG_LOSS_WEIGHT", 1.0)) == 1.0, "BBOX_REG_LOSS_WEIGHT != 1 is not implemented"
        assert cfg.MODEL.ROI_BOX_HEAD.get("BBOX_REG_LOSS_TYPE", "smooth_l1") == "smooth_l1" and \
            float(cfg.MODEL.ROI_BOX_HEAD.get("SMOOTH_L1_BETA", 0.0)) == 0.0, "only smooth_l1 with beta 0 (= L1) is implemented"
        assert all(cfg.WSL.REFINE_REG), "WSL.REFINE_REG must be all True (every refinement head regresses boxes)"
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        ch = input_shape[in_features[0]].channels
        pooler = ROIPooler(output_size=res, scales=scales, sampling_ratio=cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                           pooler_type=cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE)
        head_cls = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)
        box_head = head_cls(**head_cls.from_config(cfg, ShapeSpec(channels=ch, height=res, width=res)))
        predictor = WSDDNOutputLayers(**WSDDNOutputLayers.from_config(cfg, box_head.output_shape))
        refinery = [OICROutputLayers(**OICROutputLayers.from_config(cfg, box_head.output_shape, k))
                    for k in range(cfg.WSL.REFINE_NUM)]
        return dict(box_in_features=in_features, box_pooler=pooler, box_head=box_head, box_predictor=predictor,
                    refine_K=cfg.WSL.REFINE_NUM, refine_mist=cfg.WSL.REFINE_MIST, mist_p=cfg.WSL.MIST_P,
                    mist_thre=cfg.WSL.MIST_THRE, mist_type=cfg.WSL.MIST_TYPE, refine_reg=cfg.WSL.REFINE_REG,
                    box_refinery=refinery, cls_agnostic_bbox_reg=cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
                    pooler_type=cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE, cfg=cfg, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                    iou_thresholds=cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, iou_labels=cfg.MODEL.ROI_HEADS.IOU_LABELS,
                    bbox_reg_weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS,
                    test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                    test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
                    test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE, compute_dtype=dtype, seed=cfg.get("SEED", -1))

    # ------------------------------------------------------------------ parameter packing
    def _flat_params(self):
        ps = [self.box_head.fc1.weight, self.box_head.fc1.bias, self.box_head.fc2.weight, self.box_head.fc2.bias,
              self.box_predictor.cls.weight, self.box_predictor.cls.bias, self.box_predictor.det.weight,
              self.box_predictor.det.bias]
        for r in self.box_refinery:
            ps += [r.cls_score.weight, r.cls_score.bias, r.bbox_pred.weight, r.bbox_pred.bias]
        return ps

    def _col_layout(self):
        """column ranges of the packed predictor matrix: [cls K | det K | k x (cls_score K+1 | bbox_pred 4K)]"""
        K = self.num_classes
        cols = {"cls": 0, "det": K}
        for k in range(self.refine_K):
            base = 2 * K + k * (5 * K + 1)
            cols[f"cls_score{k}"] = base
            cols[f"bbox_pred{k}"] = base + K + 1
        return cols

    def _flatten_head_params(self, device):
        """Re-home the 10 predictor weights / biases as row slices of ONE (ld_head, 4096) f32 matrix and ONE bias vector
        (the way cuDNN RNNs / DDP buckets flatten weights): names, shapes, optimizer and state_dict are unchanged, but the
        fused predictor GEMM's operand is then a single staging copy and its bias needs none.  Done lazily and re-done
        whenever something (`.to()`, `load_state_dict(assign=True)`) gave the parameters new storage."""
        ps = self._flat_params()[4:]
        D = ps[0].shape[1]
        flat_w = torch.zeros(self.ld_head, D, device=device, dtype=torch.float32)
        flat_b = torch.zeros(self.ld_head, device=device, dtype=torch.float32)
        row = 0
        with torch.no_grad():
            for i in range(0, len(ps), 2):
                w, b = ps[i], ps[i + 1]
                n = w.shape[0]
                flat_w[row:row + n].copy_(w); flat_b[row:row + n].copy_(b)
                w.data = flat_w[row:row + n]; b.data = flat_b[row:row + n]
                row += n
        assert row == self.n_head_cols
        self._head_flat = (flat_w, flat_b)
        ops.invalidate_all_staged()

    def _head_flat_ok(self, params, device):
        flat = getattr(self, "_head_flat", None)
        if flat is None or flat[0].device != device:
            return False
        row, esz = 0, 4
        for i in range(4, len(params), 2):
            w, b = params[i], params[i + 1]
            if w.dtype != torch.float32 or w.data_ptr() != flat[0].data_ptr() + row * flat[0].shape[1] * esz or \
                    b.data_ptr() != flat[1].data_ptr() + row * esz:
                return False
            row += w.shape[0]
        return True

    def _pack_head_weights(self, params, device):
        """10 (out, 4096) f32 masters -> one (ld_head, 4096) compute-dtype operand + f32 bias vector.  The operand is a
        persistent buffer registered in ops.STAGING per predictor weight (row slice), so after an optimizer step it is
        already current."""
        if not self._head_flat_ok(params, device):
            self._flatten_head_params(device)
            self._stage_cache.pop("heads", None)
        flat_w, flat_b = self._head_flat
        ws = [params[i] for i in range(4, len(params), 2)]
        keys = [ops.param_key(w) for w in ws]
        hit = self._stage_cache.get("heads")
        if hit is not None and hit[0] == keys and hit[1].dtype == self.compute_dtype:
            return hit[1], flat_b
        Wh = torch.empty(self.ld_head, flat_w.shape[1], device=device, dtype=self.compute_dtype)
        ops.convert_2d(flat_w, Wh, self.ld_head, flat_w.shape[1])
        self._stage_cache["heads"] = (keys, Wh)
        row = 0
        for j, w in enumerate(ws):
            n = w.shape[0]
            if w.requires_grad:
                ops.register_staging(w, 1, self.compute_dtype, stage0=Wh[row:row + n], d0=Wh.shape[1], ld0=Wh.stride(0),
                                     stamp=lambda pk, j=j, keys=keys: keys.__setitem__(j, pk))
            row += n
        return Wh, flat_b

    def _head_weights_t(self, Wh, device):
        """(D, ld_head) transposed copy of the packed predictor operand for the logits' data gradient, rebuilt (one 64x64-tiled
        launch over 3.6 MB) when a predictor weight changed; None when ld_head is not a multiple of 64 (the transposing kernel's
        tile) — the data gradient then reads Wh K-strided."""
        LD, D = Wh.shape
        if LD % 64 or D % 64 or not hasattr(self, "_head_flat"):
            return None
        keys = list(self._stage_cache["heads"][0])
        hit = self._stage_cache.get("heads_t")
        if hit is not None and hit[0] == keys and hit[1].dtype == Wh.dtype and hit[1].device == device:
            return hit[1]
        buf = hit[1] if (hit is not None and hit[1].dtype == Wh.dtype and hit[1].device == device) else _padded(D, LD, device, Wh.dtype)
        ops.convert_2d_t(self._head_flat[0], buf, LD, D)
        self._stage_cache["heads_t"] = (keys, buf)
        return buf

    def _staged_matrix(self, name, w, device, transposed=False):
        """persistent compute-dtype copy (padded row pitch) of an fc weight; see _pack_head_weights.  transposed=True
        also keeps the (cols, rows) transposed copy: the data-gradient GEMM then reads the weight K-contiguous like the
        forward does (as a K-strided operand of 49 KiB row pitch it ran 25 % slower).  Returns copy or (copy, copy^T)."""
        dt_ = self.compute_dtype
        rows, cols = w.shape
        transposed = transposed and rows % 64 == 0 and cols % 64 == 0
        key = [ops.param_key(w)]
        hit = self._stage_cache.get(name)
        if hit is not None and hit[0] == key and hit[1].dtype == dt_ and hit[1].device == device and \
                (hit[2] is not None or not transposed):
            return (hit[1], hit[2]) if transposed else hit[1]
        reuse = hit is not None and tuple(hit[1].shape) == (rows, cols) and hit[1].dtype == dt_ and hit[1].device == device
        buf = hit[1] if reuse else _padded(rows, cols, device, dt_)
        ops.convert_2d(w.detach(), buf, rows, cols)
        buf_t = None
        if transposed:
            buf_t = hit[2] if (reuse and hit[2] is not None) else _padded(cols, rows, device, dt_)
            ops.convert_2d_t(w.detach(), buf_t, rows, cols)
        self._stage_cache[name] = (key, buf, buf_t)
        if w.requires_grad:
            ops.register_staging(w, 3 if transposed else 1, dt_, stage0=buf, stage1=buf_t, d0=cols, ld0=buf.stride(0),
                                 ld1=0 if buf_t is None else buf_t.stride(0),
                                 stamp=lambda pk, key=key: key.__setitem__(0, pk))
        return (buf, buf_t) if transposed else buf

    def _x3_weight(self, name, w_master, staged_f32):
        """the three-piece bf16 operand (B side, K along the columns) of a staged f32 weight matrix, rebuilt when the master changed"""
        key = ops.param_key(w_master)
        hit = self._x3_cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, ops.split_bf16x3(staged_f32, 1, out=None if hit is None else hit[1]))
            self._x3_cache[name] = hit
        return hit[1]

    def _fc_gemm(self, A, B, C, M, N, K, ep=None, tag=None, a_kstrided=False, b_kstrided=False, wname=None, wmaster=None):
        """an fc-family GEMM: the compute dtype's own sw_gemm, or (fp32 mode with fp32x3) the six-product bf16x3 form; wname / wmaster:
        B is a staged weight whose split is cached per update"""
        if self.fp32x3 and A.dtype == torch.float32 and K % 8 == 0 and M % 4 == 0 and N % 4 == 0:
            B3 = self._x3_weight(wname, wmaster, B) if wname is not None else None
            return ops.gemm_f32x3(A, B, C, M, N, K, a_kstrided=a_kstrided, b_kstrided=b_kstrided, ep=ep, tag=tag, B3=B3)
        return ops.gemm(A, B, C, M, N, K, a_kstrided=a_kstrided, b_kstrided=b_kstrided, ep=ep, tag=tag)

    # ------------------------------------------------------------------ training forward (explicit)
    def _train_forward(self, inp, feats, params):
        """feats: 2 NHWC maps per image (scale 1, scale 2), each a batch of 2 (view, flipped view).  Rows of every stacked
        matrix are image-major: image b owns rows [off_b, off_b + 4 R_b), view v of it [off_b + v R_b, off_b + (v+1) R_b)."""
        dt_ = self.compute_dtype
        K, V, RK = self.num_classes, 4, self.refine_K
        B, Rs, offs, M = inp["B"], inp["R"], inp["off"], inp["M"]
        feats = [f if f.dtype == dt_ else f.to(dt_) for f in feats]
        dev = feats[0].device
        C = feats[0].shape[3]
        P = self.box_pooler.output_size
        D0 = C * P * P
        # --- ROIPool (+ objectness prior fused) straight into the stacked fc6 operand
        # row pitch D0 + 64: with the natural 49 KiB pitch the rows of an fc6 operand tile start on 4 of the 16 L2 channels
        pooled = _padded(M, D0, dev, dt_, pad=64)
        argmax = _padded(M, D0, dev, ops.roi_argmax_dtype(max(f.shape[1] for f in feats), max(f.shape[2] for f in feats)),
                         pad=64)
        for b in range(B):
            R, off = Rs[b], offs[b]
            for s in range(2):
                r0 = off + 2 * s * R
                ops.roi_pool_fwd(feats[2 * b + s], inp["rois"][b][s], pooled[r0:r0 + 2 * R], argmax[r0:r0 + 2 * R],
                                 self.box_pooler.scale, P, P, row_scale=inp["obj"][b][2 * s:2 * s + 2].reshape(-1),
                                 row_scale_add=1.0, tag="roi_fwd")
        # --- fc6 / fc7 with bias + ReLU + dropout fused (box_head.py:82-91)
        fc1w, fc1b, fc2w, fc2b = params[0], params[1], params[2], params[3]
        D1, D2 = fc1w.shape[0], fc2w.shape[0]
        training_dropout = self.training and self.train_dropout
        masks = [None, None]
        hashes = [None, None]
        if training_dropout:
            if self.debug_drop_masks is not None:
                assert B == 1
                masks = [torch.cat([self.debug_drop_masks[v][l].to(dev) for v in range(V)], 0).contiguous() for l in range(2)]
            else:                                     # decided inside the fc6 / fc7 epilogues: same stream, no mask tensors
                seed = self._dropout_stream_seed()
                ctr = self._drop_counter_device(dev)
                hashes = [(seed, 0, 0.5, ctr), (seed, M * D1, 0.5, ctr)]       # position = device counter + offset inside the step
        W1 = self._staged_matrix("fc1", fc1w, dev, transposed=inp["need_grad"])
        W1, W1T = W1 if isinstance(W1, tuple) else (W1, None)
        # fc7's weight also keeps a transposed copy when a backward follows: its data gradient then reads the weight K-contiguous
        # (NT form, ping-pong loop) like fc6's — as a K-strided operand it ran 289 us against the forward's 261
        W2 = self._staged_matrix("fc2", fc2w, dev, transposed=inp["need_grad"])
        W2, W2T = W2 if isinstance(W2, tuple) else (W2, None)
        h1 = _padded(M, D1, dev, dt_)
        self._fc_gemm(pooled, W1, h1, M, D1, D0, ep=ops.make_epilogue(bias=fc1b, relu=True, drop_mask=masks[0], drop_hash=hashes[0], out_dtype=dt_),
                      tag="fc6_fwd", wname="fc1", wmaster=fc1w)
        h2 = _padded(M, D2, dev, dt_)
        self._fc_gemm(h1, W2, h2, M, D2, D1, ep=ops.make_epilogue(bias=fc2b, relu=True, drop_mask=masks[1], drop_hash=hashes[1],
                                                              out_dtype=dt_), wname="fc2", wmaster=fc2w)
        if hashes[0] is not None:
            ops.counter_add(hashes[0][3], M * (D1 + D2))                       # in stream order, after both readers
        # --- all 10 predictor matrices as one GEMM, f32 logits
        Wh, bh = self._pack_head_weights(params, dev)
        LD = self.ld_head
        logits = torch.empty(M, LD, device=dev, dtype=torch.float32)
        ops.gemm(h2, Wh, logits, M, LD, D2, ep=ops.make_epilogue(bias=bh, out_dtype=torch.float32))
        cols = self._col_layout()
        n_loss = 1 + 2 * RK
        K1 = K + 1
        col_stride = 5 * K + 1
        loss_view = torch.empty(B, n_loss, V, device=dev, dtype=torch.float32)     # every entry is written below
        # unit logit gradients: the loss kernels write every real column; the padding columns are never read (scale_cols_loss)
        dlogits = torch.empty(M, LD, device=dev, dtype=torch.float32) if inp["need_grad"] else None
        ones = inp["ones"]
        images = []
        for b in range(B):
            R, off, G = Rs[b], offs[b], inp["G"][b]
            lg = logits[off:off + V * R]
            dl = None if dlogits is None else dlogits[off:off + V * R]
            # --- WSDDN MIL scores + loss (+ unit gradient) and the view-averaged scores = round 0's mining input
            scores = torch.empty(V, R, K, device=dev, dtype=torch.float32)
            # mining scores of all rounds, one (R, K+1) matrix each: round 0 <- mean WSDDN scores (K columns used), round k+1
            # <- mean softmax of refinement head k.  They depend on the logits only, so the rounds are mined side by side.
            mine_scores = torch.empty(RK, R, K1, device=dev, dtype=torch.float32)
            ops.wsddn_mil(lg, V, R, K, cols["cls"], cols["det"], inp["gt_onehot"][b], scores, loss_view[b, 0], dl, ones,
                          mean_scores=mine_scores[0])
            if RK > 1:
                ops.oicr_mean_probs(lg, V, R, K, RK - 1, cols["cls_score0"], col_stride, mine_scores[1:])
            # --- K refinement rounds: mine pseudo-GT, label (one workgroup per round), then all rounds' losses at once
            top_k = max(int(R * self.mist_p), 1)                      # roi_heads_oicrplus.py:659-660
            ws = torch.empty(ops.mine_workspace_bytes(R, top_k, G, RK), device=dev, dtype=torch.uint8)
            lab_c = torch.empty(RK, R, device=dev, dtype=torch.int32); lab_w = torch.empty(RK, R, device=dev, dtype=torch.float32)
            lab_i = torch.empty(RK, R, device=dev, dtype=torch.int32); cnt = torch.empty(RK, device=dev, dtype=torch.int32)
            pi = torch.empty(RK, top_k * G, device=dev, dtype=torch.int32); pc = torch.empty_like(pi)
            ps = torch.empty(RK, top_k * G, device=dev, dtype=torch.float32)
            boxes = inp["boxes"][b]                                   # (4, R, 4) f32
            ops.oicr_mine_label(mine_scores, inp["gt_int32"][b], boxes[0], K, top_k, self.mist_thre, 0.01,
                                self.iou_thresholds[0], self.iou_thresholds[1], lab_c, lab_w, lab_i, cnt, pi, pc, ps, ws)
            ops.oicr_refine_loss(lg, V, R, K, cols["cls_score0"], cols["bbox_pred0"], boxes, lab_c, lab_w, lab_i,
                                 inp["pred_view"], self.bbox_reg_weights, loss_view[b, 1:], dl, ones, n_rounds=RK,
                                 col_stride=col_stride)
            images.append({"scores": scores, "mine_scores": mine_scores, "fc7": h2[off:off + V * R], "logits": lg,
                           "rounds": [dict(lab_class=lab_c[k], lab_weight=lab_w[k], lab_index=lab_i[k],
                                           pgt_count=cnt[k:k + 1], pgt_index=pi[k], pgt_class=pc[k], pgt_score=ps[k])
                                      for k in range(RK)]})
        out = torch.empty(n_loss + 2, device=dev, dtype=torch.float32)     # losses | their sum | finite flag
        losses, total, finite = out[:n_loss], out[n_loss:n_loss + 1], out[n_loss + 1:]
        ops.loss_finalize(loss_view, losses, out[n_loss:])
        aux = dict(images[0])              # the reference's one-image-per-GPU view of the intermediates ...
        aux["images"] = images             # ... and every image's when the batch holds more
        self.last_aux = aux
        self._last_finite = finite
        return dict(losses=losses, total=total, feats=feats, pooled=pooled, argmax=argmax, h1=h1, h2=h2, W1=W1, W1T=W1T, W2=W2,
                    W2T=W2T, Wh=Wh, dlogits=dlogits, inp=inp, train_dropout=training_dropout)

    def _col_to_loss(self, device):
        """loss index of every packed logit column (each column belongs to exactly one loss term)."""
        K = self.num_classes
        idx = torch.zeros(self.ld_head, dtype=torch.int32)
        idx[: 2 * K] = 0
        for k in range(self.refine_K):
            base = 2 * K + k * (5 * K + 1)
            idx[base: base + K + 1] = 1 + 2 * k
            idx[base + K + 1: base + 5 * K + 1] = 2 + 2 * k
        return idx.to(device)

    # ------------------------------------------------------------------ training backward (explicit)
    def _handle(self, device, fresh=False):
        """1-element tensor that carries an autograd edge between the heads' nodes (never read).  Forward outputs must be
        fresh tensors; the backward side reuses one zero per device."""
        if fresh:
            return torch.empty(1, device=device, dtype=torch.float32)
        z = getattr(self, "_zero_handle", None)
        if z is None or z.device != device:
            z = self._zero_handle = torch.zeros(1, device=device, dtype=torch.float32)
        return z

    def _train_backward_top(self, st, params_top, g_losses, g_total):
        """stage 1: loss cotangents -> predictor and fc7 gradients, dZ1 (kept in `st`) — params_top = fc7 W, b, predictor pairs"""
        dt_ = self.compute_dtype
        inp = st["inp"]
        B, M = inp["B"], inp["M"]
        LD = self.ld_head
        pooled, h1, h2, W2, Wh = st["pooled"], st["h1"], st["h2"], st["W2"], st["Wh"]
        dev = pooled.device
        D1, D2 = h1.shape[1], h2.shape[1]
        # cotangent of each loss (+ the total's) -> its logit columns, x 1/B (mean over the images); unit gradients -> compute dtype
        if not hasattr(self, "_c2l") or self._c2l.device != dev:
            self._c2l = self._col_to_loss(dev)
        if g_losses is not None:
            g_losses = g_losses.contiguous().float()
        if g_total is not None:
            g_total = g_total.contiguous().float()
        dl = torch.empty(M, LD, device=dev, dtype=dt_)
        ops.scale_cols_loss(st["dlogits"], g_losses, g_total, self._c2l, 1.0 / B, dl, M, LD, self.n_head_cols)
        rs = 2.0 if st["train_dropout"] else 1.0
        # predictor matrices
        dbh = torch.empty(LD, device=dev, dtype=torch.float32); ops.colsum(dl, M, LD, dbh)
        dWh = torch.empty(LD, D2, device=dev, dtype=torch.float32)
        ops.gemm(dl, h2, dWh, LD, D2, M, a_kstrided=True, b_kstrided=True, splitk=4)      # slabs + ordered fold (deterministic)
        dz2 = _padded(M, D2, dev, dt_)
        WhT = self._head_weights_t(Wh, dev)
        if WhT is not None:                  # NT on the transposed packed predictor matrix (K = ld_head contiguous): 116 -> ~70 us
            ops.gemm(dl, WhT, dz2, M, D2, LD, ep=ops.make_epilogue(relu_ref=h2, ref_scale=rs, out_dtype=dt_))
        else:
            ops.gemm(dl, Wh, dz2, M, D2, LD, b_kstrided=True, ep=ops.make_epilogue(relu_ref=h2, ref_scale=rs, out_dtype=dt_))
        # fc7
        db2 = ops.grad_target(self.box_head.fc2.bias, (D2,), dev); ops.colsum(dz2, M, D2, db2)
        # weight gradients read dZ^T (one 64x64-tiled transpose, 65 MB) so that the GEMM's A operand is K-contiguous: the
        # forward-style kernel instead of transposing both operands on the fly inside LDS (fc6: 1.63 -> ~1.3 ms)
        # (needs 16-byte K pieces: M a multiple of 8 bf16 / 4 f32 rows — else both operands stay K-strided)
        epc = 8 if dt_ == torch.bfloat16 else 4
        wgrad_nn = M % epc == 0

        def dz_t(dz, D):
            return ops.transpose_2d(dz, torch.empty(D, M + 8 * epc, device=dev, dtype=dt_)[:, :M], M, D)
        dW2 = ops.grad_target(self.box_head.fc2.weight, (D2, D1), dev)
        if wgrad_nn:
            self._fc_gemm(dz_t(dz2, D2), h1, dW2, D2, D1, M, b_kstrided=True)
        else:
            ops.gemm(dz2, h1, dW2, D2, D1, M, a_kstrided=True, b_kstrided=True)
        dz1 = _padded(M, D1, dev, dt_)
        if st.get("W2T") is not None:
            self._fc_gemm(dz2, st["W2T"], dz1, M, D1, D2, ep=ops.make_epilogue(relu_ref=h1, ref_scale=rs, out_dtype=dt_),
                          wname="fc2T", wmaster=self.box_head.fc2.weight)
        else:
            ops.gemm(dz2, W2, dz1, M, D1, D2, b_kstrided=True, ep=ops.make_epilogue(relu_ref=h1, ref_scale=rs, out_dtype=dt_))
        st["dz1"] = dz1
        # split the packed gradients back onto the 10 predictor tensors (row slices are contiguous views)
        dparams = [dW2, db2]
        row = 0
        for i in range(2, len(params_top), 2):
            n = params_top[i].shape[0]
            dparams += [dWh[row:row + n], dbh[row:row + n]]
            row += n
        return [g if p.requires_grad else None for g, p in zip(dparams, params_top)]

    def _fused_fc1_plan(self, M, D0, D1):
        """-> (entry, momentum, finish) when fc1.weight's update can run in its weight-gradient GEMM's epilogue, else None"""
        opt = getattr(self, "_fused_opt", None)
        if (opt is None or self.compute_dtype != torch.bfloat16 or (M % 8) or getattr(self, "fp32x3", False)
                or getattr(self, "_fc6_panels", None) is not None or getattr(self, "_staged", False)
                or not ops.gemm_sgd_fused_supported(torch.bfloat16, D1, D0, M, False, True)):
            return None
        w = self.box_head.fc1.weight
        if not w.requires_grad or w.grad is not None:
            return None
        return opt.fused_update_entry(w)

    def _fused_fc1_wgrad(self, st):
        """the deferred weight-gradient GEMM of fc6 with fc1.weight's SGD update in its epilogue (sw_epilogue.sgd_fused)"""
        plan = st.pop("fused_fc1")
        entry, momentum, finish = plan
        pooled, dz1 = st["pooled"], st["dz1"]
        dev, M = pooled.device, st["inp"]["M"]
        D0, D1 = pooled.shape[1], dz1.shape[1]

        def nn():
            dzt = ops.transpose_2d(dz1, torch.empty(D1, M + 64, device=dev, dtype=torch.bfloat16)[:, :M], M, D1)
            scratch = torch.empty(D1, D0, device=dev, dtype=torch.float32)      # only the peeled tail columns are ever written
            ep = ops.attach_sgd_fused(ops.make_epilogue(out_dtype=torch.float32), entry, momentum, 1.0)
            ops.gemm(dzt, pooled, scratch, D1, D0, M, b_kstrided=True, ep=ep)
        ops._launch("fc6_wgrad_sgd", nn)
        finish()

    def _train_backward_fc6(self, st, defer_fused=False):
        """stage 2: the fc6 weight and bias gradients (411 MB of the step's 544 MB of gradients)"""
        dt_ = self.compute_dtype
        pooled, dz1 = st["pooled"], st["dz1"]
        dev, M = pooled.device, st["inp"]["M"]
        D0, D1 = pooled.shape[1], dz1.shape[1]
        epc = 8 if dt_ == torch.bfloat16 else 4
        db1 = ops.grad_target(self.box_head.fc1.bias, (D1,), dev); ops.colsum(dz1, M, D1, db1)
        plan = self._fused_fc1_plan(M, D0, D1)
        if plan is not None:
            st["fused_fc1"] = plan
            if not defer_fused:                           # nobody asks for the feature gradient: no data-gradient GEMM will read the copies
                self._fused_fc1_wgrad(st)
            return None, db1
        dW1 = ops.grad_target(self.box_head.fc1.weight, (D1, D0), dev)
        panels = getattr(self, "_fc6_panels", None)       # (n, callback): the data-parallel reducer's SW_DDP_FC1_PANELS
        if panels is not None and panels[0] > 1 and D1 % (256 * panels[0]) == 0:
            # the weight gradient in n row panels (whole 256-row tile rows each, so every output tile is computed exactly as in the
            # one-launch form: same bits); after each the reducer starts that panel's all-reduce — the 411 MB leave in n pieces
            # while the later panels still compute.  The panel count depends on D1 only: every rank issues the same collectives
            # whatever its own proposal count (which only picks the GEMM form).
            rows = D1 // panels[0]
            dzt = ops.transpose_2d(dz1, torch.empty(D1, M + 8 * epc, device=dev, dtype=dt_)[:, :M], M, D1) if M % epc == 0 else None
            for i in range(panels[0]):
                if dzt is not None:
                    ops.gemm(dzt[i * rows:(i + 1) * rows], pooled, dW1[i * rows:(i + 1) * rows], rows, D0, M, b_kstrided=True)
                else:
                    ops.gemm(dz1[:, i * rows:(i + 1) * rows], pooled, dW1[i * rows:(i + 1) * rows], rows, D0, M, a_kstrided=True, b_kstrided=True)
                panels[1](i, dW1[i * rows:(i + 1) * rows])
        elif M % epc == 0:                               # dZ^T as in stage 1; the tagged region holds the transpose too: one "fc6_wgrad" measurement
            def nn():
                dzt = ops.transpose_2d(dz1, torch.empty(D1, M + 8 * epc, device=dev, dtype=dt_)[:, :M], M, D1)
                self._fc_gemm(dzt, pooled, dW1, D1, D0, M, b_kstrided=True)
            ops._launch("fc6_wgrad", nn)
        else:
            ops.gemm(dz1, pooled, dW1, D1, D0, M, a_kstrided=True, b_kstrided=True, tag="fc6_wgrad")
        return dW1, db1

    def _train_backward_pool(self, st, feat_req):
        """stage 3: fc6 data gradient and the ROIPool backward of every view batch that wants a feature gradient"""
        dt_ = self.compute_dtype
        inp = st["inp"]
        B, Rs, offs, M = inp["B"], inp["R"], inp["off"], inp["M"]
        pooled, dz1, W1 = st["pooled"], st["dz1"], st["W1"]
        dev = pooled.device
        D0, D1 = pooled.shape[1], dz1.shape[1]
        dfeats = [None] * len(feat_req)
        if any(feat_req):
            dpooled = _padded(M, D0, dev, dt_, pad=64)          # same pitch as argmax (one pitch per ROIPool call)
            amax = ops.fill_zero(torch.empty(1, device=dev, dtype=torch.float32))   # max|dpooled| -> fixed-point scale of the ROI scatter
            if st["W1T"] is not None:       # NT: B = W1^T (D0 x D1), K-contiguous
                self._fc_gemm(dz1, st["W1T"], dpooled, M, D0, D1, ep=ops.make_epilogue(out_dtype=dt_, absmax_out=amax),
                              tag="fc6_dgrad", wname="fc1T", wmaster=self.box_head.fc1.weight)
            else:
                ops.gemm(dz1, W1, dpooled, M, D0, D1, b_kstrided=True, ep=ops.make_epilogue(out_dtype=dt_, absmax_out=amax),
                         tag="fc6_dgrad")
            P = self.box_pooler.output_size
            for b in range(B):
                R, off = Rs[b], offs[b]
                for s in range(2):
                    if not feat_req[2 * b + s]:
                        continue
                    f = st["feats"][2 * b + s]
                    df = torch.empty_like(f)
                    r0 = off + 2 * s * R
                    ops.roi_pool_bwd(dpooled[r0:r0 + 2 * R], st["argmax"][r0:r0 + 2 * R], inp["rois"][b][s], df, P, P,
                                     row_scale=inp["obj"][b][2 * s:2 * s + 2].reshape(-1), row_scale_add=1.0, relu_ref=f,
                                     dout_absmax=amax, tag="roi_bwd", spatial_scale=self.box_pooler.scale)
                    dfeats[2 * b + s] = df
        if "fused_fc1" in st:                              # the data gradient above was the last reader of fc1.weight's copies
            self._fused_fc1_wgrad(st)
        return dfeats

    # ------------------------------------------------------------------ public forward
    def _prepare_inputs(self, proposals_list, targets1, device, need_grad):
        """proposals_list: the 4 views' lists of per-image Instances (index aligned); targets1: per-image Instances of view 1.
        The reference takes ONE image per GPU (roi_heads_oicrplus.py:193); more images are the same computation per image
        (MIL softmax, mining and NMS are per image) on stacked rows, their losses averaged — what DDP forms over as many
        ranks."""
        K = self.num_classes
        B = len(proposals_list[0])
        assert B >= 1 and all(len(p) == B for p in proposals_list)
        _, gt_ints, gt_oh = get_image_level_gt(targets1, K)
        if not hasattr(self, "_consts") or self._consts[0].device != device:
            self._consts = (torch.ones(max(2, 2 * self.refine_K), device=device),
                            torch.tensor([0, 1, 2, 2], dtype=torch.int32, device=device))
        Rs, Gs, offs, boxes, obj, rois = [], [], [], [], [], []
        off = 0
        for b in range(B):
            props = [p[b] for p in proposals_list]
            R = len(props[0])
            assert all(len(p) == R for p in props), "the 4 proposal sets are index aligned (dataset_mapper.py:353-361)"
            bl = [p.proposal_boxes.tensor.to(device=device, dtype=torch.float32).contiguous() for p in props]
            ol = [p.objectness_logits.to(device=device, dtype=torch.float32).contiguous() for p in props]
            bx = torch.empty(4, R, 4, device=device, dtype=torch.float32)
            ob = torch.empty(4, R, device=device, dtype=torch.float32)
            ro = torch.empty(2, 2 * R, 5, device=device, dtype=torch.float32)
            if device.type == "cuda":
                ops.pack_views(bl, ol, bx, ob, ro)
            else:                                      # host-side plumbing tests only
                bx.copy_(torch.stack(bl)); ob.copy_(torch.stack(ol))
                ro[..., 1:] = bx.view(2, 2 * R, 4); ro[:, :R, 0] = 0; ro[:, R:, 0] = 1
            Rs.append(R); Gs.append(int(gt_ints[b].numel())); offs.append(off); boxes.append(bx); obj.append(ob); rois.append(ro)
            off += 4 * R
        # the image-level labels are the only host data of the step: stage them through pinned memory so that the copy is
        # asynchronous (a pageable H2D copy blocks the host until the stream drains = one full pipeline bubble per step)
        nG = sum(Gs)
        if self._prestaged_labels is not None:           # a captured step reads the labels from a static device buffer that
            devbuf, staged_G = self._prestaged_labels    # stage_labels filled in stream order BEFORE the capture / replay
            assert tuple(staged_G) == tuple(Gs), "the class counts are kernel arguments of the captured step"
        elif device.type == "cuda":
            host = torch.empty(nG + B * K, dtype=torch.float32, pin_memory=True)
            self._fill_label_host(host, gt_ints, gt_oh, nG)
            devbuf = host.to(device, non_blocking=True)
        else:
            devbuf = torch.cat([torch.cat(gt_ints).to(torch.int32).view(torch.float32), gt_oh.reshape(-1)])
        gt_i32, gt_onehot, g0 = [], [], 0
        for b in range(B):
            gt_i32.append(devbuf[g0:g0 + Gs[b]].view(torch.int32)); g0 += Gs[b]
            gt_onehot.append(devbuf[nG + b * K: nG + (b + 1) * K])
        return dict(B=B, R=Rs, G=Gs, off=offs, M=off, boxes=boxes, obj=obj, rois=rois, gt_int32=gt_i32, gt_onehot=gt_onehot,
                    ones=self._consts[0], pred_view=self._consts[1], need_grad=need_grad)

    @staticmethod
    def _fill_label_host(host, gt_ints, gt_oh, nG):
        host[:nG].view(torch.int32).copy_(torch.cat(gt_ints).to(torch.int32))
        host[nG:].copy_(gt_oh.reshape(-1))

    def stage_labels(self, targets1, static_dev):
        """Upload the image-level labels of `targets1` into the static device buffer a captured step reads them from (layout of
        _prepare_inputs: all class lists, then one K-wide one-hot per image), asynchronously and in stream order — call it on
        the stream that will replay the graph, before the replay.  Returns the per-image class counts (graph key)."""
        K = self.num_classes
        _, gt_ints, gt_oh = get_image_level_gt(targets1, K)
        Gs = tuple(int(g.numel()) for g in gt_ints)
        nG = sum(Gs)
        host = torch.empty(nG + len(Gs) * K, dtype=torch.float32, pin_memory=True)
        self._fill_label_host(host, gt_ints, gt_oh, nG)
        static_dev[:host.numel()].copy_(host, non_blocking=True)
        self._prestaged_labels = (static_dev[:host.numel()], Gs)
        return Gs

    def forward(self, images_list, features_list, proposals_list, targets_list=(None, None, None, None), prepared=None):
        """training: features_list = [features1, features2] ({"plain5": NCHW view} of scale 1 / scale 2, as the reference
        passes them, roi_heads_oicrplus.py:149-188) or, for several images per GPU, one such pair per image
        ([f1_img0, f2_img0, f1_img1, ...]); each holds the view and its flipped copy as a batch of 2."""
        if not self.training:
            pred_instances, all_scores, all_boxes = self._forward_box_test(features_list, proposals_list, targets_list)
            return pred_instances, {}, all_scores, all_boxes
        key = self.box_in_features[0]
        feats = [(f[key] if isinstance(f, dict) else f) for f in features_list]
        # NCHW views of NHWC storage (what the backbone hands out) go back to NHWC without a copy
        feats = [f.permute(0, 2, 3, 1).contiguous() for f in feats]
        targets1 = targets_list[0]
        # `prepared`: the meta-architecture may build the (feature independent) ROI / label tensors before it queues the
        # backbone, so that those tiny kernels do not sit between the backbone and ROIPool on the critical path
        inp = prepared if prepared is not None else self._prepare_inputs(proposals_list, targets1, feats[0].device,
                                                                         need_grad=torch.is_grad_enabled())
        assert len(feats) == 2 * inp["B"]
        self.__dict__["_gt_int64"] = None
        self._gt_int32 = inp["gt_int32"]            # `gt_classes_img_int` (the reference's attribute) converts on access
        params = self._flat_params()              # fc6 W, b, fc7 W, b, then the predictors' (W, b) pairs
        box = [None]
        # `_staged` (set by trainer._NativeDDP): the backward is run in stages with collectives between them, so the graph is CUT at
        # the edges between the three nodes and at the feature tensors: each cut is a detached leaf that collects the gradient of
        # the stage above; the caller resumes below it with `lower.backward(leaf.grad)` (an `inputs=` partial backward also runs the
        # node BELOW a non-leaf input, i.e. the stage that was to wait).  _cuts = (feature pairs, (h0, leaf), (h1, leaf)).
        staged = getattr(self, "_staged", False) and torch.is_grad_enabled()

        def cut(t):
            return t.detach().requires_grad_(True) if (staged and t.requires_grad) else t
        feats_in = [cut(f) for f in feats]
        h0 = _HeadsPoolFunction.apply(self, inp, params, box, *feats_in)
        h0c = cut(h0)
        h1 = _HeadsFc6Function.apply(self, box, h0c, params[0], params[1])
        h1c = cut(h1)
        vec, total = _HeadsLossFunction.apply(self, box, h1c, *params[2:])
        self.__dict__["_cuts"] = ([(f, c) for f, c in zip(feats, feats_in) if c is not f], (h0, h0c), (h1, h1c)) if staged else None
        names = loss_names(self.refine_K)
        losses = LossDict(names, vec, total, self._last_finite)
        self.iter = self.iter + 1
        if has_event_storage() and not (feats[0].is_cuda and torch.cuda.is_current_stream_capturing()):
            # (inside a hipGraph capture these are static graph outputs: the trainer records snapshots of them after each replay)
            st = get_event_storage()
            for k, r in enumerate(self.last_aux["rounds"]):
                st.put_scalar(f"roi_head/num_pgt_r{k}", r["pgt_count"])      # device scalars: no host sync here
        return None, losses

    # ------------------------------------------------------------------ inference (roi_heads_oicrplus.py:432-475)
    @torch.no_grad()
    def _forward_box_test(self, features, proposals, targets_list=None):
        from .inference import oicr_inference
        return oicr_inference(self, features, proposals)
