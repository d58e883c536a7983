"""Stage-3 (Unbiased-Teacher semi-supervised) training step — the model-independent half (BASELINE config #5;
reference unbias/ubteacher/engine/trainer.py:436-549 `run_step_full_semisup`, :361-434 pseudo-labelling, :588-604 teacher EMA).

What is here: the step's control flow behind the reference's interface (`model(data, branch=...) -> (record_dict, proposals_rpn,
proposals_roih, _)` for student and teacher), the teacher refresh as ONE multi-tensor HIP launch per 48 parameters
(`sw_ema_multi`), pseudo-label thresholding as a device-side stable compaction (`sw_threshold_select`: no `.nonzero()` host
sync per image), the loss weighting, the ROI heads' focal classification loss (`FocalLoss`, `fast_rcnn_focal_loss`: one HIP
kernel for loss + gradient, fast_rcnn.py:73-105), the teacher / student container (`EnsembleTSModel`, ts_ensemble.py) and the
branch dispatch of the meta-architecture over arbitrary parts (`TwoStagePseudoLabRCNN`, meta_arch/rcnn.py:8-107).  The detector the
reference plugs in as student / teacher — the ResNet-50-FPN Faster R-CNN — is `frcnn.TwoStagePseudoLabGeneralizedRCNN`; the step
takes any module with that call signature.  Parity: the step on the real detector is checked against oracle/frcnn_oracle.py, which
is pinned by fixtures the reference's own detector wrote (tests/golden/make_stage3_golden.py, tests/test_gpu_stage3.py)."""
from typing import Dict, Optional

import weakref

import copy
import os
import torch

from . import ops
from .structures import Boxes, Instances


_EMA_PLANS = weakref.WeakKeyDictionary()       # teacher module -> (weakref to the student, storage pointers, EmaPlan, rest)


_MODULE_LISTS = weakref.WeakKeyDictionary()     # model -> list of its modules (the tree walk of .parameters() / .buffers() is the cost)


def _storage_ptrs(m):
    """where every parameter and buffer of `m` lives now.  Read from the modules' own dicts (a replaced Parameter / buffer object, a
    .to() or a load that re-allocates all change it); the list of modules itself is cached — m.parameters() + m.buffers() walk the
    module tree with a memo set each, 1.8 ms of host time per iteration for the two ResNet-50-FPN detectors (cProfile,
    tools/diag/stage3_host_profile.py), and a module ADDED to a model between two EMA updates is not something training does."""
    mods = _MODULE_LISTS.get(m)
    if mods is None:
        mods = _MODULE_LISTS[m] = [x for x in m.modules() if x is not m]          # (not the key itself: the entry must die with the model)
    out = []
    for mod in [m] + mods:
        for t in mod._parameters.values():
            if t is not None:
                out.append(t.data_ptr())
        for t in mod._buffers.values():
            if t is not None:
                out.append(t.data_ptr())
    return out


@torch.no_grad()
def update_teacher_model(student: torch.nn.Module, teacher: torch.nn.Module, keep_rate: float = 0.996):
    """trainer.py:588-604.  Every float32 entry of the state dicts in one fused pass; other dtypes (integer buffers such as
    BatchNorm's num_batches_tracked, which the reference blends in floating point and load_state_dict casts back) are blended
    the reference's way with torch ops.  The pairing of the two state dicts (two ~600-entry dict builds, 11 ms of host time on the
    ResNet-50-FPN detector) is kept between calls for as long as both models' tensors stay where they are."""
    if isinstance(student, torch.nn.parallel.DistributedDataParallel):
        student = student.module
    ptrs = (_storage_ptrs(student), _storage_ptrs(teacher))
    # the plan lives OUTSIDE the modules (weakly keyed by the teacher, holding the student weakly): a module attribute made the
    # teacher own the student and carry ctypes pointer arrays that copy.deepcopy / torch.save of the teacher cannot handle
    hit = _EMA_PLANS.get(teacher)
    if hit is None or hit[0]() is not student or hit[1] != ptrs:
        sd_s, sd_t = student.state_dict(), teacher.state_dict()
        fused_t, fused_s, rest = [], [], []
        for k, v in sd_t.items():
            if k not in sd_s:
                raise Exception("{} is not found in student model".format(k))
            s = sd_s[k]
            if v.is_cuda and v.dtype == torch.float32 and s.dtype == torch.float32 and v.is_contiguous() and s.is_contiguous():
                fused_t.append(v); fused_s.append(s)
            else:
                rest.append((v, s))
        hit = (weakref.ref(student), ptrs, ops.EmaPlan(fused_t, fused_s), rest, list(teacher.parameters()))
        _EMA_PLANS[teacher] = hit
    for v, s in hit[3]:
        v.copy_(s * (1 - keep_rate) + v * keep_rate)
    hit[2].run(keep_rate)
    # the kernel wrote the TEACHER's state behind torch's version counters: its cached compute-dtype copies are stale — the teacher's
    # only (a wholesale ops.invalidate_all_staged() here re-staged every frozen weight of every model alive in the process once per
    # iteration: the student's stem / res2, a VGG / OICR model next to it)
    ops.PARAM_EPOCH += 1
    for p_ in hit[4]:
        ops.mark_updated(p_)
    ops.BUFFER_EPOCH += 1


def threshold_bbox(data_inst: Optional[dict], proposals: Instances, thres: float = 0.7, proposal_type: str = "roih",
                   has_multi_label: bool = False) -> Instances:
    """trainer.py:361-400.  Returns Instances with `gt_boxes` (+ `gt_classes`, `scores` | `objectness_logits`); the kept count is
    read back once per image (the reference's boolean indexing syncs too) to size the result."""
    out = Instances(proposals.image_size)
    if proposal_type == "rpn":
        cnt, b, _, sc, _ = ops.threshold_select(proposals.objectness_logits.float().contiguous(), None,
                                                proposals.proposal_boxes.tensor.float().contiguous(), thres)
        n = int(cnt.item())
        out.gt_boxes = Boxes(b[:n]); out.objectness_logits = sc[:n]
    elif proposal_type == "roih":
        pad_cnt = getattr(proposals, "_sw_count", None)
        if pad_cnt is not None and thres < 0:
            # padded detections of a speculative iteration (frcnn: rows beyond the count have score 0): a negative threshold would keep them
            proposals = proposals[:int(pad_cnt.item())]
        allowed = None
        if has_multi_label:
            allowed = torch.as_tensor(data_inst["multi_label"], dtype=torch.int32).to(proposals.scores.device)
        cnt, b, c, sc, _ = ops.threshold_select(proposals.scores.float().contiguous(), proposals.pred_classes.to(torch.int32).contiguous(),
                                                proposals.pred_boxes.tensor.float().contiguous(), thres, allowed)
        n = int(cnt.item())
        out.gt_boxes = Boxes(b[:n]); out.gt_classes = c[:n].to(torch.int64); out.scores = sc[:n]
    else:
        raise ValueError("Unkown pseudo label boxes methods")
    return out


def process_pseudo_label(unlabel_data, proposals, cur_threshold, proposal_type, method="thresholding", has_multi_label=False):
    """trainer.py:402-420"""
    if method != "thresholding":
        raise ValueError("Unkown pseudo label boxes methods")
    insts, total = [], 0.0
    for d, p in zip(unlabel_data, proposals):
        inst = threshold_bbox(d, p, cur_threshold, proposal_type, has_multi_label)
        total += len(inst)
        insts.append(inst)
    return insts, total / max(len(proposals), 1)


def loss_weights(record, unsup_loss_weight: float) -> Dict[str, float]:
    """the factor weight_losses (below) multiplies each loss of the record by"""
    out = {}
    for key in record:
        if key[:4] == "loss":
            out[key] = 0.0 if key in ("loss_rpn_loc_pseudo", "loss_box_reg_pseudo") else (float(unsup_loss_weight) if key[-6:] == "pseudo" else 1.0)
    return out


class _WeightedSumFn(torch.autograd.Function):
    """(total, weighted) of scalar losses in ONE launch (sw_weighted_sum; backward sw_scale_scalars): the reference's loop of
    multiplications followed by sum(loss_dict.values()) (trainer.py:520-540) was 2 n framework launches and n more in the backward per
    iteration, all in the stretch between the forward's last kernel and the backward's first, where nothing else keeps the GPU busy.
    Same f32 products and the same order of additions: total and every weighted loss bit-identical."""

    @staticmethod
    def forward(ctx, weights, *vals):
        out = torch.empty(len(vals) + 1, device=vals[0].device, dtype=torch.float32)
        ops.weighted_sum([v.detach() for v in vals], weights, out)
        ctx.weights = weights
        ctx.mark_non_differentiable(out)
        return out[len(vals)], out

    @staticmethod
    def backward(ctx, g_total, _g_out):
        g = torch.empty(len(ctx.weights), device=g_total.device, dtype=torch.float32)
        ops.scale_scalars(g_total.contiguous(), ctx.weights, g)
        return (None,) + tuple(g[i] for i in range(len(ctx.weights)))


def weighted_total(record, weights: Dict[str, float]):
    """-> (sum of weights[k] * record[k] in the record's key order, {k: weights[k] * record[k]} detached)"""
    keys = [k for k in record if k in weights]
    vals = [record[k] for k in keys]
    if vals and all(torch.is_tensor(v) and v.is_cuda and v.dtype == torch.float32 and v.numel() == 1 for v in vals) and len(vals) <= 32:
        total, out = _WeightedSumFn.apply(tuple(weights[k] for k in keys), *vals)
        return total, {k: out[i] for i, k in enumerate(keys)}
    loss_dict = {k: record[k] * weights[k] for k in keys}
    return sum(loss_dict.values()), loss_dict


def weight_losses(record: Dict[str, torch.Tensor], unsup_loss_weight: float) -> Dict[str, torch.Tensor]:
    """trainer.py:520-534: pseudo box-regression losses x 0, other *_pseudo losses x UNSUP_LOSS_WEIGHT, supervised x 1"""
    out = {}
    for key, v in record.items():
        if key[:4] == "loss":
            if key in ("loss_rpn_loc_pseudo", "loss_box_reg_pseudo"):
                out[key] = v * 0
            elif key[-6:] == "pseudo":
                out[key] = v * unsup_loss_weight
            else:
                out[key] = v * 1
    return out


class _FocalLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, gamma):
        x = logits.detach().float().contiguous()
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        dl = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ops.focal_loss(x, target.to(torch.int32).contiguous(), gamma, loss, dl)
        ctx.dl, ctx.in_dtype = dl, logits.dtype
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.dl is None else (ctx.dl * g).to(ctx.in_dtype)), None, None


class FocalLoss(torch.nn.Module):
    """fast_rcnn.py:85-105: `forward(input, target)` = sum_r (1 - p_r)^gamma * CE_r  (the caller divides by the row count; here the
    division is part of the kernel and undone for the module's own contract)."""

    def __init__(self, weight=None, gamma=1.0, num_classes=80):
        super().__init__()
        assert gamma >= 0 and weight is None
        self.gamma, self.num_classes = gamma, num_classes

    def forward(self, input, target):
        return _FocalLossFunction.apply(input, target, self.gamma) * input.shape[0]


def fast_rcnn_focal_loss(pred_class_logits, gt_classes, gamma=1.5):
    """FastRCNNFocalLoss.comput_focal_loss (fast_rcnn.py:73-83): focal loss summed over the sampled proposals / their number;
    `0.0 * logits.sum()` for an empty batch, as the reference."""
    if gt_classes.numel() == 0:
        return 0.0 * pred_class_logits.sum()
    return _FocalLossFunction.apply(pred_class_logits, gt_classes, gamma)


class EnsembleTSModel(torch.nn.Module):
    """ts_ensemble.py:6-15: the checkpointed container `modelTeacher` / `modelStudent` (DDP / DataParallel wrappers peeled off)."""

    def __init__(self, modelTeacher, modelStudent):
        super().__init__()
        wrappers = (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)
        self.modelTeacher = modelTeacher.module if isinstance(modelTeacher, wrappers) else modelTeacher
        self.modelStudent = modelStudent.module if isinstance(modelStudent, wrappers) else modelStudent


class TwoStagePseudoLabRCNN(torch.nn.Module):
    """The branch dispatch of TwoStagePseudoLabGeneralizedRCNN.forward (meta_arch/rcnn.py:8-107) over any backbone / proposal
    generator / ROI heads with the reference's call signatures (PseudoLabRPN.forward(images, features, gt, compute_loss=,
    compute_val_loss=), StandardROIHeadsPseudoLab.forward(images, features, proposals, targets, compute_loss=, branch=,
    compute_val_loss=)); `preprocess` maps batched_inputs to the images object the parts take."""

    def __init__(self, backbone, proposal_generator, roi_heads, preprocess, inference=None):
        super().__init__()
        self.backbone, self.proposal_generator, self.roi_heads = backbone, proposal_generator, roi_heads
        self.preprocess_image, self._inference = preprocess, inference

    def forward(self, batched_inputs, branch="supervised", given_proposals=None, val_mode=False):
        if (not self.training) and (not val_mode):
            return self._inference(batched_inputs)
        images = self.preprocess_image(batched_inputs)
        gt_instances = [x["instances"] for x in batched_inputs] if "instances" in batched_inputs[0] else None
        features = self.backbone(getattr(images, "tensor", images))
        if branch == "supervised":
            proposals_rpn, proposal_losses = self.proposal_generator(images, features, gt_instances)
            _, detector_losses = self.roi_heads(images, features, proposals_rpn, gt_instances, branch=branch)
            losses = {}
            losses.update(detector_losses); losses.update(proposal_losses)
            return losses, [], [], None
        if branch == "unsup_data_weak":                       # the teacher's pass: proposals and predictions, no losses
            proposals_rpn, _ = self.proposal_generator(images, features, None, compute_loss=False)
            proposals_roih, roi_predictions = self.roi_heads(images, features, proposals_rpn, targets=None, compute_loss=False,
                                                             branch=branch)
            return {}, proposals_rpn, proposals_roih, roi_predictions
        if branch == "val_loss":
            proposals_rpn, proposal_losses = self.proposal_generator(images, features, gt_instances, compute_val_loss=True)
            _, detector_losses = self.roi_heads(images, features, proposals_rpn, gt_instances, branch=branch, compute_val_loss=True)
            losses = {}
            losses.update(detector_losses); losses.update(proposal_losses)
            return losses, [], [], None
        raise ValueError(f"unknown branch {branch!r}")


class SemiSupStep:
    """`run_step_full_semisup` (trainer.py:436-549) over any student / teacher pair with the reference's branch interface."""

    def __init__(self, model, model_teacher, optimizer, *, burn_up_step, teacher_update_iter=1, ema_keep_rate=0.9996,
                 bbox_threshold=0.7, unsup_loss_weight=4.0, burn_up_with_strong_aug=True, has_multi_label=False, fuse_grad_sums=True,
                 lockstep=True, overlap_teacher=None, speculate=True):
        self.model, self.model_teacher, self.optimizer = model, model_teacher, optimizer
        self.fuse_grad_sums = fuse_grad_sums          # ops.grad_scope around backward (False: autograd sums the two passes' weight gradients)
        core = getattr(model, "module", model)                                 # (DistributedDataParallel wraps the student)
        self.lockstep = bool(lockstep) and hasattr(getattr(getattr(core, "backbone", None), "bottom_up", None), "forward_lockstep")
        self.burn_up_step, self.teacher_update_iter, self.ema_keep_rate = burn_up_step, teacher_update_iter, ema_keep_rate
        self.bbox_threshold, self.unsup_loss_weight = bbox_threshold, unsup_loss_weight
        self.burn_up_with_strong_aug, self.has_multi_label = burn_up_with_strong_aug, has_multi_label
        # speculate: the iteration's count read-backs are assumed and confirmed once (run_step; frcnn.Speculation).  SW_S3_SPECULATE=0
        # turns it off.  With the reads gone the GPU side decides the iteration, so the teacher's pass goes on a second stream beside the
        # student's backbones and labelled-batch heads (lockstep form only: that is the call that can take the pseudo labels late) and
        # its backbone replays as a hipGraph — each measured WITHOUT gain while the reads were in (16.0-16.9 vs 15.7-16.3 ms: the
        # issuing thread kept waiting at them), together 16.0 -> 13.1 ms once they were stubbed out.  overlap_teacher=None follows
        # `speculate`; SW_S3_TEACHER_STREAM=0/1 and SW_S3_BACKBONE_GRAPH=0/1 override.
        self.speculate = bool(speculate) and os.environ.get("SW_S3_SPECULATE", "1") != "0"
        self.spec_misses, self._spec_pause = 0, 0
        ov = self.speculate if overlap_teacher is None else bool(overlap_teacher)
        env = os.environ.get("SW_S3_TEACHER_STREAM")
        self.overlap_teacher = (ov if env is None else env == "1") and self.lockstep
        tcore = getattr(model_teacher, "module", model_teacher)
        if self.speculate and hasattr(tcore, "graph_nograd_backbone"):
            tcore.graph_nograd_backbone = True
        self._side = None
        self.iter = 0

    # ------------------------------------------------------------------ one iteration
    def _samplers(self):
        core = getattr(self.model, "module", self.model)
        seen, out = set(), []
        for holder in (core, getattr(core, "proposal_generator", None), getattr(core, "roi_heads", None)):
            sp = getattr(holder, "sampler", None)
            if sp is not None and id(sp) not in seen:
                seen.add(id(sp)); out.append(sp)
        return out

    def run_step(self, data):
        """One iteration.  With `speculate` (default on a GPU) the forward / backward runs inside a frcnn.Speculation ledger — the
        count read-backs of the training path are assumed, not read — and ONE read before the optimizer step confirms them; a wrong
        assumption (an image whose NMS left fewer proposals than the cap, a sampler that could not fill its batch, a non-finite RPN
        output) discards the attempt, rewinds the label samplers and repeats the iteration with the reading code, which also raises
        what the reference raises.  After a miss the next 20 iterations do not speculate."""
        self._ema_update()
        dev = next(self.model_teacher.parameters()).device
        if self.speculate and dev.type == "cuda" and self._spec_pause == 0:
            from .frcnn import Speculation
            snaps = [(sp, copy.deepcopy(sp.__dict__)) for sp in self._samplers()]
            with Speculation() as ledger:
                out = self._attempt(data)
            if ledger.holds():
                self.optimizer.step()
                self.iter += 1
                return out
            self.spec_misses += 1
            self._spec_pause = 20
            for sp, snap in snaps:
                sp.__dict__.clear(); sp.__dict__.update(snap)
        elif self._spec_pause > 0:
            self._spec_pause -= 1
        out = self._attempt(data)
        self.optimizer.step()
        self.iter += 1
        return out

    def _attempt(self, data):
        if self.fuse_grad_sums:
            # the student's two passes share every weight: inside the scope their gradients are summed in the kernels, and the uses of
            # a 3x3 weight counted during the forward passes run as one grouped launch (ops.grad_scope)
            with ops.grad_scope():
                return self._forward_backward(data)
        return self._forward_backward(data)

    def _ema_update(self):
        if self.iter < self.burn_up_step:
            return
        if self.iter == self.burn_up_step and self.burn_up_step > 0:
            update_teacher_model(self.model, self.model_teacher, keep_rate=0.00)
        elif (self.iter - self.burn_up_step) % self.teacher_update_iter == 0:
            update_teacher_model(self.model, self.model_teacher, keep_rate=self.ema_keep_rate)

    def _forward_backward(self, data):
        label_q, label_k, unlabel_q, unlabel_k = data
        if self.iter < self.burn_up_step:
            batch = list(label_q) + list(label_k) if self.burn_up_with_strong_aug else label_k
            record, _, _, _ = self.model(batch, branch="supervised")
            weights = {k: 1.0 for k in record if k[:4] == "loss"}
        else:
            record = {}

            def teacher_pass():
                with torch.no_grad():
                    _, props_rpn, props_roih, _ = self.model_teacher(unlabel_k, branch="unsup_data_weak")
                return props_rpn, props_roih

            def pseudo_labels(props_rpn, props_roih):
                # (trainer.py:490-494 also thresholds the RPN's proposals into `joint_proposal_dict["proposals_pseudo_rpn"]`, which
                # nothing reads afterwards: not computed — it cost two launches and a count read-back per iteration)
                pseudo_roih, _ = process_pseudo_label(unlabel_k, props_roih, self.bbox_threshold, "roih",
                                                      has_multi_label=self.has_multi_label)
                for d in list(unlabel_q) + list(unlabel_k):                                           # remove_label
                    d.pop("instances", None)
                for dq, dk, lab in zip(unlabel_q, unlabel_k, pseudo_roih):                            # add_label
                    dq["instances"] = lab; dk["instances"] = lab
                return pseudo_roih
            dev = next(self.model_teacher.parameters()).device
            if self.overlap_teacher and dev.type == "cuda":
                # Teacher on the side stream, student on the caller's: the GPU runs the teacher's ~150 small launches (one 800x1216 image:
                # res4 / res5 / FPN kernels of a few hundred workgroups) between the student's.  Its labels are asked for by the student's
                # call only when the pseudo-labelled batch's heads are next (second_targets): by then both student backbones and the
                # labelled batch's RPN / ROI heads are queued.  The thresholding's count read-backs wait for the SIDE stream only.
                main = torch.cuda.current_stream(dev)
                if self._side is None:
                    # (normal priority: a high-priority stream starved — 24.8 instead of 13.0 ms per iteration — in a process whose earlier
                    # streams had used up the four hardware queues; tools/diag/s3_in_bench.py)
                    self._side = ops.worker_stream("side", dev)
                side = self._side
                side.wait_stream(main)                               # the EMA update above wrote the teacher's weights on `main`
                with torch.cuda.stream(side):
                    props = teacher_pass()

                def late_targets():
                    with torch.cuda.stream(side):
                        labs = pseudo_labels(*props)
                    main.wait_stream(side)
                    for lab in labs:                                 # allocated on `side`, read by kernels queued on `main`
                        for t in (lab.gt_boxes.tensor, lab.gt_classes, lab.scores):
                            t.record_stream(main)
                    return labs
                (rec_label, _, _, _), (rec_unlabel, _, _, _) = self.model(list(label_q) + list(label_k), branch="supervised",
                                                                          second=list(unlabel_q), second_targets=late_targets)
                side.wait_stream(main)                               # (the next teacher pass starts behind this step's readers anyway)
            elif self.lockstep:
                pseudo_labels(*teacher_pass())
                # both student passes through ONE call: their backbones run in lockstep (frcnn: forward(..., second=...))
                (rec_label, _, _, _), (rec_unlabel, _, _, _) = self.model(list(label_q) + list(label_k), branch="supervised",
                                                                          second=list(unlabel_q))
            else:
                pseudo_labels(*teacher_pass())
                rec_label, _, _, _ = self.model(list(label_q) + list(label_k), branch="supervised")
                rec_unlabel, _, _, _ = self.model(unlabel_q, branch="supervised")
            record.update(rec_label)
            record.update({k + "_pseudo": v for k, v in rec_unlabel.items()})
            weights = loss_weights(record, self.unsup_loss_weight)
        losses, loss_dict = weighted_total(record, weights)
        self.optimizer.zero_grad()
        from . import frcnn as _fr
        if _fr.SPECULATE is not None:
            _fr.SPECULATE.seal()                      # the forward is queued: compare its counts beside the backward (frcnn.Speculation.seal)
        losses.backward()
        if ops.GRAD_SCOPE is not None:
            ops.GRAD_SCOPE.finish()                   # a queued weight gradient that never ran must not reach the optimizer
        return record, loss_dict
