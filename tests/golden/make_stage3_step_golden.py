"""Generate tests/golden/stage3_step.npz by RUNNING the reference's own Stage-3 step
(/root/reference/unbias/ubteacher/engine/trainer.py: UBTeacherTrainer.run_step_full_semisup :436-549, threshold_bbox :362-400,
process_pseudo_label :402-420, remove_label / add_label :422-432, _update_teacher_model :589-604) on the reference's own student and
teacher detectors (ref_shim_d2.build_reference_model) — build container only:

    python tests/golden/make_stage3_step_golden.py

How the trainer file is loaded.  Its module-level imports pull in the detectron2 engine / data / evaluation / checkpoint stack and
ubteacher's data / hooks / checkpoint / solver packages (absent third-party dependencies behind them: fvcore, yacs, pycocotools, …).
None of that is touched by the five methods above, so — the technique ref_shim_d2.py already uses for the modeling files — those
modules are registered as PLACEHOLDERS (empty classes / None: imported, never called), the file is executed from where it lies, and
the methods run UNBOUND on an object created without __init__ that holds what they read: model, model_teacher, optimizer, cfg.SEMISUPNET,
iter, has_multi_label, a one-batch data iterator.  `_write_metrics` (logging over comm.gather) is replaced by a recorder that keeps the
record dict the step hands it.  torch.randperm inside detectron2/modeling/sampling.py is the closed-form oracle.frcnn_oracle.Perm as
in make_stage3_golden.py.

Three iterations from the closed-form weights tag "s3s" (BURN_UP_STEP 1, UNSUP_LOSS_WEIGHT 2, EMA_KEEP_RATE 0.9996, BBOX_THRESHOLD 0.7,
SGD lr 1e-5 momentum 0.9):  0 = burn-in (labelled strong + weak views, supervised branch),  1 = the copy step (teacher <- student, keep
rate 0) + a semi-supervised step,  2 = one EMA update + a semi-supervised step.  Stored per iteration: the record dict (every loss the
step logs), the weighted sum `losses.backward()` was called on, the teacher's detections before and after the 0.7 threshold, gradient samples of the student after backward (the loss weighting x0 /
x UNSUP_LOSS_WEIGHT / x1 acts there), samples of the student's weights after the optimizer step and of the teacher's tensors after
its update.  Data, no source text."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import ref_shim_d2  # noqa: E402
from ref_shim_d2 import _load, _pkg  # noqa: E402
from oracle import frcnn_oracle as FO  # noqa: E402

ns = ref_shim_d2.install()
Boxes, Instances = ns.boxes.Boxes, ns.instances.Instances
UB = ref_shim_d2.UB

K = 20
SIZES = [(96, 128), (128, 112)]
TAG = "s3s"
LR, MOM = 1e-5, 0.9
CFG = dict(BURN_UP_STEP=1, BURN_UP_WITH_STRONG_AUG=True, TEACHER_UPDATE_ITER=1, EMA_KEEP_RATE=0.9996, BBOX_THRESHOLD=0.7,
           UNSUP_LOSS_WEIGHT=2.0, HAS_MULTI_LABEL=False)
STRIDE = 1009
WATCH = ["roi_heads.box_head.fc1.weight", "roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.weight",
         "proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.anchor_deltas.weight",
         "proposal_generator.rpn_head.objectness_logits.weight", "backbone.fpn_output3.weight",
         "backbone.bottom_up.res5.2.conv3.weight", "backbone.bottom_up.res3.0.conv1.weight"]
WATCH_FULL = ["roi_heads.box_predictor.cls_score.bias", "roi_heads.box_predictor.bbox_pred.bias",
              "proposal_generator.rpn_head.objectness_logits.bias", "proposal_generator.rpn_head.anchor_deltas.bias"]


def load_trainer_module():
    """placeholders for what trainer.py imports and the step never touches; then the file itself"""
    class _Stub:                                             # base classes of the trainer classes: imported, never initialised
        pass
    fv = _pkg("fvcore.nn.precise_bn"); fv.get_bn_modules = None
    comm = sys.modules["detectron2.utils.comm"]
    comm.get_world_size = lambda: 1
    comm.get_local_rank = lambda: 0
    comm.is_main_process = lambda: True
    comm.gather = lambda d, dst=0: [d]
    sys.modules["detectron2.utils"].comm = comm
    ck = _pkg("detectron2.checkpoint"); ck.DetectionCheckpointer = None
    eng = _pkg("detectron2.engine")
    eng.DefaultTrainer = type("DefaultTrainer", (_Stub,), {}); eng.SimpleTrainer = type("SimpleTrainer", (_Stub,), {})
    eng.TrainerBase = type("TrainerBase", (_Stub,), {}); eng.hooks = types.SimpleNamespace()
    tl = _pkg("detectron2.engine.train_loop"); tl.AMPTrainer = None
    evm = _pkg("detectron2.evaluation"); evm.COCOEvaluator = evm.verify_results = evm.PascalVOCDetectionEvaluator = None
    dm = _pkg("detectron2.data.dataset_mapper"); dm.DatasetMapper = None
    db = _pkg("detectron2.data.build"); db.build_detection_train_loader = None
    mk = _pkg("detectron2.structures.masks"); mk.BitMasks = None
    _pkg("ubteacher.data")
    ub = _pkg("ubteacher.data.build")
    ub.build_detection_semisup_train_loader = ub.build_detection_test_loader = ub.build_detection_semisup_train_loader_two_crops = None
    um = _pkg("ubteacher.data.dataset_mapper"); um.DatasetMapperTwoCropSeparate = None
    _pkg("ubteacher.engine", UB + "/engine")
    hk = _pkg("ubteacher.engine.hooks"); hk.LossEvalHook = None
    _load("ubteacher.modeling.meta_arch.ts_ensemble", UB + "/modeling/meta_arch/ts_ensemble.py")     # the real (15-line) file
    _pkg("ubteacher.checkpoint"); dc = _pkg("ubteacher.checkpoint.detection_checkpoint"); dc.DetectionTSCheckpointer = None
    _pkg("ubteacher.solver"); sb = _pkg("ubteacher.solver.build"); sb.build_lr_scheduler = None
    return _load("ubteacher.engine.trainer", UB + "/engine/trainer.py")


def load_params(model, P):
    sd = model.state_dict()
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k].copy_(torch.from_numpy(v))


def batch(tag, n_gt):
    out = []
    for i, (h, w) in enumerate(SIZES):
        d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")), "height": h, "width": w}
        if n_gt:
            b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
            inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b)); inst.gt_classes = torch.from_numpy(c)
            d["instances"] = inst
        out.append(d)
    return out


def main():
    torch.manual_seed(0)
    mod = load_trainer_module()
    T = mod.UBTeacherTrainer
    P = FO.make_params(K, tag=TAG, head_scale=14.0)
    student = ref_shim_d2.build_reference_model(ns, K); load_params(student, P); student.train()
    teacher = ref_shim_d2.build_reference_model(ns, K); load_params(teacher, P); teacher.train()
    named = dict(student.named_parameters())
    opt = torch.optim.SGD([p for p in student.parameters() if p.requires_grad], lr=LR, momentum=MOM)
    me = object.__new__(T)                                   # no __init__: the step reads exactly the attributes set here
    me.model, me.model_teacher, me.optimizer = student, teacher, opt
    me.cfg = types.SimpleNamespace(SEMISUPNET=types.SimpleNamespace(**CFG))
    me.has_multi_label = CFG["HAS_MULTI_LABEL"]
    me._trainer = types.SimpleNamespace(iter=0, _data_loader_iter=None)
    records = []
    me._write_metrics = lambda metrics: records.append({k: (float(v.detach()) if isinstance(v, torch.Tensor) else float(v))
                                                        for k, v in metrics.items()})
    grads = {}
    orig_step = opt.step

    def spy_step(*a, **k):                                   # between losses.backward() and the update: the weighted gradient
        grads.clear()
        for n in WATCH:
            grads[n] = named[n].grad.numpy().ravel()[::STRIDE].copy()
        for n in WATCH_FULL:
            grads[n] = named[n].grad.numpy().copy()
        return orig_step(*a, **k)
    opt.step = spy_step
    pseudo = {}
    orig_add = T.add_label

    def spy_add(self, unlabled_data, label):                 # what process_pseudo_label produced, as it is attached to the views
        pseudo["boxes"] = [l.gt_boxes.tensor.numpy().copy() for l in label]
        pseudo["classes"] = [l.gt_classes.numpy().copy() for l in label]
        pseudo["scores"] = [l.scores.numpy().copy() for l in label]
        return orig_add(self, unlabled_data, label)
    me.add_label = types.MethodType(spy_add, me)
    dets = {}
    orig_ppl = T.process_pseudo_label

    def spy_ppl(self, unlabel_data_k, proposals, cur_threshold, proposal_type, psedo_label_method=""):
        if proposal_type == "roih":                          # the teacher's detections BEFORE the threshold (input of threshold_bbox)
            dets["boxes"] = [p.pred_boxes.tensor.numpy().copy() for p in proposals]
            dets["scores"] = [p.scores.numpy().copy() for p in proposals]
            dets["classes"] = [p.pred_classes.numpy().copy() for p in proposals]
        else:
            dets["rpn_logits"] = [p.objectness_logits.numpy().copy() for p in proposals]
        r = orig_ppl(self, unlabel_data_k, proposals, cur_threshold, proposal_type, psedo_label_method)
        if proposal_type == "rpn":
            dets["rpn_kept"] = [len(x) for x in r[0]]
        return r
    me.process_pseudo_label = types.MethodType(spy_ppl, me)
    totals = []
    orig_bw = torch.Tensor.backward

    def spy_bw(self, *a, **k):                               # `losses.backward()` (:547): the weighted sum the step differentiates
        totals.append(float(self.detach()))
        return orig_bw(self, *a, **k)
    out = {"K": np.array(K), "sizes": np.array(SIZES), "head_scale": np.array(14.0), "lr": np.array(LR), "momentum": np.array(MOM),
           "stride": np.array(STRIDE), "watch": np.array(WATCH), "watch_full": np.array(WATCH_FULL)}
    for k, v in CFG.items():
        out["cfg/" + k] = np.array(v)
    perm = FO.Perm(TAG)
    sd_t = lambda: {k: v.detach().numpy().copy() for k, v in teacher.state_dict().items()}
    sd_s = lambda: {k: v.detach().numpy().copy() for k, v in student.state_dict().items()}
    with ns.events.EventStorage(0), ref_shim_d2_patch(perm):
        for it in range(3):
            me.iter = it
            data = (batch(TAG + "_lq", 2), batch(TAG + "_lk", 3), batch(TAG + "_uq", 0), batch(TAG + "_uk", 0))
            me._trainer._data_loader_iter = iter([data])
            before_s, before_t = sd_s(), sd_t()
            torch.Tensor.backward = spy_bw
            try:
                T.run_step_full_semisup(me)
            finally:
                torch.Tensor.backward = orig_bw
            out[f"it{it}/total_loss"] = np.array(totals[-1])
            rec = records[-1]
            for k, v in rec.items():
                out[f"it{it}/record/{k}"] = np.array(v)
            for n, g in grads.items():
                out[f"it{it}/grad/{n}"] = g
            after_s, after_t = sd_s(), sd_t()
            for n in WATCH:
                out[f"it{it}/student/{n}"] = after_s[n].ravel()[::STRIDE].copy()
            out[f"it{it}/perm_k"] = np.array(perm.k)
            if it >= 1:
                for i in range(len(SIZES)):
                    out[f"it{it}/pseudo_boxes{i}"] = pseudo["boxes"][i]; out[f"it{it}/pseudo_classes{i}"] = pseudo["classes"][i]
                    out[f"it{it}/pseudo_scores{i}"] = pseudo["scores"][i]
                    out[f"it{it}/det_boxes{i}"] = dets["boxes"][i]; out[f"it{it}/det_scores{i}"] = dets["scores"][i]
                    out[f"it{it}/det_classes{i}"] = dets["classes"][i]
                    out[f"it{it}/rpn_logits{i}"] = dets["rpn_logits"][i]
                out[f"it{it}/rpn_kept"] = np.array(dets["rpn_kept"])
                for n in WATCH:
                    out[f"it{it}/teacher/{n}"] = after_t[n].ravel()[::STRIDE].copy()
                # the rule itself, checked on EVERY tensor here so that the fixture need not hold them all
                keep = 0.0 if it == 1 else CFG["EMA_KEEP_RATE"]
                for k_, v in after_t.items():
                    want = (torch.from_numpy(before_s[k_]) * (1 - keep) + torch.from_numpy(before_t[k_]) * keep).numpy()
                    assert np.array_equal(v, want), k_
            else:
                assert all(np.array_equal(before_t[k_], v) for k_, v in after_t.items())          # burn-in leaves the teacher alone
            losses = {k: v for k, v in rec.items() if k.startswith("loss")}
            print(f"[stage3 step] iteration {it}: " + ", ".join(f"{k} {v:.6f}" for k, v in losses.items()) +
                  (f"; pseudo boxes {[len(b) for b in pseudo['boxes']]}" if it >= 1 else ""))
    np.savez_compressed(os.path.join(HERE, "stage3_step.npz"), **out)
    # ---- the restated step functions (oracle/semisup_oracle.py) against what the reference's methods just did
    from oracle import semisup_oracle as SO
    assert [SO.teacher_action(i, 1, 1) for i in range(3)] == ["burn_in", "copy", "ema"]
    print("wrote stage3_step.npz:", len(out), "arrays")


class ref_shim_d2_patch:
    """torch.randperm of detectron2/modeling/sampling.py -> the closed-form permutation (as make_stage3_golden.PatchRandperm)"""

    def __init__(self, perm):
        self.perm = perm

    def __enter__(self):
        self.orig = torch.randperm
        ns.sampling.torch.randperm = lambda n, device=None: torch.from_numpy(self.perm(int(n)))
        return self

    def __exit__(self, *a):
        ns.sampling.torch.randperm = self.orig


if __name__ == "__main__":
    main()
