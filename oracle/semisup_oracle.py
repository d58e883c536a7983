"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the model-independent parts of the Stage-3 Unbiased-Teacher step (BASELINE config #5):
  /root/reference/unbias/ubteacher/engine/trainer.py
    :361-400  threshold_bbox            (score / objectness > threshold, optional multi-label class filter)
    :436-549  run_step_full_semisup     (burn-in, teacher refresh schedule, loss weighting: *_pseudo box losses x 0, other
                                         *_pseudo losses x UNSUP_LOSS_WEIGHT, supervised x 1)
    :588-604  _update_teacher_model     (teacher = student * (1 - keep) + teacher * keep, float32)
  /root/reference/unbias/ubteacher/modeling/roi_heads/fast_rcnn.py
    :73-105   comput_focal_loss / FocalLoss.forward   (CE = cross_entropy(x, t); p = exp(-CE); sum (1 - p)^gamma CE / N)

PARITY PINNING: pinned (round 5).  tests/golden/make_stage3_step_golden.py loads the reference's trainer file from where it lies (its
engine / data / evaluation / checkpoint imports registered as placeholders: imported, never called — the technique of ref_shim_d2.py)
and RUNS UBTeacherTrainer.run_step_full_semisup, threshold_bbox, process_pseudo_label, add_label and _update_teacher_model as unbound
methods on the reference's own student / teacher for three iterations (burn-in, copy step, EMA step); tests/golden/stage3_step.npz
holds what they logged and produced.  tests/test_oracle_stage3_step_cpu.py checks every function below against it: thresholded
detections bit for bit, the weighted sum the reference differentiated, the teacher tensors after copy and EMA bit for bit.
"""
import numpy as np


def threshold_bbox(scores, classes, boxes, thres=0.7, multi_label=None):
    """:361-400 ('roih' branch; 'rpn' = classes None).  Returns (boxes, classes, scores, kept index) in input order."""
    scores = np.asarray(scores, np.float32)
    valid = scores > np.float32(thres)
    if multi_label is not None and classes is not None:
        ml = set(int(c) for c in multi_label)
        valid &= np.array([int(c) in ml for c in classes], bool)
    idx = np.nonzero(valid)[0]
    return np.asarray(boxes, np.float32)[idx], (None if classes is None else np.asarray(classes)[idx]), scores[idx], idx


def update_teacher(teacher: dict, student: dict, keep_rate: float) -> dict:
    """:588-604 in float32: value = student * (1 - keep) + teacher * keep (python-float scalars times f32 tensors, as torch does)"""
    out = {}
    for k, v in teacher.items():
        if k not in student:
            raise Exception("{} is not found in student model".format(k))
        out[k] = (student[k].astype(np.float32) * np.float32(1 - keep_rate) + v.astype(np.float32) * np.float32(keep_rate)).astype(np.float32)
    return out


def weight_losses(record: dict, unsup_weight: float) -> dict:
    """:520-534"""
    out = {}
    for key, v in record.items():
        if key[:4] == "loss":
            if key == "loss_rpn_loc_pseudo" or key == "loss_box_reg_pseudo":
                out[key] = v * 0
            elif key[-6:] == "pseudo":
                out[key] = v * unsup_weight
            else:
                out[key] = v * 1
    return out


def teacher_action(it: int, burn_up_step: int, update_iter: int):
    """:455-466 -> 'burn_in' (supervised only), 'copy' (keep_rate 0), 'ema', or 'none'"""
    if it < burn_up_step:
        return "burn_in"
    if it == burn_up_step and burn_up_step > 0:
        return "copy"
    if (it - burn_up_step) % update_iter == 0:
        return "ema"
    return "none"


def focal_loss(logits, targets, gamma=1.5):
    """fast_rcnn.py:73-105 in float64 (value and d loss / d logits): F.cross_entropy(reduction="none"), p = exp(-CE),
    sum((1 - p)^gamma * CE) / N"""
    x = np.asarray(logits, np.float64)
    t = np.asarray(targets, np.int64)
    n = x.shape[0]
    m = x.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(x - m).sum(1))
    ce = lse - x[np.arange(n), t]
    p = np.exp(-ce)
    om = 1.0 - p
    loss = (om ** gamma * ce).sum() / n
    sm = np.exp(x - lse[:, None])
    onehot = np.zeros_like(x); onehot[np.arange(n), t] = 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        dw = np.where(om > 0, gamma * om ** (gamma - 1.0) * p * ce, 0.0)
    g = (om ** gamma + dw)[:, None] * (sm - onehot) / n
    return loss, g
