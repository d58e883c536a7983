"""oracle/semisup_oracle.py (the Stage-3 step functions: threshold, loss weights, teacher schedule + EMA) and oracle/frcnn_oracle.py's
detector, against tests/golden/stage3_step.npz — written by RUNNING the reference's own UBTeacherTrainer.run_step_full_semisup /
threshold_bbox / process_pseudo_label / _update_teacher_model (unbias/ubteacher/engine/trainer.py:362-604) on the reference's student
and teacher (tests/golden/make_stage3_step_golden.py).  CPU only."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import frcnn_oracle as FO  # noqa: E402
from oracle import semisup_oracle as SO  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "stage3_step.npz"))
SIZES = [tuple(int(v) for v in s) for s in G["sizes"]]
K = int(G["K"])


def _record(it):
    pre = f"it{it}/record/"
    return {k[len(pre):]: float(G[k]) for k in G.files if k.startswith(pre)}


def test_teacher_schedule_threshold_and_loss_weights_match_the_reference_run():
    burn, upd = int(G["cfg/BURN_UP_STEP"]), int(G["cfg/TEACHER_UPDATE_ITER"])
    assert [SO.teacher_action(i, burn, upd) for i in range(3)] == ["burn_in", "copy", "ema"]
    # iteration 0 logs the 4 supervised losses only; 1 and 2 also the 4 *_pseudo ones (trainer.py:453-464, 512-517)
    assert set(k for k in _record(0) if k.startswith("loss")) == {"loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"}
    thr, w = float(G["cfg/BBOX_THRESHOLD"]), float(G["cfg/UNSUP_LOSS_WEIGHT"])
    for it in (1, 2):
        rec = _record(it)
        assert len([k for k in rec if k.endswith("_pseudo")]) == 4
        # threshold_bbox (:362-400) on the teacher's own detections -> the pseudo labels the reference attached
        for i in range(len(SIZES)):
            b, c, s, _ = SO.threshold_bbox(G[f"it{it}/det_scores{i}"], G[f"it{it}/det_classes{i}"], G[f"it{it}/det_boxes{i}"], thr)
            assert np.array_equal(b, G[f"it{it}/pseudo_boxes{i}"]) and np.array_equal(c, G[f"it{it}/pseudo_classes{i}"])
            assert np.array_equal(s, G[f"it{it}/pseudo_scores{i}"])
            _, _, s_rpn, _ = SO.threshold_bbox(G[f"it{it}/rpn_logits{i}"], None, np.zeros((len(G[f"it{it}/rpn_logits{i}"]), 4), np.float32), thr)
            assert len(s_rpn) == int(G[f"it{it}/rpn_kept"][i])
        # the weighting (:520-534): the sum the reference called backward() on
        weighted = SO.weight_losses(rec, w)
        assert weighted["loss_rpn_loc_pseudo"] == 0 and weighted["loss_box_reg_pseudo"] == 0
        assert abs(sum(weighted.values()) - float(G[f"it{it}/total_loss"])) <= 2e-6 * abs(float(G[f"it{it}/total_loss"]))
    rec0 = _record(0)
    assert abs(sum(SO.weight_losses(rec0, w).values()) - float(G["it0/total_loss"])) <= 2e-6 * float(G["it0/total_loss"])


def test_teacher_ema_matches_the_reference_run():
    """_update_teacher_model (:589-604): iteration 1 copies (keep rate 0), iteration 2 blends with EMA_KEEP_RATE — on the sampled tensors
    the fixture holds (the generator asserted the rule on EVERY tensor of the reference's teacher)"""
    keep = float(G["cfg/EMA_KEEP_RATE"])
    for n in G["watch"]:
        s0, s1 = G[f"it0/student/{n}"], G[f"it1/student/{n}"]                  # the student after iterations 0 and 1
        t1, t2 = G[f"it1/teacher/{n}"], G[f"it2/teacher/{n}"]
        assert np.array_equal(t1, s0), n                                          # copy of the student as it entered iteration 1
        want = SO.update_teacher({"w": t1}, {"w": s1}, keep)["w"]
        assert np.array_equal(want, t2), n


def test_detector_oracle_reproduces_the_burn_in_iteration_of_the_reference_run():
    """iteration 0 of the reference's step = the supervised branch on label_q + label_k at the closed-form weights: oracle losses within
    1e-5 of what the reference logged, and the gradient samples of the weighted sum (all weights 1 here) within 1e-4"""
    P = FO.make_params(K, tag="s3s", head_scale=float(G["head_scale"]))
    imgs, gts = [], []
    for tag, n_gt in (("s3s_lq", 2), ("s3s_lk", 3)):
        for i, (h, w) in enumerate(SIZES):
            imgs.append(FO.make_image(h, w, f"{tag}{i}")); gts.append(FO.make_gt(h, w, n_gt, K, f"{tag}{i}"))
    losses, _, grads = FO.supervised_forward(P, imgs, gts, K, FO.Perm("s3s"), want_grads=True)
    rec = _record(0)
    for k, v in losses.items():
        assert abs(v - rec[k]) <= 1e-5 * abs(rec[k]), (k, v, rec[k])
    stride = int(G["stride"])
    for n in G["watch"]:
        want = G[f"it0/grad/{n}"]
        got = grads[str(n)].ravel()[::stride]
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-12, n
    for n in G["watch_full"]:
        want = G[f"it0/grad/{n}"]
        assert np.abs(grads[str(n)] - want).max() <= 1e-4 * np.abs(want).max() + 1e-12, n
