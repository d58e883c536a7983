"""Generate tests/golden/stage3_*.npz by RUNNING the reference's Stage-3 detector
(/root/reference/unbias/ubteacher/modeling/** over /root/reference/detectron2/detectron2/modeling/**, loaded through
ref_shim_d2.py) on closed-form inputs — build container only:

    python tests/golden/make_stage3_golden.py

Cases
  a  "supervised" branch (the burn-in step and both halves of the semi-supervised step, unbias/ubteacher/engine/trainer.py:
     453-464,512-517): 2 images 96x128 / 128x112 padded to 128x128, K = 20, 3 + 2 ground-truth boxes -> the 4 losses, RPN anchor
     labels, proposals, sampled ROIs and their classes, FPN level of every ROI, predictions, gradient samples.
  w  "unsup_data_weak" branch of the TEACHER (trainer.py:478-486): RPN proposals + ROI-head detections with peaky heads, then the
     0.7 score threshold of `process_pseudo_label` (trainer.py:361-403) -> pseudo boxes.
  e  eval mode (`GeneralizedRCNN.inference`, detectron2/detectron2/modeling/meta_arch/rcnn.py:177-219, what VOCeval consumes): the
     same model as w switched to eval (test top-k of the RPN), dataset "height" / "width" DIFFERENT from the network input size,
     through the reference's own `_postprocess` / `detector_postprocess` (modeling/postprocessing.py:9-59): rescaled, clipped boxes.

torch.randperm inside detectron2/modeling/sampling.py is replaced by the closed-form permutation oracle.frcnn_oracle.Perm
(the sampling of the reference is not reproducible across implementations otherwise); everything else is the reference's code.
The script also checks oracle/frcnn_oracle.py against what the reference produced (losses, integer outputs)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import ref_shim_d2  # noqa: E402
from oracle import frcnn_oracle as FO  # noqa: E402

ns = ref_shim_d2.install()
Boxes, Instances = ns.boxes.Boxes, ns.instances.Instances

K = 20
SIZES = [(96, 128), (128, 112)]
N_GT = [3, 2]
GRAD_FULL = ["proposal_generator.rpn_head.objectness_logits.bias", "proposal_generator.rpn_head.anchor_deltas.weight",
             "roi_heads.box_predictor.cls_score.bias", "roi_heads.box_predictor.bbox_pred.bias", "roi_heads.box_head.fc2.bias",
             "backbone.fpn_lateral5.bias", "backbone.fpn_output2.bias"]
GRAD_SAMPLED = ["roi_heads.box_head.fc1.weight", "roi_heads.box_predictor.cls_score.weight", "proposal_generator.rpn_head.conv.weight",
                "backbone.fpn_output3.weight", "backbone.fpn_lateral2.weight", "backbone.bottom_up.res5.2.conv3.weight",
                "backbone.bottom_up.res4.0.shortcut.weight", "backbone.bottom_up.res3.1.conv2.weight",
                "backbone.bottom_up.res3.0.conv1.weight"]
STRIDE = 997


def load_params(model, P):
    sd = model.state_dict()
    missing = [k for k in sd if k not in P and "anchor_generator.cell_anchors" not in k]       # buffers built by the module itself
    assert not missing, missing[:5]
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, tuple(sd[k].shape), v.shape)
        sd[k].copy_(torch.from_numpy(v))


def inputs(tag, with_gt=True):
    data, gts = [], []
    for i, ((h, w), n) in enumerate(zip(SIZES, N_GT)):
        img = FO.make_image(h, w, f"{tag}{i}")
        d = {"image": torch.from_numpy(img), "height": h, "width": w}
        if with_gt:
            b, c = FO.make_gt(h, w, n, K, f"{tag}{i}")
            inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b)); inst.gt_classes = torch.from_numpy(c)
            d["instances"] = inst
            gts.append((b, c))
        data.append(d)
    return data, gts


class PatchRandperm:
    def __init__(self, perm):
        self.perm = perm

    def __enter__(self):
        self.orig = torch.randperm
        ns.sampling.torch.randperm = lambda n, device=None: torch.from_numpy(self.perm(int(n)))
        return self

    def __exit__(self, *a):
        ns.sampling.torch.randperm = self.orig


def same_box_set(a, b, atol=1e-2):
    """the two (n, 4) box lists hold the same boxes (rows may be permuted: near-equal scores order differently)"""
    if len(a) != len(b):
        return False
    if len(a) == 0:
        return True
    d = np.abs(a[:, None, :] - b[None, :, :]).max(2)
    return bool((d.min(1) <= atol).all() and (d.min(0) <= atol).all())


def run_supervised():
    P = FO.make_params(K, tag="s3a", head_scale=5.0)
    model = ref_shim_d2.build_reference_model(ns, K)
    load_params(model, P)
    model.train()
    data, gts = inputs("s3a")
    captured = {}
    heads = model.roi_heads
    orig_fb = heads._forward_box

    def spy_forward_box(features, proposals, *a, **k):
        captured["sampled"] = proposals
        return orig_fb(features, proposals, *a, **k)
    heads._forward_box = spy_forward_box
    orig_pred = heads.box_predictor.forward

    def spy_pred(x):
        out = orig_pred(x)
        captured["scores"], captured["deltas"] = out[0].detach().numpy().copy(), out[1].detach().numpy().copy()
        return out
    heads.box_predictor.forward = spy_pred
    orig_ls = model.proposal_generator.label_and_sample_anchors

    def spy_ls(anchors, gt_instances):
        r = orig_ls(anchors, gt_instances)
        captured["rpn_labels"] = [t.numpy().copy() for t in r[0]]
        return r
    model.proposal_generator.label_and_sample_anchors = spy_ls
    orig_pp = model.proposal_generator.predict_proposals

    def spy_pp(*a, **k):
        r = orig_pp(*a, **k)
        captured["proposals"] = [(p.proposal_boxes.tensor.numpy().copy(), p.objectness_logits.numpy().copy()) for p in r]
        return r
    model.proposal_generator.predict_proposals = spy_pp
    with ns.events.EventStorage(0), PatchRandperm(FO.Perm("s3a")):
        losses, _, _, _ = model(data, branch="supervised")
        total = sum(losses.values())
        total.backward()
    out = {"K": np.array(K), "sizes": np.array(SIZES), "n_gt": np.array(N_GT), "head_scale": np.array(5.0)}
    for k, v in losses.items():
        out["loss/" + k] = np.array(float(v.detach()))
    for i in range(len(SIZES)):
        out[f"rpn_labels{i}"] = captured["rpn_labels"][i].astype(np.int8)
        out[f"prop_boxes{i}"], out[f"prop_logits{i}"] = captured["proposals"][i]
        s = captured["sampled"][i]
        out[f"samp_boxes{i}"] = s.proposal_boxes.tensor.numpy().copy()
        out[f"samp_classes{i}"] = s.gt_classes.numpy().copy()
        out[f"samp_gt_boxes{i}"] = s.gt_boxes.tensor.numpy().copy()
    out["scores"], out["deltas"] = captured["scores"], captured["deltas"]
    sd = dict(model.named_parameters())
    for k in GRAD_FULL:
        out["grad/" + k] = sd[k].grad.numpy().copy()
    for k in GRAD_SAMPLED:
        out["grads/" + k] = sd[k].grad.numpy().ravel()[::STRIDE].copy()
    out["frozen"] = np.array([k for k, p in sd.items() if not p.requires_grad])
    np.savez_compressed(os.path.join(HERE, "stage3_a.npz"), **out)
    # ---- the oracle against the reference
    ol, oaux, og = FO.supervised_forward(P, [d["image"].numpy() for d in data], gts, K, FO.Perm("s3a"), want_grads=True)
    worst = max(abs(ol[k] - float(out["loss/" + k])) / abs(float(out["loss/" + k])) for k in ol)
    ok = all(np.array_equal(oaux["rpn_labels"][i], out[f"rpn_labels{i}"]) for i in range(2))
    okp = all(same_box_set(oaux["proposals"][i]["boxes"], out[f"prop_boxes{i}"]) for i in range(2))
    oks = all(np.array_equal(oaux["sampled"][i]["gt_classes"], out[f"samp_classes{i}"]) for i in range(2))
    gerr = max(float(np.abs(og[k] - out["grad/" + k]).max() / (np.abs(out["grad/" + k]).max() + 1e-30)) for k in GRAD_FULL)
    print(f"[stage3 a] losses {{{', '.join('%s %.6f' % (k, float(out['loss/' + k])) for k in ol)}}}; oracle: worst loss rel err {worst:.2e}, "
          f"rpn labels equal {ok}, proposals equal {okp} ({[len(out[f'prop_boxes{i}']) for i in range(2)]}), sampled classes equal {oks} "
          f"(fg {[int((out[f'samp_classes{i}'] < K).sum()) for i in range(2)]} of {[len(out[f'samp_classes{i}']) for i in range(2)]}), "
          f"grad err {gerr:.2e}")
    assert worst < 1e-5 and ok and okp and oks and gerr < 1e-4


def run_weak():
    P = FO.make_params(K, tag="s3w", head_scale=12.0)
    model = ref_shim_d2.build_reference_model(ns, K)
    load_params(model, P)
    model.train()                                  # the teacher is never switched to eval (trainer.py:474-477)
    data, _ = inputs("s3w", with_gt=False)
    with ns.events.EventStorage(0), torch.no_grad():
        _, prop_rpn, prop_roih, _ = model(data, branch="unsup_data_weak")
    out = {"K": np.array(K), "sizes": np.array(SIZES), "head_scale": np.array(12.0)}
    for i in range(len(SIZES)):
        out[f"prop_boxes{i}"] = prop_rpn[i].proposal_boxes.tensor.numpy().copy()
        out[f"prop_logits{i}"] = prop_rpn[i].objectness_logits.numpy().copy()
        out[f"det_boxes{i}"] = prop_roih[i].pred_boxes.tensor.numpy().copy()
        out[f"det_scores{i}"] = prop_roih[i].scores.numpy().copy()
        out[f"det_classes{i}"] = prop_roih[i].pred_classes.numpy().copy()
        # process_pseudo_label / threshold_bbox (trainer.py:361-403): keep detections with score > 0.7
        keep = out[f"det_scores{i}"] > 0.7
        out[f"pseudo_boxes{i}"] = out[f"det_boxes{i}"][keep]
        out[f"pseudo_classes{i}"] = out[f"det_classes{i}"][keep]
    np.savez_compressed(os.path.join(HERE, "stage3_w.npz"), **out)
    oprops, odets = FO.weak_forward(P, [d["image"].numpy() for d in data], K)
    okp = all(same_box_set(oprops[i]["boxes"], out[f"prop_boxes{i}"]) for i in range(2))
    okd = all(np.array_equal(odets[i]["pred_classes"], out[f"det_classes{i}"]) and
              np.allclose(odets[i]["scores"], out[f"det_scores{i}"], rtol=1e-4, atol=1e-6) for i in range(2))
    print(f"[stage3 w] proposals {[len(out[f'prop_boxes{i}']) for i in range(2)]}, detections {[len(out[f'det_scores{i}']) for i in range(2)]}, "
          f"pseudo boxes (score > 0.7) {[len(out[f'pseudo_boxes{i}']) for i in range(2)]}; oracle: proposals equal {okp}, detections equal {okd}")
    assert okp and okd


EVAL_OUT = [(131, 175), (100, 70)]                 # dataset sizes of the two eval inputs: an up-scale and an anisotropic down-scale


def run_eval():
    P = FO.make_params(K, tag="s3w", head_scale=12.0)
    model = ref_shim_d2.build_reference_model(ns, K)
    load_params(model, P)
    model.eval()
    data, _ = inputs("s3w", with_gt=False)
    for d, (oh, ow) in zip(data, EVAL_OUT):
        d["height"], d["width"] = oh, ow
    with ns.events.EventStorage(0), torch.no_grad():
        res = model(data)
        raw = model.inference(data, do_postprocess=False)
    out = {"K": np.array(K), "sizes": np.array(SIZES), "out_sizes": np.array(EVAL_OUT), "head_scale": np.array(12.0)}
    for i, r in enumerate(res):
        inst = r["instances"]
        assert tuple(inst.image_size) == EVAL_OUT[i]
        out[f"det_boxes{i}"] = inst.pred_boxes.tensor.numpy().copy()
        out[f"det_scores{i}"] = inst.scores.numpy().copy()
        out[f"det_classes{i}"] = inst.pred_classes.numpy().copy()
        out[f"raw_boxes{i}"] = raw[i].pred_boxes.tensor.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "stage3_e.npz"), **out)
    print(f"[stage3 e] eval detections {[len(out[f'det_scores{i}']) for i in range(2)]} (raw {[len(out[f'raw_boxes{i}']) for i in range(2)]}) "
          f"in frames {EVAL_OUT}; top scores {[out[f'det_scores{i}'][:2].round(3).tolist() for i in range(2)]}")


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["a", "w", "e"]
    if "e" in which:
        run_eval()
    if "a" in which:
        run_supervised()
    if "w" in which:
        run_weak()
