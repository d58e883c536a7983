"""Generate the committed golden fixtures by running the REFERENCE's own Python
(/root/reference, imported through ref_shim.py) on closed-form inputs.

Run in the build container only:   python tests/golden/make_golden.py
Outputs: tests/golden/e2e_<case>.npz, tests/golden/mining_<case>.npz, tests/golden/infer_s0.npz, tests/golden/input_a.npz

The fixtures hold inputs' recipe (detgen tags/shapes -> regenerated, not stored), the
reference's outputs (losses, mined pseudo-GT, labels, scores, gradient samples) and nothing
from the reference's source text.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import ref_shim  # noqa: E402
from oracle import oicr_oracle as O  # noqa: E402

ns = ref_shim.install()
Boxes, Instances = ns.boxes.Boxes, ns.instances.Instances
EventStorage = ns.events.EventStorage

E2E_CASES = {
    # name: (H, W, R, n_gt, dan_dim, head_scale)
    "s0": (96, 128, 37, 2, (256, 256), 30.0),
    "s1": (128, 160, 64, 3, (256, 256), 60.0),
    "c0": (112, 144, 150, 5, (256, 256), 6.0, 80),       # COCO-shaped head: 80 classes, top-p% = 15 proposals per class
}
GRAD_KEYS_FULL = ["roi_heads.box_predictor.cls.weight", "roi_heads.box_predictor.det.bias",
                  "roi_heads.box_refinery_3.bbox_pred.weight", "roi_heads.box_refinery_0.cls_score.weight",
                  "roi_heads.box_head.fc1.bias", "roi_heads.box_head.fc2.bias",
                  "backbone.plain5.0.conv3.bias", "backbone.plain3.0.conv1.bias"]
GRAD_KEYS_SAMPLED = ["roi_heads.box_head.fc1.weight", "roi_heads.box_head.fc2.weight",
                     "backbone.plain5.0.conv3.weight", "backbone.plain5.0.conv1.weight",
                     "backbone.plain4.0.conv2.weight", "backbone.plain3.0.conv1.weight",
                     "backbone.plain3.0.conv3.weight"]
SAMPLE_STRIDE = 997


def load_params(model, P):
    sd = model.state_dict()
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k].copy_(torch.from_numpy(v))


def to_batched_inputs(views, gt):
    d = {}
    for name, v in zip(["1", "1_flip", "2", "2_flip"], views):
        h, w = v["image"].shape[1:]
        p = Instances((h, w)); p.proposal_boxes = Boxes(torch.from_numpy(v["boxes"]))
        p.objectness_logits = torch.from_numpy(v["obj"])
        t = Instances((h, w)); t.gt_boxes = Boxes(torch.zeros(len(gt), 4)); t.gt_classes = torch.from_numpy(gt)
        d["image" + name] = torch.from_numpy(v["image"]); d["proposals" + name] = p; d["instances" + name] = t
    return [d]


class DropoutPatch:
    """Replace F.dropout by recorded keep-masks, consumed in call order
    (box_head.py:90: view order [1,1_flip,2,2_flip], fc1 then fc2)."""

    def __init__(self, masks):
        self.q = [torch.from_numpy(m).float() for v in masks for m in v]
        self.i = 0

    def __enter__(self):
        self.orig = F.dropout

        def fake(x, p=0.5, training=True, inplace=False):
            if not training:
                return x
            m = self.q[self.i]; self.i += 1
            return x * m / (1 - p)
        F.dropout = fake
        return self

    def __exit__(self, *a):
        F.dropout = self.orig


def run_e2e(case):
    H, W, R, n_gt, dan, hs = E2E_CASES[case][:6]
    K = E2E_CASES[case][6] if len(E2E_CASES[case]) > 6 else 20
    P = O.make_params(K, dan, tag="p" + case, head_scale=hs)
    views, gt = O.make_views(H, W, R, n_gt=n_gt, K=K, tag="v" + case)
    masks = O.make_masks(R, dan, tag="m" + case)
    model = ref_shim.build_reference_model(ns, K, dan)
    load_params(model, P)
    model.train()
    heads = model.roi_heads
    rec = {"pgt": [], "lab": [], "fc7": [], "wsddn": [], "refine": [[] for _ in range(4)], "plain5": []}
    orig_mist, orig_label = heads.get_pgt_mist, heads.label_and_sample_proposals

    def mist(*a, **k):
        t = orig_mist(*a, **k); rec["pgt"].append(t[0]); return t

    def label(*a, **k):
        t = orig_label(*a, **k); rec["lab"].append(t[0]); return t
    heads.get_pgt_mist, heads.label_and_sample_proposals = mist, label
    model.backbone.register_forward_hook(lambda m, i, o: rec["plain5"].append(o["plain5"].detach().clone()))
    heads.box_head.register_forward_hook(lambda m, i, o: rec["fc7"].append(o.detach().clone()))
    heads.box_predictor.register_forward_hook(lambda m, i, o: rec["wsddn"].append(o[0].detach().clone()))
    for k in range(4):
        heads.box_refinery[k].register_forward_hook(
            lambda m, i, o, k=k: rec["refine"][k].append((o[0].detach().clone(), o[1].detach().clone())))
    with EventStorage(0), DropoutPatch(masks):
        losses = model(to_batched_inputs(views, gt))
        total = sum(losses.values())
        total.backward()
    out = {"meta_case": case, "H": H, "W": W, "R": R, "n_gt": n_gt, "K": K, "dan": np.array(dan),
           "head_scale": hs, "gt": gt}
    for k, v in losses.items():
        out["loss/" + k] = np.float64(v.item())
    f1, f2 = rec["plain5"]
    out["plain5_v0_sample"] = f1[0].numpy().ravel()[::SAMPLE_STRIDE]
    out["plain5_v3_sample"] = f2[1].numpy().ravel()[::SAMPLE_STRIDE]
    out["plain5_shape"] = np.array(f1.shape)
    for v in range(4):
        out[f"fc7_v{v}"] = rec["fc7"][v].numpy()
        out[f"wsddn_v{v}"] = rec["wsddn"][v].numpy()
    for k in range(4):
        t, l = rec["pgt"][k], rec["lab"][k]
        out[f"r{k}/pgt_index"] = t.gt_index.numpy(); out[f"r{k}/pgt_classes"] = t.gt_classes.numpy()
        out[f"r{k}/pgt_scores"] = t.gt_scores.numpy(); out[f"r{k}/pgt_boxes"] = t.gt_boxes.tensor.numpy()
        out[f"r{k}/gt_classes"] = l.gt_classes.numpy(); out[f"r{k}/gt_index"] = l.gt_index.numpy()
        out[f"r{k}/gt_weights"] = l.gt_weights.numpy()
        for v in range(4):
            out[f"r{k}/logits_v{v}"] = rec["refine"][k][v][0].numpy()
    sd = dict(model.named_parameters())
    for k in GRAD_KEYS_FULL:
        out["grad/" + k] = sd[k].grad.numpy()
    for k in GRAD_KEYS_SAMPLED:
        out["grads/" + k] = sd[k].grad.numpy().ravel()[::SAMPLE_STRIDE]
    frozen = [k for k, p in sd.items() if not p.requires_grad]
    out["frozen"] = np.array(frozen)
    np.savez_compressed(os.path.join(HERE, f"e2e_{case}.npz"), **out)

    # cross-check the oracle restatement against the reference right here
    ol, aux, og = O.oicr_plus_iteration(P, views, gt, masks, K=K, want_grads=True)
    worst = max(abs(ol[k] - losses[k].item()) / max(abs(losses[k].item()), 1e-12) for k in ol)
    gerr = max(np.abs(og[k] - sd[k].grad.numpy()).max() / (np.abs(sd[k].grad.numpy()).max() + 1e-20)
               for k in GRAD_KEYS_FULL + GRAD_KEYS_SAMPLED)
    same_idx = all(np.array_equal(aux["rounds"][k]["pgt"]["index"], rec["pgt"][k].gt_index.numpy()) and
                   np.array_equal(aux["rounds"][k]["labels"]["gt_classes"], rec["lab"][k].gt_classes.numpy())
                   for k in range(4))
    print(f"[e2e {case}] losses:", {k: round(v.item(), 6) for k, v in losses.items()})
    print(f"[e2e {case}] pgt sizes:", [len(rec['pgt'][k]) for k in range(4)],
          "fg counts:", [int(((rec['lab'][k].gt_classes >= 0) & (rec['lab'][k].gt_classes < K)).sum()) for k in range(4)])
    print(f"[e2e {case}] oracle-vs-reference: loss rel err {worst:.2e}, grad rel-to-max err {gerr:.2e}, "
          f"integer outputs identical: {same_idx}")
    assert worst < 1e-5 and gerr < 1e-4 and same_idx


def run_mining(case, R, K, G, seed):
    """Stage-level vectors: reference get_pgt_mist + label_and_sample_proposals on peaky synthetic
    scores (exercises threshold mask, cross-class NMS and ignore/bg/fg labelling)."""
    from oracle import detgen
    model = ref_shim.build_reference_model(ns, K, (8, 8))
    heads = model.roi_heads
    views, _ = O.make_views(256, 320, R, n_gt=G, K=K, tag=f"mine{case}")
    boxes = views[0]["boxes"]
    gt = np.sort(np.unique(detgen.randint(f"mine{case}gt", (G,), 0, K)))
    out = {"R": R, "K": K, "gt": gt, "boxes_tag": f"mine{case}"}
    for variant, ncol in (("wsddn", K), ("refine", K + 1)):
        raw = detgen.uniform(f"mine{case}{variant}", (R, ncol)) ** 8       # peaky
        if variant == "refine":
            sc = raw / raw.sum(1, keepdims=True)                           # rows sum to 1 like a softmax
        else:
            sc = raw / raw.sum(0, keepdims=True) * 3.0                     # columns sum to 3: many >= 0.05
        sc = sc.astype(np.float32)
        heads.gt_classes_img_int = [torch.from_numpy(gt)]
        p = Instances((256, 320)); p.proposal_boxes = Boxes(torch.from_numpy(boxes))
        p.objectness_logits = torch.from_numpy(views[0]["obj"])
        with EventStorage(0):
            t = heads.get_pgt_mist([p.proposal_boxes], torch.from_numpy(sc), [p], top_pro=0.10, thres=0.05)
            l = heads.label_and_sample_proposals([p], t)[0]
        t = t[0]
        out[f"{variant}/scores"] = sc
        out[f"{variant}/pgt_index"] = t.gt_index.numpy(); out[f"{variant}/pgt_classes"] = t.gt_classes.numpy()
        out[f"{variant}/pgt_scores"] = t.gt_scores.numpy()
        out[f"{variant}/gt_classes"] = l.gt_classes.numpy(); out[f"{variant}/gt_index"] = l.gt_index.numpy()
        out[f"{variant}/gt_weights"] = l.gt_weights.numpy(); out[f"{variant}/gt_boxes"] = l.gt_boxes.tensor.numpy()
        o = O.get_pgt_mist(sc, boxes, gt)
        ol = O.label_proposals(o, boxes, K)
        ok = (np.array_equal(o["index"], t.gt_index.numpy()) and np.array_equal(ol["gt_classes"], l.gt_classes.numpy())
              and np.array_equal(ol["gt_index"], l.gt_index.numpy()) and np.array_equal(ol["gt_weights"], l.gt_weights.numpy()))
        cls = l.gt_classes.numpy()
        print(f"[mining {case}/{variant}] pre-nms->kept {len(o['pre_nms']['index'])}->{len(o['index'])}; "
              f"fg {int(((cls >= 0) & (cls < K)).sum())} ig {int((cls == -1).sum())} bg {int((cls == K).sum())}; "
              f"oracle identical: {ok}")
        assert ok
    np.savez_compressed(os.path.join(HERE, f"mining_{case}.npz"), **out)


def run_infer(case="s0"):
    H, W, R, n_gt, dan, hs = E2E_CASES[case]
    K = 20
    P = O.make_params(K, dan, tag="p" + case, head_scale=hs)
    views, gt = O.make_views(H, W, R, n_gt=n_gt, K=K, tag="v" + case)
    model = ref_shim.build_reference_model(ns, K, dan)
    load_params(model, P)
    model.eval()
    v = views[0]
    p = Instances((H, W)); p.proposal_boxes = Boxes(torch.from_numpy(v["boxes"])); p.objectness_logits = torch.from_numpy(v["obj"])
    with torch.no_grad(), EventStorage(0):
        res, all_scores, all_boxes = model.inference([{"image": torch.from_numpy(v["image"]), "proposals": p}],
                                                     do_postprocess=False)
    inst = res[0]
    out = dict(pred_boxes=inst.pred_boxes.tensor.numpy(), scores=inst.scores.numpy(),
               pred_classes=inst.pred_classes.numpy(), all_scores=all_scores[0].numpy() if isinstance(all_scores, (list, tuple)) else all_scores.numpy())
    np.savez_compressed(os.path.join(HERE, f"infer_{case}.npz"), **out)
    o = O.oicr_plus_inference(P, v["image"], v["boxes"], v["obj"], K=K)
    n = len(out["scores"])
    print(f"[infer {case}] {n} dets; oracle dets {len(o['scores'])}; "
          f"max|score diff| {np.abs(np.sort(o['scores'])[::-1][:n] - np.sort(out['scores'])[::-1]).max():.2e}")


def input_case_boxes(seed=0, n=600, h=375, w=500):
    """synthetic proposal list in an (h, w) image: integer-cornered boxes (selective-search style) with exact duplicates,
    sub-pixel near-duplicates that only collide after a resize, zero-area boxes and boxes on the border"""
    rng = np.random.RandomState(seed)
    x1 = rng.randint(0, w - 2, n); y1 = rng.randint(0, h - 2, n)
    x2 = np.minimum(x1 + rng.randint(1, w, n), w - 1); y2 = np.minimum(y1 + rng.randint(1, h, n), h - 1)
    b = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
    b[50:80] = b[10:40]                                   # exact duplicates
    b[100:120] = b[200:220] + np.float32(0.24)            # collide after rounding
    b[130:140, 2] = b[130:140, 0]                         # zero width
    b[140:150, 3] = b[140:150, 1]                         # zero height
    b[150:155] = [0, 0, w - 1, h - 1]                     # whole image, repeated
    b[160:170, 2] = b[160:170, 0] + 1                     # 1 px wide: may collapse under a down-scale + round
    logits = rng.rand(n).astype(np.float32)
    return b, logits


def run_input(case="a"):
    """masks of the reference's own `Boxes.clip / unique_boxes / nonempty` (structures/boxes.py) on the four views of
    `DatasetMapperMultiInput`; the fvcore box transforms are the restated rule (oracle/input_oracle.py)."""
    from oracle import input_oracle as IO
    if not hasattr(np, "int"):
        np.int = int                                       # boxes.py:224 uses the alias numpy 2 removed
    h, w = 375, 500
    boxes, logits = input_case_boxes(0, 600, h, w)
    hw1 = IO.shortest_edge_shape(h, w, 480, 2000)
    hw2 = IO.shortest_edge_shape(h, w, 1200, sys.maxsize)
    out = {"orig_hw": np.array([h, w]), "hw1": np.array(hw1), "hw2": np.array(hw2), "boxes": boxes, "logits": logits}
    joint = None
    for name, hw, flip in (("1", hw1, False), ("2", hw2, False), ("1_flip", hw1, True), ("2_flip", hw2, True)):
        tb = Boxes(torch.from_numpy(IO.apply_box(boxes, (h, w), hw, flip)))
        tb.clip(hw)
        final_keep = torch.zeros(len(tb)).bool()
        final_keep[tb.unique_boxes()] = True
        final_keep = final_keep & tb.nonempty(threshold=0)
        out["boxes" + name] = tb.tensor.numpy().copy()
        out["keep" + name] = final_keep.numpy().copy()
        joint = final_keep if joint is None else joint & final_keep
    out["keep"] = joint.numpy()
    np.savez_compressed(os.path.join(HERE, f"input_{case}.npz"), **out)
    print(f"[input {case}] views {hw1} {hw2}; kept {int(joint.sum())} of {len(boxes)}; "
          + " ".join(f"{k}={int(out['keep' + k].sum())}" for k in ("1", "2", "1_flip", "2_flip")))


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["e2e", "mining", "infer", "input"]
    if "input" in which:
        run_input("a")
    if "mining" in which:
        run_mining("a", 500, 20, 3, 0)
        run_mining("b", 2000, 80, 5, 1)
    if "e2e" in which:
        for c in E2E_CASES:
            if c in which or not any(w in E2E_CASES for w in which):
                run_e2e(c)
    if "infer" in which:
        run_infer("s0")
