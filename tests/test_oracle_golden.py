"""CPU: the oracle restatement vs the fixtures generated from the reference's own Python
(tests/golden/make_golden.py) and vs the known-answer vectors of the reference's unit tests."""
import os

import numpy as np
import pytest
import torch

from oracle import oicr_oracle as O


@pytest.mark.parametrize("case", ["s0", "s1", "c0"])
def test_e2e_losses_grads_and_integer_outputs(case, golden_dir):
    g = np.load(os.path.join(golden_dir, f"e2e_{case}.npz"), allow_pickle=False)
    K, R, H, W = int(g["K"]), int(g["R"]), int(g["H"]), int(g["W"])
    dan = tuple(int(x) for x in g["dan"])
    P = O.make_params(K, dan, tag="p" + case, head_scale=float(g["head_scale"]))
    views, gt = O.make_views(H, W, R, n_gt=int(g["n_gt"]), K=K, tag="v" + case)
    masks = O.make_masks(R, dan, tag="m" + case)
    assert np.array_equal(gt, g["gt"])
    losses, aux, grads = O.oicr_plus_iteration(P, views, gt, masks, K=K, want_grads=True)
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(v - ref) <= 1e-6 * max(abs(ref), 1e-6), (k, v, ref)      # fp32 losses: 1e-6 rel on CPU
    for k in range(4):                                                      # integer outputs: bit exact
        r = aux["rounds"][k]
        assert np.array_equal(r["pgt"]["index"], g[f"r{k}/pgt_index"])
        assert np.array_equal(r["pgt"]["classes"], g[f"r{k}/pgt_classes"])
        assert np.array_equal(r["labels"]["gt_classes"], g[f"r{k}/gt_classes"])
        assert np.array_equal(r["labels"]["gt_index"], g[f"r{k}/gt_index"])
        np.testing.assert_allclose(r["labels"]["gt_weights"], g[f"r{k}/gt_weights"], rtol=1e-6)
    for v in range(4):
        np.testing.assert_allclose(aux["wsddn_scores"][v], g[f"wsddn_v{v}"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(aux["fc7"][v], g[f"fc7_v{v}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(aux["plain5"][0].ravel()[::997], g["plain5_v0_sample"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(aux["plain5"][3].ravel()[::997], g["plain5_v3_sample"], rtol=1e-5, atol=1e-5)
    for key in g.files:
        if key.startswith("grad/"):
            ref = g[key]; got = grads[key[5:]]
            assert np.abs(got - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-20), key
        elif key.startswith("grads/"):
            ref = g[key]; got = grads[key[6:]].ravel()[::997]
            assert np.abs(got - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-20), key


@pytest.mark.parametrize("case", ["a", "b"])
def test_mining_and_labels_bit_exact(case, golden_dir):
    g = np.load(os.path.join(golden_dir, f"mining_{case}.npz"), allow_pickle=False)
    R, K = int(g["R"]), int(g["K"])
    views, _ = O.make_views(256, 320, R, n_gt=len(g["gt"]), K=K, tag=str(g["boxes_tag"]))
    boxes = views[0]["boxes"]
    for variant in ("wsddn", "refine"):
        o = O.get_pgt_mist(g[f"{variant}/scores"], boxes, g["gt"])
        l = O.label_proposals(o, boxes, K)
        assert np.array_equal(o["index"], g[f"{variant}/pgt_index"])
        assert np.array_equal(o["classes"], g[f"{variant}/pgt_classes"])
        assert np.array_equal(o["scores"], g[f"{variant}/pgt_scores"])
        assert np.array_equal(l["gt_classes"], g[f"{variant}/gt_classes"])
        assert np.array_equal(l["gt_index"], g[f"{variant}/gt_index"])
        assert np.array_equal(l["gt_weights"], g[f"{variant}/gt_weights"])
        assert np.array_equal(l["gt_boxes"], g[f"{variant}/gt_boxes"])


def test_vector_nms_equals_the_line_by_line_restatement():
    """nms_keep (numpy vector form used at full size) == nms_keep_scalar (nms_cpu.cpp restated pair by pair), incl. ties,
    degenerate boxes and both thresholds of the path (0.01 mining, 0.3 inference)"""
    rng = np.random.RandomState(3)
    for n, thr in ((1, 0.01), (37, 0.01), (300, 0.3), (300, 0.01), (64, 0.5)):
        xy = rng.rand(n, 2).astype(np.float32) * 200
        wh = rng.rand(n, 2).astype(np.float32) * 120
        wh[rng.rand(n) < 0.1] = 0                                   # zero-area boxes (0/0 -> NaN > thr is False)
        b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        s = np.round(rng.rand(n).astype(np.float32), 1)             # many exact ties
        assert np.array_equal(O.nms_keep(b, s, thr), O.nms_keep_scalar(b, s, thr)), (n, thr)


def test_inference_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "infer_s0.npz"))
    e = np.load(os.path.join(golden_dir, "e2e_s0.npz"))
    dan = tuple(int(x) for x in e["dan"])
    P = O.make_params(20, dan, tag="ps0", head_scale=float(e["head_scale"]))
    views, _ = O.make_views(int(e["H"]), int(e["W"]), int(e["R"]), n_gt=int(e["n_gt"]), K=20, tag="vs0")
    o = O.oicr_plus_inference(P, views[0]["image"], views[0]["boxes"], views[0]["obj"], K=20)
    assert np.array_equal(o["classes"], g["pred_classes"])
    np.testing.assert_allclose(o["scores"], g["scores"], rtol=1e-6)
    np.testing.assert_allclose(o["boxes"], g["pred_boxes"], rtol=1e-5, atol=1e-4)


# ---- known-answer vectors held by the reference's own unit tests (data only) ----------------
def test_kat_pairwise_iou():
    """uwsod/tests/structures/test_boxes.py:149-173"""
    b1 = np.array([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]], np.float32)
    b2 = np.array([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.5, 1.0], [0.0, 0.0, 1.0, 0.5], [0.0, 0.0, 0.5, 0.5],
                   [0.5, 0.5, 1.0, 1.0], [0.5, 0.5, 1.5, 1.5]], np.float32)
    exp = np.array([[1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]] * 2, np.float32)
    np.testing.assert_allclose(O.pairwise_iou(b1, b2), exp, rtol=1e-6)


def test_kat_matcher():
    """uwsod/tests/modeling/test_matcher.py:14-27 (thresholds [0.3,0.7], labels [0,-1,1], low-quality on)"""
    iou = np.array([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]], np.float32)
    m, l = O.matcher(iou, thresholds=(0.3, 0.7), labels=(0, -1, 1), allow_low_quality_matches=True)
    assert m.tolist() == [1, 1, 2, 0]
    assert l.tolist() == [-1, 1, 0, 1]


def test_kat_box2box_roundtrip():
    """uwsod/tests/modeling/test_box2box_transform.py:16-30: apply_deltas(get_deltas(src,dst),src)==dst"""
    torch.manual_seed(0)

    def rb(n):
        xy = torch.rand(n, 2) * 100
        wh = torch.rand(n, 2) * 100 + 1
        return torch.cat([xy, xy + wh], 1)
    src, dst = rb(10), rb(10)
    d = O.get_deltas(src, dst)
    np.testing.assert_allclose(O.apply_deltas(d, src).numpy(), dst.numpy(), rtol=1e-4, atol=1e-3)


def test_roipool_c_oracle_edge_cases():
    """Empty bins -> 0 / -1; strict '>' keeps the first max; malformed ROI forced to 1x1
    (ROILoopPool_cpu.cpp:35-38,56-58,66)."""
    feat = np.zeros((1, 2, 6, 8), np.float32)
    feat[0, 0] = np.arange(48).reshape(6, 8)
    feat[0, 1] = 5.0                                   # all ties -> first index of each bin
    rois = np.array([[0, 0, 0, 63, 47], [0, 40, 40, 8, 8], [0, 1000, 1000, 1100, 1100]], np.float32)
    out, arg = O.roi_pool_fwd(feat, rois, 1.0 / 8, 7, 7)
    assert out.shape == (3, 2, 7, 7)
    assert out[0, 0].max() == 47 and arg[0, 0, 5, 6] == 47
    assert (arg[0, :, 6, :] == -1).all()                      # bin row clipped away by the image edge
    assert arg[0, 1, 0, 0] == 0 and out[0, 1, 0, 0] == 5.0
    assert (arg[2] == -1).all() and (out[2] == 0).all()          # fully outside -> empty bins
    assert (arg[1, 0] == arg[1, 0, 0, 0]).all()                  # 1x1 ROI: every bin sees the same pixel
    g = np.ones_like(out)
    gi = O.roi_pool_bwd(g, arg, rois, feat.shape)
    assert gi.sum() == (arg >= 0).sum()
