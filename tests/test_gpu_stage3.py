"""GPU: the Stage-3 detector (sos-wsod_amd/frcnn.py: ResNet-50-FPN Faster R-CNN of the Unbiased-Teacher step) against
(a) the fixtures written by RUNNING the reference's own detector (tests/golden/make_stage3_golden.py -> stage3_a / stage3_w) and
(b) the oracle (oracle/frcnn_oracle.py, itself pinned by those fixtures) — kernels first, then the branches, then one burn-in and
one semi-supervised iteration of semisup.SemiSupStep on the real modules (unbias/ubteacher/engine/trainer.py:436-549)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import frcnn_oracle as FO  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def ops():
    import sos_wsod_amd  # noqa: F401
    import sos_wsod_amd.ops as ops
    return ops


class _Keys:
    """the closed-form sampling keys of the fixtures (oracle.frcnn_oracle.Perm) as the product's sampler"""

    def __init__(self, tag):
        self.perm = FO.Perm(tag)

    def next_seed(self):
        from oracle import detgen
        k = self.perm.k
        self.perm.k += 1
        return detgen.fnv1a64(f"{self.perm.tag}perm{k}")


def _same_boxes(a, b, atol=1e-2):
    if len(a) != len(b):
        return False
    if len(a) == 0:
        return True
    d = np.abs(a[:, None, :] - b[None, :, :]).max(2)
    return bool((d.min(1) <= atol).all() and (d.min(0) <= atol).all())


def _model(K, P, tag, dtype=torch.float32):
    from sos_wsod_amd.frcnn import TwoStagePseudoLabGeneralizedRCNN
    m = TwoStagePseudoLabGeneralizedRCNN(num_classes=K, compute_dtype=dtype, sampler=_Keys(tag)).cuda()
    sd = m.state_dict()
    assert set(sd) == set(P), (sorted(set(sd) - set(P))[:5], sorted(set(P) - set(sd))[:5])       # the reference's state-dict names
    with torch.no_grad():
        for k, v in P.items():
            assert tuple(sd[k].shape) == tuple(v.shape), k
            sd[k].copy_(torch.from_numpy(v))
    return m


def _inputs(tag, t, K, with_gt=True):
    from sos_wsod_amd.structures import Boxes, Instances
    data, gts = [], []
    n_gt = t["n_gt"] if "n_gt" in t.files else [0] * len(t["sizes"])
    for i, ((h, w), n) in enumerate(zip(t["sizes"], n_gt)):
        h, w = int(h), int(w)
        d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).cuda(), "height": h, "width": w}
        if with_gt:
            b, c = FO.make_gt(h, w, int(n), K, f"{tag}{i}")
            inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
            d["instances"] = inst
            gts.append((b, c))
        data.append(d)
    return data, gts


# ------------------------------------------------------------------------------------------ kernels
def test_roi_align_forward_backward_against_the_c_oracle(ops):
    torch.manual_seed(0)
    N, C, H, W, R = 2, 16, 20, 28, 120
    feat = torch.randn(N, C, H, W)
    x1 = torch.rand(R) * 180; y1 = torch.rand(R) * 120
    rois = torch.stack([(torch.arange(R) % N).float(), x1, y1, x1 + 2 + torch.rand(R) * 150, y1 + 2 + torch.rand(R) * 100], 1)
    rois[:4, 1:] = torch.tensor([[-30.0, -20.0, 10.0, 12.0], [200.0, 140.0, 260.0, 190.0], [5.0, 5.0, 5.5, 5.5], [0.0, 0.0, 224.0, 160.0]])
    scale = 1.0 / 8
    ref = FO.roi_align_fwd(feat.numpy(), rois.numpy(), scale)
    f = feat.permute(0, 2, 3, 1).contiguous().cuda()
    sel = torch.arange(R, dtype=torch.int32).cuda()
    out = torch.zeros(R, C * 49, device="cuda")
    ops.roi_align_fwd(f, rois.cuda(), sel, out, scale)
    np.testing.assert_allclose(out.cpu().numpy().reshape(ref.shape), ref, rtol=1e-5, atol=1e-6)
    # only the listed rows are written
    out2 = torch.full((R, C * 49), 7.0, device="cuda")
    ops.roi_align_fwd(f, rois.cuda(), sel[::2].contiguous(), out2, scale)
    assert torch.equal(out2[1::2], torch.full_like(out2[1::2], 7.0)) and torch.equal(out2[::2], out[::2])
    g = torch.randn(R, C, 7, 7)
    gref = FO.roi_align_bwd(g.numpy(), rois.numpy(), scale, feat.shape)
    d = torch.zeros(N, H, W, C, device="cuda")
    ops.roi_align_bwd(g.view(R, -1).cuda().contiguous(), rois.cuda(), sel, d, scale)
    np.testing.assert_allclose(d.permute(0, 3, 1, 2).cpu().numpy(), gref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_align_backward_fixed_point_form_is_reproducible(ops, dtype):
    """sw_roi_align_bwd_fx (round 6): the ROIAlign backward as 64-bit fixed-point integer accumulation — against the C oracle (the
    reference's sequential CPU loop, ROIAlign_cpu.cpp:286-400), BIT-equal over repeated runs with 2000 heavily overlapping ROIs (the f32
    atomic form is checked to be the close-but-not-reproducible one it replaces), zero gradients, and a NaN in the gradient poisons the map"""
    torch.manual_seed(3)
    N, C, H, W, R = 2, 32, 25, 38, 2000
    x1 = torch.rand(R) * 240; y1 = torch.rand(R) * 150
    rois = torch.stack([(torch.arange(R) % N).float(), x1, y1, x1 + 2 + torch.rand(R) * 200, y1 + 2 + torch.rand(R) * 120], 1)
    rois[:4, 1:] = torch.tensor([[-30.0, -20.0, 10.0, 12.0], [280.0, 180.0, 330.0, 230.0], [5.0, 5.0, 5.5, 5.5], [0.0, 0.0, 304.0, 200.0]])
    scale = 1.0 / 8
    g = (torch.randn(R, C, 7, 7) * torch.exp(torch.randn(R, 1, 1, 1) * 2)).to(dtype)       # heavy-tailed magnitudes
    gref = FO.roi_align_bwd(g.float().numpy(), rois.numpy(), scale, (N, C, H, W))
    gc, rc = g.view(R, -1).cuda().contiguous(), rois.cuda()
    sel = torch.arange(R, dtype=torch.int32).cuda()
    amax = ops.absmax(gc)
    assert float(amax) == float(g.float().abs().max())

    def run():
        acc = torch.zeros(N, H, W, C, device="cuda", dtype=torch.int64)
        ops.roi_align_bwd_fx(gc, rc, sel, acc, scale, amax)
        return ops.fx_to_float(acc, amax, torch.empty(N, H, W, C, device="cuda", dtype=torch.float32))
    a, b, c = run(), run(), run()
    assert torch.equal(a, b) and torch.equal(a, c)
    ref = torch.from_numpy(gref).permute(0, 2, 3, 1)
    assert float((a.cpu().double() - ref.double()).abs().max() / ref.abs().max()) < 3e-5     # (the oracle sums 2000 ROIs in f32, sequentially)
    d1 = torch.zeros(N, H, W, C, device="cuda"); ops.roi_align_bwd(gc, rc, sel, d1, scale)   # the float-atomic form: the same up to rounding order
    assert float((a - d1).abs().max() / d1.abs().max()) < 3e-5
    out16 = ops.fx_to_float(torch.zeros(N, H, W, C, device="cuda", dtype=torch.int64), amax, torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16))
    assert float(out16.float().abs().max()) == 0.0
    z = torch.zeros_like(gc); az = ops.absmax(z)
    acc = torch.zeros(N, H, W, C, device="cuda", dtype=torch.int64)
    ops.roi_align_bwd_fx(z, rc, sel, acc, scale, az)
    assert float(ops.fx_to_float(acc, az, torch.empty(N, H, W, C, device="cuda")).abs().max()) == 0.0
    gn = gc.clone(); gn[7, 11] = float("nan"); an = ops.absmax(gn)
    acc = torch.zeros(N, H, W, C, device="cuda", dtype=torch.int64)
    ops.roi_align_bwd_fx(gn, rc, sel, acc, scale, an)
    assert torch.isnan(ops.fx_to_float(acc, an, torch.empty(N, H, W, C, device="cuda"))).all()


def test_stem_pool_and_join_kernels_against_torch(ops):
    torch.manual_seed(1)
    # stem: 7x7 s2 p3 + affine + ReLU, then 3x3 s2 p1 max pool (resnet.py:334-359)
    N, H, W = 2, 70, 96
    x = torch.randn(N, 3, H, W) * 50
    w = torch.randn(64, 3, 7, 7) * 0.05; sc = torch.rand(64) + 0.5; sh = torch.randn(64) * 0.1
    ref = F.relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=3) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    x4 = torch.zeros(N, H, W, 4); x4[..., :3] = x.permute(0, 2, 3, 1)
    y = ops.stem_conv7x7(x4.cuda(), w.cuda(), sc.cuda(), sh.cuda(), torch.empty(N, ref.shape[2], ref.shape[3], 64, device="cuda"))
    assert float((y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()) < 1e-5
    pref = F.max_pool2d(ref.float(), kernel_size=3, stride=2, padding=1)
    p = ops.maxpool3x3s2(y, torch.empty(N, pref.shape[2], pref.shape[3], 64, device="cuda"))
    assert float((p.cpu().permute(0, 3, 1, 2) - pref).abs().max()) < 1e-4
    # subsample / scatter (odd sizes), residual join, FPN top-down join and its backward
    a = torch.randn(2, 9, 13, 16)
    s = ops.subsample2x(a.cuda(), torch.empty(2, 5, 7, 16, device="cuda"))
    assert torch.equal(s.cpu(), a[:, ::2, ::2])
    back = ops.scatter2x(s, torch.full((2, 9, 13, 16), 5.0, device="cuda")).cpu()
    want = torch.zeros_like(a); want[:, ::2, ::2] = a[:, ::2, ::2]
    assert torch.equal(back, want)
    b = torch.randn_like(a)
    assert torch.equal(ops.add_relu(a.cuda(), b.cuda(), torch.empty_like(a).cuda()).cpu(), F.relu(a + b))
    top = torch.randn(2, 4, 6, 8); lat = torch.randn(2, 8, 12, 8)
    up = ops.upsample2x_add(lat.cuda(), top.cuda(), torch.empty_like(lat).cuda()).cpu()
    want = lat + F.interpolate(top.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(up, want)
    g = torch.randn_like(lat)
    ds = ops.downsample2x_sum(g.cuda(), torch.empty_like(top).cuda()).cpu()
    want = F.avg_pool2d(g.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1) * 4
    assert float((ds - want).abs().max()) < 1e-5
    # padded preprocess (rcnn.py:220-228 + image_list.py)
    img = torch.randint(0, 256, (3, 37, 50), dtype=torch.uint8)
    o = ops.preprocess_pad(img.cuda(), torch.empty(64, 64, 4, device="cuda"), FO.PIXEL_MEAN, FO.PIXEL_STD).cpu()
    ref, _ = FO.preprocess([img.numpy()])
    assert torch.equal(o[:37, :50, :3].permute(2, 0, 1), ref[0, :, :37, :50]) and float(o[37:].abs().max()) == 0 and float(o[..., 3].abs().max()) == 0


def test_rpn_loss_kernel_against_the_oracle(ops):
    torch.manual_seed(2)
    N = 2
    anchors = np.concatenate(FO.grid_anchors([(10, 10), (5, 5), (3, 3), (2, 2), (1, 1)]), 0)
    A = len(anchors)
    logits = torch.randn(N, A) * 2; deltas = torch.randn(N, A, 4) * 0.3
    labels = torch.randint(-1, 2, (N, A))
    gt = torch.from_numpy(anchors)[None].repeat(N, 1, 1) + torch.randn(N, A, 4) * 2
    gt[..., 2:] = torch.maximum(gt[..., 2:], gt[..., :2] + 1)
    ref = FO.rpn_losses(anchors, [logits.clone().requires_grad_(True)], [deltas.clone().requires_grad_(True)],
                        [labels[i].numpy() for i in range(N)], [gt[i].numpy() for i in range(N)], batch_size=256)
    lt = logits.clone().requires_grad_(True); dt_ = deltas.clone().requires_grad_(True)
    r2 = FO.rpn_losses(anchors, [lt], [dt_], [labels[i].numpy() for i in range(N)], [gt[i].numpy() for i in range(N)], batch_size=256)
    (r2["loss_rpn_cls"] + 3.0 * r2["loss_rpn_loc"]).backward()
    out = torch.empty(2, device="cuda"); dl = torch.empty(N * A, device="cuda"); dd = torch.empty(N * A, 4, device="cuda")
    ops.rpn_loss(logits.reshape(-1).cuda(), deltas.reshape(-1, 4).cuda(), labels.reshape(-1).to(torch.int8).cuda(),
                 torch.from_numpy(anchors).cuda(), gt.reshape(-1, 4).cuda(), FO.RPN_BBOX_WEIGHTS, 1.0 / (256 * N), out, dl, dd)
    assert abs(float(out[0]) - float(ref["loss_rpn_cls"])) <= 1e-5 * abs(float(ref["loss_rpn_cls"]))
    assert abs(float(out[1]) - float(ref["loss_rpn_loc"])) <= 1e-5 * abs(float(ref["loss_rpn_loc"]))
    np.testing.assert_allclose(dl.cpu().numpy().reshape(N, A), lt.grad.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(3.0 * dd.cpu().numpy().reshape(N, A, 4), dt_.grad.numpy(), rtol=1e-4, atol=1e-8)


# ------------------------------------------------------------------------------------------ branches against the reference fixtures
def test_supervised_branch_matches_the_reference_generated_fixture(golden_dir):
    t = np.load(os.path.join(golden_dir, "stage3_a.npz"))
    K = int(t["K"])
    P = FO.make_params(K, tag="s3a", head_scale=float(t["head_scale"]))
    model = _model(K, P, "s3a")
    model.train()
    data, gts = _inputs("s3a", t, K)
    losses, _, _, _ = model(data, branch="supervised")
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    assert set(losses) == {"loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"}
    lab = model.proposal_generator.last_labels.cpu().numpy()
    # Proposals whose objectness logits tie to within a few ulps (the fixture holds a pair at 0.0953579 / 0.09535759) may swap places
    # against the reference's CPU run; the label sampling is positional, so such a swap exchanges one sampled background ROI.  Rows
    # are therefore compared exactly wherever the sampled box is the fixture's, the rest (<= 2 of 512) must be fixture proposals.
    swapped = 0
    keep_rows = []
    for i in range(2):
        assert np.array_equal(lab[i], t[f"rpn_labels{i}"]), i                                   # sampled anchor labels: bit exact
        s = model.roi_heads.last_sampled[i]
        assert np.array_equal(s.gt_classes.cpu().numpy(), t[f"samp_classes{i}"]), i              # sampled proposals' classes: bit exact
        got, want = s.proposal_boxes.tensor.cpu().numpy(), t[f"samp_boxes{i}"]
        same = np.abs(got - want).max(1) <= 1e-2
        pool = np.concatenate([t[f"prop_boxes{i}"], gts[i][0]], 0)
        for r in np.nonzero(~same)[0]:
            assert np.abs(pool - got[r]).max(1).min() <= 1e-2, (i, r, got[r])
        assert (~same).sum() <= 2, (i, int((~same).sum()))
        swapped += int((~same).sum())
        keep_rows.append(same)
    keep_rows = np.concatenate(keep_rows)
    for k, v in losses.items():
        ref = float(t["loss/" + k])
        tol = 1e-4 if (swapped == 0 or k.startswith("loss_rpn")) else 1e-3
        assert abs(float(v) - ref) <= tol * abs(ref), (k, float(v), ref, swapped)
    K1 = K + 1
    lg = model.roi_heads.last_logits.detach().cpu().numpy()
    np.testing.assert_allclose(lg[keep_rows, :K1], t["scores"][keep_rows], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(lg[keep_rows, K1:5 * K + 1], t["deltas"][keep_rows], rtol=1e-3, atol=1e-3)
    sd = dict(model.named_parameters())
    worst = ("", 0.0)
    for key in t.files:
        if key.startswith("grad/"):
            ref, got = t[key], sd[key[5:]].grad.cpu().numpy()
        elif key.startswith("grads/"):
            ref, got = t[key], sd[key[6:]].grad.cpu().numpy().ravel()[::997]
        else:
            continue
        err = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
        worst = max(worst, (key, err), key=lambda x: x[1])
        assert err <= (2e-3 if swapped == 0 else 5e-3), (key, err, swapped)      # (one exchanged ROI of 1024 moves the head's gradients by ~1e-3)
    for name in t["frozen"]:
        assert sd[str(name)].grad is None                                                        # FREEZE_AT 2: stem + res2
    print(f"stage-3 supervised branch: losses {dict((k, round(float(v), 6)) for k, v in losses.items())}; sampled ROIs exchanged by near-tied proposals: "
          f"{swapped}; worst gradient error {worst[1]:.1e} ({worst[0]})")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_fused_bottleneck_node_equals_the_layer_by_layer_form(golden_dir, mode):
    """BottleneckBlock as ONE autograd node (ReLU masks in the data-gradient epilogues, the shortcut gradient as the residual of
    conv1's data gradient, one subsample / scatter per stride-2 block) against the layer-by-layer nodes it replaces, on the fixture's
    supervised branch: losses identical (same forward kernels); fp32 gradients within 1e-5 of the tensor's largest element (a stage
    output feeds the next block and an FPN lateral: its three gradient terms are summed as lateral + (conv1 + shortcut) instead of
    (lateral + shortcut) + conv1), bf16 within 2e-2 (the block-input gradient is rounded once instead of twice)."""
    import sos_wsod_amd.frcnn as F
    t = np.load(os.path.join(golden_dir, "stage3_a.npz"))
    K = int(t["K"])
    P = FO.make_params(K, tag="s3a", head_scale=float(t["head_scale"]))
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    res = {}
    old = F.FUSED_BLOCKS
    try:
        for fused in (True, False):
            F.FUSED_BLOCKS = fused
            model = _model(K, P, "s3a", dtype=dtype)
            model.train()
            data, _ = _inputs("s3a", t, K)
            losses, _, _, _ = model(data, branch="supervised")
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            res[fused] = ({k: float(v) for k, v in losses.items()},
                          {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        F.FUSED_BLOCKS = old
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])
    assert set(res[True][1]) == set(res[False][1])
    worst = ("", 0.0)
    for n, g in res[True][1].items():
        r = res[False][1][n]
        err = float((g - r).abs().max() / (r.abs().max() + 1e-30))
        worst = max(worst, (n, err), key=lambda x: x[1])
        assert err <= (1e-5 if mode == "fp32" else 2e-2), (n, err)
    print(f"fused bottleneck node vs layer by layer ({mode}): {len(res[True][1])} gradients, worst relative difference {worst[1]:.1e} {worst[0]}")


def test_supervised_branch_bf16_mode_stays_close_to_the_fp32_fixture(golden_dir):
    """bf16 storage (activations, staged weights, gradients at layer boundaries; f32 accumulation, f32 logits and losses): the RPN
    losses — continuous in the features — within 1e-2 of the reference-generated fp32 values; the ROI-head losses depend on WHICH
    proposals survive top-k / NMS / sampling and are bounded at 5 % (classification) / 10 % (box regression: the few foreground
    ROIs) (printed); every gradient finite."""
    t = np.load(os.path.join(golden_dir, "stage3_a.npz"))
    K = int(t["K"])
    P = FO.make_params(K, tag="s3a", head_scale=float(t["head_scale"]))
    model = _model(K, P, "s3a", dtype=torch.bfloat16)
    model.train()
    data, _ = _inputs("s3a", t, K)
    losses, _, _, _ = model(data, branch="supervised")
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    rel = {k: abs(float(v) - float(t["loss/" + k])) / abs(float(t["loss/" + k])) for k, v in losses.items()}
    print("stage-3 bf16 vs the fp32 fixture, relative loss differences:", {k: "%.1e" % v for k, v in rel.items()})
    assert rel["loss_rpn_cls"] <= 1e-2 and rel["loss_rpn_loc"] <= 1e-2, rel            # measured 7e-5 / 1e-3
    assert rel["loss_cls"] <= 5e-2 and rel["loss_box_reg"] <= 1e-1, rel                # measured 4e-4 .. 8e-4 / 5e-3 .. 5e-2
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)


def test_teacher_weak_branch_and_pseudo_labels_match_the_reference_generated_fixture(golden_dir):
    from sos_wsod_amd.semisup import process_pseudo_label
    t = np.load(os.path.join(golden_dir, "stage3_w.npz"))
    K = int(t["K"])
    P = FO.make_params(K, tag="s3w", head_scale=float(t["head_scale"]))
    model = _model(K, P, "s3w")
    model.train()                                   # the teacher stays in training mode (trainer.py:474-477): train top-k counts
    data, _ = _inputs("s3w", t, K, with_gt=False)
    with torch.no_grad():
        _, props, dets, _ = model(data, branch="unsup_data_weak")
    pseudo, _ = process_pseudo_label(data, dets, 0.7, "roih")
    for i in range(2):
        assert _same_boxes(props[i].proposal_boxes.tensor.cpu().numpy(), t[f"prop_boxes{i}"])
        assert np.array_equal(dets[i].pred_classes.cpu().numpy(), t[f"det_classes{i}"])
        np.testing.assert_allclose(dets[i].scores.cpu().numpy(), t[f"det_scores{i}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dets[i].pred_boxes.tensor.cpu().numpy(), t[f"det_boxes{i}"], rtol=1e-4, atol=1e-2)
        assert len(pseudo[i]) == len(t[f"pseudo_boxes{i}"]) == 3
        assert np.array_equal(pseudo[i].gt_classes.cpu().numpy(), t[f"pseudo_classes{i}"])
        np.testing.assert_allclose(pseudo[i].gt_boxes.tensor.cpu().numpy(), t[f"pseudo_boxes{i}"], rtol=1e-4, atol=1e-2)


def test_eval_mode_inference_rescales_to_the_dataset_frame_like_the_reference(golden_dir):
    """`GeneralizedRCNN.inference` -> `_postprocess` -> `detector_postprocess` (detectron2/detectron2/modeling/meta_arch/rcnn.py:177-259,
    modeling/postprocessing.py:9-59): eval mode (RPN test top-k), "height" / "width" different from the network input; fixture
    written by running the reference model in eval mode with its own detector_postprocess (make_stage3_golden.py case e)"""
    t = np.load(os.path.join(golden_dir, "stage3_e.npz"))
    K = int(t["K"])
    P = FO.make_params(K, tag="s3w", head_scale=float(t["head_scale"]))
    model = _model(K, P, "s3w")
    model.eval()
    data, _ = _inputs("s3w", t, K, with_gt=False)
    for d, (oh, ow) in zip(data, t["out_sizes"]):
        d["height"], d["width"] = int(oh), int(ow)
    res = model(data)
    raw = model.inference(data, do_postprocess=False)
    for i, r in enumerate(res):
        inst = r["instances"]
        assert tuple(inst.image_size) == tuple(int(v) for v in t["out_sizes"][i])
        assert np.array_equal(inst.pred_classes.cpu().numpy(), t[f"det_classes{i}"])
        np.testing.assert_allclose(inst.scores.cpu().numpy(), t[f"det_scores{i}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(inst.pred_boxes.tensor.cpu().numpy(), t[f"det_boxes{i}"], rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(raw[i].pred_boxes.tensor.cpu().numpy(), t[f"raw_boxes{i}"], rtol=1e-4, atol=1e-2)
        b = inst.pred_boxes.tensor
        assert float(b[:, 0::2].max()) <= t["out_sizes"][i][1] and float(b[:, 1::2].max()) <= t["out_sizes"][i][0]
    # inputs without "height" / "width": the network-input frame, as the reference defaults
    plain = model([{"image": d["image"]} for d in data])
    for i in range(2):
        assert tuple(plain[i]["instances"].image_size) == tuple(int(v) for v in t["sizes"][i])


# ------------------------------------------------------------------------------------------ index-side kernels (csrc/proposals.hip)
def _seed(tag, k):
    from oracle import detgen
    return detgen.fnv1a64(f"{tag}perm{k}")


@pytest.mark.parametrize("hw,n_gt", [((96, 128), [3, 0]), ((160, 224), [5, 1]), ((800, 1216), [4, 2])])
def test_rpn_label_anchors_kernel_against_the_oracle(ops, hw, n_gt):
    """sw_rpn_label_anchors: IoU matching with low-quality matches + the random-key label sampling on the device, against
    oracle.frcnn_oracle.rpn_label_and_sample (rpn.py:305-360, matcher.py:60-126, sampling.py:8-54) — labels of EVERY anchor equal,
    matched boxes equal; up to 242 991 anchors per image, an image without ground truth included"""
    h, w = hw
    grids = [((h + s - 1) // s, (w + s - 1) // s) for s in (4, 8, 16, 32, 64)]
    anchors = np.concatenate(FO.grid_anchors(grids), 0).astype(np.float32)
    A = anchors.shape[0]
    rng = np.random.RandomState(h + sum(n_gt))
    gts = []
    for n in n_gt:
        x1 = rng.rand(n) * (w - 40); y1 = rng.rand(n) * (h - 40)
        b = np.stack([x1, y1, x1 + 16 + rng.rand(n) * (w - x1 - 16), y1 + 16 + rng.rand(n) * (h - y1 - 16)], 1).astype(np.float32)
        if n:
            b[0] = anchors[A // 3]                              # one box that IS an anchor: IoU 1.0, ties with the low-quality rule
        gts.append(b)
    tag = f"lab{h}"
    perm = FO.Perm(tag)
    want_labels, want_matched = FO.rpn_label_and_sample(anchors, gts, perm)
    seeds = [_seed(tag, k) for k in range(2 * len(gts))]
    cat = torch.from_numpy(np.concatenate(gts, 0)).cuda() if sum(n_gt) else torch.zeros(0, 4, device="cuda")
    labels, matched = ops.rpn_label_anchors(torch.from_numpy(anchors).cuda(), cat, n_gt, seeds, 256, 64)
    torch.cuda.synchronize()
    for i in range(len(gts)):
        got = labels[i].cpu().numpy().astype(np.int64)
        assert np.array_equal(got, want_labels[i]), (i, int((got != want_labels[i]).sum()))
        assert (got == 1).sum() <= 64 and (got >= 0).sum() == min(256, A)
        pos = got == 1
        assert np.array_equal(matched[i].cpu().numpy()[pos], want_matched[i][pos])
        if n_gt[i]:
            assert np.array_equal(matched[i].cpu().numpy(), want_matched[i])


def test_rpn_select_pack_kernel_against_the_oracle(ops):
    """sw_rpn_select_pack: per (image, level) the pre_topk best logits with torch.sort(descending, stable) ties, decoded, clipped and
    packed for the per-level NMS — against oracle find_top_rpn_proposals' selection stage (proposal_utils.py:22-106).  Logits are
    quantised so that thousands of exact ties straddle the cut; one level is shorter than pre_topk; a NaN delta must clear `finite`."""
    rng = np.random.RandomState(5)
    N, pre = 2, 1000
    n_l = [30000, 7500, 1900, 480, 120]
    H, W = 200, 304
    anchors, logits, deltas = [], [], []
    for n in n_l:
        x1 = rng.rand(n) * (W - 20); y1 = rng.rand(n) * (H - 20)
        anchors.append(np.stack([x1 - 10, y1 - 10, x1 + 8 + rng.rand(n) * 120, y1 + 8 + rng.rand(n) * 90], 1).astype(np.float32))
        logits.append((np.round(rng.randn(N, n) * 8) / 8).astype(np.float32))            # many exact ties
        deltas.append((rng.randn(N, n, 4) * 0.4).astype(np.float32))
    logits[1][0, :3000] = 2.5                                                             # 3000 tied values right at the top of a level
    img_hw = torch.tensor([[H, W], [H - 40, W - 64]], dtype=torch.int32).cuda()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    sc, bx, fin = ops.rpn_select_pack([t(l) for l in logits], [t(d) for d in deltas], [t(a) for a in anchors], pre, (1.0, 1.0, 1.0, 1.0),
                                      float(np.log(1000.0 / 16)), img_hw)
    torch.cuda.synchronize()
    assert all(v != 0 for v in fin.tolist())
    L = len(n_l)
    sc, bx = sc.cpu().numpy(), bx.cpu().numpy()
    for img in range(N):
        h, w = (int(v) for v in img_hw[img].tolist())
        for l, n in enumerate(n_l):
            k = min(n, pre)
            idx = FO._sort_desc_stable(logits[l][img])[:k]
            props = FO.O.apply_deltas(torch.from_numpy(deltas[l][img]), torch.from_numpy(anchors[l]), (1.0, 1.0, 1.0, 1.0)).numpy()[idx]
            rows = slice(l * pre, l * pre + k)
            got_b = bx[img, rows, 4 * l:4 * l + 4]
            np.testing.assert_allclose(got_b, props, rtol=1e-5, atol=1e-3)
            assert np.array_equal(bx[img, rows, :4], got_b)                              # the box is repeated in every level's columns
            cb = props.copy(); cb[:, 0::2] = cb[:, 0::2].clip(0, w); cb[:, 1::2] = cb[:, 1::2].clip(0, h)
            keep = ((cb[:, 2] - cb[:, 0]) > 0) & ((cb[:, 3] - cb[:, 1]) > 0)
            want_s = np.where(keep, logits[l][img][idx], -np.inf).astype(np.float32)
            got_s = sc[img, rows, l]
            edge = np.abs((cb[:, 2] - cb[:, 0])) < 1e-3                                    # expf ulps may move a box across "empty"
            assert np.array_equal(got_s[~edge], want_s[~edge]), (img, l)
            other = np.delete(sc[img, rows], l, axis=1)
            assert np.isneginf(other).all() and np.isneginf(sc[img, l * pre + k:(l + 1) * pre]).all()
    deltas[2][1, 7, 2] = np.nan
    logits[2][1, 7] = 100.0                                                               # make sure the NaN row is selected
    _, _, fin = ops.rpn_select_pack([t(l) for l in logits], [t(d) for d in deltas], [t(a) for a in anchors], pre, (1.0, 1.0, 1.0, 1.0),
                                    float(np.log(1000.0 / 16)), img_hw)
    assert fin.tolist()[0] != 0 and fin.tolist()[1] == 0


@pytest.mark.parametrize("P,n_gt,append", [(300, [3, 0], True), (1000, [5, 2], True), (1000, [4, 1], False), (2000, [6, 3], True)])
def test_roi_label_sample_kernel_against_the_oracle(ops, P, n_gt, append):
    """sw_roi_label_sample against oracle.frcnn_oracle.roi_label_and_sample (roi_heads.py:324-375): sampled rows in the same ORDER
    (foreground by random key, then background), classes, boxes and matched ground truth equal; device-side proposal counts"""
    K = 20
    rng = np.random.RandomState(P + sum(n_gt))
    H, W = 400, 600
    props, gts, cnts = [], [], []
    for n in n_gt:
        x1 = rng.rand(n) * (W - 150); y1 = rng.rand(n) * (H - 150)
        gb = np.stack([x1, y1, x1 + 60 + rng.rand(n) * 80, y1 + 60 + rng.rand(n) * 80], 1).astype(np.float32)
        gc = rng.randint(0, K, n)
        cnt = P - rng.randint(0, 40)                              # the RPN kept fewer than the stride for this image
        x1 = rng.rand(cnt) * (W - 30); y1 = rng.rand(cnt) * (H - 30)
        pb = np.stack([x1, y1, x1 + 10 + rng.rand(cnt) * 200, y1 + 10 + rng.rand(cnt) * 200], 1).astype(np.float32)
        if n:
            pb[:40] = gb[rng.randint(0, n, 40)] + rng.randn(40, 4).astype(np.float32) * 4     # foreground candidates
        props.append(pb); gts.append((gb, gc)); cnts.append(cnt)
    tag = f"roi{P}{int(append)}"
    want = FO.roi_label_and_sample([{"boxes": p} for p in props], gts, K, FO.Perm(tag), append_gt=append)
    N = len(n_gt)
    buf = np.zeros((N, P, 4), np.float32)
    for i, p in enumerate(props):
        buf[i, :len(p)] = p
    cat_b = torch.from_numpy(np.concatenate([g[0] for g in gts], 0)).cuda()
    cat_c = torch.from_numpy(np.concatenate([g[1] for g in gts], 0).astype(np.int32)).cuda()
    cnt, idx, cls, both = ops.roi_label_sample(torch.tensor(cnts, dtype=torch.int32).cuda(), torch.from_numpy(buf).cuda(), cat_b, cat_c, n_gt,
                                               [_seed(tag, k) for k in range(2 * N)], append, 0.5, K, 512, 128)
    bx, gb = both[0], both[1]
    torch.cuda.synchronize()
    for i in range(N):
        n = int(cnt[i])
        assert n == len(want[i]["sampled_idx"]) == min(512, cnts[i] + (n_gt[i] if append else 0))
        assert np.array_equal(idx[i, :n].cpu().numpy(), want[i]["sampled_idx"])
        assert np.array_equal(cls[i, :n].cpu().numpy(), want[i]["gt_classes"])
        assert np.array_equal(bx[i, :n].cpu().numpy(), want[i]["boxes"])
        assert np.array_equal(gb[i, :n].cpu().numpy(), want[i]["gt_boxes"])
        assert (cls[i, :n] != K).sum() <= 128


def test_roi_label_sample_short_batch_more_foreground_than_the_cap_and_few_background(ops):
    """sampling.py:36-47: num_pos = min(n_pos, B * frac), num_neg = min(n_neg, B - num_pos).  An image with MORE foreground candidates than
    the cap and FEWER background ones than B - cap returns fewer rows than min(B, candidates) (200 fg + 100 bg + 2 gt -> 128 + 100 = 228,
    not 302): the kernel's count says so, rows beyond it are class -1 / empty boxes (never stale memory), and the module level
    (`label_and_sample_proposals`) hands on exactly the counted rows — ADVICE r4: the host used to assume min(B, candidates)."""
    from sos_wsod_amd.frcnn import StandardROIHeadsPseudoLab
    from sos_wsod_amd.structures import Boxes, Instances
    K, B = 20, 512
    rng = np.random.RandomState(77)
    H, W = 400, 600
    gb = np.array([[50, 60, 250, 260], [300, 100, 520, 330]], np.float32); gc = np.array([3, 11])
    fg = gb[rng.randint(0, 2, 200)] + rng.randn(200, 4).astype(np.float32) * 3          # IoU with its gt box well above 0.5
    x1 = rng.rand(100) * 40; y1 = 340 + rng.rand(100) * 20
    bg = np.stack([x1, y1, x1 + 12, y1 + 12], 1).astype(np.float32)                      # far from both gt boxes
    pb = np.concatenate([fg, bg], 0)[rng.permutation(300)].astype(np.float32)
    tag = "roishort"
    want = FO.roi_label_and_sample([{"boxes": pb}], [(gb, gc)], K, FO.Perm(tag), append_gt=True)
    assert len(want[0]["sampled_idx"]) == 228
    buf = torch.full((1, 320, 4), 7.0, device="cuda"); buf[0, :300] = torch.from_numpy(pb).cuda()
    cnt, idx, cls, both = ops.roi_label_sample(torch.tensor([300], dtype=torch.int32).cuda(), buf, torch.from_numpy(gb).cuda(),
                                               torch.from_numpy(gc.astype(np.int32)).cuda(), [2], [_seed(tag, 0), _seed(tag, 1)], True, 0.5, K, B, 128)
    torch.cuda.synchronize()
    n = int(cnt[0])
    assert n == 228
    assert np.array_equal(idx[0, :n].cpu().numpy(), want[0]["sampled_idx"])
    assert np.array_equal(cls[0, :n].cpu().numpy(), want[0]["gt_classes"])
    assert np.array_equal(both[0, 0, :n].cpu().numpy(), want[0]["boxes"])
    assert (cls[0, n:] == -1).all() and (idx[0, n:] == -1).all() and float(both[:, 0, n:].abs().max()) == 0.0
    # the module level: Instances of exactly the counted rows, the dense fast path (every image a full batch) not taken
    heads = StandardROIHeadsPseudoLab(K, _Keys(tag))
    prop = Instances((H, W)); prop.proposal_boxes = Boxes(torch.from_numpy(pb).cuda())
    tgt = Instances((H, W)); tgt.gt_boxes = Boxes(torch.from_numpy(gb).cuda()); tgt.gt_classes = torch.from_numpy(gc).cuda()
    out = heads.label_and_sample_proposals([prop], [tgt], True)
    assert len(out[0]) == 228 and out[0]._sw_dense[3] == 228
    assert np.array_equal(out[0].gt_classes.cpu().numpy(), want[0]["gt_classes"])
    assert np.array_equal(out[0].proposal_boxes.tensor.cpu().numpy(), want[0]["boxes"])


def test_index_side_kernels_take_more_than_eight_images(ops):
    """the per-image tables of csrc/proposals.hip travel by value in the kernel arguments (8 images / 40 segments); a
    supervised call of voc_ssod.yaml on one GPU is label_q + label_k = 16 images.  The entry points now walk image ranges: 11 images
    through sw_rpn_label_anchors, sw_rpn_select_pack, sw_roi_label_sample and sw_roi_assign_levels must equal the same images run in
    two calls of <= 8 (which the tests above pin against the oracle)."""
    N = 11
    rng = np.random.RandomState(3)
    h, w = 96, 128
    grids = [((h + s - 1) // s, (w + s - 1) // s) for s in (4, 8, 16, 32, 64)]
    anchors = torch.from_numpy(np.concatenate(FO.grid_anchors(grids), 0).astype(np.float32)).cuda()
    n_gt = [int(v) for v in rng.randint(0, 4, N)]
    gts = []
    for n in n_gt:
        x1 = rng.rand(n) * (w - 40); y1 = rng.rand(n) * (h - 40)
        gts.append(np.stack([x1, y1, x1 + 16 + rng.rand(n) * 20, y1 + 16 + rng.rand(n) * 20], 1).astype(np.float32))
    cat = torch.from_numpy(np.concatenate(gts, 0)).cuda()
    seeds = [_seed("many", k) for k in range(2 * N)]
    split = 6
    g0 = sum(n_gt[:split])

    def two(fn_all, fn_a, fn_b):
        a, b = fn_a(), fn_b()
        return fn_all(), [torch.cat([x, y], 0) for x, y in zip(a, b)]
    # anchors -> labels
    got, want = two(lambda: ops.rpn_label_anchors(anchors, cat, n_gt, seeds, 256, 64),
                    lambda: ops.rpn_label_anchors(anchors, cat[:g0], n_gt[:split], seeds[:2 * split], 256, 64),
                    lambda: ops.rpn_label_anchors(anchors, cat[g0:].contiguous(), n_gt[split:], seeds[2 * split:], 256, 64))
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    # top-k selection + decode
    n_l = [3000, 800, 200, 60, 20]
    lv_anchors, logits, deltas = [], [], []
    for n in n_l:
        x1 = rng.rand(n) * (w - 20); y1 = rng.rand(n) * (h - 20)
        lv_anchors.append(torch.from_numpy(np.stack([x1, y1, x1 + 8 + rng.rand(n) * 60, y1 + 8 + rng.rand(n) * 40], 1).astype(np.float32)).cuda())
        logits.append(torch.from_numpy((np.round(rng.randn(N, n) * 8) / 8).astype(np.float32)).cuda())
        deltas.append(torch.from_numpy((rng.randn(N, n, 4) * 0.3).astype(np.float32)).cuda())
    img_hw = torch.tensor([[h, w]] * N, dtype=torch.int32).cuda()
    sel = lambda a, b: ops.rpn_select_pack([l[a:b].contiguous() for l in logits], [d[a:b].contiguous() for d in deltas], lv_anchors, 500,
                                           (1.0, 1.0, 1.0, 1.0), float(np.log(1000.0 / 16)), img_hw[a:b].contiguous())
    got, want = two(lambda: sel(0, N), lambda: sel(0, split), lambda: sel(split, N))
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    # the single-tensor (anchor order) form of the same call
    lg1 = torch.cat(logits, 1).contiguous(); dl1 = torch.cat(deltas, 1).contiguous()
    one = ops.rpn_select_pack(lg1, dl1, lv_anchors, 500, (1.0, 1.0, 1.0, 1.0), float(np.log(1000.0 / 16)), img_hw)
    for x, y in zip(one, got):
        assert torch.equal(x, y)
    # ROI matching + sampling
    P = 600
    buf = torch.from_numpy(np.concatenate([rng.rand(N, P, 2) * 60, 70 + rng.rand(N, P, 2) * 50], 2).astype(np.float32)).cuda()
    pc = torch.from_numpy(rng.randint(500, P + 1, N).astype(np.int32)).cuda()
    cc = torch.from_numpy(rng.randint(0, 20, sum(n_gt)).astype(np.int32)).cuda()
    rs = lambda a, b: ops.roi_label_sample(pc[a:b].contiguous(), buf[a:b].contiguous(), cat[sum(n_gt[:a]):sum(n_gt[:b])].contiguous(),
                                           cc[sum(n_gt[:a]):sum(n_gt[:b])].contiguous(), n_gt[a:b], seeds[2 * a:2 * b], True, 0.5, 20, 512, 128)
    full, a_, b_ = rs(0, N), rs(0, split), rs(split, N)
    assert torch.equal(full[0], torch.cat([a_[0], b_[0]])) and torch.equal(full[1], torch.cat([a_[1], b_[1]]))
    assert torch.equal(full[2], torch.cat([a_[2], b_[2]])) and torch.equal(full[3], torch.cat([a_[3], b_[3]], 1))
    # FPN levels over 11 images / 11 * 1000 boxes (> the old 8192-row cap)
    boxes = torch.from_numpy(np.concatenate([rng.rand(N, 1000, 2) * 300, 310 + rng.rand(N, 1000, 2) * 400], 2).astype(np.float32)).cuda()
    rois, lv, sl, cnt = ops.roi_assign_levels(boxes, [1000] * N, [i * 4000 for i in range(N)])
    torch.cuda.synchronize()
    assert np.array_equal(rois[:, 0].cpu().numpy(), np.repeat(np.arange(N), 1000).astype(np.float32))
    assert torch.equal(rois[:, 1:], boxes.view(-1, 4))
    b = boxes.view(-1, 4)
    want_lv = torch.clamp(torch.floor(4 + torch.log2(torch.sqrt((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])) / 224 + 1e-8)), 2, 5).to(torch.int32) - 2
    assert int((lv != want_lv).sum()) <= 2                                # (log2 ulps at a level edge)
    for l in range(4):
        n = int(cnt[l])
        assert torch.equal(sl[l, :n], torch.nonzero(lv == l).flatten().to(torch.int32))
    with pytest.raises(ValueError, match="at most 64 images"):
        ops.roi_assign_levels(boxes, [1] * 65, [0] * 65)


# ------------------------------------------------------------------------------------------ the step on the real modules
def test_semisup_step_burn_in_and_semi_supervised_iteration_on_the_real_detector():
    """unbias/ubteacher/engine/trainer.py:436-549 with the real student / teacher: iteration 0 = burn-in (supervised branch on the
    strong + weak labelled views), iteration 1 = teacher copy (keep rate 0), teacher's weak pass, 0.7 thresholding, student on the
    labelled views and on the strongly augmented unlabelled views with the pseudo boxes, loss weights (pseudo box losses x 0, other
    pseudo losses x UNSUP_LOSS_WEIGHT 2).  Every recorded loss against the oracle evaluated at the student's / teacher's current
    weights with the same sampling keys: 1e-4 relative."""
    from sos_wsod_amd.semisup import SemiSupStep
    from sos_wsod_amd.structures import Boxes, Instances
    K = 20
    P = FO.make_params(K, tag="s3s", head_scale=14.0)
    student, teacher = _model(K, P, "s3s"), _model(K, P, "s3s")
    student.train(); teacher.train()
    keys = student.sampler
    student.proposal_generator.sampler = student.roi_heads.sampler = keys
    opt = torch.optim.SGD([p for p in student.parameters() if p.requires_grad], lr=1e-5, momentum=0.9)      # (the fixture is touchy: at 1e-4 one step pushes every teacher score below 0.7)
    step = SemiSupStep(student, teacher, opt, burn_up_step=1, ema_keep_rate=0.9996, bbox_threshold=0.7, unsup_loss_weight=2.0)
    sizes = [(96, 128), (128, 112)]

    def batch(tag, n_gt):
        out, gts, imgs = [], [], []
        for i, (h, w) in enumerate(sizes):
            img = FO.make_image(h, w, f"{tag}{i}")
            d = {"image": torch.from_numpy(img).cuda(), "height": h, "width": w}
            if n_gt:
                b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
                inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
                d["instances"] = inst
                gts.append((b, c))
            out.append(d); imgs.append(img)
        return out, gts, imgs
    lq, gq, iq = batch("s3s_lq", 2); lk, gk, ik = batch("s3s_lk", 3)
    uq, _, iuq = batch("s3s_uq", 0); uk, _, iuk = batch("s3s_uk", 0)

    def weights(m):
        return {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    # ---- iteration 0: burn-in
    W0 = weights(student)
    record, loss_dict = step.run_step((lq, lk, uq, uk))
    ref, _, _ = FO.supervised_forward(W0, iq + ik, gq + gk, K, FO.Perm("s3s"))
    for k in ref:
        # RPN losses 1e-4; ROI-head losses 2e-3 for the reason given at iteration 1 below (position-based sampling over proposals
        # whose logits tie to ~1e-7: since res5's K = 2048 convolutions run K-split the features differ from the oracle's by an ulp)
        tol = 1e-4 if "rpn" in k else 2e-3
        assert abs(float(record[k]) - ref[k]) <= tol * abs(ref[k]), (k, float(record[k]), ref[k])
    print("burn-in iteration:", {k: (round(float(record[k]), 6), round(float(ref[k]), 6)) for k in ref})
    assert any(not np.array_equal(W0[k], v) for k, v in weights(student).items())                    # the optimizer moved the student
    # ---- iteration 1: teacher <- student, pseudo labels, student on labelled + pseudo-labelled data
    W1 = weights(student)
    perm = FO.Perm("s3s"); perm.k = keys.perm.k                                                       # the key stream continues
    record, loss_dict = step.run_step((lq, lk, uq, uk))
    for k, v in weights(teacher).items():
        assert np.array_equal(v, W1[k]), k                                                            # keep rate 0: an exact copy
    _, dets = FO.weak_forward(W1, iuk, K)
    pseudo = [(d["pred_boxes"][d["scores"] > 0.7], d["pred_classes"][d["scores"] > 0.7]) for d in dets]
    assert sum(len(p[0]) for p in pseudo) > 0
    forced = []
    for d, p in zip(uq, pseudo):
        inst = d["instances"]                                                                         # add_label put the pseudo boxes on the strong views
        assert len(inst) == len(p[0]) and np.array_equal(inst.gt_classes.cpu().numpy(), p[1])
        np.testing.assert_allclose(inst.gt_boxes.tensor.cpu().numpy(), p[0], rtol=1e-4, atol=1e-2)
        # the student's pass on the unlabelled views is then checked from the HIP teacher's own boxes: anchors whose IoU with a
        # pseudo box sits within 1e-5 of the 0.3 / 0.7 thresholds would otherwise flip their label (measured: loss_rpn_cls_pseudo 3e-3)
        forced.append((inst.gt_boxes.tensor.cpu().numpy(), p[1]))
    pseudo = forced
    ref_l, _, _ = FO.supervised_forward(W1, iq + ik, gq + gk, K, perm)
    ref_u, _, _ = FO.supervised_forward(W1, iuq, pseudo, K, perm)
    want = dict(ref_l); want.update({k + "_pseudo": v for k, v in ref_u.items()})
    assert set(k for k in record if k.startswith("loss")) == set(want)
    for k, v in want.items():
        # RPN losses 1e-4.  The ROI-head losses of this iteration get 2e-3: the label sampling picks candidates by POSITION in
        # the proposal list (sampling.py:49-53 indexes the list with randperm), and two RPN logits equal to ~1e-7 order
        # differently on the two sides (measured: one swapped pair -> one other background row among 512, loss_cls 3e-4);
        # with identical sampled sets the bar is 1e-4 (the reference-generated fixture).
        tol = 1e-4 if "rpn" in k else 2e-3
        assert abs(float(record[k]) - v) <= tol * abs(v) + 1e-7, (k, float(record[k]), v)
    assert float(loss_dict["loss_box_reg_pseudo"]) == 0.0 and float(loss_dict["loss_rpn_loc_pseudo"]) == 0.0
    assert abs(float(loss_dict["loss_cls_pseudo"]) - 2.0 * want["loss_cls_pseudo"]) <= 4e-3 * abs(want["loss_cls_pseudo"])
    print("semi-supervised iteration:", {k: round(float(v), 5) for k, v in record.items() if k.startswith("loss")})


def test_semisup_step_against_the_reference_run_of_its_own_trainer_methods(golden_dir):
    """tests/golden/stage3_step.npz was written by RUNNING the reference's UBTeacherTrainer.run_step_full_semisup (with its
    threshold_bbox / process_pseudo_label / add_label / _update_teacher_model, unbias/ubteacher/engine/trainer.py:362-604) on the
    reference's own student and teacher for three iterations: burn-in, the copy step + a semi-supervised step, one EMA update + a
    semi-supervised step.  `semisup.SemiSupStep` on the HIP detectors, same closed-form weights / data / sampling keys / SGD, must log
    the same record dict, attach the same pseudo labels, differentiate the same weighted sum and leave the same teacher.
    Round 6: the ROIAlign backward accumulates in 64-bit fixed point (csrc/detector.hip roi_align_bwd_fx_kernel), so the iteration is
    bitwise reproducible — the float-atomic form left the student different in the last bits from run to run, the pseudo boxes jittered
    by ~1e-4 px and position-keyed anchor sampling flipped (loss_rpn_loc_pseudo 0.678 .. 0.7065 over runs: a 6e-2 bar).  Measured now
    (tools/diag/s3_ref_dev.py, profiles/r06_stage3_reproducibility.txt; identical over 3 runs), relative to the reference's values:
    iterations 0-1 every loss <= 2.1e-4 (loss_cls; the pseudo losses <= 2.5e-6); iteration 2 — the third step of a free-running
    trajectory — <= 2.4e-3 except loss_rpn_loc_pseudo 1.3e-2 (0.6995 vs 0.6904: one anchor whose IoU with a pseudo box sits within
    1e-5 of a threshold, and with it the positions of the later candidates in the sampler's list).  Bars: iterations 0-1 RPN 1e-4,
    ROI heads 5e-4, pseudo 1e-4, total 5e-4; iteration 2 5e-3, loss_rpn_loc_pseudo 2e-2, total 2e-3."""
    from sos_wsod_amd.semisup import SemiSupStep
    from sos_wsod_amd.structures import Boxes, Instances
    G = np.load(os.path.join(golden_dir, "stage3_step.npz"))
    K = int(G["K"])
    sizes = [tuple(int(v) for v in s_) for s_ in G["sizes"]]
    P = FO.make_params(K, tag="s3s", head_scale=float(G["head_scale"]))
    student, teacher = _model(K, P, "s3s"), _model(K, P, "s3s")
    student.train(); teacher.train()
    student.proposal_generator.sampler = student.roi_heads.sampler = student.sampler
    opt = torch.optim.SGD([p for p in student.parameters() if p.requires_grad], lr=float(G["lr"]), momentum=float(G["momentum"]))
    step = SemiSupStep(student, teacher, opt, burn_up_step=int(G["cfg/BURN_UP_STEP"]), teacher_update_iter=int(G["cfg/TEACHER_UPDATE_ITER"]),
                       ema_keep_rate=float(G["cfg/EMA_KEEP_RATE"]), bbox_threshold=float(G["cfg/BBOX_THRESHOLD"]),
                       unsup_loss_weight=float(G["cfg/UNSUP_LOSS_WEIGHT"]), burn_up_with_strong_aug=bool(G["cfg/BURN_UP_WITH_STRONG_AUG"]))

    def batch(tag, n_gt):
        out = []
        for i, (h, w) in enumerate(sizes):
            d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).cuda(), "height": h, "width": w}
            if n_gt:
                b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
                inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
                d["instances"] = inst
            out.append(d)
        return out
    named = dict(student.named_parameters())
    stride = int(G["stride"])
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    trajectory = []
    for it in range(3):
        data = (batch("s3s_lq", 2), batch("s3s_lk", 3), batch("s3s_uq", 0), batch("s3s_uk", 0))
        s_before, t_before = sd(student), sd(teacher)
        record, loss_dict = step.run_step(data)
        torch.cuda.synchronize()
        want = {k[len(f"it{it}/record/"):]: float(G[k]) for k in G.files if k.startswith(f"it{it}/record/") and "/loss" in k}
        assert set(k for k in record if k.startswith("loss")) == set(want), (it, sorted(record), sorted(want))
        # iteration 2 is the THIRD step of a free-running trajectory on a fixture with peaky heads (loss_cls falls 5.0 -> 4.3 -> 1.8 in two
        # SGD steps): the two implementations' students differ by then (measured: loss_cls 0.6 %), so its bars are 2e-2; the step
        # logic under test at iteration 2 is the EMA, checked on the teacher's tensors below
        loose = it == 2
        for k, v in want.items():
            if loose:
                # the anchor sampler picks candidates by POSITION in the candidate list (sampling.py:49-53): one anchor whose IoU with a
                # pseudo box crosses 0.3 / 0.7 moves every later candidate's key and with it the sampled set
                tol = 2e-2 if k == "loss_rpn_loc_pseudo" else 5e-3
            else:
                tol = 1e-4 if (k.endswith("_pseudo") or "rpn" in k) else 5e-4
            assert abs(float(record[k]) - v) <= tol * abs(v) + 1e-7, (it, k, float(record[k]), v)
        total = float(sum(float(v) for v in loss_dict.values()))
        assert abs(total - float(G[f"it{it}/total_loss"])) <= (2e-3 if loose else 5e-4) * abs(float(G[f"it{it}/total_loss"])), (it, total)
        trajectory.append({k: float(v) for k, v in record.items() if k.startswith("loss")})
        if it == 0:
            assert all(torch.equal(t_before[k], v) for k, v in sd(teacher).items())                  # burn-in leaves the teacher alone
            for n in list(G["watch"]) + list(G["watch_full"]):                                          # gradient of the (unit-weighted) sum
                g_ = named[str(n)].grad.detach().cpu().numpy()
                wg = G[f"it0/grad/{n}"]
                got = g_.ravel()[::stride] if str(n) in set(G["watch"]) else g_
                assert np.abs(got - wg).max() <= 2e-3 * np.abs(wg).max() + 1e-12, (n, float(np.abs(got - wg).max() / np.abs(wg).max()))
            continue
        assert float(loss_dict["loss_box_reg_pseudo"]) == 0.0 and float(loss_dict["loss_rpn_loc_pseudo"]) == 0.0
        # the pseudo labels add_label attached to the strong views
        for i, d in enumerate(data[2]):
            inst = d["instances"]
            assert np.array_equal(inst.gt_classes.cpu().numpy(), G[f"it{it}/pseudo_classes{i}"]), (it, i)
            np.testing.assert_allclose(inst.gt_boxes.tensor.cpu().numpy(), G[f"it{it}/pseudo_boxes{i}"], rtol=1e-4, atol=2e-2)
            np.testing.assert_allclose(inst.scores.cpu().numpy(), G[f"it{it}/pseudo_scores{i}"], rtol=2e-3)
        # the teacher after its update: the rule on every tensor, and the reference's own values on the sampled ones
        keep = 0.0 if it == 1 else float(G["cfg/EMA_KEEP_RATE"])
        t_after = sd(teacher)
        for k, v in t_after.items():
            if keep == 0.0:
                assert torch.equal(v, s_before[k]), k
            elif v.dtype == torch.float32:
                rule = s_before[k] * (1 - keep) + t_before[k] * keep
                assert float((v - rule).abs().max()) <= 1e-6 * float(rule.abs().max()) + 1e-12, k
        for n in G["watch"]:
            ref_t = G[f"it{it}/teacher/{n}"]
            got = t_after[str(n)].cpu().numpy().ravel()[::stride]
            assert np.abs(got - ref_t).max() <= 1e-5 * np.abs(ref_t).max(), n
        # the weighted gradient (x UNSUP_LOSS_WEIGHT on the pseudo classification / objectness terms, x 0 on the pseudo box terms): the
        # sampled ROI / anchor sets may differ by a row (above), so a bound that a wrong weight (x1 or x4 instead of x2: > 25 %) cannot meet
        for n in ("roi_heads.box_predictor.cls_score.bias", "proposal_generator.rpn_head.objectness_logits.bias",
                  "roi_heads.box_predictor.bbox_pred.bias", "proposal_generator.rpn_head.anchor_deltas.bias"):
            g_ = named[n].grad.detach().cpu().numpy(); wg = G[f"it{it}/grad/{n}"]
            bar = 2e-1 if "rpn" in n else (1e-1 if loose else 5e-2)          # (RPN: the sampled anchor set may differ, above)
            assert np.abs(g_ - wg).max() <= bar * np.abs(wg).max(), (it, n, float(np.abs(g_ - wg).max() / np.abs(wg).max()))
        print(f"iteration {it}:", {k: (round(float(record[k]), 5), round(v, 5)) for k, v in want.items()})
    # ---- the same three iterations on a fresh student / teacher pair: every logged loss and every student weight bit for bit (round 6)
    final = sd(student)
    student2, teacher2 = _model(K, P, "s3s"), _model(K, P, "s3s")
    student2.train(); teacher2.train()
    student2.proposal_generator.sampler = student2.roi_heads.sampler = student2.sampler
    opt2 = torch.optim.SGD([p for p in student2.parameters() if p.requires_grad], lr=float(G["lr"]), momentum=float(G["momentum"]))
    step2 = SemiSupStep(student2, teacher2, opt2, burn_up_step=int(G["cfg/BURN_UP_STEP"]), teacher_update_iter=int(G["cfg/TEACHER_UPDATE_ITER"]),
                        ema_keep_rate=float(G["cfg/EMA_KEEP_RATE"]), bbox_threshold=float(G["cfg/BBOX_THRESHOLD"]),
                        unsup_loss_weight=float(G["cfg/UNSUP_LOSS_WEIGHT"]), burn_up_with_strong_aug=bool(G["cfg/BURN_UP_WITH_STRONG_AUG"]))
    for it in range(3):
        record, _ = step2.run_step((batch("s3s_lq", 2), batch("s3s_lk", 3), batch("s3s_uq", 0), batch("s3s_uk", 0)))
        again = {k: float(v) for k, v in record.items() if k.startswith("loss")}
        assert again == trajectory[it], (it, {k: (again[k], trajectory[it][k]) for k in again if again[k] != trajectory[it][k]})
    torch.cuda.synchronize()
    assert all(torch.equal(v, final[k]) for k, v in sd(student2).items())


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_weight_gradients_summed_in_the_kernels_equal_autograd_sums(mode):
    """ops.grad_scope: in a semi-supervised iteration every student weight is used by two forward passes (and the RPN head's 3x3
    convolution by five levels in each); inside the scope the second and later weight-gradient kernels of a parameter add to the
    first one's buffer (split-K fold / GEMM epilogue with the buffer as its own residual) and autograd's accumulator receives ONE
    gradient.  Same f32 additions as autograd's own sums: every gradient within 1e-5 (fp32; see the note at the comparison)."""
    from sos_wsod_amd.semisup import SemiSupStep
    from sos_wsod_amd.structures import Boxes, Instances
    K = 20
    P = FO.make_params(K, tag="s3g", head_scale=14.0)
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16
    sizes = [(96, 128), (128, 112)]

    def batch(tag, n_gt):
        out = []
        for i, (h, w) in enumerate(sizes):
            d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).cuda(), "height": h, "width": w}
            if n_gt:
                b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
                inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
                d["instances"] = inst
            out.append(d)
        return out
    grads = {}
    import sos_wsod_amd.ops as ops
    calls = {True: [0, 0], False: [0, 0]}           # grouped launches (3x3, 1x1) per setting: the fused step must actually take them
    real3, real1 = ops.conv3x3_wgrad_grouped, ops.gemm_kk_grouped
    for fuse in (True, False):
        def spy3(*a, _f=fuse, **k):
            calls[_f][0] += 1
            return real3(*a, **k)

        def spy1(*a, _f=fuse, **k):
            calls[_f][1] += 1
            return real1(*a, **k)
        ops.conv3x3_wgrad_grouped, ops.gemm_kk_grouped = spy3, spy1
        student, teacher = _model(K, P, "s3g", dtype=dtype), _model(K, P, "s3g", dtype=dtype)
        student.train(); teacher.train()
        student.proposal_generator.sampler = student.roi_heads.sampler = student.sampler
        opt = torch.optim.SGD([p for p in student.parameters() if p.requires_grad], lr=0.0)
        step = SemiSupStep(student, teacher, opt, burn_up_step=0, bbox_threshold=0.0, unsup_loss_weight=2.0, fuse_grad_sums=fuse)
        record, _ = step.run_step((batch("s3g_lq", 2), batch("s3g_lk", 3), batch("s3g_uq", 0), batch("s3g_uk", 0)))
        torch.cuda.synchronize()
        assert "loss_cls_pseudo" in record
        grads[fuse] = {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}
    ops.conv3x3_wgrad_grouped, ops.gemm_kk_grouped = real3, real1
    # the uses were counted (ops.CountedFunction) and the last use of a weight ran all pairs as ONE grouped launch: >= one launch per
    # 3x3 weight with several uses (13 bottleneck conv2 + the RPN head + 4 FPN outputs) and per bottleneck block's 1x1 weights
    assert calls[True][0] >= 10 and calls[True][1] >= 10, calls
    assert set(grads[True]) == set(grads[False]) and len(grads[True]) > 60
    # the sums themselves are the same f32 additions in the same order; the comparison is not bit-exact because the ROIAlign backward
    # scatters with f32 atomics (run-to-run noise of ~1e-7 in everything below the pooler); 1e-5 of the tensor's largest element
    diff = [(n, float((g - grads[False][n]).abs().max() / (grads[False][n].abs().max() + 1e-30))) for n, g in grads[True].items()]
    bad = [t for t in diff if t[1] > (1e-5 if mode == "fp32" else 2e-2)]
    assert not bad, (len(bad), sorted(bad, key=lambda t: -t[1])[:5])
    print(f"grad_scope vs autograd sums ({mode}): worst relative difference {max(t[1] for t in diff):.1e} over {len(diff)} tensors")
    assert all(float(g.abs().max()) > 0 for g in grads[True].values())


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_student_passes_in_lockstep_equal_two_calls(mode):
    """SemiSupStep(lockstep=True): the student's labelled and pseudo-labelled batches (different image sizes) go through ONE model
    call whose backbone runs both in lockstep — rows of all maps back to back, every 1x1 convolution one GEMM, every 3x3 one
    multi-problem launch (ResNet.forward_lockstep) — against the reference's two calls (trainer.py:527-538).  Rows are independent
    in every layer: fp32 — the 8 losses agree to 1e-5, every gradient to 1e-4 of the tensor's largest element (ROIAlign's backward
    scatters with atomics); bf16 — a map of few tiles runs another form of the 3x3 kernel inside the shared launch (last-bit
    differences), so the RPN losses agree to 2e-3, the ROI-head losses within the sampling-sensitivity bounds of the bf16 fixture test (5e-2
    classification, 2e-1 box regression: a handful of foreground ROIs); the gradients are printed with a sanity bound."""
    from sos_wsod_amd.semisup import SemiSupStep
    from sos_wsod_amd.structures import Boxes, Instances
    K = 20
    P = FO.make_params(K, tag="s3l", head_scale=14.0)
    dtype = torch.float32 if mode == "fp32" else torch.bfloat16

    def batch(tag, sizes, n_gt):
        out = []
        for i, (h, w) in enumerate(sizes):
            d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).cuda(), "height": h, "width": w}
            if n_gt:
                b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
                inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
                d["instances"] = inst
            out.append(d)
        return out
    res = {}
    for lock in (True, False):
        student, teacher = _model(K, P, "s3l", dtype=dtype), _model(K, P, "s3l", dtype=dtype)
        student.train(); teacher.train()
        student.proposal_generator.sampler = student.roi_heads.sampler = student.sampler
        opt = torch.optim.SGD([p for p in student.parameters() if p.requires_grad], lr=0.0)
        step = SemiSupStep(student, teacher, opt, burn_up_step=0, bbox_threshold=0.0, unsup_loss_weight=2.0, lockstep=lock)
        assert step.lockstep == lock
        # labelled views 96x128 (+ one 128x112), unlabelled views of ANOTHER size: the two batches pad to different grids
        record, _ = step.run_step((batch("s3l_lq", [(96, 128)], 2), batch("s3l_lk", [(128, 112)], 3),
                                   batch("s3l_uq", [(160, 96)], 0), batch("s3l_uk", [(160, 96)], 0)))
        torch.cuda.synchronize()
        res[lock] = ({k: float(v) for k, v in record.items() if k.startswith("loss")},
                     {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None})
    tl, tg = (1e-5, 1e-4) if mode == "fp32" else (2e-3, 5e-1)
    assert set(res[True][0]) == set(res[False][0]) and len(res[True][0]) == 8
    for k, v in res[False][0].items():
        # bf16: the ROI-head losses depend on WHICH proposals survive top-k / NMS / sampling (last-bit feature differences reorder
        # near-tied proposals): the bound of test_supervised_branch_bf16_mode_stays_close_to_the_fp32_fixture
        tol = tl if (mode == "fp32" or "rpn" in k) else (2e-1 if "box_reg" in k else 5e-2)
        assert abs(res[True][0][k] - v) <= tol * abs(v) + 1e-7, (k, res[True][0][k], v)
    assert set(res[True][1]) == set(res[False][1])
    worst = max(((n, float((g - res[False][1][n]).abs().max() / (res[False][1][n].abs().max() + 1e-30))) for n, g in res[True][1].items()),
                key=lambda t: t[1])
    assert worst[1] <= tg, worst
    print(f"lockstep vs two calls ({mode}): losses {res[True][0]}; worst gradient difference {worst[1]:.1e} ({worst[0]})")


def test_teacher_on_a_side_stream_and_its_backbone_as_a_graph_give_the_same_step():
    """SemiSupStep(overlap_teacher=True): the teacher's weak pass runs on a second stream and the student's call asks for the pseudo
    labels only when the pseudo-labelled batch's heads are next (frcnn forward(second_targets=...)).  Same kernels, same order per
    stream, the labels handed over behind a stream wait.  Three iterations at learning rate 0 with a teacher that starts AWAY from the
    student (every iteration's EMA update, keep rate 0.5, moves it: the update has to wait for the previous iteration's teacher
    readers, the teacher's pass for the update) — nothing in the losses depends on the gradients (whose ROIAlign backward scatters
    with float atomics), so the 8 losses of every iteration and the teacher's parameters must be EQUAL to the sequential form's."""
    from sos_wsod_amd.semisup import SemiSupStep
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.structures import Boxes, Instances
    K = 20
    P, PT = FO.make_params(K, tag="s3l", head_scale=14.0), FO.make_params(K, tag="s3o_teacher", head_scale=14.0)

    def batch(tag, sizes, n_gt):
        out = []
        for i, (h, w) in enumerate(sizes):
            d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).cuda(), "height": h, "width": w}
            if n_gt:
                b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
                inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
                d["instances"] = inst
            out.append(d)
        return out
    res = {}
    # (overlap, backbone graph): the teacher's no-grad backbone pass is a hipGraph replay from its third call with one input shape on
    # (frcnn._features) — iteration 0 launches, iteration 1 captures and replays, iteration 2 replays, with the EMA moving the weights
    # under the graph every time; the baseline launches everything
    for overlap, graph in ((True, True), (False, True), (False, False)):
        os.environ["SW_S3_BACKBONE_GRAPH"] = "1" if graph else "0"
        student, teacher = _model(K, P, "s3l"), _model(K, PT, "s3l")
        student.train(); teacher.train()
        student.proposal_generator.sampler = student.roi_heads.sampler = student.sampler
        opt = HipSGD([p for p in student.parameters() if p.requires_grad], 0.0, momentum=0.9)
        step = SemiSupStep(student, teacher, opt, burn_up_step=0, bbox_threshold=0.0, unsup_loss_weight=2.0, ema_keep_rate=0.5,
                           overlap_teacher=overlap, speculate=False)
        assert step.overlap_teacher == overlap and not step.speculate
        recs = []
        for it in range(3):
            record, _ = step.run_step((batch(f"s3o{it}_lq", [(96, 128)], 2), batch(f"s3o{it}_lk", [(128, 112)], 3),
                                       batch(f"s3o{it}_uq", [(160, 96)], 0), batch(f"s3o{it}_uk", [(160, 96)], 0)))
            recs.append({k: float(v) for k, v in record.items() if k.startswith("loss")})
        torch.cuda.synchronize()
        assert (len([v for v in teacher.__dict__.get("_bb_graphs", {}).values() if v != 1]) == 1) == graph
        res[(overlap, graph)] = (recs, {n: p.detach().clone() for n, p in teacher.named_parameters()})
    os.environ.pop("SW_S3_BACKBONE_GRAPH", None)
    base = res[(False, False)]
    for variant in ((True, True), (False, True)):
        for it in range(3):
            assert len(res[variant][0][it]) == 8
            for k, v in base[0][it].items():
                assert res[variant][0][it][k] == v, (variant, it, k, res[variant][0][it][k], v)
        assert all(torch.equal(a, base[1][n]) for n, a in res[variant][1].items())
    assert base[0][0] != base[0][1] != base[0][2]                                     # the teacher did move between the iterations


def test_backbone_graph_goes_with_the_weight_stage_it_was_captured_on():
    """a captured no-grad backbone graph (frcnn._features) holds the addresses of the WeightStage's staged buffers.  Re-homing a
    parameter (here: new storage with new values, as .to() / load_state_dict(assign=True) / a flat-master re-pack do) rebuilds the stage;
    the graphs captured on the old one must go, or the teacher would replay stale weights with no error.  After the rebuild the replayed
    features equal the plain launches' bit for bit — and differ from the old weights' features."""
    K = 20
    P = FO.make_params(K, tag="s3l", head_scale=14.0)
    os.environ["SW_S3_BACKBONE_GRAPH"] = "1"
    try:
        m = _model(K, P, "s3l"); m.train()
        h, w = 160, 96
        b = [{"image": torch.from_numpy(FO.make_image(h, w, "bbg0")).cuda(), "height": h, "width": w}]

        def leaves(f):
            return list(f.values()) if isinstance(f, dict) else list(f)
        with torch.no_grad():
            m.refresh_staged_weights()
            x4, _ = m.preprocess_image(b)
            assert all(torch.equal(a, c) for a, c in zip(leaves(m._features(x4)), leaves(m.backbone(x4))))   # allow_graph off: plain launches
            for _ in range(3):
                old = [t.clone() for t in leaves(m._features(x4, allow_graph=True))]
            assert len([v for v in m.__dict__["_bb_graphs"].values() if v != 1]) == 1
            name, prm = next((n, q) for n, q in m.named_parameters() if "res3" in n and n.endswith("conv2.weight"))
            prm.data = (prm.data * 1.5).clone()                                     # re-homed AND changed
            m.refresh_staged_weights()
            assert not [v for v in m.__dict__.get("_bb_graphs", {}).values() if v != 1], "graphs of the old stage kept"
            for _ in range(3):                                                        # launches, capture + replay, replay
                new = [t.clone() for t in leaves(m._features(x4, allow_graph=True))]
            assert len([v for v in m.__dict__["_bb_graphs"].values() if v != 1]) == 1
            ref = leaves(m.backbone(x4))
            torch.cuda.synchronize()
            assert all(torch.equal(a, c) for a, c in zip(new, ref)), name
            assert any(not torch.equal(a, c) for a, c in zip(new, old))
    finally:
        os.environ.pop("SW_S3_BACKBONE_GRAPH", None)


@pytest.mark.parametrize("caps_reached", [True, False])
def test_speculative_iteration_equals_the_reading_one(caps_reached):
    """SemiSupStep(speculate=True): the count read-backs of the training path (proposals left by the RPN's NMS + finite flags, rows the
    ROI sampler filled, the teacher's detection count) are assumed to sit at their caps and confirmed by ONE read before the optimizer
    step (frcnn.Speculation).  caps_reached: caps set so that these small images reach them (NMS cap 100, ROI batch 16) — no miss, the
    padded rows (zero boxes / class -1 / score 0) never show; not reached (the default caps 1000 / 512 on a 160x96 image): every
    speculative attempt is discarded, the samplers are rewound and the iteration repeated by the reading code, then 20 iterations
    without speculation.  Either way, against a run that never speculates (fp32, teacher on the main stream in both): three
    iterations at learning rate 0 with a teacher that starts away from the student and moves every iteration (EMA 0.5) give EQUAL
    losses in every iteration and an equal teacher; one iteration at learning rate 1e-4 leaves the student's parameters within 1e-5
    (the ROIAlign backward's float atomics are the run-to-run noise of the gradients; a free-running second iteration would amplify
    it through the pseudo labels)."""
    from sos_wsod_amd.semisup import SemiSupStep
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.structures import Boxes, Instances
    K = 20
    P, PT = FO.make_params(K, tag="s3l", head_scale=14.0), FO.make_params(K, tag="s3o_teacher", head_scale=14.0)

    def batch(tag, sizes, n_gt):
        out = []
        for i, (h, w) in enumerate(sizes):
            d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).cuda(), "height": h, "width": w}
            if n_gt:
                b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
                inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
                d["instances"] = inst
            out.append(d)
        return out
    res = {}
    for spec in (True, False):
        for lr, n_it in ((0.0, 3), (1e-4, 1)):
            student, teacher = _model(K, P, "s3l"), _model(K, PT, "s3l")
            student.train(); teacher.train()
            student.proposal_generator.sampler = student.roi_heads.sampler = student.sampler
            if caps_reached:
                for m in (student, teacher):
                    m.proposal_generator.post_nms_topk = (100, 100)
                    m.roi_heads.batch_size_per_image = 16
            opt = HipSGD([p for p in student.parameters() if p.requires_grad], lr, momentum=0.9)
            step = SemiSupStep(student, teacher, opt, burn_up_step=0, bbox_threshold=0.0, unsup_loss_weight=2.0, ema_keep_rate=0.5,
                               overlap_teacher=False, speculate=spec)
            assert step.speculate == spec
            recs = []
            for it in range(n_it):
                record, _ = step.run_step((batch(f"s3o{it}_lq", [(96, 128)], 2), batch(f"s3o{it}_lk", [(128, 112)], 3),
                                           batch(f"s3o{it}_uq", [(160, 96)], 0), batch(f"s3o{it}_uk", [(160, 96)], 0)))
                recs.append({k: float(v) for k, v in record.items() if k.startswith("loss")})
            torch.cuda.synchronize()
            if spec:
                assert step.spec_misses == (0 if caps_reached else 1), step.spec_misses  # (after a miss: 20 iterations of the reading code)
            res[(spec, lr)] = (recs, {n: p.detach().clone() for n, p in student.named_parameters()},
                               {n: p.detach().clone() for n, p in teacher.named_parameters()})
    a, b = res[(True, 0.0)], res[(False, 0.0)]
    for it in range(3):
        assert len(a[0][it]) == 8
        for k, v in b[0][it].items():
            assert a[0][it][k] == v, (it, k, a[0][it][k], v)
    assert a[0][0] != a[0][1] != a[0][2] and all(torch.equal(t, b[2][n]) for n, t in a[2].items())
    a, b = res[(True, 1e-4)], res[(False, 1e-4)]
    for k, v in b[0][0].items():
        assert a[0][0][k] == v, (k, a[0][0][k], v)
    worst = max(float((t - b[1][n]).abs().max() / (b[1][n].abs().max() + 1e-30)) for n, t in a[1].items())
    assert worst <= 1e-5, worst
    assert any(not torch.equal(t, torch.from_numpy(P[n]).cuda()) for n, t in a[1].items() if n in P)             # the step did move the student
    print(f"speculative vs reading iteration (caps reached: {caps_reached}): worst parameter difference after a step {worst:.1e}")


def test_detector_trains_the_same_under_hipsgd_and_torch_sgd(golden_dir):
    """Stage 3's solver is the same SGD (momentum 0.9, weight decay) as Stage 1's: the fused HipSGD and torch.optim.SGD drive the
    detector to the same parameters and losses over 3 supervised steps.  HipSGD writes the parameters behind torch's version
    counters (ops.PARAM_EPOCH), torch's SGD bumps them: both must invalidate the detector's staged weight copies (a stale copy would
    freeze the losses)."""
    from sos_wsod_amd.solver import HipSGD
    t = np.load(os.path.join(golden_dir, "stage3_a.npz"))
    K = int(t["K"])
    P = FO.make_params(K, tag="s3a", head_scale=float(t["head_scale"]))
    runs = {}
    for name in ("hip", "torch"):
        model = _model(K, P, "s3a")
        model.train()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = (HipSGD(params, 1e-4, momentum=0.9, weight_decay=1e-4) if name == "hip"
               else torch.optim.SGD(params, lr=1e-4, momentum=0.9, weight_decay=1e-4))
        hist = []
        for it in range(3):
            model.proposal_generator.sampler = model.roi_heads.sampler = _Keys("s3a")       # the same sampling keys every step, both runs
            data, _ = _inputs("s3a", t, K)
            losses, _, _, _ = model(data, branch="supervised")
            opt.zero_grad()
            sum(losses.values()).backward()
            opt.step()
            hist.append({k: float(v.detach()) for k, v in losses.items()})
        runs[name] = (hist, {n: p.detach().clone() for n, p in model.named_parameters()})
    for it, (a, b) in enumerate(zip(runs["hip"][0], runs["torch"][0])):
        for k in a:
            # step 0 starts from identical parameters: equal.  Later the two updates differ in the last bits (fused multiply-adds),
            # which can exchange proposals tied to an ulp and with them a sampled ROI (see the fixture test): RPN losses 1e-4, ROI 1e-2
            tol = 1e-6 if it == 0 else (1e-4 if "rpn" in k else 1e-2)
            assert abs(a[k] - b[k]) <= tol * abs(b[k]) + 1e-8, (it, k, a[k], b[k])
    assert all(abs(runs["hip"][0][0][k] - runs["hip"][0][2][k]) > 1e-6 * abs(runs["hip"][0][0][k]) for k in ("loss_cls", "loss_rpn_cls"))   # the staged weights followed the updates
    for n, p in runs["hip"][1].items():
        q = runs["torch"][1][n]
        assert float((p - q).norm()) <= 1e-4 * float(q.norm()) + 1e-9, n


def test_small_map_weight_gradient_kernel(ops):
    """sw_conv3x3_wgrad_small (FPN p5 / p6 of small images) against torch's conv2d weight gradient, with the FrozenBN scale"""
    torch.manual_seed(3)
    for (n, H, W, cin, cout), dtype in (((2, 4, 4, 64, 32), torch.float32), ((1, 2, 3, 32, 48), torch.float32), ((2, 3, 2, 64, 64), torch.bfloat16)):
        x = torch.randn(n, H, W, cin, device="cuda").to(dtype); dy = torch.randn(n, H, W, cout, device="cuda").to(dtype)
        scale = torch.rand(cout, device="cuda") + 0.5
        dw = torch.empty(cout, cin, 3, 3, device="cuda")
        ops.conv3x3_wgrad_small(x, dy, dw, cout_scale=scale)
        w = torch.zeros(cout, cin, 3, 3, device="cuda", requires_grad=True)
        F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
        want = w.grad * scale.view(-1, 1, 1, 1)
        assert float((dw - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_weight_staging_kernel_layouts_and_frozenbn_fold(ops, dtype):
    """sw_stage_weights_multi (ops.StagePlan): kind 0 (rows as they lie), 1 (OIHW -> [co][tap][ci]), 2 (OIHW -> [ci][8 - tap][co]),
    3 (f32 copy), with and without the FrozenBN fold w * scale[co]; channel counts that do and do not fill the kernel's blocks
    (448 < 512 input channels: two ranges per output channel; 72 output channels: a partial 64-channel tile; cols % 4 != 0).
    Layout and fold product: every staged element bit for bit against `w * scale` with the scale the kernel wrote; that scale /
    shift against layers/batch_norm.py:52-60 evaluated on the CPU (scale = bn_w * (1 / sqrt(var + eps))) within 3e-7 relative
    (the device's sqrt / divide differ from the host's in the last bit for a few channels)."""
    torch.manual_seed(11)
    dev = "cuda"
    entries, want = [], []

    def bn_of(c):
        return (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev), torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.1)
    for kind, co, ci, with_bn in [(0, 72, 130, True), (0, 256, 64, False), (0, 40, 12544, True), (1, 72, 128, True), (1, 64, 512, False),
                                  (1, 24, 40, True), (2, 72, 128, True), (2, 512, 20, False), (2, 130, 7, True), (3, 1, 104, False)]:
        w = torch.randn(co, ci, 3, 3, device=dev) if kind in (1, 2) else torch.randn(co, ci, device=dev)
        dst = torch.full((w.numel(),), float("nan"), device=dev, dtype=torch.float32 if kind == 3 else dtype)
        e = dict(kind=kind, w=w, dst=dst)
        if with_bn:
            e.update(bn=bn_of(co), scale=torch.empty(co, device=dev), shift=torch.empty(co, device=dev))
        entries.append(e)
    ops.StagePlan(entries, dtype).run()
    torch.cuda.synchronize()
    bad = []
    for e in entries:
        w, kind = e["w"], e["kind"]
        weff = w
        if "bn" in e:
            bw, bb, bm, bv = (t.cpu() for t in e["bn"])
            sc = bw * (1.0 / torch.sqrt(bv + 1e-5))
            sh = bb - bm * sc
            ok_f = bool(torch.allclose(e["scale"].cpu(), sc, rtol=3e-7, atol=0)) and bool(torch.allclose(e["shift"].cpu(), sh, rtol=3e-6, atol=1e-6))
            weff = w * e["scale"].view(-1, *([1] * (w.dim() - 1)))
        else:
            ok_f = True
        if kind == 1:
            ref = weff.permute(0, 2, 3, 1).reshape(-1)                       # [co][tap][ci]
        elif kind == 2:
            ref = weff.flip(2, 3).permute(1, 2, 3, 0).reshape(-1)            # [ci][8 - tap][co]
        else:
            ref = weff.reshape(-1)
        ref = ref.to(e["dst"].dtype)
        if not (bool(torch.equal(e["dst"], ref)) and ok_f):
            bad.append((kind, tuple(w.shape), "bn" in e, int((e["dst"].float() != ref.float()).sum()), ok_f))
    assert not bad, bad


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_levels_convolution_node_against_single_launches_and_with_unused_outputs(golden_dir, dtype):
    """frcnn._Conv3x3LevelsFn (the FPN output convolutions / the RPN head's shared convolution as ONE launch each way,
    sw_conv3x3_multi; fp32: the one-by-one path behind the same node): outputs and every gradient equal the per-level
    `Conv.forward` calls (fp32 bit for bit, bf16 to the last bit of a result: see below); a loss that uses ONE of the outputs
    leaves the other levels' input gradients exactly zero instead of failing on their missing cotangents."""
    import sos_wsod_amd.frcnn as F
    t = np.load(os.path.join(golden_dir, "stage3_a.npz"))
    K = int(t["K"])
    P = FO.make_params(K, tag="s3a", head_scale=float(t["head_scale"]))
    model = _model(K, P, "s3a", dtype=dtype)
    model.train()
    model.refresh_staged_weights()
    fpn = model.backbone
    convs = [getattr(fpn, f"fpn_output{s}") for s in F.FPN_STAGES]
    torch.manual_seed(2)
    xs = [torch.randn(2, h, w, 256, device="cuda").to(dtype).requires_grad_(True) for h, w in ((40, 56), (20, 28), (10, 14), (5, 7))]
    gs = [torch.randn(2, h, w, 256, device="cuda").to(dtype) for h, w in ((40, 56), (20, 28), (10, 14), (5, 7))]

    def grads():
        out = [x.grad.clone() for x in xs] + [c.weight.grad.clone() for c in convs] + [c.bias.grad.clone() for c in convs]
        for x in xs:
            x.grad = None
        for c in convs:
            c.weight.grad = None; c.bias.grad = None
        return out
    outs = F._conv3x3_levels(convs, xs)
    torch.autograd.backward(list(outs), gs)
    got = grads()
    ref_o = [c(x) for c, x in zip(convs, xs)]
    torch.autograd.backward(ref_o, gs)
    want = grads()
    torch.cuda.synchronize()
    # fp32: the same launches behind the node: identical.  bf16: alone, a map of few tiles runs the kernel's two-K-group form
    # (partial sums of alternate channel chunks exchanged at the end), in the shared launch the one-group form: the f32 sums are
    # associated differently, so a result may differ in its last bf16 bit
    def close(a, b):
        if dtype == torch.float32:
            return bool(torch.equal(a, b))
        return float((a.float() - b.float()).abs().max()) <= 1.6e-2 * float(b.float().abs().max()) + 1e-6
    assert all(close(a, b) for a, b in zip(outs, ref_o))
    assert all(close(a, b) for a, b in zip(got, want)), [float((a.float() - b.float()).abs().max() / b.float().abs().max()) for a, b in zip(got, want)]
    # a shared layer on every level (the RPN head's form) and a loss on one output only
    outs = F._conv3x3_levels([convs[0]] * 4, xs, relu=True)
    outs[1].float().sum().backward()
    torch.cuda.synchronize()
    assert xs[1].grad is not None and float(xs[1].grad.abs().max()) > 0
    assert all(float(xs[i].grad.abs().max()) == 0 for i in (0, 2, 3))
    ref = convs[0](xs[1].detach().requires_grad_(True), relu=True)
    w_before = convs[0].weight.grad.clone()
    convs[0].weight.grad = None
    ref.float().sum().backward()
    assert close(w_before, convs[0].weight.grad)
