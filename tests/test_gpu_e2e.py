"""GPU end-to-end parity of one OICR+ iteration (forward + backward through the C-ABI kernels)
against (a) the golden fixtures generated from the reference's own Python and (b) the CPU oracle.

fp32 mode (exact f32 MFMA): losses within 1e-4 relative (north_star's bar), integer outputs bit exact,
gradients within 2e-4 of the tensor's max.
bf16 mode: compared with the oracle emulating the same bf16 storage points; losses within 2e-2 relative
(bf16 has 8 significant bits; the pseudo-label sets must still be identical for these fixtures)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oicr_oracle as O  # noqa: E402  (checker only)
from helpers import build_model, load_params, to_batched_inputs  # noqa: E402


def _setup(case, golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, f"e2e_{case}.npz"), allow_pickle=False)
    K, R, H, W = int(g["K"]), int(g["R"]), int(g["H"]), int(g["W"])
    dan = tuple(int(x) for x in g["dan"])
    P = O.make_params(K, dan, tag="p" + case, head_scale=float(g["head_scale"]))
    views, gt = O.make_views(H, W, R, n_gt=int(g["n_gt"]), K=K, tag="v" + case)
    masks = O.make_masks(R, dan, tag="m" + case)
    model = build_model(K, dan, dtype)
    load_params(model, P)
    model.train()
    model.roi_heads.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
    return g, P, views, gt, masks, model


@pytest.mark.parametrize("case", ["s0", "s1", "c0"])
def test_fp32_iteration_matches_reference_golden(case, golden_dir):
    from sos_wsod_amd.events import EventStorage
    g, P, views, gt, masks, model = _setup(case, golden_dir, torch.float32)
    K = int(g["K"])
    with EventStorage(0):
        losses = model(to_batched_inputs(views, gt))
        total = sum(losses.values())
        total.backward()
    torch.cuda.synchronize()
    assert set(losses.keys()) == {k[5:] for k in g.files if k.startswith("loss/")}
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(v.item() - ref) <= 1e-4 * abs(ref), (k, v.item(), ref)
    aux = model.roi_heads.last_aux
    for k in range(4):
        r = aux["rounds"][k]
        n = int(r["pgt_count"].item())
        assert np.array_equal(r["pgt_index"][:n].cpu().numpy(), g[f"r{k}/pgt_index"])
        assert np.array_equal(r["pgt_class"][:n].cpu().numpy(), g[f"r{k}/pgt_classes"])
        assert np.array_equal(r["lab_class"].cpu().numpy(), g[f"r{k}/gt_classes"])
        assert np.array_equal(r["lab_index"].cpu().numpy(), g[f"r{k}/gt_index"])
        np.testing.assert_allclose(r["lab_weight"].cpu().numpy(), g[f"r{k}/gt_weights"], rtol=5e-3)   # weights down to 1e-29 = exp(-65): |logit| * 1e-5 relative
    R = int(g["R"])
    for v in range(4):
        np.testing.assert_allclose(aux["scores"][v].cpu().numpy(), g[f"wsddn_v{v}"], rtol=2e-3, atol=1e-8)   # peaky softmax of |logit|~50 amplifies f32 summation-order noise
        np.testing.assert_allclose(aux["fc7"][v * R:(v + 1) * R].cpu().numpy(), g[f"fc7_v{v}"], rtol=1e-4, atol=1e-4)
    sd = dict(model.named_parameters())
    for key in g.files:
        if key.startswith("grad/"):
            ref, got = g[key], sd[key[5:]].grad.cpu().numpy()
        elif key.startswith("grads/"):
            ref, got = g[key], sd[key[6:]].grad.cpu().numpy().ravel()[::997]
        else:
            continue
        # + 1e-7 absolute: e.g. d/d(det.bias) is analytically 0 (softmax over proposals is shift invariant), both sides hold noise.
        # Backbone gradients get 2e-2: ONE ROIPool argmax that flips between two feature values equal to ~1e-6 relative
        # (f32 summation order, GPU vs CPU conv) re-routes that bin's gradient to the neighbouring pixel — measured on
        # this fixture: 1 flip in 3.7 M bins carrying 8 % of max|dfeat|.  The backward CHAIN itself is checked to 1e-4 in
        # test_backbone_backward_matches_autograd and the ROIPool scatter in test_gpu_kernels.py.
        tol = 2e-2 if "backbone" in key else 2e-4
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max() + 1e-7, key
    for name in g["frozen"]:
        assert sd[str(name)].grad is None


@pytest.mark.parametrize("case", ["s0", "c0"])
def test_bf16_iteration_close_to_bf16_emulating_oracle(case, golden_dir):
    from sos_wsod_amd.events import EventStorage
    g, P, views, gt, masks, model = _setup(case, golden_dir, torch.bfloat16)
    K = int(g["K"])
    ol, oaux, ograds = O.oicr_plus_iteration(P, views, gt, masks, K=K, bf16=True, want_grads=True)
    with EventStorage(0):
        losses = model(to_batched_inputs(views, gt))
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    for k, v in losses.items():
        assert abs(v.item() - ol[k]) <= 2e-2 * abs(ol[k]) + 1e-5, (k, v.item(), ol[k])
    aux = model.roi_heads.last_aux
    for k in range(4):
        n = int(aux["rounds"][k]["pgt_count"].item())
        assert np.array_equal(aux["rounds"][k]["pgt_index"][:n].cpu().numpy(), oaux["rounds"][k]["pgt"]["index"])
    # EVERY gradient of the benchmarked mode against the oracle's autograd over the same bf16 storage points.  `_rb` (x.to(bf16)
    # .float()) rounds the GRADIENT at the same points too (autograd of the two casts), so the oracle's backward carries the bf16
    # rounding of dZ / dpooled / dfeat like the HIP path; the one storage point it lacks is the bf16 copy of dlogits (measured
    # on this fixture by rounding it in the oracle: 0.1-0.4 % per tensor).
    # The bar VERDICT r2 asked for (relative L2 <= 2e-2, cosine >= 0.999 per tensor) cannot be met by ANY two bf16 evaluations
    # of this network, the oracle against itself included: f32 inputs that differ in the last bit round differently from the
    # first layers on, the flips feed the next layer's sums and cause more flips, and within a few layers the two runs'
    # activations are ~0.5 % apart (measured: HIP vs oracle fc6 input 0.54 %, fc7 0.60 %; oracle vs oracle with the backbone
    # weights perturbed by 1e-6: fc7 0.69 %).  With |logit| ~ 50 that moves the softmax over PROPOSALS by percents, ROIPool argmax
    # positions move to neighbouring pixels (dfeat: cosine 0.71), and the weight gradients — sums over proposals / pixels with
    # heavy cancellation — separate by 5-12 % (backbone) and up to 19 % (cls / det weights) for the ORACLE PAIR.
    # So the floor is measured here, with the oracle itself, and the HIP path must stay within it:
    #   relative L2 (HIP, oracle) <= 2e-2 + 2 x relative L2 (oracle with weights perturbed by 1e-6, oracle)   for every tensor.
    # Teacher forced at fc7 (fc7_override: everything downstream evaluated at the HIP path's own fc7 values) the predictor
    # gradients lose the softmax amplification and DO meet the 2e-2 / 0.999 bar (0.1-0.3 % measured): asserted as well.
    # d(det.bias) is analytically 0 (the softmax over proposals is shift invariant): noise on both sides, absolute check.
    sd = dict(model.named_parameters())
    R = int(g["R"])
    fc7 = aux["fc7"].float().cpu().numpy()
    _, _, fgrads = O.oicr_plus_iteration(P, views, gt, masks, K=K, bf16=True, want_grads=True,
                                         fc7_override=[fc7[v * R:(v + 1) * R] for v in range(4)])
    floors = []
    for seed in (0, 1):
        rng = np.random.RandomState(seed)
        Pp = {k: (v * (1 + 1e-6 * rng.randn(*v.shape)).astype(np.float32) if k.startswith("backbone.") and k.endswith("weight") else v)
              for k, v in P.items()}
        floors.append(O.oicr_plus_iteration(Pp, views, gt, masks, K=K, bf16=True, want_grads=True)[2])

    def rel_cos(got, ref):
        got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
        nr = np.linalg.norm(ref)
        return float(np.linalg.norm(got - ref) / (nr + 1e-300)), float((got * ref).sum() / (np.linalg.norm(got) * nr + 1e-300))
    rows, bad = [], []
    for name, p in sd.items():
        if not p.requires_grad:
            assert p.grad is None
            continue
        assert ograds.get(name) is not None and p.grad is not None, name
        got = p.grad.double().cpu().numpy()
        if name.endswith("box_predictor.det.bias"):
            assert np.abs(got).max() <= 1e-4 and np.abs(ograds[name]).max() <= 1e-4, name
            continue
        free, forced = rel_cos(got, ograds[name]), rel_cos(got, fgrads[name])
        floor = max(rel_cos(f[name], ograds[name])[0] for f in floors)
        rows.append((name, free, forced, floor))
        if free[0] > 2e-2 + 2.0 * floor:
            bad.append((name, "free", free, floor))
        if name.startswith("roi_heads.box_predictor") or name.startswith("roi_heads.box_refinery"):
            if not (forced[0] <= 2e-2 and forced[1] >= 0.999):
                bad.append((name, "forced", forced))
    print("bf16 gradients vs the bf16-emulating oracle: name | free-running rel L2, cos | teacher-forced at fc7 rel L2, cos | oracle's own floor")
    for n, a, b, f in rows:
        print("   %-50s %.3e %.6f   %.3e %.6f   %.3e" % (n, a[0], a[1], b[0], b[1], f))
    assert not bad, bad


def test_backbone_standalone_api_and_roipooler(golden_dir):
    """Tier-1 API: backbone(x NCHW f32) -> {"plain5": (N,512,h,w)}; ROIPooler([feat], [Boxes]) -> (R,512,7,7)."""
    from sos_wsod_amd.structures import Boxes
    g, P, views, gt, masks, model = _setup("s0", golden_dir, torch.float32)
    x = torch.stack([O.preprocess(torch.from_numpy(views[0]["image"])), O.preprocess(torch.from_numpy(views[1]["image"]))])
    out = model.backbone(x.cuda())
    assert set(out.keys()) == {"plain5"}
    f = out["plain5"]
    assert tuple(f.shape) == tuple(int(v) for v in g["plain5_shape"])
    np.testing.assert_allclose(f[0].detach().contiguous().cpu().numpy().ravel()[::997], g["plain5_v0_sample"], rtol=1e-4, atol=1e-3)
    shp = model.backbone.output_shape()["plain5"]
    assert shp.channels == 512 and shp.stride == 8 and model.backbone.size_divisibility == 0
    pooled = model.roi_heads.box_pooler([f[0:1]], [Boxes(torch.from_numpy(views[0]["boxes"]).cuda())])
    ref, _ = O.roi_pool_fwd(f[0:1].detach().contiguous().cpu().numpy(), O.boxes_to_rois(torch.from_numpy(views[0]["boxes"])).numpy(), 1 / 8)
    assert np.array_equal(pooled.detach().cpu().numpy(), ref)




@pytest.mark.parametrize("dtype", [torch.float32])
def test_backbone_backward_matches_autograd(dtype):
    """explicit VGG backward (wgrad / flipped-weight dgrad / pool routing / fused ReLU masks) vs torch autograd
    on the oracle's VGG, same weights, random upstream gradient."""
    P = O.make_params(20, (8, 8), tag="bbk")
    model = build_model(20, (8, 8), dtype)
    load_params(model, P)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 72, 88, generator=gen) * 60
    Pt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in P.items() if k.startswith("backbone")}
    f_ref = O.vgg16_forward(x, Pt)
    gy = torch.randn(f_ref.shape, generator=gen)
    f_ref.backward(gy)
    f = model.backbone(x.cuda())["plain5"]
    assert (f.detach().cpu() - f_ref.detach()).abs().max() <= 2e-4 * f_ref.abs().max()
    f.backward(gy.cuda())
    sd = dict(model.named_parameters())
    worst = {}
    for k, p in Pt.items():
        got = sd[k].grad
        if k.startswith(("backbone.plain1", "backbone.plain2")):
            assert got is None
            continue
        err = float((got.cpu() - p.grad).abs().max() / p.grad.abs().max())
        worst[k] = err
    print({k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-4, worst


def test_inference_matches_reference_golden(golden_dir):
    """a23: _forward_box_test + fast_rcnn_inference_single_image (mean of 4 softmaxes / delta sets, decode, clip,
    score > 1e-6, per-class NMS 0.3, top-100) vs the reference's own output on the same inputs (fp32 mode)."""
    from sos_wsod_amd.structures import Boxes, Instances
    g, P, views, gt, masks, model = _setup("s0", golden_dir, torch.float32)
    ref = np.load(os.path.join(golden_dir, "infer_s0.npz"))
    model.eval()
    v = views[0]
    H, W = v["image"].shape[1:]
    p = Instances((H, W)); p.proposal_boxes = Boxes(torch.from_numpy(v["boxes"])); p.objectness_logits = torch.from_numpy(v["obj"])
    out = model([{"image": torch.from_numpy(v["image"]), "proposals": p}])
    inst = out[0]["instances"]
    assert len(inst) == len(ref["scores"])
    assert np.array_equal(inst.pred_classes.cpu().numpy(), ref["pred_classes"])        # detection order and classes: exact
    np.testing.assert_allclose(inst.scores.cpu().numpy(), ref["scores"], rtol=1e-4)
    np.testing.assert_allclose(inst.pred_boxes.tensor.cpu().numpy(), ref["pred_boxes"], rtol=1e-4, atol=1e-2)
    res, all_scores, all_boxes = model.inference([{"image": torch.from_numpy(v["image"]), "proposals": p}], do_postprocess=False)
    np.testing.assert_allclose(all_scores[0].cpu().numpy(), ref["all_scores"].reshape(all_scores[0].shape), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_staging_keeps_weight_copies_current(golden_dir, dtype):
    """The optimizer rewrites the persistent compute-dtype weight copies (ops.STAGING) in its update pass.  After 3 steps
    every registered copy must equal, bit for bit, what the staging kernels make of the CURRENT f32 master, and the
    modules' caches must report those copies as current (so the next forward launches no staging kernel)."""
    import sos_wsod_amd.ops as ops
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.trainer import Trainer
    g, P, views, gt, masks, _ = _setup("s0", golden_dir, dtype)
    data = to_batched_inputs(views, gt)
    ops.STAGING.clear()
    model = build_model(int(g["K"]), tuple(int(x) for x in g["dan"]), dtype)
    load_params(model, P)
    model.train()
    model.roi_heads.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
    w0 = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad],
                 1e-3, momentum=0.9)
    tr = Trainer(model, opt)
    for _ in range(3):
        tr.run_step(data)
    torch.cuda.synchronize()
    assert len(ops.STAGING) >= 9 + 2 + 10, len(ops.STAGING)        # conv3_1..conv5_3, fc6/fc7, 10 predictor matrices
    n_checked = 0
    for st in ops.STAGING.values():
        p = st["param"]()
        if st["kind"] == 2:
            for mode, buf in ((0, st["stage0"]), (1, st["stage1"])):
                if buf is None:
                    continue
                want = torch.zeros_like(buf)
                ops.conv_weight_prep(p.detach(), want, mode, st["d2"] if mode == 0 else None)
                assert torch.equal(buf, want)
                n_checked += 1
        else:
            assert torch.equal(st["stage0"], p.detach().to(dtype))
            n_checked += 1
    assert n_checked >= 9 + 8 + 2 + 10
    for k, v in model.named_parameters():
        if v.requires_grad:
            assert not torch.equal(v.detach(), w0[k]), k               # the steps did move the weights
    # caches are current: a forward after the steps must not rebuild anything
    bb, hd = model.backbone, model.roi_heads
    for (wid, mode), (key, buf) in bb._wk_cache.items():
        w = next(c.weight for blk in bb.blocks for c in blk.convs() if id(c.weight) == wid)
        assert key[0] == ops.param_key(w)
    assert hd._stage_cache["fc1"][0] == [ops.param_key(hd.box_head.fc1.weight)]
    assert hd._stage_cache["fc2"][0] == [ops.param_key(hd.box_head.fc2.weight)]
    assert hd._stage_cache["heads"][0] == [ops.param_key(p) for p in hd._flat_params()[4::2]]


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE.json's full size (4 views 512x512, R = 2000, K = 20, 4096-wide heads): too large for the CPU oracle inside a
# test, so the check goes through size-independent properties and through an independent GPU implementation (torch's own
# hipBLASLt matmul / conv for the contractions).
def _full_size_model_and_data():
    import bench
    model = bench.build(torch.device("cuda", 0), torch.bfloat16)
    model.train()
    return model, bench.make_inputs(torch.device("cuda", 0), 7), bench


def test_full_size_iteration_properties():
    import sos_wsod_amd.ops as ops
    from sos_wsod_amd.events import EventStorage
    model, data, bench = _full_size_model_and_data()
    R, K = bench.R, bench.K
    with EventStorage(0):
        losses = model(data)
        losses.total().backward()
    torch.cuda.synchronize()
    vec = losses.vector.detach().cpu()
    assert vec.shape == (9,) and torch.isfinite(vec).all() and (vec >= 0).all()
    aux = model.roi_heads.last_aux
    # WSDDN scores: product of a softmax over classes and one over proposals => every (view, class) column sums to <= 1
    sc = aux["scores"]
    assert sc.shape == (4, R, K) and (sc >= 0).all()
    assert (sc.sum(1) <= 1 + 1e-4).all() and (sc.sum((1, 2)) > 0).all()
    boxes = data[0]["proposals1"].proposal_boxes.tensor
    gt = set(int(c) for c in data[0]["instances1"].gt_classes.tolist())
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for rnd in aux["rounds"]:
        n = int(rnd["pgt_count"].item())
        idx = rnd["pgt_index"][:n].long(); cls = rnd["pgt_class"][:n]; s = rnd["pgt_score"][:n]
        assert n >= len(gt) and set(cls.tolist()) <= gt and (idx >= 0).all() and (idx < R).all()
        assert (s[:-1] >= s[1:]).all()                                   # kept list is score-descending
        b = boxes[idx]                                                     # NMS 0.01: kept boxes barely overlap
        lt = torch.max(b[:, None, :2], b[None, :, :2]); rb = torch.min(b[:, None, 2:], b[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        iou = inter / (area[idx][:, None] + area[idx][None, :] - inter)
        iou.fill_diagonal_(0)
        assert (iou <= 0.01 + 1e-6).all()
        lab = rnd["lab_class"]; w = rnd["lab_weight"]; li = rnd["lab_index"].long()
        assert ((lab >= -1) & (lab <= K)).all()
        # labels follow the Matcher: recompute the best IoU against the kept list and the {0,-1,1} thresholds [0.5, 0.6]
        ltp = torch.max(boxes[:, None, :2], b[None, :, :2]); rbp = torch.min(boxes[:, None, 2:], b[None, :, 2:])
        ip = (rbp - ltp).clamp(min=0).prod(-1)
        ioup = torch.where(ip > 0, ip / (area[:, None] + area[idx][None, :] - ip), torch.zeros_like(ip))
        best, bj = ioup.max(1)
        fg, bg = best >= 0.6, best < 0.5
        clear = ((best - 0.6).abs() > 1e-5) & ((best - 0.5).abs() > 1e-5)  # away from the thresholds (f32 rounding of IoU)
        assert (lab[fg & clear] == cls[bj][fg & clear]).all() and (lab[bg & clear] == K).all()
        assert (lab[~fg & ~bg & clear] == -1).all()
        assert torch.equal(w[clear & fg], s[bj][clear & fg]) and torch.equal(li[clear & fg], idx[bj][clear & fg])
    # every trainable parameter received a finite gradient of its own shape; frozen ones none
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and p.grad.shape == p.shape and torch.isfinite(p.grad).all(), name
        else:
            assert p.grad is None, name
    # ROIPool at full size: argmax points at a pixel that holds exactly the pooled value (before the objectness scale)
    hd = model.roi_heads
    f = torch.randn(2, 63, 63, 512, device="cuda").to(torch.bfloat16)
    rois = torch.cat([(torch.arange(2 * R, device="cuda") >= R).float()[:, None], torch.cat([boxes, boxes], 0)], 1).contiguous()
    out = torch.empty(2 * R, 512 * 49, device="cuda", dtype=torch.bfloat16)
    arg = torch.empty(2 * R, 512 * 49, device="cuda", dtype=ops.roi_argmax_dtype(63, 63))
    ops.roi_pool_fwd(f, rois, out, arg, hd.box_pooler.scale, 7, 7)
    a = ops.argmax_to_int32(arg).view(2 * R, 512, 49)
    img = (torch.arange(2 * R, device="cuda") >= R).long()
    sel = torch.randperm(2 * R, device="cuda")[:200]
    for r in sel.tolist():
        ar = a[r]                                                          # (512, 49)
        val = out[r].view(512, 49)
        ok = ar >= 0
        px = f[img[r]].view(63 * 63, 512)                                  # (pixels, C)
        got = px[ar.clamp(min=0), torch.arange(512, device="cuda")[:, None].expand(512, 49)]
        assert torch.equal(got[ok], val[ok]) and (val[~ok] == 0).all()


def test_full_size_contractions_match_torch_matmul():
    """fc6 forward / data gradient / weight gradient and conv5_3 at their benchmark sizes against torch's own GPU matmul
    and conv2d (hipBLASLt / MIOpen: an independent implementation), bf16 inputs, f32 accumulation: <= 1e-2 of the max."""
    import sos_wsod_amd.ops as ops
    dt = torch.bfloat16
    M, D0, D1 = 8000, 25088, 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    X = (torch.randn(M, D0, device="cuda", generator=g) * 0.5).to(dt)
    W = (torch.randn(D1, D0, device="cuda", generator=g) * 0.02).to(dt)
    dZ = (torch.randn(M, D1, device="cuda", generator=g) * 0.5).to(dt)
    rows = torch.arange(0, M, 37, device="cuda")

    def close(got, ref):
        return (got.float() - ref.float()).abs().max() <= 1e-2 * ref.float().abs().max()
    Y = torch.empty(M, D1, device="cuda", dtype=dt)
    ops.gemm(X, W, Y, M, D1, D0)
    assert close(Y[rows], X[rows].float() @ W.float().t())
    WT = W.t().contiguous()
    dX = torch.empty(M, D0, device="cuda", dtype=dt)
    ops.gemm(dZ, WT, dX, M, D0, D1, ep=ops.make_epilogue(out_dtype=dt))                       # NT on the transposed copy
    assert close(dX[rows], dZ[rows].float() @ W.float())
    dX2 = torch.empty(M, D0, device="cuda", dtype=dt)
    ops.gemm(dZ, W, dX2, M, D0, D1, b_kstrided=True, ep=ops.make_epilogue(out_dtype=dt))      # NN, K-strided operand
    assert close(dX2[rows], dZ[rows].float() @ W.float())
    dW = torch.empty(D1, D0, device="cuda")
    ops.gemm(dZ, X, dW, D1, D0, M, a_kstrided=True, b_kstrided=True)
    cols = torch.arange(0, D1, 61, device="cuda")
    assert close(dW[cols], dZ[:, cols].float().t() @ X.float())
    del X, W, dZ, Y, dX, dX2, dW, WT
    x = (torch.randn(2, 63, 63, 512, device="cuda", generator=g) * 0.5).to(dt)
    w = (torch.randn(512, 512, 3, 3, device="cuda", generator=g) * 0.02)
    wk = torch.empty(512, 9, 512, device="cuda", dtype=dt); ops.conv_weight_prep(w, wk, 0, 512)
    b = torch.randn(512, device="cuda", generator=g)
    out = torch.empty(2, 63, 63, 512, device="cuda", dtype=dt)
    ops.conv3x3(x, wk, out, 2, ops.make_epilogue(bias=b, relu=True, out_dtype=dt))
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).float(), w.to(dt).float(), b, padding=2, dilation=2))
    assert close(out.permute(0, 3, 1, 2), ref)
    dy = (torch.randn(2, 63, 63, 512, device="cuda", generator=g) * 0.5).to(dt)
    dw = torch.empty(512, 512, 3, 3, device="cuda")
    ops.conv3x3_wgrad(x, dy, dw, 2, splitk=3)
    xr = x.permute(0, 3, 1, 2).float().requires_grad_(False)
    wr = w.to(dt).float().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, None, padding=2, dilation=2).backward(dy.permute(0, 3, 1, 2).float())
    assert close(dw, wr.grad)


def test_voc_sized_nonsquare_views_run(monkeypatch):
    """a VOC-scale multi-scale pair is not square and not 512: 608x912 views (76x114 maps: the ROIPool plane no longer fits
    two workgroups per CU, the backward's fixed-point slab does not fit at all -> the other code paths) must train too"""
    import bench
    from sos_wsod_amd.events import EventStorage
    monkeypatch.setattr(bench, "H", 608); monkeypatch.setattr(bench, "W", 912); monkeypatch.setattr(bench, "R", 1500)
    dev = torch.device("cuda", 0)
    model = bench.build(dev, torch.bfloat16); model.train()
    data = bench.make_inputs(dev, 11)
    with EventStorage(0):
        losses = model(data)
        losses.total().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(losses.vector).all()
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    g3 = model.backbone.blocks[2].convs()[0].weight.grad
    assert g3.abs().max() > 0                                      # the gradient reached the first trainable conv


def test_checkpoint_load_invalidates_the_compute_copies(golden_dir, tmp_path):
    """weights loaded from a checkpoint (copy into the existing storage) must reach the next forward: the persistent
    compute-dtype copies are rebuilt, and the predictor weights stay row slices of their flat master"""
    import sos_wsod_amd.ops as ops
    from sos_wsod_amd.checkpoint import DetectionCheckpointer
    from sos_wsod_amd.events import EventStorage
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.trainer import Trainer
    dtype = torch.bfloat16
    g, P, views, gt, masks, model = _setup("s0", golden_dir, dtype)
    data = to_batched_inputs(views, gt)
    opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad],
                 1e-3, momentum=0.9)
    tr = Trainer(model, opt)
    tr.run_step(data); tr.run_step(data)
    other = build_model(int(g["K"]), tuple(int(x) for x in g["dan"]), dtype)
    load_params(other, O.make_params(int(g["K"]), tuple(int(x) for x in g["dan"]), tag="other", head_scale=float(g["head_scale"])))
    DetectionCheckpointer(other, str(tmp_path)).save("m", iteration=7)
    extra = DetectionCheckpointer(model, str(tmp_path)).resume_or_load("", resume=True)
    assert extra["iteration"] == 7
    hd = model.roi_heads
    with EventStorage(0), torch.no_grad():
        model.eval()
        model.train()
        losses = model(data)
    torch.cuda.synchronize()
    ref_sd = other.state_dict()
    assert torch.equal(hd._stage_cache["fc1"][1], ref_sd["roi_heads.box_head.fc1.weight"].to(dtype))
    flat_w, _ = hd._head_flat
    assert hd.box_predictor.cls.weight.data_ptr() == flat_w.data_ptr()                 # still a view of the flat master
    assert torch.equal(flat_w[:int(g["K"])], ref_sd["roi_heads.box_predictor.cls.weight"])
    wk = model.backbone._wk_cache[(id(model.backbone.blocks[4].convs()[2].weight), 0)][1]
    want = torch.zeros_like(wk)
    ops.conv_weight_prep(ref_sd["backbone.plain5.0.conv3.weight"], want, 0, 512)
    assert torch.equal(wk, want)
    assert torch.isfinite(losses.vector).all()


def test_tta_avg_against_the_reference_generated_fixture(golden_dir):
    """§8f row 2, pinned: tests/golden/tta_s0.npz was written by RUNNING the reference's own test_time_augmentation_avg.py
    (DatasetMapperTTAAVG :131-197, _get_augmented_boxes :343-369, _merge_detections :371-393; make_tta_golden.py) on the
    reference model of fixture s0 with 3 scales x flip.  Here: the device-built views (pixels by CRC32 = Pillow's, proposal
    boxes), the view-averaged score / box matrices, and the merged detections — same boxes (1e-4), same order, same classes."""
    import zlib
    from sos_wsod_amd.structures import Boxes, Instances
    from sos_wsod_amd.tta import DeviceTTAMapper, GeneralizedRCNNWithTTAAVG
    t = np.load(os.path.join(golden_dir, "tta_s0.npz"))
    g, P, views, gt, masks, model = _setup("s0", golden_dir, torch.float32)
    model.eval()
    v = views[0]
    h, w = v["image"].shape[1:]
    prop = Instances((h, w)); prop.proposal_boxes = Boxes(torch.from_numpy(v["boxes"]).cuda())
    prop.objectness_logits = torch.from_numpy(v["obj"]).cuda()
    inp = {"image": torch.from_numpy(np.ascontiguousarray(v["image"])).cuda(), "proposals": prop, "height": h, "width": w}
    mapper = DeviceTTAMapper(min_sizes=tuple(int(x) for x in t["min_sizes"]), max_size=int(t["max_size"]), flip=bool(t["flip"]),
                             proposal_topk=4000)
    built = mapper(dict(inp))
    assert len(built) == 6
    for i, (view, _) in enumerate(built):                     # the reference's view order: per size, plain then flipped
        assert tuple(view["image"].shape[1:]) == tuple(int(x) for x in t[f"view{i}/hw"])
        assert zlib.crc32(np.ascontiguousarray(view["image"].cpu().numpy()).tobytes()) == int(t[f"view{i}/crc"]), i
        np.testing.assert_allclose(view["proposals"].proposal_boxes.tensor.cpu().numpy(), t[f"view{i}/boxes"], rtol=1e-6, atol=1e-5)
    wrapper = GeneralizedRCNNWithTTAAVG(model, mapper)
    res = wrapper([inp])[0]["instances"]
    avg_scores, avg_boxes = wrapper.last_avg
    np.testing.assert_allclose(avg_scores.cpu().numpy(), t["avg_scores"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(avg_boxes.cpu().numpy(), t["avg_boxes"], rtol=1e-4, atol=1e-3)
    n = len(t["scores"])
    assert len(res) == n == 100
    np.testing.assert_allclose(res.scores.cpu().numpy(), t["scores"], rtol=1e-4, atol=1e-7)
    # same detections in the same order; inside a run of reference scores closer than 1e-5 relative (one pair 7e-8 apart here)
    # the order is not defined by float tolerance: such a run is compared as a set
    sc, cls, bx = t["scores"], res.pred_classes.cpu().numpy(), res.pred_boxes.tensor.cpu().numpy()
    i = 0
    while i < n:
        j = i + 1
        while j < n and abs(sc[j] - sc[j - 1]) <= 1e-5 * abs(sc[j - 1]):
            j += 1
        ours = sorted(zip(cls[i:j].tolist(), bx[i:j].tolist()))
        ref = sorted(zip(t["pred_classes"][i:j].tolist(), t["pred_boxes"][i:j].tolist()))
        assert [c for c, _ in ours] == [c for c, _ in ref], (i, j)
        np.testing.assert_allclose(np.array([b for _, b in ours]), np.array([b for _, b in ref]), rtol=1e-4, atol=1e-3)
        i = j


def test_tta_avg_merge(golden_dir):
    """TTA-avg (test_time_augmentation_avg.py:311-393): one identity view == plain inference; flip / resize views are mapped
    back by the inverse transforms (checked against a float64 restatement) and averaged before the final NMS"""
    from sos_wsod_amd.structures import Boxes, Instances
    from sos_wsod_amd.tta import DeviceTTAMapper, GeneralizedRCNNWithTTAAVG, ViewTransform
    g, P, views, gt, masks, model = _setup("s0", golden_dir, torch.float32)
    model.eval()
    v = views[0]
    h, w = v["image"].shape[1:]
    prop = Instances((h, w)); prop.proposal_boxes = Boxes(torch.from_numpy(v["boxes"]).cuda())
    prop.objectness_logits = torch.from_numpy(v["obj"]).cuda()
    inp = {"image": torch.from_numpy(np.ascontiguousarray(v["image"])).cuda(), "proposals": prop}
    plain = model.inference([inp])[0]["instances"]
    ident = GeneralizedRCNNWithTTAAVG(model, lambda d: [(d, ViewTransform((h, w), (h, w), False))])([inp])[0]["instances"]
    assert torch.equal(plain.pred_boxes.tensor, ident.pred_boxes.tensor) and torch.equal(plain.scores, ident.scores)
    assert torch.equal(plain.pred_classes, ident.pred_classes)
    # inverse transforms: resize (h,w)->(nh,nw) then flip; float64 restatement
    t = ViewTransform((h, w), (2 * h, 3 * w), True)
    b = torch.from_numpy(v["boxes"]).double()
    fwd = b * torch.tensor([3.0, 2.0, 3.0, 2.0], dtype=torch.float64)
    fwd = torch.stack([3.0 * w - fwd[:, 2], fwd[:, 1], 3.0 * w - fwd[:, 0], fwd[:, 3]], 1)
    assert torch.allclose(t.apply_box(torch.from_numpy(v["boxes"])).double(), fwd, atol=1e-3)
    assert torch.allclose(t.inverse_box(t.apply_box(torch.from_numpy(v["boxes"]))), torch.from_numpy(v["boxes"]), atol=1e-3)
    # several device-built views run end to end, detections stay inside the original image and sorted per the final NMS
    tta = GeneralizedRCNNWithTTAAVG(model, DeviceTTAMapper(min_sizes=(h, h + 32), max_size=4000, flip=True))
    out = tta([inp])[0]["instances"]
    assert out.image_size == (h, w) and len(out) > 0
    bx = out.pred_boxes.tensor
    assert (bx[:, 0] >= 0).all() and (bx[:, 2] <= w).all() and (bx[:, 1] >= 0).all() and (bx[:, 3] <= h).all()
    assert (out.scores[:-1] >= out.scores[1:]).all()


def test_multi_input_mapper_device(golden_dir):
    """the multi-view training mapper (dataset_mapper.py:272-425) on the device: proposal boxes / keep masks bit-exact
    against the masks of the reference's own Boxes class (tests/golden/input_a.npz), and its output feeds one training
    iteration directly (VOC-sized image, two drawn scales, four index-aligned proposal sets)"""
    from test_mapper_cpu import check_mapper_against_golden
    from sos_wsod_amd.events import EventStorage
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    check_mapper_against_golden("cuda")
    g = np.load(os.path.join(golden_dir, "input_a.npz"))
    h, w = (int(v) for v in g["orig_hw"])
    dev = torch.device("cuda", 0)
    img = torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
    d = {"image": img, "proposal_boxes": g["boxes"], "proposal_objectness_logits": g["logits"],
         "annotations": [{"bbox": [30.0, 40.0, 300.0, 330.0], "category_id": 4},
                         {"bbox": [200.0, 10.0, 480.0, 200.0], "category_id": 17}]}
    mapper = DeviceMultiInputMapper(min_sizes=(480, 576, 688), max_size=2000, seed=5)
    data = [mapper(d)]
    n = len(data[0]["proposals1"].proposal_boxes)
    assert all(len(data[0]["proposals" + k].proposal_boxes) == n for k in ("1", "2", "1_flip", "2_flip"))
    assert data[0]["image1"].shape != data[0]["image2"].shape and data[0]["image1"].is_cuda
    model = build_model(20, (256, 256), torch.bfloat16)
    model.train()
    with EventStorage(0):
        losses = model(data)
        losses.total().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(losses.vector).all()
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name


def test_full_size_iteration_is_reproducible():
    """two runs of the same iteration from the same state: every loss and every gradient is bitwise identical — slab split-K
    with ordered folds (conv and fc weight gradients, the GEMM tail peel), ordered bias sums, integer fixed-point ROI scatter,
    hash dropout: no floating-point atomics anywhere in the step"""
    from sos_wsod_amd.events import EventStorage
    model, data, bench = _full_size_model_and_data()
    runs = []
    for _ in range(2):
        model.roi_heads._drop_counter = 0                       # same dropout stream
        for p in model.parameters():
            p.grad = None
        with EventStorage(0):
            losses = model(data)
            losses.total().backward()
        torch.cuda.synchronize()
        runs.append((losses.vector.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    diff = [n for n, g0 in runs[0][1].items() if not torch.equal(g0, runs[1][1][n])]
    assert not diff, diff


@pytest.mark.parametrize("graph", [False, True])
def test_fused_fc1_update_equals_the_optimizer_step_bit_for_bit(monkeypatch, graph):
    """Round 6: on the single-GPU path fc1.weight's SGD update runs in the epilogue of fc6's weight-gradient GEMM (sw_epilogue.sgd_fused;
    the GEMM is deferred behind the fc6 data gradient, the last reader of the weight copies it rewrites; the 411 MB gradient is never
    written).  Full-size model, five steps with a learning-rate milestone, against the same trainer with SW_FUSE_FC1_UPDATE=0: every loss
    vector and every parameter identical bit for bit; eager and as one captured hipGraph; the fused launch really ran."""
    import bench
    import sos_wsod_amd.ops as ops
    from sos_wsod_amd.solver import HipSGD, WarmupMultiStepLR
    from sos_wsod_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    data = [bench.make_inputs(dev, 100 + i) for i in range(2)]
    calls = {"n": 0}
    real = ops.attach_sgd_fused

    def spy(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(ops, "attach_sgd_fused", spy)

    def run(fuse):
        monkeypatch.setenv("SW_FUSE_FC1_UPDATE", "1" if fuse else "0")
        model = bench.build(dev, torch.bfloat16); model.train()
        gs = [{"params": [p], "lr": 2e-3 if n.endswith(".bias") else 1e-3, "weight_decay": 0.0 if n.endswith(".bias") else 5e-4}
              for n, p in model.named_parameters() if p.requires_grad]
        opt = HipSGD(gs, 1e-3, momentum=0.9)
        tr = Trainer(model, opt, scheduler=WarmupMultiStepLR(opt, [3], gamma=0.5, warmup_iters=0), use_graph=graph)
        assert (model.roi_heads.__dict__.get("_fused_opt") is opt) == fuse
        losses = []
        for i in range(5):
            losses.append(tr.run_step(data[i % 2]).vector.detach().clone())
        tr.finish()
        torch.cuda.synchronize()
        w1 = model.roi_heads.box_head.fc1.weight
        assert ("momentum_buffer" in opt.state[w1])
        out = (losses, {n: p.detach().clone() for n, p in model.named_parameters()}, opt.state[w1]["momentum_buffer"].clone())
        del tr, opt, model
        return out
    n0 = calls["n"]
    lf, wf, mf = run(True)
    assert calls["n"] - n0 >= (2 if graph else 5), calls                      # (a replayed step launches from the graph: no Python call)
    n1 = calls["n"]
    lu, wu, mu = run(False)
    assert calls["n"] == n1
    for i, (a, b) in enumerate(zip(lf, lu)):
        assert torch.equal(a, b), (i, a, b)
    diff = [n for n in wf if not torch.equal(wf[n], wu[n])]
    assert not diff, diff
    assert torch.equal(mf, mu)
    assert not torch.equal(wf["roi_heads.box_head.fc1.weight"], bench.build(dev, torch.bfloat16).roi_heads.box_head.fc1.weight)


def test_train_loop_from_proposal_file_to_checkpoint(tmp_path):
    """the pieces of §8f in one loop, as train_net_multi.py wires them: proposal pickle -> device mapper -> Trainer.run_step
    (HipSGD + WarmupMultiStepLR) -> DetectionCheckpointer.save -> resume into a fresh model -> identical next iteration"""
    import pickle
    from sos_wsod_amd.checkpoint import DetectionCheckpointer
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    from sos_wsod_amd.proposals import load_proposals_into_dataset
    from sos_wsod_amd.solver import HipSGD, WarmupMultiStepLR
    from sos_wsod_amd.trainer import Trainer
    rng = np.random.RandomState(0)
    dev = torch.device("cuda", 0)
    h, w, n = 120, 160, 90
    ids, boxes, scores, records = [], [], [], []
    for i in range(3):
        x1 = rng.randint(0, w - 30, n); y1 = rng.randint(0, h - 30, n)
        b = np.stack([x1, y1, np.minimum(x1 + rng.randint(16, 90, n), w - 1), np.minimum(y1 + rng.randint(16, 70, n), h - 1)], 1)
        ids.append(f"{i:06d}"); boxes.append(b.astype(np.float32)); scores.append(rng.rand(n).astype(np.float32))
        records.append({"image_id": f"{i:06d}", "image": torch.randint(0, 256, (3, h, w), dtype=torch.uint8, device=dev),
                        "annotations": [{"bbox": [10.0, 10.0, 100.0, 90.0], "category_id": int(rng.randint(0, 20))}]})
    pkl = str(tmp_path / "props.pkl")
    pickle.dump({"indexes": ids, "boxes": boxes, "scores": scores}, open(pkl, "wb"))
    records = load_proposals_into_dataset(records, pkl)
    mapper = DeviceMultiInputMapper(min_sizes=(96, 128, 160), max_size=400, proposal_topk=80, seed=3)

    def make(seed):
        torch.manual_seed(seed)
        model = build_model(20, (256, 256), torch.bfloat16); model.train()
        groups = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
                  for nm, p in model.named_parameters() if p.requires_grad]
        opt = HipSGD(groups, 1e-3, momentum=0.9)
        sch = WarmupMultiStepLR(opt, [3, 5], warmup_iters=2)
        return model, opt, sch
    model, opt, sch = make(0)
    batches = [[mapper(records[i % 3])] for i in range(5)]
    tr = Trainer(model, opt, scheduler=sch)
    for i in range(3):
        ld = tr.run_step(batches[i])
        assert torch.isfinite(ld.vector).all()
    ck = DetectionCheckpointer(model, str(tmp_path), optimizer=opt, scheduler=sch)
    ck.save("model_0000002", iteration=2)
    model.roi_heads._drop_counter = 1000
    ref = tr.run_step(batches[3]).vector.detach().clone()
    ref_w = model.roi_heads.box_head.fc1.weight.detach().clone()
    # a fresh process would do exactly this
    model2, opt2, sch2 = make(1)
    ck2 = DetectionCheckpointer(model2, str(tmp_path), optimizer=opt2, scheduler=sch2)
    extra = ck2.resume_or_load("", resume=True)
    assert extra["iteration"] == 2 and sch2.last_epoch == 3
    model2.roi_heads._drop_counter = 1000
    tr2 = Trainer(model2, opt2, scheduler=sch2)
    got = tr2.run_step(batches[3]).vector.detach()
    assert torch.equal(got, ref)                                   # resumed run reproduces the original bit for bit
    assert torch.equal(model2.roi_heads.box_head.fc1.weight.detach(), ref_w)


def test_baseline_config1_fp32_parity_against_the_oracle():
    """BASELINE.json configs[0] / SURVEY §8d #1: one view set, 512x512 (+ 640x640 second scale), 500 proposals, K = 20, fp32 —
    the CPU oracle against the HIP path at full image size: losses within 1e-4 relative (north_star's bar), pseudo-label
    indices / classes and proposal labels bit exact.  (dan 1024: the contraction width does not change what is tested;
    the CPU side stays at a few seconds)"""
    from sos_wsod_amd.events import EventStorage
    K, R, H, W, dan = 20, 500, 512, 512, (1024, 1024)
    P = O.make_params(K, dan, tag="pcfg1", head_scale=30.0)
    views, gt = O.make_views(H, W, R, n_gt=2, K=K, scale2=1.25, tag="vcfg1")
    masks = O.make_masks(R, dan, tag="mcfg1")
    assert views[2]["image"].shape[1:] == (640, 640)
    ol, oaux, _ = O.oicr_plus_iteration(P, views, gt, masks, K=K)
    model = build_model(K, dan, torch.float32)
    load_params(model, P)
    model.train()
    model.roi_heads.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
    with EventStorage(0):
        losses = model(to_batched_inputs(views, gt))
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    for k, v in losses.items():
        assert abs(v.item() - ol[k]) <= 1e-4 * abs(ol[k]) + 1e-7, (k, v.item(), ol[k])
    aux = model.roi_heads.last_aux
    for k in range(4):
        r, o = aux["rounds"][k], oaux["rounds"][k]
        n = int(r["pgt_count"].item())
        assert np.array_equal(r["pgt_index"][:n].cpu().numpy(), o["pgt"]["index"])
        assert np.array_equal(r["pgt_class"][:n].cpu().numpy(), o["pgt"]["classes"])
        assert np.array_equal(r["lab_class"].cpu().numpy(), o["labels"]["gt_classes"])
        assert np.array_equal(r["lab_index"].cpu().numpy(), o["labels"]["gt_index"])


def test_step_graph_replay_equals_eager_steps(golden_dir):
    """Trainer(use_graph=True): after the second sight of an input signature the whole step (forward, backward, HipSGD) replays
    as one hipGraph.  Same start, same data sequence, same dropout stream => losses and weights of every step are bitwise those
    of the eager run, for new images / proposals / labels copied into the graph's static inputs each step."""
    from sos_wsod_amd.solver import HipSGD, WarmupMultiStepLR
    from sos_wsod_amd.trainer import Trainer
    K, dan, R, H, W = 20, (256, 256), 80, 96, 128
    P = O.make_params(K, dan, tag="pgraph", head_scale=3.0)

    def batches():
        out = []
        for i in range(6):
            views, _ = O.make_views(H, W, R, n_gt=2, K=K, tag=f"vgraph{i}")
            out.append(to_batched_inputs(views, np.array([(3 + i) % K, (11 + 2 * i) % K])))      # two classes each: one signature
            for k, v in out[-1][0].items():                                                       # device-resident inputs
                if k.startswith("image"):
                    out[-1][0][k] = v.cuda()
                elif k.startswith("proposals"):
                    v.proposal_boxes.tensor = v.proposal_boxes.tensor.cuda(); v.objectness_logits = v.objectness_logits.cuda()
        return out

    def run(use_graph):
        model = build_model(K, dan, torch.bfloat16)
        load_params(model, P)
        model.train()
        model.roi_heads.seed = 77
        groups = [{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad]
        opt = HipSGD(groups, 1e-3, momentum=0.9)
        # the learning rate moves DURING the replayed steps (milestones at 3 and 4): it reaches the captured update through the
        # optimizer's device buffer, not through kernel arguments — still one capture, still bitwise the eager run
        sched = WarmupMultiStepLR(opt, [3, 4], gamma=0.5, warmup_iters=0)
        tr = Trainer(model, opt, scheduler=sched, use_graph=use_graph, check_finite_every=2, metrics_period=1)
        losses = []
        for b in batches():
            ld = tr.run_step(b)
            losses.append(ld.vector.detach().clone())
        tr.finish()
        torch.cuda.synchronize()
        return tr, losses, {n: p.detach().clone() for n, p in model.named_parameters()}
    tr_e, le, we = run(False)
    tr_g, lg, wg = run(True)
    assert tr_e._graphs is None and tr_g._graphs is not None
    assert tr_g._graphs.captures == 1 and tr_g._graphs.replays == 4          # steps 0, 1 eager (step 1 = first sight), 2..5 replayed
    assert abs(tr_g.optimizer.param_groups[0]["lr"] - 0.25e-3) < 1e-12
    for i, (a, b) in enumerate(zip(le, lg)):
        assert torch.equal(a, b), (i, a, b)
    # the metric sink holds every step's OWN values (snapshots of the graph's static outputs, not views of them)
    for tr_, ls in ((tr_e, le), (tr_g, lg)):
        hist = tr_.storage.history("loss_cls")
        assert [it for it, _ in hist] == list(range(6))
        assert [float(v) for _, v in hist] == [float(l[0]) for l in ls]
        assert [float(v) for _, v in tr_.storage.history("total_loss")] == pytest.approx([float(l.sum()) for l in ls], rel=1e-6)
    assert len(tr_g.storage.history("roi_head/num_pgt_r0")) == 6
    bad = [n for n in we if not torch.equal(we[n], wg[n])]
    assert not bad, bad
    assert tr_g.raw_model.roi_heads._drop_counter == tr_e.raw_model.roi_heads._drop_counter > 0
    # the reference's attribute (roi_heads_oicrplus.py:206) is built on access, from the labels of the last forward
    gt = tr_e.raw_model.roi_heads.gt_classes_img_int
    assert isinstance(gt, list) and gt[0].dtype == torch.int64 and sorted(gt[0].tolist()) == sorted({(3 + 5) % K, (11 + 10) % K})
    # replay switched off: the same trainer runs the step eagerly on its stream (bench.py's launch-by-launch comparison) and
    # continues the same trajectory as an all-eager trainer given the same extra batch
    extra = batches()[0]
    tr_g._graphs.enabled = False
    la, lb = tr_g.run_step(extra).vector.clone(), tr_e.run_step(extra).vector.clone()
    assert tr_g._graphs.replays == 4 and torch.equal(la, lb)


def test_two_images_per_gpu_equal_the_mean_of_two_single_image_iterations():
    """The reference asserts one image per GPU (rcnn_multi.py:148); B images here are B times the same computation on stacked
    rows with the losses averaged — what DDP forms over B ranks.  Two images of DIFFERENT view sizes, proposal counts and label
    sets: losses and every gradient of the B = 2 iteration equal the mean of the two B = 1 iterations (fp32, dropout off: the
    hash stream indexes rows of the stacked matrix, so the second image would draw another mask)."""
    from sos_wsod_amd.events import EventStorage
    K, dan = 20, (256, 256)
    P = O.make_params(K, dan, tag="pb2", head_scale=6.0)
    va, ga = O.make_views(96, 128, 70, n_gt=2, K=K, tag="vb2a")
    vb, gb = O.make_views(112, 96, 53, n_gt=3, K=K, tag="vb2b")
    da, db = to_batched_inputs(va, ga)[0], to_batched_inputs(vb, gb)[0]
    model = build_model(K, dan, torch.float32)
    load_params(model, P)
    model.train()
    model.roi_heads.train_dropout = False

    def run(batch):
        for p in model.parameters():
            p.grad = None
        with EventStorage(0):
            ld = model(batch)
            ld.total().backward()
        torch.cuda.synchronize()
        return ld.vector.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    la, gra = run([da])
    lb, grb = run([db])
    l2, gr2 = run([da, db])
    want = (la + lb) / 2
    assert ((l2 - want).abs() <= 1e-5 * want.abs() + 1e-7).all(), (l2, want)
    assert set(gr2) == set(gra)
    for n in gr2:
        ref = (gra[n] + grb[n]) / 2
        assert float((gr2[n] - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-8, n
    aux = model.roi_heads.last_aux
    assert len(aux["images"]) == 2 and aux["images"][1]["scores"].shape == (4, 53, K)


@pytest.mark.parametrize("R,K,n_gt,H,W", [(1, 20, 1, 96, 128), (7, 3, 2, 96, 128), (37, 1, 1, 80, 112), (23, 20, 5, 112, 96)])
def test_small_and_degenerate_view_sets_against_the_oracle(R, K, n_gt, H, W):
    """The ragged end of the input range (the reference's loader can hand over any of these): ONE proposal per view (top-k =
    max(int(R * 0.1), 1) = 1, every softmax over proposals is over a single row), fewer proposals than ROIs a wave holds, a single
    class (K = 1: 2-column refinement heads, 4-column box deltas), more gt classes than the usual 1-2, non-square small maps.  Losses
    1e-4 against the CPU oracle, pseudo labels and proposal labels bit exact, every head gradient within 2e-3 of its scale (the
    bound of the full-size fp32 test).  Backbone gradients: with a handful of ROIs ONE pooling argmax that lands on the other of two
    pixels tied to the last ulp (the CPU convolution and the MFMA one sum in different orders) is visible — measured at R = 1:
    conv5_3's weight gradient is off by 1.3e-2 in exactly one output channel and <= 1.2e-4 in the other 511, its bias gradient
    (blind to WHERE a gradient lands) by 1.1e-4 — so they are held to 2e-2 in relative L2 and conv5_3's bias to 2e-3."""
    from sos_wsod_amd.events import EventStorage
    dan = (256, 256)
    tag = f"edge{R}_{K}_{n_gt}"
    P = O.make_params(K, dan, tag="p" + tag, head_scale=20.0)
    views, gt = O.make_views(H, W, R, n_gt=min(n_gt, K), K=K, scale2=1.25, tag="v" + tag)
    masks = O.make_masks(R, dan, tag="m" + tag)
    ol, oaux, ograd = O.oicr_plus_iteration(P, views, gt, masks, K=K, want_grads=True)
    model = build_model(K, dan, torch.float32)
    load_params(model, P)
    model.train()
    model.roi_heads.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
    with EventStorage(0):
        losses = model(to_batched_inputs(views, gt))
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    assert set(losses) == set(ol)
    for k, v in losses.items():
        assert np.isfinite(v.item()) and abs(v.item() - ol[k]) <= 1e-4 * abs(ol[k]) + 1e-7, (k, v.item(), ol[k])
    aux = model.roi_heads.last_aux
    for k in range(4):
        r, o = aux["rounds"][k], oaux["rounds"][k]
        n = int(r["pgt_count"].item())
        assert n == len(o["pgt"]["index"]) >= 1
        assert np.array_equal(r["pgt_index"][:n].cpu().numpy(), o["pgt"]["index"])
        assert np.array_equal(r["pgt_class"][:n].cpu().numpy(), o["pgt"]["classes"])
        assert np.array_equal(r["lab_class"].cpu().numpy(), o["labels"]["gt_classes"])
        assert np.array_equal(r["lab_index"].cpu().numpy(), o["labels"]["gt_index"])
    sd = dict(model.named_parameters())
    checked, worst = 0, ("", 0.0)
    for name, g in ograd.items():
        if name in sd and sd[name].grad is not None:
            got = sd[name].grad.cpu().numpy()
            scale = np.abs(g).max()
            if name.startswith("backbone.") and name != "backbone.plain5.0.conv3.bias":
                err = float(np.linalg.norm(got - g) / (np.linalg.norm(g) + 1e-30))
                assert err <= 2e-2, (name, err)
            elif name == "roi_heads.box_predictor.det.bias":
                # a constant added to a column of the detection stream cancels in the softmax over proposals: the true gradient is 0 and
                # both sides hold rounding noise, measured against the scale of the weight gradient next to it
                ref_scale = np.abs(ograd["roi_heads.box_predictor.det.weight"]).max()
                err = float(max(np.abs(got).max(), np.abs(g).max()) / (ref_scale + 1e-30)) if ref_scale > 0 else 0.0
                assert err <= 1e-4, (name, err)
            else:
                err = float(np.abs(got - g).max() / (scale + 1e-30)) if scale > 1e-6 else 0.0
                assert np.abs(got - g).max() <= 2e-3 * scale + 1e-7, (name, err)
            worst = max(worst, (name, err), key=lambda x: x[1])
            checked += 1
    assert checked >= 20
    print(f"R={R} K={K} gt={list(gt)}: worst gradient error {worst[1]:.1e} ({worst[0]})")


def test_trainer_skips_images_without_labels_and_iter_size_accumulates():
    """train_net_multi.py:121-127 (an image whose `instances1` is empty is skipped: the next one from the loader is taken) and
    :146-168 (ITER_SIZE: losses / ITER_SIZE, the optimizer steps when iter % ITER_SIZE == 0 — so, as written there, at iteration 0
    already and then after every second one): a loader that yields [unlabelled, A, unlabelled, unlabelled, B, C] through
    Trainer(iter_size=2) ends where a replica ends that is handed A, B, C."""
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.structures import Instances, Boxes
    from sos_wsod_amd.trainer import Trainer
    K, dan = 20, (256, 256)
    P = O.make_params(K, dan, tag="pskip", head_scale=20.0)

    def item(tag, labelled=True):
        views, gt = O.make_views(96, 128, 40, n_gt=2, K=K, scale2=1.25, tag=tag)
        d = to_batched_inputs(views, gt)
        if not labelled:
            for n in ["1", "1_flip", "2", "2_flip"]:
                h, w = views[0]["image"].shape[1:]
                t = Instances((h, w)); t.gt_boxes = Boxes(torch.zeros(0, 4)); t.gt_classes = torch.zeros(0, dtype=torch.int64)
                d[0]["instances" + n] = t
        return d

    def fresh():
        m = build_model(K, dan, torch.float32); load_params(m, P); m.train()
        m.roi_heads.seed = 77
        groups = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
                  for nm, p in m.named_parameters() if p.requires_grad]
        return m, HipSGD(groups, 1e-3, momentum=0.9)
    A, B, C = item("skipA"), item("skipB"), item("skipC")
    m1, o1 = fresh()
    tr = Trainer(m1, o1, iter_size=2, use_graph=False)
    tr.data_iter = iter([item("skipU0", False), A, item("skipU1", False), item("skipU2", False), B, C])
    w0 = m1.roi_heads.box_head.fc1.weight.detach().clone()
    tr.run_step()
    w1 = m1.roi_heads.box_head.fc1.weight.detach().clone()
    assert not torch.equal(w1, w0)                                              # iteration 0: 0 % 2 == 0, the reference steps
    tr.run_step()
    assert torch.equal(m1.roi_heads.box_head.fc1.weight.detach(), w1)          # iteration 1 only accumulates
    tr.run_step()
    assert not torch.equal(m1.roi_heads.box_head.fc1.weight.detach(), w1)      # iteration 2 applies B + C
    m2, o2 = fresh()
    tr2 = Trainer(m2, o2, iter_size=2, use_graph=False)
    tr2.run_step(A); tr2.run_step(B); tr2.run_step(C)
    torch.cuda.synchronize()
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1.detach(), p2.detach()), n1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tta_flip_pairs_as_one_batch_equal_single_view_calls(dtype):
    """GeneralizedRCNNWithTTAAVG(batch_views=True) runs a scale and its flip as one batch of two: the view-averaged score / box matrices
    equal the one-view-per-call form (fp32: 1e-5; bf16: the batch-2 convolutions may take another kernel variant, 2e-2) and the
    merged detections agree"""
    from sos_wsod_amd.structures import Boxes, Instances
    from sos_wsod_amd.tta import DeviceTTAMapper, GeneralizedRCNNWithTTAAVG
    K, dan = 20, (256, 256)
    P = O.make_params(K, dan, tag="pttab", head_scale=20.0)
    model = build_model(K, dan, dtype); load_params(model, P); model.eval()
    g = torch.Generator().manual_seed(4)
    H, W, R = 120, 160, 150
    x1 = torch.rand(R, generator=g) * (W - 32); y1 = torch.rand(R, generator=g) * (H - 32)
    b = torch.stack([x1, y1, x1 + 16 + torch.rand(R, generator=g) * (W - x1 - 16), y1 + 16 + torch.rand(R, generator=g) * (H - y1 - 16)], 1)
    p = Instances((H, W)); p.proposal_boxes = Boxes(b.cuda()); p.objectness_logits = torch.rand(R, generator=g).cuda()
    inp = {"image": torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).cuda(), "proposals": p, "height": H, "width": W}
    mapper = DeviceTTAMapper(min_sizes=(96, 128, 176), max_size=400, flip=True)
    outs = {}
    for mode in (True, False):
        tta = GeneralizedRCNNWithTTAAVG(model, mapper, batch_views=mode)
        det = tta([inp])[0]["instances"]
        outs[mode] = (tta.last_avg[0].clone(), tta.last_avg[1].clone(), det)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, c in zip(outs[True][:2], outs[False][:2]):
        assert float((a - c).abs().max()) <= tol * float(c.abs().max())
    if dtype == torch.float32:
        da, dc = outs[True][2], outs[False][2]
        assert len(da) == len(dc) and torch.equal(da.pred_classes, dc.pred_classes)
        assert float((da.pred_boxes.tensor - dc.pred_boxes.tensor).abs().max()) <= 1e-2
