"""Randomised parity sweep of the integer / index kernels against the oracle (development aid, not a test: the pytest suite
holds the fixed cases).  ROIPool forward over random map sizes (plane / band / narrow-slab / gather forms), dtypes, ROI sets with
degenerate and out-of-image boxes: bins, argmax and values bit-exact vs oracle/roipool_oracle.c, the backward within the
storage type's rounding; mining + labelling over random
proposal counts, class counts, tied scores: kept indices / classes / scores / labels bit-exact vs oracle.get_pgt_mist +
label_proposals; the inference post-processing (per-class NMS, merge, top-k) vs the oracle's NMS.   usage: fuzz_parity.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
from oracle import oicr_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_roi = n_mine = 0


def fuzz_roi():
    H, W = int(rng.randint(4, 181)), int(rng.randint(4, 221))
    C = int(rng.choice([8, 16, 24, 40]))
    n = int(rng.randint(1, 4))
    R = int(rng.randint(1, 400))
    dtype = torch.bfloat16 if rng.rand() < 0.7 else torch.float32
    feat = torch.from_numpy(rng.randn(n, C, H, W).astype(np.float32))
    if rng.rand() < 0.5:
        feat = feat.relu()                                        # ties at zero
    feat = feat.to(dtype).float()
    x1 = rng.rand(R) * W * 8 - 20; y1 = rng.rand(R) * H * 8 - 20
    bw = rng.rand(R) ** 2 * W * 8 * 1.3; bh = rng.rand(R) ** 2 * H * 8 * 1.3
    boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
    k = min(R, 6)
    boxes[:k] = np.array([[0, 0, 1e4, 1e4], [-900, -700, 8 * W + 500, 8 * H + 900], [3.9, 3.9, 4.1, 4.1], [100, 100, 90, 90],
                          [8 * W - 1, 8 * H - 1, 8 * W + 40, 8 * H + 40], [0, 8 * H - 9, 8 * W, 8 * H - 1]], np.float32)[:k]
    rois = np.concatenate([rng.randint(0, n, (R, 1)).astype(np.float32), boxes], 1).astype(np.float32)
    ref_out, ref_arg = O.roi_pool_fwd(feat.numpy(), rois, 1.0 / 8)
    f = feat.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    adt = ops.roi_argmax_dtype(H, W) if rng.rand() < 0.7 else torch.int32
    out = torch.empty(R, C * 49, device="cuda", dtype=dtype); arg = torch.empty(R, C * 49, device="cuda", dtype=adt)
    ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7)
    got_arg = ops.argmax_to_int32(arg).cpu().numpy().reshape(ref_arg.shape)
    assert np.array_equal(got_arg, ref_arg), ("argmax", H, W, C, n, R, dtype, adt)
    assert torch.equal(out.cpu().float().reshape(ref_out.shape), torch.from_numpy(ref_out).to(dtype).float()), ("values", H, W, C, n, R, dtype)
    # backward (fixed-point scatter) on the same argmax: against the C oracle's float scatter-add
    dout = torch.from_numpy(rng.randn(R, C * 49).astype(np.float32)).to(dtype)
    ref_d = O.roi_pool_bwd(dout.float().numpy().reshape(R, C, 7, 7), ref_arg, rois, (n, C, H, W))
    dfeat = torch.full((n, H, W, C), float("nan"), device="cuda", dtype=dtype)
    ops.roi_pool_bwd(dout.cuda(), arg, torch.from_numpy(rois).cuda(), dfeat, 7, 7)
    got_d = dfeat.permute(0, 3, 1, 2).float().cpu().numpy()
    tol = (2e-2 if dtype == torch.bfloat16 else 2e-5) * max(np.abs(ref_d).max(), 1e-6)
    assert np.isfinite(got_d).all() and np.abs(got_d - ref_d).max() <= tol, ("bwd", H, W, C, n, R, dtype, np.abs(got_d - ref_d).max(), tol)


def fuzz_mine():
    R = int(rng.randint(1, 700)); K = int(rng.choice([3, 20, 80])); G = int(rng.randint(1, min(K, 9) + 1))
    views, _ = O.make_views(256, 320, R, n_gt=1, K=K, tag=f"fz{rng.randint(1 << 30)}")
    boxes = views[0]["boxes"]
    sc = rng.rand(R, K + 1).astype(np.float32)
    if rng.rand() < 0.5:
        sc = np.round(sc, 1)                                       # heavy ties (also with the 0.05 threshold region)
    if rng.rand() < 0.3:
        sc *= 0.06
    gt = np.sort(rng.choice(K, G, replace=False)).astype(np.int64)
    top_k = max(int(R * 0.10), 1)
    pgt = O.get_pgt_mist(sc[:, :K], boxes, gt)
    lab = O.label_proposals(pgt, boxes, K)
    dev = "cuda"
    i32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.int32)
    lab_c, lab_w, lab_i, cnt = i32(R), torch.empty(R, device=dev), i32(R), i32(1)
    pi, pc, ps = i32(top_k * G), i32(top_k * G), torch.empty(top_k * G, device=dev)
    ws = torch.empty(ops.mine_workspace_bytes(R, top_k, G), dtype=torch.uint8, device=dev)
    ops.oicr_mine_label(torch.from_numpy(sc).cuda(), torch.from_numpy(gt.astype(np.int32)).cuda(), torch.from_numpy(boxes).cuda(), K, top_k,
                        0.05, 0.01, 0.5, 0.6, lab_c, lab_w, lab_i, cnt, pi, pc, ps, ws)
    n = int(cnt.item())
    tag = (R, K, G)
    assert n == len(pgt["index"]), ("count", tag, n, len(pgt["index"]))
    assert np.array_equal(pi[:n].cpu().numpy(), pgt["index"]), ("index", tag)
    assert np.array_equal(pc[:n].cpu().numpy(), pgt["classes"]), ("classes", tag)
    assert np.array_equal(ps[:n].cpu().numpy(), pgt["scores"]), ("scores", tag)
    assert np.array_equal(lab_c.cpu().numpy(), lab["gt_classes"]), ("labels", tag)
    assert np.array_equal(lab_i.cpu().numpy(), lab["gt_index"]), ("gt_index", tag)
    assert np.array_equal(lab_w.cpu().numpy(), lab["gt_weights"]), ("weights", tag)


def fuzz_detect():
    """per-class NMS on class-offset clipped boxes + global top-k (fast_rcnn_oicr.py:86-148): detections and their order"""
    R = int(rng.randint(1, 500)); K = int(rng.choice([3, 20, 80]))
    sc = torch.softmax(torch.from_numpy(rng.randn(R, K + 1).astype(np.float32)) * float(rng.choice([1, 3, 8])), 1)
    if rng.rand() < 0.4:
        sc = torch.round(sc * 50) / 50                              # tied scores
    H, W = int(rng.randint(60, 400)), int(rng.randint(60, 400))
    ctr = torch.from_numpy(rng.rand(R, K, 2).astype(np.float32)) * torch.tensor([float(W), float(H)])
    wh = torch.from_numpy(rng.rand(R, K, 2).astype(np.float32)) * 120 + 4
    bx = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).reshape(R, 4 * K)
    thr, nms_thr, topk = float(rng.choice([1e-5, 0.02, 0.2])), float(rng.choice([0.3, 0.5])), int(rng.choice([5, 100]))
    s = sc[:, :-1].numpy(); pb = bx.numpy().reshape(R, K, 4).copy()
    pb[..., 0::2] = pb[..., 0::2].clip(0, W); pb[..., 1::2] = pb[..., 1::2].clip(0, H)
    r_idx, c_idx = np.nonzero(s > np.float32(thr))
    cnt, dboxes, dscores, dclasses, drows = ops.detect_postprocess(sc.cuda(), bx.cuda(), H, W, thr, nms_thr, topk)
    n = int(cnt.item())
    if len(r_idx) == 0:
        assert n == 0, ("detect empty", R, K)
        return
    bsel, ssel = pb[r_idx, c_idx], s[r_idx, c_idx]
    off = c_idx.astype(np.float32) * np.float32(bsel.max() + 1)
    keep = O.nms_keep((bsel + off[:, None]).astype(np.float32), ssel, nms_thr)[:topk]
    tag = (R, K, H, W, thr, nms_thr, topk)
    assert n == len(keep), ("detect count", tag, n, len(keep))
    assert np.array_equal(dclasses[:n].cpu().numpy(), c_idx[keep]) and np.array_equal(drows[:n].cpu().numpy(), r_idx[keep]), ("detect order", tag)
    assert np.array_equal(dscores[:n].cpu().numpy(), ssel[keep]) and np.array_equal(dboxes[:n].cpu().numpy(), bsel[keep]), ("detect values", tag)


n_det = 0
while time.time() < t_end:
    fuzz_roi(); n_roi += 1
    fuzz_mine(); n_mine += 1
    fuzz_detect(); n_det += 1
print(f"fuzz ok: {n_roi} ROIPool cases, {n_mine} mining cases, {n_det} detection post-processing cases")
