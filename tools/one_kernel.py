"""Run ONE hot-path kernel shape a few times (for rocprofv3 --pmc passes).
usage: one_kernel.py conv5_3|conv3_2|wgrad*|wgrad_grouped|roi_fwd|roi_bwd|fc6_fwd|fc6_dgrad|fc6_wgrad [repeats]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
which = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt, dev = torch.bfloat16, "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
if which == "conv5_3":
    x = rnd(2, 63, 63, 512); wk = rnd(512, 9, 512); b = torch.zeros(512, device=dev); out = torch.empty(2, 63, 63, 512, device=dev, dtype=dt)
    ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
    f = lambda: ops.conv3x3(x, wk, out, 2, ep)
elif which == "conv3_2":
    x = rnd(2, 128, 128, 256); wk = rnd(256, 9, 256); b = torch.zeros(256, device=dev); out = torch.empty(2, 128, 128, 256, device=dev, dtype=dt)
    ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
    f = lambda: ops.conv3x3(x, wk, out, 1, ep)
elif which.startswith("wgrad") and which != "wgrad_grouped":
    n_, H_, W_, cin, cout, dil, sk = {"wgrad5": (2, 63, 63, 512, 512, 2, 3), "wgrad3": (2, 128, 128, 256, 256, 1, 8), "wgrad3s16": (2, 128, 128, 256, 256, 1, 16), "wgrad4": (2, 64, 64, 512, 512, 1, 3), "wgrad4s8": (2, 64, 64, 512, 512, 1, 8)}[which]
    x = rnd(n_, H_, W_, cin); dy = rnd(n_, H_, W_, cout); dw = torch.empty(cout, cin, 3, 3, device=dev); ws = torch.empty(cout * 9 * cin, device=dev)
    f = lambda: ops.conv3x3_wgrad(x, dy, dw, dil, splitk=sk, workspace=ws)
elif which == "wgrad_grouped":          # every conv weight gradient of one backward pass (9 layers x 2 view batches), one launch
    from sos_wsod_amd.backbone_vgg import _wgrad_grouped_splits, _wgrad_grouped_target, _wgrad_direct_splits
    L = [(128, 128, 128, 256, 1), (128, 128, 256, 256, 1), (128, 128, 256, 256, 1), (64, 64, 256, 512, 1), (64, 64, 512, 512, 1),
         (64, 64, 512, 512, 1), (63, 63, 512, 512, 2), (63, 63, 512, 512, 2), (63, 63, 512, 512, 2)]
    T = _wgrad_grouped_target([(2 * H * W, co, 9 * ci) for H, W, ci, co, _ in L for _v in range(2)], 64)
    direct = os.environ.get("SW_WGRAD_DIRECT", "1") != "0"         # the direct kernel's own split plan (what the backbone passes)
    plan = iter(_wgrad_direct_splits([(2, H, W, ci, co, dil) for H, W, ci, co, dil in L for _v in range(2)]))
    probs = []
    for H, W, ci, co, dil in L:
        for _v in range(2):
            x, dy = rnd(2, H, W, ci), rnd(2, H, W, co)
            ns = next(plan) if direct else _wgrad_grouped_splits(2 * H * W, 64, T)
            probs.append((x, dy, torch.empty(ops.conv3x3_wgrad_nslab(x, co, ns), co * 9 * ci, device=dev), dil, ns))
    f = lambda: ops.conv3x3_wgrad_grouped(probs)
elif which in ("roi_fwd", "roi_bwd"):   # one ROIPool call of the step: 2 x 2000 ROIs (view + flipped view) on a 2 x 63 x 63 x 512 map
    import bench
    R = 2000
    d = bench.make_inputs(torch.device(dev), 3)[0]
    bx = torch.cat([d["proposals1"].proposal_boxes.tensor, d["proposals1_flip"].proposal_boxes.tensor], 0)
    rois = torch.cat([(torch.arange(2 * R, device=dev) >= R).float()[:, None], bx], 1).contiguous()
    feat = torch.relu(rnd(2, 63, 63, 512))
    out = torch.empty(2 * R, 25088 + 64, device=dev, dtype=dt)[:, :25088]
    arg = torch.empty(2 * R, 25088 + 64, device=dev, dtype=torch.int16)[:, :25088]
    obj = torch.rand(2 * R, device=dev)
    ops.roi_pool_fwd(feat, rois, out, arg, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0)
    if which == "roi_fwd":
        f = lambda: ops.roi_pool_fwd(feat, rois, out, arg, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0)
    else:
        dout = rnd(2 * R, 25088 + 64)[:, :25088]; df = torch.empty_like(feat); amax = ops.absmax(dout)
        f = lambda: ops.roi_pool_bwd(dout, arg, rois, df, 7, 7, row_scale=obj, row_scale_add=1.0, relu_ref=feat, dout_absmax=amax)
else:
    # the box head's matrices as the step holds them: row pitches padded by 128 elements (roi_heads_oicrplus._padded)
    M, D0, D1 = 8000, 25088, 4096
    pad = lambda r, c: rnd(r, c + 128)[:, :c]
    X = rnd(M, D0); W1 = pad(D1, D0); dZ = pad(M, D1)
    if which == "fc6_fwd":
        Y = torch.empty(M, D1 + 128, device=dev, dtype=dt)[:, :D1]; f = lambda: ops.gemm(X, W1, Y, M, D1, D0)
    elif which == "fc6_dgrad":       # NT on the transposed weight copy
        W1T = pad(D0, D1); dX = torch.empty(M, D0, device=dev, dtype=dt)
        f = lambda: ops.gemm(dZ, W1T, dX, M, D0, D1, ep=ops.make_epilogue(out_dtype=dt))
    else:                            # weight gradient as the step runs it: dZ^T (transpose kernel) x pooled, A K-contiguous
        dW = torch.empty(D1, D0, device=dev); dZT = torch.empty(D1, M + 64, device=dev, dtype=dt)[:, :M]
        def f():
            ops.transpose_2d(dZ, dZT, M, D1)
            ops.gemm(dZT, X, dW, D1, D0, M, b_kstrided=True)
for _ in range(n): f()
torch.cuda.synchronize()
