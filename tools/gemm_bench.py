"""Micro-benchmark of the MFMA GEMM / implicit-GEMM conv kernels on the hot path's shapes (GPU only).
usage: python tools/gemm_bench.py [filter]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops  # noqa: E402

dt = torch.bfloat16
dev = "cuda"
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


def check(C, ref, name):
    err = ((C.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
    return err


rows = []
M, D0, D1 = 8000, 25088, 4096
if "fc" in flt or not flt:
    X = rnd(M, D0); W1 = rnd(D1, D0); Y = torch.empty(M, D1, device=dev, dtype=dt)
    t = timeit(lambda: ops.gemm(X, W1, Y, M, D1, D0))
    ref = (X[:256].float() @ W1.float().t())
    rows.append(("fc6_fwd  NT 8000x4096x25088", t, 2.0 * M * D1 * D0, check(Y[:256], ref, "")))
    dZ = rnd(M, D1); dX = torch.empty(M, D0, device=dev, dtype=dt)
    t = timeit(lambda: ops.gemm(dZ, W1, dX, M, D0, D1, b_kstrided=True, ep=ops.make_epilogue(out_dtype=dt)))
    ref = dZ[:256].float() @ W1.float()
    rows.append(("fc6_dgrad NN 8000x25088x4096", t, 2.0 * M * D1 * D0, check(dX[:256], ref, "")))
    dW = torch.empty(D1, D0, device=dev, dtype=torch.float32)
    t = timeit(lambda: ops.gemm(dZ, X, dW, D1, D0, M, a_kstrided=True, b_kstrided=True))
    ref = dZ[:, :256].float().t() @ X.float()
    rows.append(("fc6_wgrad TN 4096x25088x8000", t, 2.0 * M * D1 * D0, check(dW[:256], ref, "")))
    del X, W1, Y, dZ, dX, dW
    H1 = rnd(M, D1); W2 = rnd(D1, D1); Y = torch.empty(M, D1, device=dev, dtype=dt)
    t = timeit(lambda: ops.gemm(H1, W2, Y, M, D1, D1))
    rows.append(("fc7_fwd  NT 8000x4096x4096", t, 2.0 * M * D1 * D1, check(Y[:256], H1[:256].float() @ W2.float().t(), "")))
if "conv" in flt or not flt:
    for name, n, H, W, cin, cout, dil in [("conv5_3 2x63x63 512->512 d2", 2, 63, 63, 512, 512, 2),
                                          ("conv4_2 2x64x64 512->512", 2, 64, 64, 512, 512, 1),
                                          ("conv3_2 2x128x128 256->256", 2, 128, 128, 256, 256, 1),
                                          ("conv2_2 2x256x256 128->128", 2, 256, 256, 128, 128, 1),
                                          ("conv1_2 2x512x512 64->64", 2, 512, 512, 64, 64, 1)]:
        x = rnd(n, H, W, cin); w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        wk = torch.empty(cout, 9, cin, device=dev, dtype=dt); ops.conv_weight_prep(w, wk, 0, cin)
        b = torch.zeros(cout, device=dev); out = torch.empty(n, H, W, cout, device=dev, dtype=dt)
        ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
        t = timeit(lambda: ops.conv3x3(x, wk, out, dil, ep))
        ref = torch.relu(torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).float(), w.to(dt).float(), None, padding=dil, dilation=dil))
        err = check(out[:1].permute(0, 3, 1, 2), ref, "")
        fl = 2.0 * n * H * W * cout * 9 * cin
        rows.append((name + " fwd", t, fl, err))
        dy = rnd(n, H, W, cout); dw = torch.empty(cout, cin, 3, 3, device=dev)
        npix = n * H * W
        tiles = ((cout + 127) // 128) * ((9 * cin + 127) // 128)
        for splitk in [int(v) for v in os.environ.get("SK", "0").split(",")]:
            if splitk == 0:            # the backbone's policy (backbone_vgg.py)
                splitk = max(1, min(8, (300 + tiles - 1) // tiles, max(1, npix // 1024)))
            t = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, dil, splitk=splitk))
            rows.append((name + f" wgrad sk{splitk}", t, fl, float("nan")))
for name, t, fl, err in rows:
    print(f"{name:42s} {t * 1e3:9.1f} us  {fl / t / 1e9:8.1f} TFLOP/s  relerr {err:.1e}")
