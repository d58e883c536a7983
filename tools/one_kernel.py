#!/usr/bin/env python
"""Run ONE hot-path GEMM/conv shape a few times (for rocprofv3 --pmc passes).  usage: one_kernel.py conv5_3|fc6_fwd|fc6_dgrad|fc6_wgrad"""
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
elif which.startswith("wgrad"):
    n_, H_, W_, cin, cout, dil, sk = {"wgrad5": (2, 63, 63, 512, 512, 2, 3), "wgrad3": (2, 128, 128, 256, 256, 1, 8), "wgrad3s16": (2, 128, 128, 256, 256, 1, 16), "wgrad4": (2, 64, 64, 512, 512, 1, 3), "wgrad4s8": (2, 64, 64, 512, 512, 1, 8)}[which]
    x = rnd(n_, H_, W_, cin); dy = rnd(n_, H_, W_, cout); dw = torch.empty(cout, cin, 3, 3, device=dev); ws = torch.empty(cout * 9 * cin, device=dev)
    f = lambda: ops.conv3x3_wgrad(x, dy, dw, dil, splitk=sk, workspace=ws)
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
    else:
        dW = torch.empty(D1, D0, device=dev); f = lambda: ops.gemm(dZ, X, dW, D1, D0, M, a_kstrided=True, b_kstrided=True)
for _ in range(n): f()
torch.cuda.synchronize()
