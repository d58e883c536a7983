"""Weight-gradient GEMMs of the Stage-3 detector's 1x1 convolutions / fc layers (dW (out, in) = dY^T (out, P) @ X (P, in), K = P
pixels): the K-strided form `frcnn._LinearFn.backward` launches, at several split-K factors, against explicit transposes + the
K-contiguous form.  Shapes: ResNet-50 + FPN laterals + box head at 800 x 1216, 2 images."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops

dt, dev = torch.bfloat16, "cuda"
nimg = int(os.environ.get("NIMG", 2))
P4, P8, P16, P32 = [nimg * (800 // s) * (1216 // s) for s in (4, 8, 16, 32)]
shapes = [("lat2 256<-256", 256, 256, P4), ("res3.c1 128<-256", 128, 256, P8), ("res3.c3 512<-128", 512, 128, P8),
          ("res3.sc 512<-256", 512, 256, P8), ("res3.c1 128<-512", 128, 512, P8), ("lat3 256<-512", 256, 512, P8),
          ("res4.c1 256<-1024", 256, 1024, P16), ("res4.c3 1024<-256", 1024, 256, P16), ("res4.sc 1024<-512", 1024, 512, P16),
          ("lat4 256<-1024", 256, 1024, P16), ("res5.c1 512<-2048", 512, 2048, P32), ("res5.c3 2048<-512", 2048, 512, P32),
          ("lat5 256<-2048", 256, 2048, P32), ("rpn.obj 8<-256", 8, 256, P4), ("rpn.delta 16<-256", 16, 256, P4),
          ("fc1 1024<-12544", 1024, 12544, 512 * nimg), ("fc2 1024<-1024", 1024, 1024, 512 * nimg), ("pred 104<-1024", 104, 1024, 512 * nimg)]


def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3


for name, M, N, P in shapes:
    Pp = (P + 7) // 8 * 8
    gs = torch.randn(Pp, M, device=dev).to(dt); x = torch.randn(Pp, N, device=dev).to(dt)
    dw = torch.zeros(M, N, device=dev)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    cur = max(1, min(64, 512 // tiles, Pp // 512))
    res = []
    for sk in sorted({cur, 1, 4, 16, 64, 128, 256}):
        if sk > max(1, Pp // 256): continue
        try:
            us = t(lambda: ops.gemm(gs, x, dw, M, N, Pp, a_kstrided=True, b_kstrided=True, splitk=sk))
            res.append(f"{'*' if sk == cur else ''}sk{sk}:{us:.0f}")
        except Exception as e:
            res.append(f"sk{sk}:ERR")
    ref = dw.clone()
    gT = torch.empty(M, Pp, device=dev, dtype=dt); xT = torch.empty(N, Pp, device=dev, dtype=dt)
    tt = t(lambda: (ops.transpose_2d(gs, gT, Pp, M), ops.transpose_2d(x, xT, Pp, N)))
    nt = []
    for sk in (1, 4, 16, 64):
        if sk > max(1, Pp // 256): continue
        us = t(lambda: ops.gemm(gT, xT, dw, M, N, Pp, splitk=sk))
        nt.append(f"sk{sk}:{us:.0f}")
    err = float((dw - ref).abs().max() / ref.abs().max())
    gf = 2.0 * M * N * Pp / 1e9
    print(f"{name:20s} P={Pp:6d} {gf:7.1f} GF | TN " + " ".join(res) + f" | transposes {tt:.0f} + NT " + " ".join(nt) + f" | rel diff {err:.1e}", flush=True)
