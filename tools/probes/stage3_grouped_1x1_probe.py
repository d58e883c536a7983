"""1x1 weight gradients of one bottleneck block (conv1, conv3, shortcut) for the student's two passes: one sw_gemm per weight and
pass (K-split + fold each) against ONE sw_gemm_kk_grouped launch over all of them + ONE sw_splitk_fold_multi."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
from sos_wsod_amd.backbone_vgg import _wgrad_grouped_target, _wgrad_grouped_splits
dt, dev = torch.bfloat16, "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
CAND = (4, 6, 8, 12, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160)
# (name, pixels per image, cin, mid, cout, has shortcut)
for name, P1, cin, mid, cout, sc in [("res3.0", 15200, 256, 128, 512, True), ("res3.1", 15200, 512, 128, 512, False), ("res4.0", 3800, 512, 256, 1024, True),
                                     ("res4.1", 3800, 1024, 256, 1024, False), ("res5.0", 950, 1024, 512, 2048, True), ("res5.1", 950, 2048, 512, 2048, False)]:
    uses = []
    for n in (2, 1):
        P = n * P1
        x2 = torch.randn(P, cin, device=dev).to(dt); h2 = torch.randn(P, mid, device=dev).to(dt)
        gs = torch.randn(P, cout, device=dev).to(dt); dh1 = torch.randn(P, mid, device=dev).to(dt)
        uses.append([(gs, h2), (dh1, x2)] + ([(gs, x2)] if sc else []))
    nw = len(uses[0])
    dws = [torch.empty(a.shape[1], b.shape[1], device=dev) for a, b in uses[0]]
    def single():
        for u, use in enumerate(uses):
            for (a, b), dw in zip(use, dws):
                P, ld = a.shape; D = b.shape[1]
                tiles = ((ld + 127) // 128) * ((D + 127) // 128)
                ep = ops.make_epilogue(out_dtype=torch.float32, residual=dw) if u else None
                ops.gemm(a, b, dw, ld, D, P, a_kstrided=True, b_kstrided=True, splitk=max(1, min(64, 512 // tiles, P // 512)), ep=ep)
    us1 = t(single)
    res = []
    for T in (0, 4, 8, 16, 32):
        shapes = [(a.shape[0], a.shape[1], b.shape[1]) for use in uses for a, b in use]
        target = _wgrad_grouped_target(shapes, 64, candidates=CAND) if T == 0 else T
        probs, folds = [], []
        for w in range(nw):
            ns = [_wgrad_grouped_splits(use[w][0].shape[0], 64, target) for use in uses]
            nsl = [ops.gemm_kk_nslab(dt, use[w][0].shape[0], s) for use, s in zip(uses, ns)]
            M, N = uses[0][w][0].shape[1], uses[0][w][1].shape[1]
            ws = torch.empty(sum(nsl), M * N, device=dev)
            off = 0
            for use, s, k in zip(uses, ns, nsl):
                probs.append((use[w][0], use[w][1], ws[off:], s)); off += k
            folds.append((ws, sum(nsl), dws[w], None, False))
        def grouped():
            ops.gemm_kk_grouped(probs); ops.splitk_fold_multi(folds)
        us2 = t(grouped)
        res.append(f"T={target}:{us2:6.1f}")
    print(f"{name}: {2 * nw} launches + folds {us1:7.1f} us | grouped " + "  ".join(res), flush=True)
