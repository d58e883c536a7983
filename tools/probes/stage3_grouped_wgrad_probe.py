"""Would the grouped 256x256 weight-gradient launch (sw_conv3x3_wgrad_grouped, one launch for a list of problems + ONE fold over
all their slabs) pay for the Stage-3 detector's repeated uses of a 3x3 weight?  Sets: the RPN head's convolution (5 levels x
batch 2 and batch 1), an FPN output convolution (2 uses), res4 / res5 conv2 (2 uses).  Against one sw_conv3x3_wgrad per use."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
from sos_wsod_amd.backbone_vgg import _wgrad_grouped_target, _wgrad_grouped_splits
dt, dev = torch.bfloat16, "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
LV = [(200, 304), (100, 152), (50, 76), (25, 38), (13, 19)]
sets = {"rpn conv 256 (5 levels x n=2,1)": [(n, h, w, 256) for n in (2, 1) for h, w in LV],
        "fpn out p2 256 (n=2,1)": [(2, 200, 304, 256), (1, 200, 304, 256)], "fpn out p4 256 (n=2,1)": [(2, 50, 76, 256), (1, 50, 76, 256)],
        "res4 conv2 256 (n=2,1)": [(2, 50, 76, 256), (1, 50, 76, 256)], "res5 conv2 512 (n=2,1)": [(2, 25, 38, 512), (1, 25, 38, 512)],
        "res3 conv2 128 (n=2,1)": [(2, 100, 152, 128), (1, 100, 152, 128)]}
for name, probs in sets.items():
    C = probs[0][3]
    xs = [(torch.randn(n, h, w, C, device=dev).to(dt), torch.randn(n, h, w, C, device=dev).to(dt)) for n, h, w, _ in probs]
    dw = torch.empty(C, C, 3, 3, device=dev); dw2 = torch.empty_like(dw)
    def single():
        for i, (x, dy) in enumerate(xs):
            n, H, W, _ = x.shape
            tiles = ((C + 127) // 128) * ((9 * C + 127) // 128)
            ops.conv3x3_wgrad(x, dy, dw, 1, splitk=max(1, min(32, 512 // tiles, max(1, n * H * W // 1024))), accumulate=i > 0)
    us1 = t(single)
    res = []
    for T in (0, 16, 24, 32, 48, 64, 96):
        shapes = [(x.shape[0] * x.shape[1] * x.shape[2], C, 9 * C) for x, _ in xs]
        target = _wgrad_grouped_target(shapes, 64) if T == 0 else T
        splits = [_wgrad_grouped_splits(s[0], 64, target) for s in shapes]
        nsl = [ops.conv3x3_wgrad_nslab(x, C, sp) for (x, _), sp in zip(xs, splits)]
        ws = torch.empty(sum(nsl), C * 9 * C, device=dev)
        offs = [sum(nsl[:i]) for i in range(len(nsl))]
        def grouped():
            ops.conv3x3_wgrad_grouped([(x, dy, ws[o:], 1, sp) for (x, dy), o, sp in zip(xs, offs, splits)])
            ops.conv3x3_wgrad_fold(ws, sum(nsl), dw2)
        try:
            us2 = t(grouped)
        except Exception as e:
            res.append(f"T={target}: {type(e).__name__}"); continue
        res.append(f"T={target}:{us2:6.1f} ({sum(nsl)} slabs)")
    single(); grouped(); torch.cuda.synchronize()
    err = float((dw - dw2).abs().max() / dw.abs().max())
    print(f"{name:34s} one per use {us1:7.1f} us | grouped " + "  ".join(res) + f" | rel diff {err:.1e}", flush=True)
