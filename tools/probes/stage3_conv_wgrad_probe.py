"""3x3 weight gradients of the Stage-3 detector's shapes (sw_conv3x3_wgrad: K-split gather GEMM + fold) for several split counts.
usage: stage3_conv_wgrad_probe.py   (GPU only)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
shapes = [("res3.conv2", 100, 152, 128), ("res4.conv2", 50, 76, 256), ("res5.conv2", 25, 38, 512), ("fpn/rpn p2", 200, 304, 256),
          ("fpn/rpn p3", 100, 152, 256), ("fpn/rpn p4", 50, 76, 256), ("fpn/rpn p5", 25, 38, 256), ("rpn p6", 13, 19, 256)]
for name, H, W, C in shapes:
    for n in (1, 2):
        x = torch.randn(n, H, W, C, device=dev).to(dt); dy = torch.randn(n, H, W, C, device=dev).to(dt)
        dw = torch.empty(C, C, 3, 3, device=dev)
        tiles = ((C + 127) // 128) * ((9 * C + 127) // 128)
        npix = n * H * W
        cur = max(1, min(32, 512 // tiles, max(1, npix // 1024)))
        line = f"{name:11s} n={n} tiles={tiles:3d} now sk={cur:2d}:"
        for sk in sorted(set([1, 2, 3, 4, 6, 8, 12, 16, 24, 32, cur])):
            if sk > max(1, npix // 128): continue
            try:
                us = t(lambda: ops.conv3x3_wgrad(x, dy, dw, 1, splitk=sk))
            except Exception as e:
                line += f"  {sk}:err"; continue
            line += f"  {sk}:{us:5.1f}{'*' if sk == cur else ''}"
        print(line, flush=True)
