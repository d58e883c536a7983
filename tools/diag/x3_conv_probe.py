"""fp32 mode: bf16x3 convolutions (backbone.fp32x3) against the exact-f32 MFMA path: outputs and every gradient of one forward/backward."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_model
torch.manual_seed(0)
m = build_model(20, (256, 256), torch.float32).cuda().train()
bb = m.backbone
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(2, H, W, 8, device="cuda"); x[..., 3:] = 0
res = {}
for mode in (False, True):
    bb.fp32x3 = mode
    for p in bb.parameters(): p.grad = None
    outs = bb.forward_views([x]) if hasattr(bb, "forward_views") else None
    y = outs[0] if isinstance(outs, (list, tuple)) else outs
    g = torch.randn(y.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    y.backward(g)
    torch.cuda.synchronize()
    res[mode] = (y.detach().clone(), {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None})
y0, g0 = res[False]; y1, g1 = res[True]
print("out: max rel err", float((y1 - y0).abs().max() / y0.abs().max()), "finite", bool(torch.isfinite(y1).all()))
for n in g0:
    e = float((g1[n] - g0[n]).abs().max() / (g0[n].abs().max() + 1e-30))
    print(f"{n:28s} {e:.2e} finite {bool(torch.isfinite(g1[n]).all())}")
