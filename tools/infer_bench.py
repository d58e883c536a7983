"""Inference throughput (not a BASELINE metric): single 512x512 view, and the reference's TTA (6 scales x h-flip) of a 375x500 image,
2000 proposals.  Shows where the test-time path spends its time."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from sos_wsod_amd.structures import Boxes, Instances
from sos_wsod_amd.tta import GeneralizedRCNNWithTTAAVG
dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16).eval()
def make(H, W, R=2000, seed=0):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.rand(R, generator=g) * (W - 32); y1 = torch.rand(R, generator=g) * (H - 32)
    b = torch.stack([x1, y1, x1 + 16 + torch.rand(R, generator=g) * (W - x1 - 16), y1 + 16 + torch.rand(R, generator=g) * (H - y1 - 16)], 1)
    p = Instances((H, W)); p.proposal_boxes = Boxes(b.to(dev)); p.objectness_logits = torch.rand(R, generator=g).to(dev)
    return {"image": torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev), "proposals": p, "height": H, "width": W}
def t(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    x = make(512, 512)
    print(f"single view 512x512, R=2000: {t(lambda: model.inference([x]), 20):.2f} ms per image")
    tta = GeneralizedRCNNWithTTAAVG(model)
    y = make(375, 500)
    print(f"TTA 12 views of a 375x500 image, R=2000: {t(lambda: tta([y]), 5):.1f} ms per image")
