"""where the host time of Stage-1 inference goes (single view and the 12-view TTA): cProfile of steady-state calls + the thread's CPU time"""
import cProfile, pstats, io, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
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
with torch.no_grad():
    x = make(512, 512); tta = GeneralizedRCNNWithTTAAVG(model); y = make(375, 500)
    for name, fn, n in (("single view", lambda: model.inference([x]), 30), ("TTA", lambda: tta([y]), 6)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.thread_time()
        for _ in range(n): fn()
        c1 = time.thread_time(); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"{name}: {(t1 - t0) / n * 1e3:.2f} ms per call, issuing thread's CPU time {(c1 - c0) / n * 1e3:.2f} ms")
        pr = cProfile.Profile(); pr.enable()
        for _ in range(n): fn()
        pr.disable(); torch.cuda.synchronize()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:2600])
