"""sw_oicr_mine_label alone (4 refinement rounds, as the step launches it): R proposals, G image-level classes, top 10 % per class"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
dev = "cuda"
def run(R, K, G, rounds=4):
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(R, generator=g) * 480; y1 = torch.rand(R, generator=g) * 480
    boxes = torch.stack([x1, y1, x1 + 24 + torch.rand(R, generator=g) * 200, y1 + 24 + torch.rand(R, generator=g) * 200], 1).to(dev)
    scores = torch.softmax(torch.randn(rounds, R, K + 1, generator=g), -1).to(dev)
    gt = torch.randperm(K, generator=g)[:G].sort().values.int().to(dev)
    top_k = max(int(R * 0.10), 1); n_slots = top_k * G
    i32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.int32)
    f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    out = (i32(rounds, R), f32(rounds, R), i32(rounds, R), i32(rounds), i32(rounds, n_slots), i32(rounds, n_slots), f32(rounds, n_slots))
    ws = torch.empty(ops.mine_workspace_bytes(R, top_k, G, rounds), device=dev, dtype=torch.uint8)
    fn = lambda: ops.oicr_mine_label(scores, gt, boxes, K, top_k, 0.05, 0.01, 0.5, 0.6, *out, ws)
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(20)]; b.record(); torch.cuda.synchronize()
    print(f"mine_label R={R} K={K} G={G}: {a.elapsed_time(b) / 20 * 1e3:.0f} us  (kept {out[3].tolist()})")
run(2000, 20, 2); run(2000, 20, 5); run(4000, 80, 8); run(10000, 80, 18)
