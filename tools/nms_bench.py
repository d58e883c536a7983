"""sw_detect_postprocess in its RPN use (frcnn.PseudoLabRPN.predict_proposals): 5 levels as classes, 2000 / 2000 / 2000 / 2000 / 741
candidates, anchor-like boxes on an 800 x 1216 image, NMS 0.7, best 1000."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
counts, sizes = [2000, 2000, 2000, 2000, 741], [32, 64, 128, 256, 512]
sc, bx, lv = [], [], []
for i, (n, s) in enumerate(zip(counts, sizes)):
    cx = torch.rand(n, generator=g) * 1216; cy = torch.rand(n, generator=g) * 800
    w = s * (0.5 + torch.rand(n, generator=g)); h = s * (0.5 + torch.rand(n, generator=g))
    bx.append(torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)); sc.append(torch.randn(n, generator=g).sort(descending=True).values)
    lv.append(torch.full((n,), i, dtype=torch.int64))
sc, bx, lv = torch.cat(sc).to(dev), torch.cat(bx).to(dev), torch.cat(lv).to(dev)
R, L = sc.numel(), 5
scores = torch.full((R, L + 1), -float("inf"), device=dev); scores[torch.arange(R, device=dev), lv] = sc
boxes = bx[:, None, :].expand(R, L, 4).reshape(R, 4 * L).contiguous()
for topk, thr in [(1000, 0.7), (64, 0.7), (1, 0.7), (1000, 0.0)]:
    def run(): return ops.detect_postprocess(scores, boxes, 800, 1216, -3.0e38, thr, topk)
    for _ in range(3): out = run()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [run() for _ in range(20)]; b.record(); torch.cuda.synchronize()
    print(f"RPN-shaped post-processing ({R} candidates, 5 levels), best {topk}, NMS {thr}: {a.elapsed_time(b) / 20 * 1e3:.0f} us per call, kept {int(out[0].item())}, checksum {float(out[1][:int(out[0].item())].sum()):.3f}")
