"""One-GPU step at BASELINE config #4's shape: 800x1333 views (F = 99x165), R = 4000 proposals, K = 80 classes"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
bench.H, bench.W, bench.R, bench.K = 800, 1333, int(os.environ.get("VR", 4000)), 80
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16); model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3, momentum=0.9)
tr = Trainer(model, opt)
data = bench.make_inputs(dev, 1)
for _ in range(3): ld = tr.run_step(data)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ld = tr.run_step(data)
torch.cuda.synchronize()
print(f"{bench.H}x{bench.W} R={bench.R} K={bench.K}: {(time.perf_counter()-t0)*200:.2f} ms/step, losses finite: {bool(torch.isfinite(ld.vector).all())}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
