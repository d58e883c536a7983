"""Step time at a VOC-like non-square size (not the benchmark configuration): which kernels fall off their fast paths?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
bench.H, bench.W, bench.R = int(os.environ.get("VH", 608)), int(os.environ.get("VW", 912)), int(os.environ.get("VR", 2000))
import sos_wsod_amd.ops as ops
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16); model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3, momentum=0.9)
tr = Trainer(model, opt)
data = bench.make_inputs(dev, 1)
for _ in range(5): tr.run_step(data)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tr.run_step(data)
torch.cuda.synchronize(); print(f"{bench.H}x{bench.W} R={bench.R}: {(time.perf_counter()-t0)*100:.2f} ms/step")
