"""cProfile of the host side of the training step (who spends the CPU time between kernel launches)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from sos_wsod_amd.solver import HipSGD  # noqa: E402
from sos_wsod_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16)
model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3,
             momentum=0.9)
tr = Trainer(model, opt)
data = bench.make_inputs(dev, 1)
for _ in range(5):
    tr.run_step(data)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    tr.run_step(data)
ti = time.perf_counter()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"CPU issue {1e3*(ti-t0)/N:.2f} ms/step; GPU done {1e3*(t1-t0)/N:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.run_step(data)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
