#!/usr/bin/env python
"""Is the host ahead of the GPU when optimizer.step() is called (event recorded right before it: already complete?)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.events import EventStorage
dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16); model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3, momentum=0.9)
data = bench.make_inputs(dev, 1)
with EventStorage(0):
    for i in range(12):
        t0 = time.perf_counter()
        ld = model(data)
        t1 = time.perf_counter()
        ld.total().backward()
        t2 = time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True); ev.record()
        done = ev.query()
        import sos_wsod_amd.ops as ops
        ea = torch.cuda.Event(enable_timing=True)
        early = ops.EARLY_GRADS
        opt.step()
        eb = torch.cuda.Event(enable_timing=True); eb.record(opt._side) if opt._side is not None else eb.record()
        ec = torch.cuda.Event(enable_timing=True); ec.record()
        opt.zero_grad()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        if i >= 4:
            print(f"   GPU: end-of-backward -> side SGD done {ev.elapsed_time(eb):.3f} ms, -> all SGD done {ev.elapsed_time(ec):.3f} ms")
        if i >= 4:
            print(f"step {i}: fwd issue {1e3*(t1-t0):.2f} ms, backward() call {1e3*(t2-t1):.2f} ms, opt {1e3*(t3-t2):.2f} ms; GPU already done at step(): {done}")
torch.cuda.synchronize()
