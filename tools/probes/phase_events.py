"""GPU time of the step's phases from events recorded on the main stream (no host syncs inside the step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.events import EventStorage
dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16); model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3, momentum=0.9)
data = bench.make_inputs(dev, 1)
hd, bb = model.roi_heads, model.backbone
marks = {}
def ev(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.setdefault(name, []).append(e)
orig_heads_fwd = hd.forward
def heads_fwd(*a, **k):
    ev("backbone_fwd_done"); r = orig_heads_fwd(*a, **k); ev("heads_fwd_done"); return r
hd.forward = heads_fwd
N = 30
with EventStorage(0):
    for i in range(N + 5):
        ev("start")
        ld = model(data)
        ld.total().backward()
        ev("backward_done")
        opt.step(); opt.zero_grad()
        ev("sgd_done")
torch.cuda.synchronize()
names = ["start", "backbone_fwd_done", "heads_fwd_done", "backward_done", "sgd_done"]
for a, b in zip(names[:-1], names[1:]):
    ts = [x.elapsed_time(y) for x, y in zip(marks[a][5:], marks[b][5:])]
    print(f"{a:>20s} -> {b:<20s} {sum(ts)/len(ts):7.3f} ms")
tot = [x.elapsed_time(y) for x, y in zip(marks["start"][5:-1], marks["start"][6:])]
print(f"{'step (start->start)':>44s} {sum(tot)/len(tot):7.3f} ms")
