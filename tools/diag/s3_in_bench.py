"""debug: tools/stage3_step.time_step inside a process that ran Stage-1 steps first (as bench.py does)"""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench, stage3_step
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
dev = torch.device("cuda", 0)
if os.environ.get("PRE", "1") == "1":
    m = bench.build(dev, torch.bfloat16); m.train()
    gs = [{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in m.parameters() if p.requires_grad]
    tr = Trainer(m, HipSGD(gs, 1e-3, momentum=0.9), use_graph=os.environ.get("GRAPH", "1") == "1")
    b = [bench.make_inputs(dev, 100 + i) for i in range(2)]
    for i in range(8): tr.run_step(b[i % 2])
    torch.cuda.synchronize()
    del tr, m, b
    torch.cuda.empty_cache()
ms, _ = stage3_step.time_step(torch.bfloat16, 800, 1216, dev=dev, warm=int(os.environ.get("WARM", "3")), n=int(os.environ.get("N", "6")))
print(f"stage3 {ms:.2f} ms, misses {stage3_step.MISSES}, host {stage3_step.HOST_MS:.1f} ms")
