"""debug: cProfile of Stage-3 iterations after a graph-mode Stage-1 Trainer lived in the process"""
import cProfile, pstats, io, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench, stage3_step as S
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
dev = torch.device("cuda", 0)
m = bench.build(dev, torch.bfloat16); m.train()
gs = [{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in m.parameters() if p.requires_grad]
tr = Trainer(m, HipSGD(gs, 1e-3, momentum=0.9), use_graph=True)
b = [bench.make_inputs(dev, 100 + i) for i in range(2)]
for i in range(8): tr.run_step(b[i % 2])
torch.cuda.synchronize()
del tr, m, b
torch.cuda.empty_cache()
step = S.make_step(torch.bfloat16, dev)
batches = S.make_batches(4 + 10, 800, 1216, dev)
for i in range(4): step.run_step(batches[i])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(10): step.run_step(batches[4 + i])
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:5000])
