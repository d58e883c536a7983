"""The VOC recipe's scale pairs (two distinct short sides from 480..1216 step 32, a 500x375 image), eager: per-iteration wall times,
to see what a NEW input signature costs (allocator growth, weight staging) against the steady state of that signature."""
import os, sys, time, random, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as B
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
dev = torch.device("cuda", 0); dt = torch.bfloat16
mm = B.build(dev, dt); mm.train()
gs = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
      for nm, p in mm.named_parameters() if p.requires_grad]
tr = Trainer(mm, HipSGD(gs, 1e-3, momentum=0.9), use_graph=False)
rnd = random.Random(1234); shorts = list(range(480, 1217, 32))
pairs = [tuple(rnd.sample(shorts, 2)) for _ in range(int(os.environ.get("PAIRS", "8")))]
for rep in range(2):
    for s1, s2 in pairs:
        dat = B.make_inputs(dev, 900 + s1, H=s1, W=int(500.0 / 375.0 * s1 + 0.5), scale2=s2 / s1)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter(); tr.run_step(dat); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(f"pass {rep} pair {s1:4d},{s2:4d}: " + " ".join(f"{x:7.2f}" for x in ts) + f"   alloc {torch.cuda.memory_allocated()/2**30:.1f} GiB reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
        del dat
