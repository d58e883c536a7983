"""diagnostic: capture the training step as a hipGraph at small sizes with parts switched off (each variant in a subprocess)"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
H, W, R, dual, dan = [int(v) for v in sys.argv[1:6]]
dev = torch.device("cuda", 0)
bench.DAN = (dan, dan)
model = bench.build(dev, torch.bfloat16); model.train(); model.dual_stream = bool(dual)
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3, momentum=0.9)
tr = Trainer(model, opt, use_graph=True)
data = [bench.make_inputs(dev, 5 + i, H=H, W=W, R=R, n_gt=2) for i in range(2)]
for i in range(5):
    tr.run_step(data[i % 2])
torch.cuda.synchronize()
print("OK", tr._graphs.captures, tr._graphs.replays)
'''
for name, args, env in [("bench-size", "512 512 2000 1 4096", {}), ("small", "96 128 80 1 256", {}), ("small-single-stream", "96 128 80 0 256", {}),
                        ("small-no-grouped", "96 128 80 1 256", {"SW_WGRAD_GROUPED": "0"}), ("mid", "256 256 500 1 1024", {}),
                        ("small-bigdan", "96 128 80 1 4096", {}), ("big-smalldan", "512 512 2000 1 256", {})]:
    r = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + CHILD] + args.split(), env=dict(os.environ, **env), capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Fatal" in l][:2]
    print(f"{name:22s} rc={r.returncode} {tail} {err}")
