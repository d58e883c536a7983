import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oicr_oracle as O
from helpers import build_model, load_params, to_batched_inputs
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
scale2, eager_first, cf, use_params = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
K, dan, R, H, W = 20, (256, 256), 80, 96, 128
P = O.make_params(K, dan, tag="pgraph", head_scale=3.0)
def batches():
    out = []
    for i in range(6):
        views, _ = O.make_views(H, W, R, n_gt=2, K=K, scale2=scale2, tag=f"vgraph{i}")
        out.append(to_batched_inputs(views, np.array([(3 + i) % K, (11 + 2 * i) % K])))
        for k, v in out[-1][0].items():
            if k.startswith("image"): out[-1][0][k] = v.cuda()
            elif k.startswith("proposals"):
                v.proposal_boxes.tensor = v.proposal_boxes.tensor.cuda(); v.objectness_logits = v.objectness_logits.cuda()
    return out
def run(use_graph):
    model = build_model(K, dan, torch.bfloat16)
    if use_params: load_params(model, P)
    model.train()
    if os.environ.get("DIAG_SEED"): model.roi_heads.seed = 77
    groups = [{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad]
    tr = Trainer(model, HipSGD(groups, 1e-3, momentum=0.9), use_graph=use_graph, check_finite_every=cf, metrics_period=cf)
    keepl = []
    for bi, b in enumerate(batches()):
        ld = tr.run_step(b)
        if os.environ.get("DIAG_SYNC"):
            torch.cuda.synchronize(); print("step", bi, "done", float(ld.vector.sum()), flush=True)
        if os.environ.get("DIAG_CLONE"): keepl.append(ld.vector.detach().clone())
    tr.finish(); torch.cuda.synchronize()
    return tr
keep = run(False) if eager_first else None
import gc
if os.environ.get('DIAG_GC'): gc.disable()
tr = run(True)
print("OK", tr._graphs.captures, tr._graphs.replays)
'''
for name, args, env in [("seed77", "1.25 1 2 1", {"DIAG_SEED": "1"}), ("clone", "1.25 1 2 1", {"DIAG_CLONE": "1"}), ("both", "1.25 1 2 1", {"DIAG_SEED": "1", "DIAG_CLONE": "1"}), ("plain", "1.25 1 2 1", {})]:
    r = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + CHILD] + args.split(), env=dict(os.environ, **env), capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "rror:" in l or "Fatal" in l][:2]
    print(f"{name:22s} rc={r.returncode} {tail} {err}")
    print("   stdout:", r.stdout.strip().splitlines()[-8:])
    print("   stderr:", [l[:200] for l in r.stderr.strip().splitlines() if "amdgpu.ids" not in l][-12:])
