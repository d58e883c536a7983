"""diagnostic: per-step losses and per-tensor weight / gradient differences, HIP fp32 vs oracle + numpy SGD (fixture s0)"""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oicr_oracle as O
from helpers import build_model, load_params, to_batched_inputs
from sos_wsod_amd.events import EventStorage
from sos_wsod_amd.solver import HipSGD
LR, MOM, WD = 1e-3, 0.9, 5e-4
g = np.load(os.path.join(ROOT, "tests/golden/e2e_s0.npz"))
K, R, H, W_ = int(g["K"]), int(g["R"]), int(g["H"]), int(g["W"]); dan = tuple(int(x) for x in g["dan"])
P = O.make_params(K, dan, tag="ps0", head_scale=float(g["head_scale"]))
views, gt = O.make_views(H, W_, R, n_gt=int(g["n_gt"]), K=K, tag="vs0")
masks = O.make_masks(R, dan, tag="ms0")
model = build_model(K, dan, torch.float32); load_params(model, P); model.train()
model.roi_heads.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
groups = [{"params": [p], "lr": 2 * LR if n.endswith(".bias") else LR, "weight_decay": 0.0 if n.endswith(".bias") else WD}
          for n, p in model.named_parameters() if p.requires_grad]
opt = HipSGD(groups, LR, momentum=MOM)
data = to_batched_inputs(views, gt)
Wn = {k: np.array(v, np.float32) for k, v in P.items()}; buf = {}
frozen = {n for n, p in model.named_parameters() if not p.requires_grad}
for step in range(3):
    with EventStorage(0):
        ld = model(data); ld.total().backward()
    hl = {k: float(v) for k, v in ld.items()}
    hg = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    ol, _, og = O.oicr_plus_iteration(Wn, views, gt, masks, K=K, want_grads=True)
    print("step", step, {k: (round(hl[k], 6), round(ol[k], 6)) for k in ("loss_cls", "loss_cls_r0", "loss_box_reg_r0")})
    for n in hg:
        e = np.abs(hg[n] - og[n]).max() / (np.abs(og[n]).max() + 1e-30)
        if e > 1e-4: print("   grad", n, "rel err %.2e" % e, "max|g| %.3e" % np.abs(og[n]).max())
    opt.step(); opt.zero_grad()
    for n in Wn:
        if n in frozen or og.get(n) is None: continue
        b = n.endswith(".bias")
        gg = og[n].astype(np.float32) + np.float32(0.0 if b else WD) * Wn[n]
        buf[n] = gg.copy() if n not in buf else np.float32(MOM) * buf[n] + gg
        Wn[n] = (Wn[n] - np.float32(2 * LR if b else LR) * buf[n]).astype(np.float32)
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        e = np.abs(p.detach().cpu().numpy() - Wn[n]).max() / (np.abs(Wn[n]).max() + 1e-30)
        if e > 1e-6: print("   weight", n, "rel err %.2e" % e, "moved %.3e" % np.abs(Wn[n] - P[n]).max())
