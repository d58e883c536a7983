#!/usr/bin/env python
"""fp32 gradients of a degenerate view set (R proposals) against the oracle, per parameter, plus d(loss)/d(plain5 features)."""
import os, sys
import numpy as np, torch
R_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
from helpers import build_model, load_params, to_batched_inputs
from oracle import oicr_oracle as O
from sos_wsod_amd.events import EventStorage
R, K, n_gt, H, W = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 20, 1, 96, 128))]
dan = (256, 256); tag = f"edge{R}_{K}_{n_gt}"
P = O.make_params(K, dan, tag="p" + tag, head_scale=20.0)
views, gt = O.make_views(H, W, R, n_gt=min(n_gt, K), K=K, scale2=1.25, tag="v" + tag)
masks = O.make_masks(R, dan, tag="m" + tag)
ol, oaux, og = O.oicr_plus_iteration(P, views, gt, masks, K=K, want_grads=True)
model = build_model(K, dan, torch.float32); load_params(model, P); model.train()
model.roi_heads.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
with EventStorage(0):
    losses = model(to_batched_inputs(views, gt)); sum(losses.values()).backward()
torch.cuda.synchronize()
for name, p in model.named_parameters():
    if p.grad is None or name not in og: continue
    g = og[name]; got = p.grad.cpu().numpy()
    sc = np.abs(g).max()
    print(f"{name:45s} scale {sc:9.3e}  max abs err {np.abs(got-g).max():9.3e}  rel {np.abs(got-g).max()/(sc+1e-30):8.2e}  relL2 {np.linalg.norm(got-g)/(np.linalg.norm(g)+1e-30):8.2e}")
# is the conv5_3 weight-gradient error spread over the output channels (arithmetic) or concentrated in a few (an argmax that went to a
# different, numerically tied pixel: the bias gradient cannot see that, the weight gradient can)?
g = og["backbone.plain5.0.conv3.weight"]; got = dict(model.named_parameters())["backbone.plain5.0.conv3.weight"].grad.cpu().numpy()
per = np.abs(got - g).reshape(g.shape[0], -1).max(1) / np.abs(g).max()
order = np.argsort(-per)
print("conv5_3 weight gradient: error per output channel, top 8:", [(int(c), float(f"{per[c]:.2e}")) for c in order[:8]], "median", float(np.median(per)))
