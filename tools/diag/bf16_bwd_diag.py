"""Where does the bf16 backward leave the bf16-emulating oracle?  Fixture s0, teacher forced at fc7: dz2 / dz1 / dpooled of the HIP
path against the oracle's autograd at the same points."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oicr_oracle as O
from helpers import build_model, load_params, to_batched_inputs
from sos_wsod_amd.events import EventStorage
import torch.nn.functional as F

case = sys.argv[1] if len(sys.argv) > 1 else "s0"
g = np.load(os.path.join(ROOT, "tests", "golden", f"e2e_{case}.npz"))
K, R, H, W = int(g["K"]), int(g["R"]), int(g["H"]), int(g["W"]); dan = tuple(int(x) for x in g["dan"])
P = O.make_params(K, dan, tag="p" + case, head_scale=float(g["head_scale"]))
views, gt = O.make_views(H, W, R, n_gt=int(g["n_gt"]), K=K, tag="v" + case)
masks = O.make_masks(R, dan, tag="m" + case)
model = build_model(K, dan, torch.bfloat16); load_params(model, P); model.train()
hd = model.roi_heads
hd.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
stash = {}
orig_top, orig_pool = hd._train_backward_top, hd._train_backward_pool
def top(st, *a, **k):
    out = orig_top(st, *a, **k); stash["dz1"] = st["dz1"].float().cpu(); stash["h1"] = st["h1"].float().cpu(); stash["h2"] = st["h2"].float().cpu()
    stash["pooled"] = st["pooled"].float().cpu(); return out
hd._train_backward_top = top
import sos_wsod_amd.ops as ops
orig_gemm = ops.gemm
def gemm(A, B, C, M, N, Kk, **kw):
    r = orig_gemm(A, B, C, M, N, Kk, **kw)
    if kw.get("tag") == "fc6_dgrad": stash["dpooled"] = C.float().cpu()
    return r
ops.gemm = gemm
orig_rb = ops.roi_pool_bwd
def rb(dout, argmax, rois, dfeat, *a, **k):
    r = orig_rb(dout, argmax, rois, dfeat, *a, **k); stash.setdefault("dfeat", []).append(dfeat.float().cpu()); return r
ops.roi_pool_bwd = rb
with EventStorage(0):
    losses = model(to_batched_inputs(views, gt)); sum(losses.values()).backward()
torch.cuda.synchronize()
fc7 = hd.last_aux["fc7"].float().cpu().numpy()
# oracle with hooks
grads = {}
orig_bhf = O.box_head_forward
def bhf(x, Pt, masks=None, bf16=False):
    x = x.flatten(1); x.retain_grad(); grads.setdefault("pooled", []).append(x)
    outs = []
    for i in (1, 2):
        w = O._rb(Pt[f"roi_heads.box_head.fc{i}.weight"], bf16)
        z = F.relu(F.linear(x, w, Pt[f"roi_heads.box_head.fc{i}.bias"]))
        if masks is not None:
            z = z * torch.from_numpy(masks[i - 1]).to(torch.float32) * 2.0
        z.retain_grad(); grads.setdefault(f"z{i}", []).append(z)
        x = O._rb(z, bf16)
    return x
O.box_head_forward = bhf
orig_vgg = O.vgg16_forward
def vgg(x, Pt, bf16=False, collect=None):
    f = orig_vgg(x, Pt, bf16=bf16, collect=collect); f.retain_grad(); grads.setdefault("feat", []).append(f); return f
O.vgg16_forward = vgg
ol, oaux, og = O.oicr_plus_iteration(P, views, gt, masks, K=K, bf16=True, want_grads=True, fc7_override=[fc7[v * R:(v + 1) * R] for v in range(4)])
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm()), float((a * b).sum() / (a.norm() * b.norm()))
# z.grad = gradient wrt the post-dropout pre-rounding activation = rounded-to-bf16 upstream gradient; dz (HIP) = that * 2 * (h > 0)
h1, h2 = stash["h1"], stash["h2"]
z2g = torch.cat([t.grad for t in grads["z2"]], 0); z1g = torch.cat([t.grad for t in grads["z1"]], 0)
print("h2 HIP vs oracle forward", rel(h2, torch.cat([t.detach() for t in grads["z2"]], 0).to(torch.bfloat16).float()))
print("h1 HIP vs oracle forward", rel(h1, torch.cat([t.detach() for t in grads["z1"]], 0).to(torch.bfloat16).float()))
print("dz1: HIP vs oracle (grad wrt z1 masked)", rel(stash["dz1"], z1g * 2.0 * (h1 > 0)))
pg = torch.cat([t.grad for t in grads["pooled"]], 0)
print("dpooled: HIP vs oracle", rel(stash["dpooled"], pg))
print("pooled fwd", rel(stash["pooled"], torch.cat([t.detach() for t in grads["pooled"]], 0)))
for i, (df, fo) in enumerate(zip(stash["dfeat"], [torch.cat([grads["feat"][0].grad]), torch.cat([grads["feat"][1].grad])])):
    print(f"dfeat scale {i}: HIP vs oracle", rel(df.permute(0, 3, 1, 2), fo))
sd = dict(model.named_parameters())
for n in ["roi_heads.box_head.fc2.weight", "roi_heads.box_head.fc2.bias", "roi_heads.box_head.fc1.weight", "roi_heads.box_head.fc1.bias", "backbone.plain5.0.conv3.weight"]:
    print(n, rel(sd[n].grad.cpu(), torch.from_numpy(og[n])))
# recompute dW2 on the host from the HIP path's own operands: is the GEMM the source?
dz2_from_oracle = None
