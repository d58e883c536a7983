#!/usr/bin/env python
"""Where the supervised-branch run differs from the reference-generated fixture stage3_a (losses, sampled anchors / proposals, logits)."""
import os, sys
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_stage3 as T
from oracle import frcnn_oracle as FO
t = np.load(os.path.join(R, "tests", "golden", "stage3_a.npz"))
K = int(t["K"])
P = FO.make_params(K, tag="s3a", head_scale=float(t["head_scale"]))
model = T._model(K, P, "s3a"); model.train()
data, gts = T._inputs("s3a", t, K)
losses, _, _, _ = model(data, branch="supervised")
sum(losses.values()).backward()
for k, v in losses.items():
    ref = float(t["loss/" + k]); print(k, float(v), ref, abs(float(v) - ref) / abs(ref))
lab = model.proposal_generator.last_labels.cpu().numpy()
for i in range(2):
    print("img", i, "rpn labels equal", np.array_equal(lab[i], t[f"rpn_labels{i}"]), "differ at", int((lab[i] != t[f"rpn_labels{i}"]).sum()))
    s = model.roi_heads.last_sampled[i]
    print("   sampled classes equal", np.array_equal(s.gt_classes.cpu().numpy(), t[f"samp_classes{i}"]),
          "box max diff", float(np.abs(s.proposal_boxes.tensor.cpu().numpy() - t[f"samp_boxes{i}"]).max()))
lg = model.roi_heads.last_logits.detach().cpu().numpy()
print("scores max diff", float(np.abs(lg[:, :K + 1] - t["scores"]).max()), "deltas", float(np.abs(lg[:, K + 1:5 * K + 1] - t["deltas"]).max()))
# --- which part of the forward moved? (a) the in-kernel FrozenBN fold against torch's arithmetic, (b) the RPN head's GEMM per level against the concatenated one
import sos_wsod_amd.ops as ops
from sos_wsod_amd import frcnn
nb = 0
for m in model.modules():
    if isinstance(m, frcnn.ConvBN) and m.k != 7:
        n = m.norm
        sc = n.weight.cpu() * torch.rsqrt(n.running_var.cpu() + 1e-5); sh = n.bias.cpu() - n.running_mean.cpu() * sc        # torch-CPU arithmetic
        st = m._st
        nb += int((st.scale.cpu() != sc).sum()) + int((st.shift.cpu() != sh).sum())
print("FrozenBN fold: elements that differ from torch's", nb)
with torch.no_grad():
    x4, sizes = model.preprocess_image(data)
    feats = model.backbone(x4)
    head = model.proposal_generator.rpn_head
    lg, dl = head(feats)
    st = head._st
    for f, l in zip(feats, lg):
        tt = head.conv(f, relu=True)
        n, H, W, C = tt.shape
        y = frcnn._LinearFn.apply(tt.reshape(n * H * W, C), st.w, st.bias, None, False, True, (3, 12), None, head.objectness_logits.weight,
                                  head.anchor_deltas.weight, head.objectness_logits.bias, head.anchor_deltas.bias)
        print("level", H, W, "logits differing between per-level and concatenated GEMM:", int((y[:, :3].reshape(n, -1) != l).sum()), "of", l.numel())
# --- the proposals themselves
model.train()
with torch.no_grad():
    props, _ = model.proposal_generator(sizes, feats, None, compute_loss=False)
for i, p in enumerate(props):
    pb, pl = p.proposal_boxes.tensor.cpu().numpy(), p.objectness_logits.cpu().numpy()
    rb, rl = t[f"prop_boxes{i}"], t[f"prop_logits{i}"]
    print("img", i, "proposals", len(pb), "fixture", len(rb))
    n = min(len(pb), len(rb))
    bad = np.nonzero(np.abs(pb[:n] - rb[:n]).max(1) > 1e-2)[0]
    print("   positions whose box differs:", bad[:20], "count", len(bad))
    for j in bad[:6]:
        print("    ", j, pb[j], pl[j], "| fixture", rb[j], rl[j])
    s = model.roi_heads.last_sampled[i].proposal_boxes.tensor.cpu().numpy()
    sb = t[f"samp_boxes{i}"]
    bad = np.nonzero(np.abs(s - sb).max(1) > 1e-2)[0]
    print("   sampled rows that differ:", bad[:20], "count", len(bad))
