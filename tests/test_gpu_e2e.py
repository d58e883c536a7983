"""GPU end-to-end parity of one OICR+ iteration (forward + backward through the C-ABI kernels)
against (a) the golden fixtures generated from the reference's own Python and (b) the CPU oracle.

fp32 mode (exact f32 MFMA): losses within 1e-4 relative (north_star's bar), integer outputs bit exact,
gradients within 2e-4 of the tensor's max.
bf16 mode: compared with the oracle emulating the same bf16 storage points; losses within 2e-2 relative
(bf16 has 8 significant bits; the pseudo-label sets must still be identical for these fixtures)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oicr_oracle as O  # noqa: E402  (checker only)
from helpers import build_model, load_params, to_batched_inputs  # noqa: E402


def _setup(case, golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, f"e2e_{case}.npz"), allow_pickle=False)
    K, R, H, W = int(g["K"]), int(g["R"]), int(g["H"]), int(g["W"])
    dan = tuple(int(x) for x in g["dan"])
    P = O.make_params(K, dan, tag="p" + case, head_scale=float(g["head_scale"]))
    views, gt = O.make_views(H, W, R, n_gt=int(g["n_gt"]), K=K, tag="v" + case)
    masks = O.make_masks(R, dan, tag="m" + case)
    model = build_model(K, dan, dtype)
    load_params(model, P)
    model.train()
    model.roi_heads.debug_drop_masks = [[torch.from_numpy(m) for m in v] for v in masks]
    return g, P, views, gt, masks, model


@pytest.mark.parametrize("case", ["s0", "s1"])
def test_fp32_iteration_matches_reference_golden(case, golden_dir):
    from sos_wsod_amd.events import EventStorage
    g, P, views, gt, masks, model = _setup(case, golden_dir, torch.float32)
    K = int(g["K"])
    with EventStorage(0):
        losses = model(to_batched_inputs(views, gt))
        total = sum(losses.values())
        total.backward()
    torch.cuda.synchronize()
    assert set(losses.keys()) == {k[5:] for k in g.files if k.startswith("loss/")}
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        assert abs(v.item() - ref) <= 1e-4 * abs(ref), (k, v.item(), ref)
    aux = model.roi_heads.last_aux
    for k in range(4):
        r = aux["rounds"][k]
        n = int(r["pgt_count"].item())
        assert np.array_equal(r["pgt_index"][:n].cpu().numpy(), g[f"r{k}/pgt_index"])
        assert np.array_equal(r["pgt_class"][:n].cpu().numpy(), g[f"r{k}/pgt_classes"])
        assert np.array_equal(r["lab_class"].cpu().numpy(), g[f"r{k}/gt_classes"])
        assert np.array_equal(r["lab_index"].cpu().numpy(), g[f"r{k}/gt_index"])
        np.testing.assert_allclose(r["lab_weight"].cpu().numpy(), g[f"r{k}/gt_weights"], rtol=1e-4)
    R = int(g["R"])
    for v in range(4):
        np.testing.assert_allclose(aux["scores"][v].cpu().numpy(), g[f"wsddn_v{v}"], rtol=2e-4, atol=1e-8)
        np.testing.assert_allclose(aux["fc7"][v * R:(v + 1) * R].cpu().numpy(), g[f"fc7_v{v}"], rtol=1e-4, atol=1e-4)
    sd = dict(model.named_parameters())
    for key in g.files:
        if key.startswith("grad/"):
            ref, got = g[key], sd[key[5:]].grad.cpu().numpy()
        elif key.startswith("grads/"):
            ref, got = g[key], sd[key[6:]].grad.cpu().numpy().ravel()[::997]
        else:
            continue
        assert np.abs(got - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-20), key
    for name in g["frozen"]:
        assert sd[str(name)].grad is None


@pytest.mark.parametrize("case", ["s0"])
def test_bf16_iteration_close_to_bf16_emulating_oracle(case, golden_dir):
    from sos_wsod_amd.events import EventStorage
    g, P, views, gt, masks, model = _setup(case, golden_dir, torch.bfloat16)
    K = int(g["K"])
    ol, oaux, _ = O.oicr_plus_iteration(P, views, gt, masks, K=K, bf16=True)
    with EventStorage(0):
        losses = model(to_batched_inputs(views, gt))
        sum(losses.values()).backward()
    torch.cuda.synchronize()
    for k, v in losses.items():
        assert abs(v.item() - ol[k]) <= 2e-2 * abs(ol[k]) + 1e-5, (k, v.item(), ol[k])
    aux = model.roi_heads.last_aux
    for k in range(4):
        n = int(aux["rounds"][k]["pgt_count"].item())
        assert np.array_equal(aux["rounds"][k]["pgt_index"][:n].cpu().numpy(), oaux["rounds"][k]["pgt"]["index"])
    sd = dict(model.named_parameters())
    gref = g["grad/roi_heads.box_head.fc2.bias"]
    got = sd["roi_heads.box_head.fc2.bias"].grad.cpu().numpy()
    cos = float((got * gref).sum() / (np.linalg.norm(got) * np.linalg.norm(gref) + 1e-30))
    assert cos > 0.99, cos


def test_backbone_standalone_api_and_roipooler(golden_dir):
    """Tier-1 API: backbone(x NCHW f32) -> {"plain5": (N,512,h,w)}; ROIPooler([feat], [Boxes]) -> (R,512,7,7)."""
    from sos_wsod_amd.structures import Boxes
    g, P, views, gt, masks, model = _setup("s0", golden_dir, torch.float32)
    x = torch.stack([O.preprocess(torch.from_numpy(views[0]["image"])), O.preprocess(torch.from_numpy(views[1]["image"]))])
    out = model.backbone(x.cuda())
    assert set(out.keys()) == {"plain5"}
    f = out["plain5"]
    assert tuple(f.shape) == tuple(int(v) for v in g["plain5_shape"])
    np.testing.assert_allclose(f[0].contiguous().cpu().numpy().ravel()[::997], g["plain5_v0_sample"], rtol=1e-4, atol=1e-3)
    shp = model.backbone.output_shape()["plain5"]
    assert shp.channels == 512 and shp.stride == 8 and model.backbone.size_divisibility == 0
    pooled = model.roi_heads.box_pooler([f[0:1]], [Boxes(torch.from_numpy(views[0]["boxes"]).cuda())])
    ref, _ = O.roi_pool_fwd(f[0:1].contiguous().cpu().numpy(), O.boxes_to_rois(torch.from_numpy(views[0]["boxes"])).numpy(), 1 / 8)
    assert np.array_equal(pooled.cpu().numpy(), ref)


