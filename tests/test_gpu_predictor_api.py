"""GPU: the predictor modules' own call API (forward / losses / predict_probs / inference of WSDDNOutputLayers and
OICROutputLayers, fast_rcnn_wsddn.py:542-589,658-681; fast_rcnn_oicr.py:504-614,702-716) against the oracle's torch-CPU
restatement of the same functions: outputs, losses and the gradients that reach the weights and the input features."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import oicr_oracle as O  # noqa: E402  (checker only)


def _rel(a, b):
    """max error relative to the reference's largest element; + 1e-7 absolute: d/d(det.bias) is analytically 0 (the softmax over
    proposals is shift invariant), both sides hold rounding noise there"""
    return float(((a - b).abs().max() - 1e-7).clamp(min=0) / (b.abs().max() + 1e-30))


def test_wsddn_layer_forward_losses_and_gradients():
    from sos_wsod_amd.fast_rcnn_wsddn import WSDDNOutputLayers
    from sos_wsod_amd.structures import Boxes, Instances
    torch.manual_seed(0)
    K, D, sizes = 20, 256, [70, 45]
    layer = WSDDNOutputLayers(D, num_classes=K).cuda()
    with torch.no_grad():
        layer.cls.weight.mul_(8); layer.det.weight.mul_(8); layer.cls.bias.uniform_(-0.1, 0.1)
    x = torch.randn(sum(sizes), D, device="cuda", requires_grad=True)
    props = []
    for n in sizes:
        p = Instances((100, 100)); p.proposal_boxes = Boxes(torch.rand(n, 4).cuda()); props.append(p)
    oh = torch.zeros(2, K, device="cuda"); oh[0, [2, 7]] = 1; oh[1, 11] = 1
    scores, deltas = layer(x, props)
    assert tuple(scores.shape) == (115, K) and tuple(deltas.shape) == (115, 4 * K) and float(deltas.abs().max()) == 0
    losses = layer.losses((scores, deltas), props, oh)
    losses["loss_cls"].backward()
    # oracle: the same arithmetic on the CPU (fast_rcnn_wsddn.py:556-567, 340-375)
    xc = x.detach().cpu().requires_grad_(True)
    Wc = {n: p.detach().cpu().requires_grad_(True) for n, p in layer.named_parameters()}
    C = F.linear(xc, Wc["cls.weight"], Wc["cls.bias"]); Dd = F.linear(xc, Wc["det.weight"], Wc["det.bias"])
    ref_scores, loss = [], 0.0
    off = 0
    for i, n in enumerate(sizes):
        s = F.softmax(C[off:off + n], dim=1) * F.softmax(Dd[off:off + n], dim=0)
        ref_scores.append(s)
        loss = loss + O.wsddn_loss(s, oh[i:i + 1].cpu())
        off += n
    loss = loss / len(sizes)
    loss.backward()
    assert _rel(scores.detach().cpu(), torch.cat(ref_scores).detach()) < 2e-5
    assert abs(float(losses["loss_cls"]) - float(loss)) <= 2e-5 * abs(float(loss))
    for n, p in layer.named_parameters():
        assert _rel(p.grad.cpu(), Wc[n].grad) < 2e-4, n
    assert _rel(x.grad.cpu(), xc.grad) < 2e-4
    parts = layer.predict_probs((scores, deltas), props)
    assert [len(t) for t in parts] == sizes


def test_oicr_layer_forward_losses_probs_inference():
    from sos_wsod_amd.fast_rcnn_oicr import OICROutputLayers
    from sos_wsod_amd.structures import Boxes, Instances
    torch.manual_seed(1)
    K, D, sizes = 20, 128, [60, 33]
    layer = OICROutputLayers(D, num_classes=K, refine_k=0, refine_reg=[True], test_score_thresh=0.05, test_nms_thresh=0.3).cuda()
    with torch.no_grad():
        layer.cls_score.weight.mul_(30); layer.bbox_pred.weight.mul_(30)
    N = sum(sizes)
    x = torch.randn(N, D, device="cuda", requires_grad=True)
    views, _ = O.make_views(120, 160, N, n_gt=2, K=K, tag="papi")
    pb = torch.from_numpy(views[0]["boxes"])
    gen = torch.Generator().manual_seed(2)
    gtb = pb[torch.randint(0, N, (N,), generator=gen)] + torch.rand(N, 4, generator=gen)
    gtc = torch.randint(-1, K + 1, (N,), generator=gen)
    gtw = torch.rand(N, generator=gen)
    props, off = [], 0
    for n in sizes:
        p = Instances((120, 160))
        p.proposal_boxes = Boxes(pb[off:off + n].cuda()); p.gt_boxes = Boxes(gtb[off:off + n].cuda())
        p.gt_classes = gtc[off:off + n].cuda(); p.gt_weights = gtw[off:off + n].cuda()
        props.append(p); off += n
    scores, deltas = layer(x)
    assert tuple(scores.shape) == (N, K + 1) and tuple(deltas.shape) == (N, 4 * K)
    losses = layer.losses((scores, deltas), props)
    (losses["loss_cls"] + 2.0 * losses["loss_box_reg"]).backward()
    xc = x.detach().cpu().requires_grad_(True)
    Wc = {n: p.detach().cpu().requires_grad_(True) for n, p in layer.named_parameters()}
    lg = F.linear(xc, Wc["cls_score.weight"], Wc["cls_score.bias"]); dl = F.linear(xc, Wc["bbox_pred.weight"], Wc["bbox_pred.bias"])
    a, b = O.oicr_losses(lg, dl, pb.numpy(), gtb.numpy(), gtc.numpy(), gtw.numpy(), K)      # fast_rcnn_oicr.py:157-352 (mean over all N)
    (a + 2.0 * b).backward()
    assert _rel(scores.detach().cpu(), lg.detach()) < 2e-5 and _rel(deltas.detach().cpu(), dl.detach()) < 2e-5
    assert abs(float(losses["loss_cls"]) - float(a)) <= 2e-5 * abs(float(a))
    assert abs(float(losses["loss_box_reg"]) - float(b)) <= 2e-5 * abs(float(b))
    for n, p in layer.named_parameters():
        assert _rel(p.grad.cpu(), Wc[n].grad) < 3e-4, n
    assert _rel(x.grad.cpu(), xc.grad) < 3e-4
    probs = layer.predict_probs((scores, deltas), props)
    ref = F.softmax(lg.detach(), dim=-1)
    assert [len(t) for t in probs] == sizes and _rel(torch.cat(probs).cpu(), ref) < 2e-5
    dets, kept = layer.inference((scores, deltas), props)
    assert len(dets) == 2
    for d, k_, p in zip(dets, kept, props):
        assert len(d) == len(k_) <= 100 and (d.scores[:-1] >= d.scores[1:]).all()
        bx = d.pred_boxes.tensor
        assert (bx[:, 0] >= 0).all() and (bx[:, 2] <= 160).all() and (bx[:, 3] <= 120).all()


def test_predictions_survive_caller_side_tensor_ops():
    """the logits travel as a Python attribute on `scores`; a caller that slices / re-concatenates the predictions
    (torch.cat over images drops attributes) must still get the same losses and gradients, and a loss the caller builds itself
    from the WSDDN scores must train the layer (the scores are a real autograd output, not a detached tensor)."""
    from sos_wsod_amd.fast_rcnn_oicr import OICROutputLayers
    from sos_wsod_amd.fast_rcnn_wsddn import WSDDNOutputLayers
    from sos_wsod_amd.structures import Boxes, Instances
    torch.manual_seed(3)
    K, D, sizes = 20, 128, [40, 25]
    N = sum(sizes)
    props = []
    for n in sizes:
        p = Instances((100, 100)); p.proposal_boxes = Boxes((torch.rand(n, 4) * 50 + torch.tensor([0.0, 0.0, 50.0, 50.0])).cuda())
        p.gt_boxes = Boxes(p.proposal_boxes.tensor + 1.0); p.gt_classes = torch.randint(-1, K + 1, (n,)).cuda()
        p.gt_weights = torch.rand(n).cuda(); props.append(p)
    oh = torch.zeros(2, K, device="cuda"); oh[0, 3] = 1; oh[1, [5, 9]] = 1
    x = torch.randn(N, D, device="cuda")

    def grads(layer):
        g = [p.grad.clone() for p in layer.parameters()]
        for p in layer.parameters():
            p.grad = None
        return g
    # ---- WSDDN: fused path vs predictions rebuilt by the caller (attribute gone) vs the caller's own loss on the scores
    wl = WSDDNOutputLayers(D, num_classes=K).cuda()
    with torch.no_grad():
        wl.cls.weight.mul_(8); wl.det.weight.mul_(8)
    s, d = wl(x, props)
    assert s.requires_grad
    l0 = wl.losses((s, d), props, oh)["loss_cls"]; l0.backward(); g0 = grads(wl)
    s, d = wl(x, props)
    s2 = torch.cat([s[:sizes[0]], s[sizes[0]:]], 0)                       # what per-image post-processing does
    assert not hasattr(s2, "_sw_logits")
    l1 = wl.losses((s2, d), props, oh)["loss_cls"]; l1.backward(); g1 = grads(wl)
    assert abs(float(l1) - float(l0)) <= 1e-5 * abs(float(l0))
    for a, b in zip(g0, g1):
        assert _rel(b, a) < 1e-4
    # reference-style code that builds its own loss from `scores` (this trained with zero gradient before)
    xc = x.detach().cpu()
    Wc = {n: p.detach().cpu().requires_grad_(True) for n, p in wl.named_parameters()}
    C = F.linear(xc, Wc["cls.weight"], Wc["cls.bias"]); Dd = F.linear(xc, Wc["det.weight"], Wc["det.bias"])
    ref = torch.cat([F.softmax(C[:sizes[0]], 1) * F.softmax(Dd[:sizes[0]], 0), F.softmax(C[sizes[0]:], 1) * F.softmax(Dd[sizes[0]:], 0)])
    wgt = torch.rand(N, K)
    (ref * wgt).sum().backward()
    s, _ = wl(x, props)
    (s * wgt.cuda()).sum().backward()
    for n, p in wl.named_parameters():
        assert _rel(p.grad.cpu(), Wc[n].grad) < 2e-4, n
    # ---- OICR: fused path vs detached-and-rejoined predictions
    ol = OICROutputLayers(D, num_classes=K, refine_k=0, refine_reg=[True]).cuda()
    with torch.no_grad():
        ol.cls_score.weight.mul_(30); ol.bbox_pred.weight.mul_(30)
    sc, dl = ol(x)
    la = ol.losses((sc, dl), props); (la["loss_cls"] + la["loss_box_reg"]).backward(); ga = grads(ol)
    sc, dl = ol(x)
    sc2, dl2 = sc[:, :].clone(), dl * 1.0                                # new tensors, graph kept, attribute lost
    lb = ol.losses((sc2, dl2), props); (lb["loss_cls"] + lb["loss_box_reg"]).backward(); gb = grads(ol)
    assert abs(float(la["loss_cls"]) - float(lb["loss_cls"])) <= 1e-6 * abs(float(la["loss_cls"]))
    for a, b in zip(ga, gb):
        assert _rel(b, a) < 1e-5
    dets, _ = ol.inference((sc.detach(), dl.detach()), props)            # detached predictions: no AttributeError
    assert len(dets) == 2
    # the staged weight copy is cached until a parameter changes
    st = ol.__dict__["_api_stage"][1]
    ol(x)
    assert ol.__dict__["_api_stage"][1] is st
    with torch.no_grad():
        ol.cls_score.weight.add_(1.0)
    ol(x)
    assert ol.__dict__["_api_stage"][1] is not st


def test_wsddn_scores_backward_kernel_against_float64():
    """sw_wsddn_scores_bwd (the stand-alone API's gradient of softmax(C, 1) * softmax(D, 0), fast_rcnn_wsddn.py:564-567) against the
    analytic expression in float64 — VOC and COCO class counts, a padded logit pitch, peaky logits"""
    import sos_wsod_amd.ops as ops
    torch.manual_seed(0)
    for R, K, pad, amp in ((300, 20, 0, 3.0), (1000, 80, 8, 12.0), (37, 5, 3, 30.0)):
        lg = (torch.randn(R, 2 * K + pad, device="cuda") * amp)[:, :2 * K]
        g = torch.randn(R, K, device="cuda")
        d = torch.zeros(R, 2 * K, device="cuda")
        ops.wsddn_scores_bwd(lg, K, g, d)
        A = torch.softmax(lg[:, :K].double(), 1); B = torch.softmax(lg[:, K:2 * K].double(), 0)
        gB, gA = g.double() * B, g.double() * A
        wc = A * (gB - (A * gB).sum(1, keepdim=True)); wd = B * (gA - (B * gA).sum(0, keepdim=True))
        assert float((d[:, :K].double() - wc).abs().max()) <= 2e-7 * float(wc.abs().max()) + 1e-12
        assert float((d[:, K:].double() - wd).abs().max()) <= 2e-7 * float(wd.abs().max()) + 1e-12
