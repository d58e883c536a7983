"""GPU: Stage-3 (Unbiased-Teacher) step pieces — the teacher EMA (sw_ema_multi) and pseudo-label thresholding
(sw_threshold_select) bit-exact against the oracle restatement of unbias/ubteacher/engine/trainer.py:361-400,588-604, and the
step's control flow (burn-in -> copy -> EMA schedule, loss weights :436-549) on a toy student / teacher pair."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import semisup_oracle as SO  # noqa: E402  (checker only)


def test_teacher_ema_bit_exact():
    from sos_wsod_amd.semisup import update_teacher_model
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(8, 70001)).cuda()
    student, teacher = mk(), mk()
    student[1].num_batches_tracked += 5
    for keep in (0.9996, 0.5, 0.0):
        t0 = {k: v.detach().cpu().numpy().copy() for k, v in teacher.state_dict().items()}
        s0 = {k: v.detach().cpu().numpy().copy() for k, v in student.state_dict().items()}
        update_teacher_model(student, teacher, keep)
        want = SO.update_teacher({k: v for k, v in t0.items() if v.dtype == np.float32}, s0, keep)
        got = teacher.state_dict()
        for k, v in want.items():
            assert np.array_equal(got[k].cpu().numpy(), v), (keep, k)
    assert int(teacher[1].num_batches_tracked) == 5                        # keep 0: the copy at the end of burn-in
    # the EMA plan is cached outside the modules: the teacher neither owns the student nor carries ctypes arrays — it still
    # deep-copies and pickles after an EMA step, and the cache entry dies with it
    import copy, gc, io
    from sos_wsod_amd import semisup
    twin = copy.deepcopy(teacher)
    assert all(torch.equal(a, b) for a, b in zip(twin.state_dict().values(), teacher.state_dict().values()))
    buf = io.BytesIO(); torch.save(teacher, buf)
    assert "_ema_plan" not in teacher.__dict__ and teacher in semisup._EMA_PLANS
    n = len(semisup._EMA_PLANS)
    del teacher, twin; gc.collect()
    assert len(semisup._EMA_PLANS) == n - 1


@pytest.mark.parametrize("n", [0, 1, 77, 1024, 2500])
def test_threshold_select_bit_exact(n):
    import sos_wsod_amd.ops as ops
    rng = np.random.RandomState(n)
    scores = np.round(rng.rand(n).astype(np.float32), 2)                   # exact ties with the threshold 0.7
    classes = rng.randint(0, 20, n).astype(np.int32)
    boxes = (rng.rand(n, 4) * 500).astype(np.float32)
    for allowed in (None, [3, 7, 19]):
        cnt, b, c, s, idx = ops.threshold_select(torch.from_numpy(scores).cuda(), torch.from_numpy(classes).cuda(),
                                                 torch.from_numpy(boxes).cuda(), 0.7,
                                                 None if allowed is None else torch.tensor(allowed, dtype=torch.int32).cuda())
        wb, wc, ws, wi = SO.threshold_bbox(scores, classes, boxes, 0.7, allowed)
        k = int(cnt.item())
        assert k == len(wi)
        assert np.array_equal(idx[:k].cpu().numpy(), wi) and np.array_equal(b[:k].cpu().numpy(), wb)
        assert np.array_equal(c[:k].cpu().numpy(), wc) and np.array_equal(s[:k].cpu().numpy(), ws)


class _Toy(torch.nn.Module):
    """student / teacher stand-in with the reference's branch interface"""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(4))
        self.calls = []

    def forward(self, data, branch="supervised"):
        from sos_wsod_amd.structures import Boxes, Instances
        self.calls.append((branch, len(data), ["instances" in d for d in data]))
        if branch == "unsup_data_weak":
            outs_rpn, outs_roih = [], []
            for d in data:
                p = Instances((100, 100)); p.proposal_boxes = Boxes(torch.rand(6, 4).cuda() * 90); p.objectness_logits = torch.linspace(0, 1, 6).cuda()
                r = Instances((100, 100)); r.pred_boxes = Boxes(torch.rand(5, 4).cuda() * 90)
                r.scores = torch.tensor([0.95, 0.3, 0.71, 0.7, 0.9]).cuda(); r.pred_classes = torch.tensor([1, 2, 3, 4, 5]).cuda()
                outs_rpn.append(p); outs_roih.append(r)
            return {}, outs_rpn, outs_roih, None
        s = self.w.sum() * 0 + 1.0
        return {"loss_cls": s * 2, "loss_box_reg": s * 3, "loss_rpn_cls": s * 5, "loss_rpn_loc": s * 7, "n_boxes": 1.0}, None, None, None


def test_semisup_step_control_flow_and_loss_weights():
    from sos_wsod_amd.semisup import SemiSupStep
    student, teacher = _Toy().cuda(), _Toy().cuda()
    with torch.no_grad():
        teacher.w.fill_(5.0)
    opt = torch.optim.SGD(student.parameters(), lr=0.0)
    step = SemiSupStep(student, teacher, opt, burn_up_step=2, teacher_update_iter=2, ema_keep_rate=0.5, bbox_threshold=0.7,
                       unsup_loss_weight=4.0)
    mk = lambda n: [{"image": None, "instances": "gt"} for _ in range(n)]
    for it in range(5):
        assert SO.teacher_action(it, 2, 2) == ["burn_in", "burn_in", "copy", "none", "ema"][it]
        data = (mk(2), mk(2), mk(3), mk(3))
        before = teacher.w.detach().clone()
        record, loss_dict = step.run_step(data)
        if it < 2:
            assert student.calls[-1][:2] == ("supervised", 4) and set(loss_dict) == {"loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"}
            assert torch.equal(teacher.w, before)
            continue
        if it == 2:
            assert torch.equal(teacher.w.detach(), student.w.detach())                     # keep_rate 0: copy
        elif it == 3:
            assert torch.equal(teacher.w, before)
        else:
            assert torch.equal(teacher.w.detach(), student.w.detach() * 0.5 + before * 0.5)
        assert teacher.calls[-1][0] == "unsup_data_weak" and teacher.calls[-1][1] == 3
        assert student.calls[-2][:2] == ("supervised", 4) and student.calls[-1][:2] == ("supervised", 3)
        lab = data[2][0]["instances"]                                                       # pseudo labels replaced the ground truth
        assert lab.gt_classes.tolist() == [1, 3, 5] and data[3][0]["instances"] is lab     # 0.95, 0.71, 0.9 pass; 0.7 does not (strict >)
        want = SO.weight_losses({k: float(v) for k, v in record.items() if isinstance(v, torch.Tensor)}, 4.0)
        assert {k: float(v) for k, v in loss_dict.items()} == pytest.approx(want)
        assert float(loss_dict["loss_box_reg_pseudo"]) == 0.0 and float(loss_dict["loss_cls_pseudo"]) == 8.0


@pytest.mark.parametrize("N,C", [(512, 21), (2048, 81), (7, 3), (1, 2)])
def test_focal_loss_vs_oracle_and_torch(N, C):
    """fast_rcnn.py:73-105: value and gradient of the ROI heads' focal loss (gamma 1.5) against the float64 restatement and against
    the reference's own torch expression evaluated in fp32 on the CPU; tolerance 1e-5 relative (f32 exp / log / pow)."""
    from sos_wsod_amd.semisup import FocalLoss, fast_rcnn_focal_loss
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, generator=g) * 3
    x[0, :] = 0.0                                            # uniform row
    if N > 2:
        x[1, 0] = 40.0                                       # saturated: p -> 1 (target 0) resp. 0
    t = torch.randint(0, C, (N,), generator=g)
    t[0] = C - 1
    if N > 2:
        t[1] = 0; t[2] = (int(x[2].argmax()) + 1) % C
    want, want_g = SO.focal_loss(x.numpy(), t.numpy(), 1.5)
    xd = x.cuda().requires_grad_(True)
    loss = fast_rcnn_focal_loss(xd, t.cuda(), 1.5)
    loss.backward()
    assert abs(float(loss) - want) <= 1e-5 * max(abs(want), 1e-3)
    assert np.abs(xd.grad.cpu().numpy() - want_g).max() <= 1e-5 * max(np.abs(want_g).max(), 1e-6)
    # the reference's expression, fp32 torch on the CPU
    xc = x.clone().requires_grad_(True)
    ce = torch.nn.functional.cross_entropy(xc, t, reduction="none")
    ref = ((1 - torch.exp(-ce)) ** 1.5 * ce).sum() / N
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-5 * max(abs(float(ref)), 1e-3)
    assert (xd.grad.cpu() - xc.grad).abs().max() <= 2e-5 * max(float(xc.grad.abs().max()), 1e-6)
    # module contract: FocalLoss.forward returns the SUM (the caller divides)
    m = FocalLoss(gamma=1.5, num_classes=C - 1)
    assert abs(float(m(x.cuda(), t.cuda())) - want * N) <= 1e-5 * max(abs(want * N), 1e-3)
    assert float(fast_rcnn_focal_loss(x.cuda()[:0], t.cuda()[:0])) == 0.0


def test_two_stage_branch_dispatch_and_ensemble():
    """meta_arch/rcnn.py:8-107 branch interface + ts_ensemble.py: what each branch calls and returns"""
    from sos_wsod_amd.semisup import EnsembleTSModel, TwoStagePseudoLabRCNN
    calls = []

    class RPN(torch.nn.Module):
        def forward(self, images, features, gt, compute_loss=True, compute_val_loss=False):
            calls.append(("rpn", gt is not None, compute_loss, compute_val_loss))
            return ["props"], ({"loss_rpn_cls": torch.ones(())} if (compute_loss and gt is not None) or compute_val_loss else {})

    class Heads(torch.nn.Module):
        def forward(self, images, features, proposals, targets=None, compute_loss=True, branch="", compute_val_loss=False):
            calls.append(("roi", targets is not None, compute_loss, branch, compute_val_loss))
            if branch == "unsup_data_weak":
                return ["roih"], "pred"
            return proposals, {"loss_cls": torch.ones(())}

    m = TwoStagePseudoLabRCNN(torch.nn.Identity(), RPN(), Heads(), preprocess=lambda b: torch.zeros(1), inference=lambda b: "inf")
    m.train()
    batch = [{"instances": "gt"}]
    rec, a, b, c = m(batch, branch="supervised")
    assert set(rec) == {"loss_cls", "loss_rpn_cls"} and a == [] and b == [] and c is None
    rec, prpn, proih, pred = m([{}], branch="unsup_data_weak")
    assert rec == {} and prpn == ["props"] and proih == ["roih"] and pred == "pred"
    rec, *_ = m(batch, branch="val_loss")
    assert set(rec) == {"loss_cls", "loss_rpn_cls"}
    assert calls == [("rpn", True, True, False), ("roi", True, True, "supervised", False), ("rpn", False, False, False),
                     ("roi", False, False, "unsup_data_weak", False), ("rpn", True, True, True), ("roi", True, True, "val_loss", True)]
    m.eval()
    assert m(batch) == "inf"
    ddp_like = torch.nn.DataParallel(torch.nn.Linear(2, 2))
    ens = EnsembleTSModel(ddp_like, torch.nn.Linear(2, 2))
    assert isinstance(ens.modelTeacher, torch.nn.Linear) and set(k.split(".")[0] for k in ens.state_dict()) == {"modelTeacher", "modelStudent"}
