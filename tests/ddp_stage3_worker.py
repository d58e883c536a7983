"""One rank of the Stage-3 data-parallel check (launched by tests/test_gpu_ddp.py, two ranks sharing cuda:0 over gloo): the
Unbiased-Teacher step (semisup.SemiSupStep, unbias/ubteacher/engine/trainer.py:436-549) with the student detector
(frcnn.TwoStagePseudoLabGeneralizedRCNN) inside DistributedDataParallel, as the reference wraps it (ubteacher/engine/trainer.py:
`DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False)`), the teacher outside.  The student is called
TWICE per iteration (labelled batch, pseudo-labelled batch) before the one backward.  Checks, on every rank:
  1. the all-reduced gradients of a semi-supervised iteration == the mean of the two ranks' single-process gradients;
  2. after a further iteration with a real learning rate the student is bit-identical on both ranks, and so is the EMA teacher."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_path = sys.argv[1]
    import sos_wsod_amd  # noqa: F401
    from oracle import frcnn_oracle as FO                       # closed-form parameters / images / sampling keys (test infrastructure)
    from sos_wsod_amd.frcnn import TwoStagePseudoLabGeneralizedRCNN
    from sos_wsod_amd.semisup import SemiSupStep
    from sos_wsod_amd.structures import Boxes, Instances
    from sos_wsod_amd.trainer import init_distributed
    rank, _, world = init_distributed(backend=os.environ["SW_DIST_BACKEND"])
    assert world == 2
    dev = torch.device("cuda", int(os.environ.get("SW_BENCH_DEVICE", 0)))
    torch.cuda.set_device(dev)
    K = 4
    P = FO.make_params(K, tag="s3ddp", head_scale=14.0)

    class Keys:
        def __init__(self, tag):
            self.perm = FO.Perm(tag)

        def next_seed(self):
            from oracle import detgen
            k = self.perm.k
            self.perm.k += 1
            return detgen.fnv1a64(f"{self.perm.tag}perm{k}")

    def fresh(tag):
        m = TwoStagePseudoLabGeneralizedRCNN(num_classes=K, compute_dtype=torch.float32, sampler=Keys(tag)).to(dev)
        sd = m.state_dict()
        with torch.no_grad():
            for k, v in P.items():
                sd[k].copy_(torch.from_numpy(v))
        return m.train()

    sizes = [(96, 128)]

    def batch(tag, n_gt):
        out = []
        for i, (h, w) in enumerate(sizes):
            d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).to(dev), "height": h, "width": w}
            if n_gt:
                b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
                inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).to(dev)); inst.gt_classes = torch.from_numpy(c).to(dev)
                d["instances"] = inst
            out.append(d)
        return out

    def data_of(r, it):
        t = f"s3ddp_r{r}_i{it}"
        return batch(t + "lq", 2), batch(t + "lk", 3), batch(t + "uq", 0), batch(t + "uk", 0)

    def step_of(student, teacher, lr):
        params = [p for p in (student.module if hasattr(student, "module") else student).parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=lr, momentum=0.9)
        return SemiSupStep(student, teacher, opt, burn_up_step=0, ema_keep_rate=0.9996, bbox_threshold=0.5, unsup_loss_weight=2.0)

    # ---- replicas: rank r's data through a single-process student / teacher, learning rate 0 (the gradients stay in .grad)
    grads, pseudo_counts = [], []
    for r in range(2):
        s, t = fresh(f"k{r}"), fresh(f"kt{r}")
        rec, _ = step_of(s, t, 0.0).run_step(data_of(r, 0))
        grads.append({n: p.grad.detach().clone() for n, p in s.named_parameters() if p.grad is not None})
        pseudo_counts.append(sorted(k for k in rec if k.endswith("_pseudo")))
    # ---- the data-parallel run
    student, teacher = fresh(f"k{rank}"), fresh(f"kt{rank}")
    ddp = torch.nn.parallel.DistributedDataParallel(student, broadcast_buffers=False)
    step = step_of(ddp, teacher, 0.0)
    rec, _ = step.run_step(data_of(rank, 0))
    errs = []
    for n, p in student.named_parameters():
        if p.grad is None:
            assert n not in grads[0], n
            continue
        want = (grads[0][n] + grads[1][n]) / 2
        errs.append(float((p.grad - want).abs().max() / (want.abs().max() + 1e-30)))
    # ---- one more iteration with a real learning rate: both ranks must hold the same student and the same EMA teacher
    for g in step.optimizer.param_groups:
        g["lr"] = 1e-5
    rec2, _ = step.run_step(data_of(rank, 1))
    torch.cuda.synchronize()

    def digest(m):
        return torch.stack([v.detach().double().sum() for v in m.state_dict().values() if v.dtype == torch.float32]).cpu()
    same = True
    for m in (student, teacher):
        d = digest(m)
        both = [torch.empty_like(d) for _ in range(2)]
        dist.all_gather(both, d)
        same = same and bool(torch.equal(both[0], both[1]))
    torch.save({"grad_err": errs, "same_across_ranks": same, "n_grads": len(errs),
                "losses": {k: float(v) for k, v in rec2.items() if k.startswith("loss")},
                "finite": all(np.isfinite(float(v)) for k, v in rec2.items() if k.startswith("loss"))}, f"{out_path}.rank{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
