"""One GPU's share of the Stage-3 (Unbiased-Teacher) semi-supervised step at BASELINE config #5's shape: per GPU one labelled and
one unlabelled image, each in a strong and a weak view (voc_ssod.yaml: 8 + 8 images over 8 GPUs) -> teacher forward on 1 image,
student forward + backward on 2 + 1 images, SGD, teacher EMA.  usage: stage3_step.py [bf16|fp32] [H W]   (also imported by bench.py)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd  # noqa
from sos_wsod_amd.frcnn import TwoStagePseudoLabGeneralizedRCNN
from sos_wsod_amd.semisup import SemiSupStep
from sos_wsod_amd.structures import Boxes, Instances

K = 20


def make_step(dtype, dev):
    torch.manual_seed(0)
    student = TwoStagePseudoLabGeneralizedRCNN(num_classes=K, compute_dtype=dtype).to(dev).train()
    teacher = TwoStagePseudoLabGeneralizedRCNN(num_classes=K, compute_dtype=dtype).to(dev).train()
    teacher.load_state_dict(student.state_dict())
    with torch.no_grad():                      # tame the random FrozenBN affine so activations stay O(1) (a trained net's do)
        for m in (student, teacher):
            for n, b in m.named_buffers():
                if n.endswith("norm.weight"):
                    b.mul_(0.02 if "stem" in n else 0.35 if "conv3" in n else 0.6 if "shortcut" in n else 1.0)
    from sos_wsod_amd.solver import HipSGD
    opt = HipSGD([p for p in student.parameters() if p.requires_grad], 1e-4, momentum=0.9)        # the fused multi-tensor update
    return SemiSupStep(student, teacher, opt, burn_up_step=1, unsup_loss_weight=2.0)


def make_batches(n, H, W, dev, seed=1):
    g = torch.Generator().manual_seed(seed)

    def view(with_gt):
        d = {"image": torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev), "height": H, "width": W}
        if with_gt:
            k = 3
            x1 = torch.rand(k, generator=g) * (W - 200); y1 = torch.rand(k, generator=g) * (H - 200)
            b = torch.stack([x1, y1, x1 + 60 + torch.rand(k, generator=g) * 140, y1 + 60 + torch.rand(k, generator=g) * 140], 1)
            inst = Instances((H, W)); inst.gt_boxes = Boxes(b.to(dev)); inst.gt_classes = torch.randint(0, K, (k,), generator=g).to(dev)
            d["instances"] = inst
        return d
    return [([view(True)], [view(True)], [view(False)], [view(False)]) for _ in range(n)]


def time_step(dtype, H=800, W=1216, dev=None, warm=3, n=8):
    """-> (ms per iteration, last record).  Inputs are synthesised before the clock starts: the step is timed, not torch.randint."""
    dev = dev or torch.device("cuda", 0)
    step = make_step(dtype, dev)
    batches = make_batches(warm + n, H, W, dev)
    for i in range(warm):
        rec, _ = step.run_step(batches[i])
    torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.thread_time()
    for i in range(n):
        rec, _ = step.run_step(batches[warm + i])
    c1 = time.thread_time()
    torch.cuda.synchronize()
    global HOST_MS, MISSES
    HOST_MS = (c1 - c0) / n * 1e3            # CPU time of the issuing thread per iteration (close to the wall time = host bound)
    MISSES = getattr(step, "spec_misses", None)
    return (time.perf_counter() - t0) / n * 1e3, rec


HOST_MS = MISSES = None


if __name__ == "__main__":
    dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
    H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (800, 1216)
    ms, rec = time_step(dtype, H, W, n=int(os.environ.get("ITERS", 8)))
    print(f"stage-3 semi-sup step {dtype} {H}x{W}: {ms:.1f} ms per iteration per GPU (4 views: teacher fwd 1, student fwd+bwd 3), "
          f"losses finite: {all(bool(torch.isfinite(v)) for k, v in rec.items() if k.startswith('loss'))}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; issuing thread's CPU time {HOST_MS:.1f} ms per iteration; speculation misses {MISSES}")
