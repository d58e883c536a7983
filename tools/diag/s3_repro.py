"""run-to-run reproducibility of the Stage-3 step on the stage3_step fixture's inputs: 4 fresh student/teacher pairs, 2 iterations each"""
import os, sys, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sos_wsod_amd  # noqa
from oracle import frcnn_oracle as FO
import test_gpu_stage3 as T
from sos_wsod_amd.semisup import SemiSupStep
from sos_wsod_amd.structures import Boxes, Instances
G = np.load(os.path.join(ROOT, "tests", "golden", "stage3_step.npz"))
K = 20; sizes = [(96, 128), (128, 112)]
P = FO.make_params(K, tag="s3s", head_scale=14.0)
def batch(tag, n_gt):
    out = []
    for i, (h, w) in enumerate(sizes):
        d = {"image": torch.from_numpy(FO.make_image(h, w, f"{tag}{i}")).cuda(), "height": h, "width": w}
        if n_gt:
            b, c = FO.make_gt(h, w, n_gt, K, f"{tag}{i}")
            inst = Instances((h, w)); inst.gt_boxes = Boxes(torch.from_numpy(b).cuda()); inst.gt_classes = torch.from_numpy(c).cuda()
            d["instances"] = inst
        out.append(d)
    return out
lock = os.environ.get("LOCKSTEP", "1") == "1"
for trial in range(4):
    student, teacher = T._model(K, P, "s3s"), T._model(K, P, "s3s")
    student.train(); teacher.train()
    student.proposal_generator.sampler = student.roi_heads.sampler = student.sampler
    opt = torch.optim.SGD([p for p in student.parameters() if p.requires_grad], lr=1e-5, momentum=0.9)
    step = SemiSupStep(student, teacher, opt, burn_up_step=1, ema_keep_rate=0.9996, bbox_threshold=0.7, unsup_loss_weight=2.0, lockstep=lock)
    for it in range(2):
        data = (batch("s3s_lq", 2), batch("s3s_lk", 3), batch("s3s_uq", 0), batch("s3s_uk", 0))
        record, loss_dict = step.run_step(data)
        torch.cuda.synchronize()
        w = float(sum(p.detach().double().sum() for p in student.parameters()))
        line = f"trial {trial} it {it} wsum {w:.12f} " + " ".join(f"{k[5:]}={float(v):.7f}" for k, v in record.items() if k.startswith("loss"))
        if it == 1:
            pb = [d["instances"].gt_boxes.tensor.double().sum().item() for d in data[2]]
            line += f" pseudo_box_sums {pb[0]:.6f} {pb[1]:.6f}"
        print(line, flush=True)
