"""Stage-3 step on the stage3_step fixture (the reference's own run): 3 fresh student / teacher pairs x 3 iterations — is the trajectory
bitwise reproducible (round 6: fixed-point ROIAlign backward), and how far is every logged loss from the reference's value?
SW_ROI_ALIGN_BWD_FX=0 shows the float-atomic form's spread."""
import os, sys, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sos_wsod_amd  # noqa
from oracle import frcnn_oracle as FO
import test_gpu_stage3 as T
from sos_wsod_amd.semisup import SemiSupStep
from sos_wsod_amd.structures import Boxes, Instances
G = np.load(os.path.join(ROOT, "tests", "golden", "stage3_step.npz"))
K = int(G["K"]); sizes = [tuple(int(v) for v in s_) for s_ in G["sizes"]]
P = FO.make_params(K, tag="s3s", head_scale=float(G["head_scale"]))
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
runs = []
for trial in range(3):
    student, teacher = T._model(K, P, "s3s"), T._model(K, P, "s3s")
    student.train(); teacher.train()
    student.proposal_generator.sampler = student.roi_heads.sampler = student.sampler
    opt = torch.optim.SGD([p for p in student.parameters() if p.requires_grad], lr=float(G["lr"]), momentum=float(G["momentum"]))
    step = SemiSupStep(student, teacher, opt, burn_up_step=int(G["cfg/BURN_UP_STEP"]), teacher_update_iter=int(G["cfg/TEACHER_UPDATE_ITER"]),
                       ema_keep_rate=float(G["cfg/EMA_KEEP_RATE"]), bbox_threshold=float(G["cfg/BBOX_THRESHOLD"]),
                       unsup_loss_weight=float(G["cfg/UNSUP_LOSS_WEIGHT"]), burn_up_with_strong_aug=bool(G["cfg/BURN_UP_WITH_STRONG_AUG"]))
    rec = []
    for it in range(3):
        record, loss_dict = step.run_step((batch("s3s_lq", 2), batch("s3s_lk", 3), batch("s3s_uq", 0), batch("s3s_uk", 0)))
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in record.items() if k.startswith("loss")}
        vals["__total"] = float(sum(float(v) for v in loss_dict.values()))
        vals["__wsum"] = float(sum(p.detach().double().sum() for p in student.parameters()))
        rec.append(vals)
    runs.append(rec)
for it in range(3):
    same = all(runs[t][it] == runs[0][it] for t in range(3))
    print(f"iteration {it}: bitwise equal over 3 runs: {same}")
    for k in sorted(runs[0][it]):
        if k == "__wsum":
            continue
        ref = float(G[f"it{it}/total_loss"]) if k == "__total" else float(G[f"it{it}/record/{k}"])
        devs = [abs(runs[t][it][k] - ref) / (abs(ref) + 1e-30) for t in range(3)]
        print(f"   {k:28s} ref {ref:12.7f}  got " + " ".join(f"{runs[t][it][k]:12.7f}" for t in range(3)) + f"   rel dev max {max(devs):.2e}")
