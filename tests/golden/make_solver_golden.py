"""Generate tests/golden/lr_sched.npz by RUNNING the reference's own scheduler
(/root/reference/uwsod/detectron2/solver/lr_scheduler.py:13-58 WarmupMultiStepLR, :91-119 _get_warmup_factor_at_iter),
loaded by path in THIS container (the file imports only math / bisect / typing / torch).  The fixture holds inputs
(milestones, gamma, warmup settings, base lrs) and the learning-rate sequence per iteration: data, no source text.

    python tests/golden/make_solver_golden.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/uwsod/detectron2/solver/lr_scheduler.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lr_sched.npz")

# (name, milestones, gamma, warmup_factor, warmup_iters, warmup_method, n_iters)
CASES = [
    ("voc", (35, 50), 0.1, 0.001, 0, "linear", 60),            # voc07_oicr_plus.yaml STEPS (35000, 50000), WARMUP_ITERS 0, scaled /1000
    ("linear", (30, 45), 0.1, 0.001, 20, "linear", 60),        # detectron2 defaults' shape (WARMUP 1000 / STEPS 30000), scaled
    ("const", (10, 12, 40), 0.5, 0.3, 7, "constant", 50),
]


def main():
    spec = importlib.util.spec_from_file_location("ref_lr_scheduler", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, ms, gamma, wf, wi, wm, n in CASES:
        params = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD([{"params": [params[0]], "lr": 1e-3}, {"params": [params[1]], "lr": 2e-3}], lr=1e-3, momentum=0.9)
        sch = mod.WarmupMultiStepLR(opt, list(ms), gamma, warmup_factor=wf, warmup_iters=wi, warmup_method=wm)
        lrs = []
        for _ in range(n):
            lrs.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sch.step()
        out[f"{name}/milestones"] = np.asarray(ms, np.int64)
        out[f"{name}/hyper"] = np.asarray([gamma, wf, wi], np.float64)
        out[f"{name}/method"] = np.asarray(wm)
        out[f"{name}/lrs"] = np.asarray(lrs, np.float64)
    np.savez(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("lrs")})


if __name__ == "__main__":
    main()
