"""Wall-clock split of one training step into phases. Synchronises between phases, so the sum exceeds the pipelined
step bench.py reports; use it to see where the step goes, not as a throughput number."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from sos_wsod_amd.events import EventStorage  # noqa: E402
from sos_wsod_amd.solver import HipSGD  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16)
model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3,
             momentum=0.9)
data = bench.make_inputs(dev, 1)
side = torch.cuda.Stream()


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


for it in range(6):
    x = data[0]
    t0 = T()
    x1 = model._views_to_nhwc([x["image1"], x["image1_flip"]])
    x2 = model._views_to_nhwc([x["image2"], x["image2_flip"]])
    model.backbone.stage_all_weights(True)
    t1 = T()
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        f2 = model.backbone.forward_nhwc(x2)
    f1 = model.backbone.forward_nhwc(x1)
    main.wait_stream(side)
    t2 = T()
    props = [[x["proposals1"]], [x["proposals1_flip"]], [x["proposals2"]], [x["proposals2_flip"]]]
    gts = [[x["instances1"]], None, None, None]
    with EventStorage(0):
        _, losses = model.roi_heads([None] * 4, [{"plain5": f1.permute(0, 3, 1, 2)}, {"plain5": f2.permute(0, 3, 1, 2)}],
                                    props, gts)
    t3 = T()
    total = sum(losses.values())
    gf = torch.autograd.grad(total, [f1, f2], retain_graph=True) if os.environ.get("SPLIT_BWD") else None
    t4 = T()
    total.backward()
    t5 = T()
    opt.step()
    opt.zero_grad()
    t6 = T()
    if it >= 2:
        print(f"prep+stage {1e3*(t1-t0):.2f}  backbone_fwd {1e3*(t2-t1):.2f}  heads_fwd {1e3*(t3-t2):.2f}  "
              f"heads_bwd_only {1e3*(t4-t3):.2f}  backward {1e3*(t5-t4):.2f}  sgd {1e3*(t6-t5):.2f}  "
              f"sum {1e3*(t6-t0):.2f} ms", flush=True)

# CPU issue time vs GPU completion time over 10 pipelined steps (no syncs inside)
with EventStorage(0):
    for rep in range(2):
        t0 = T()
        for it in range(10):
            losses = model(data)
            sum(losses.values()).backward()
            opt.step()
            opt.zero_grad()
        ti = time.perf_counter()
        t1 = T()
        print(f"10 steps: CPU issued in {1e2*(ti-t0):.2f} ms/step, GPU done in {1e2*(t1-t0):.2f} ms/step", flush=True)
