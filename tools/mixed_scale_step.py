"""Training steps on mapper-built views of a VOC-sized image (375x500, 2000 proposals) at the recipe's scale pairs:
which kernels fall off their fast paths when the two view pairs have different, non-square sizes?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import sos_wsod_amd.ops as ops
from sos_wsod_amd.mapper import DeviceMultiInputMapper
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16); model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3, momentum=0.9)
tr = Trainer(model, opt)
rng = np.random.RandomState(0)
h, w, R = 375, 500, int(os.environ.get("VR", 2000))
x1 = rng.randint(0, w - 20, R); y1 = rng.randint(0, h - 20, R)
x2 = np.minimum(x1 + rng.randint(12, w // 2, R), w - 1); y2 = np.minimum(y1 + rng.randint(12, h // 2, R), h - 1)
d = {"image": torch.randint(0, 256, (3, h, w), dtype=torch.uint8, device=dev),
     "proposal_boxes": np.stack([x1, y1, x2, y2], 1).astype(np.float32), "proposal_objectness_logits": rng.rand(R).astype(np.float32),
     "annotations": [{"bbox": [30.0, 40.0, 300.0, 330.0], "category_id": 4}, {"bbox": [200.0, 10.0, 480.0, 200.0], "category_id": 17}]}
mapper = DeviceMultiInputMapper(proposal_topk=4000)
pairs = [((480, 640), (576, 768)), ((688, 917), (864, 1152)), ((1000, 1333), (1200, 1600)), ((480, 640), (1200, 1600))]
which = os.environ.get("PAIR")
for i, shapes in enumerate(pairs):
    if which is not None and int(which) != i:
        continue
    data = [mapper(d, shapes=shapes)]
    n = len(data[0]["proposals1"].proposal_boxes)
    for _ in range(3): tr.run_step(data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): tr.run_step(data)
    torch.cuda.synchronize()
    print(f"views {shapes[0]} + {shapes[1]}, {n} proposals: {(time.perf_counter() - t0) * 200:.2f} ms/step, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
