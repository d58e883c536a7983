"""One-GPU training step at a named shape, for rocprofv3 (tools/profile_shapes.sh) and quick timing:
   python tools/shape_step.py recipe|coco|headline|fp32|fp32x3 [steps]
recipe   = a 500x375 VOC image at the recipe's MEAN scale: short sides 832 and 864 (voc07_oicr_plus.yaml:30 draws two distinct sides from
           480..1216 step 32: mean 848) -> views 832x1109 + 864x1152, maps 104x139 and 108x144, R = 2000, K = 20, eager launches
coco     = BASELINE configs[3] per GPU: 4 views 800x1333 (99x165 maps), R = 4000, K = 80, FREEZE_AT 3
headline = BASELINE configs[1]: 4 views 512x512"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer

RECIPE_MEAN = (832, 864)


def make(which, dev):
    if which == "coco":
        model = bench.build(dev, torch.bfloat16, K=80, freeze_at=3)
        data = [bench.make_inputs(dev, 500 + i, H=800, W=1333, R=4000, K=80, n_gt=5) for i in range(2)]
        graph = True
    elif which == "recipe":
        model = bench.build(dev, torch.bfloat16)
        s1, s2 = RECIPE_MEAN
        data = [bench.make_inputs(dev, 900 + s1 + i, H=s1, W=int(500.0 / 375.0 * s1 + 0.5), scale2=s2 / s1) for i in range(2)]
        graph = False
    elif which in ("fp32", "fp32x3"):            # BASELINE configs[1] in the reference's precision; fp32x3: fc + conv GEMMs as bf16x3
        model = bench.build(dev, torch.float32)
        model.roi_heads.fp32x3 = which == "fp32x3"
        model.backbone.fp32x3 = which == "fp32x3" and os.environ.get("SW_FP32X3_CONV", "1") != "0"
        data = [bench.make_inputs(dev, 100 + i) for i in range(2)]
        graph = False
    else:
        model = bench.build(dev, torch.bfloat16)
        data = [bench.make_inputs(dev, 100 + i) for i in range(2)]
        graph = True
    model.train()
    gs = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
          for nm, p in model.named_parameters() if p.requires_grad]
    return Trainer(model, HipSGD(gs, 1e-3, momentum=0.9), use_graph=graph and os.environ.get("EAGER") != "1"), data


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "recipe"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    tr, data = make(which, dev)
    for i in range(6):
        tr.run_step(data[i % 2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        ld = tr.run_step(data[i % 2])
    torch.cuda.synchronize()
    print(f"{which}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step over {n} steps (+6 warm-up), losses finite: "
          f"{bool(torch.isfinite(ld.vector).all())}", flush=True)


if __name__ == "__main__":
    main()
