"""Every conv weight gradient of one backward pass (7 trainable layers x 2 view batches) as the grouped launch + folds, ALONE, at the
headline (512x512), recipe-mean (832x1109 + 864x1152) and COCO (800x1333) view sizes:   python tools/wgrad_shapes.py
SW_WGRAD_DIRECT=0 selects the implicit-GEMM path; TS=... lists K-tile targets per item (default: the backbone's own choice)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
from sos_wsod_amd.backbone_vgg import _wgrad_grouped_splits, _wgrad_grouped_target, _wgrad_direct_splits
dt, dev = torch.bfloat16, "cuda"


def maps(H, W):
    p = lambda v: (v - 2) // 2 + 1
    h3, w3 = p(p(H)), p(p(W)); h4, w4 = p(h3), p(w3)
    return [(h3, w3, 128, 256, 1), (h3, w3, 256, 256, 1), (h3, w3, 256, 256, 1), (h4, w4, 256, 512, 1), (h4, w4, 512, 512, 1),
            (h4, w4, 512, 512, 1), (h4, w4, 512, 512, 2), (h4, w4, 512, 512, 2), (h4, w4, 512, 512, 2)]


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n


tag = os.environ.get("TAG", "-")
for name, views in [("headline", [(512, 512), (512, 512)]), ("recipe", [(832, 1109), (864, 1152)]), ("coco", [(800, 1333), (800, 1333)])]:
    layers = [maps(*v) for v in views]
    data = []
    for li in range(9):
        for v in range(2):
            h, w, cin, cout, dil = layers[v][li]
            data.append(((torch.randn(2, h, w, cin, device=dev) * .5).to(dt), (torch.randn(2, h, w, cout, device=dev) * .5).to(dt), dil))
    flops = sum(2.0 * x.numel() * 9 * dy.shape[3] for x, dy, _ in data)
    shapes = [(x.shape[0] * x.shape[1] * x.shape[2], dy.shape[3], 9 * x.shape[3]) for x, dy, _ in data]
    targets = [int(v) for v in os.environ["TS"].split(",")] if "TS" in os.environ else [_wgrad_grouped_target(shapes, 64)]
    if os.environ.get("SW_WGRAD_DIRECT", "1") != "0" and "TS" not in os.environ:
        targets = [0]                                            # 0: the direct kernel's own split plan (backbone_vgg._wgrad_direct_splits)
    plan = _wgrad_direct_splits([(x.shape[0], x.shape[1], x.shape[2], x.shape[3], dy.shape[3], dil) for x, dy, dil in data])
    for T in targets:
        probs, folds, nslabs = [], [], 0
        for i in range(0, len(data), 2):
            cout, cin = data[i][1].shape[3], data[i][0].shape[3]
            ns = [_wgrad_grouped_splits(d[0].shape[0] * d[0].shape[1] * d[0].shape[2], 64, T) for d in data[i:i + 2]] if T else plan[i:i + 2]
            nsl = [ops.conv3x3_wgrad_nslab(d[0], cout, s) for d, s in zip(data[i:i + 2], ns)]
            ws = torch.empty(sum(nsl), cout * 9 * cin, device=dev); off = 0
            for d, s, k in zip(data[i:i + 2], ns, nsl):
                probs.append((d[0], d[1], ws[off:], d[2], s)); off += k
            folds.append((ws, sum(nsl), torch.empty(cout, cin, 3, 3, device=dev)))
            nslabs += sum(nsl)
        tg = timeit(lambda: ops.conv3x3_wgrad_grouped(probs))
        tf = timeit(lambda: ops.conv3x3_wgrad_fold_multi(folds))
        mb = sum(ws.numel() * 4 for ws, _, _ in folds) / 1e6
        print(f"{tag:8s} {name:8s} T={T:3d}: grouped launch {tg*1e3:7.1f} us = {flops/tg/1e9:5.0f} TF/s ({flops/tg/1e9/2500:.3f})   "
              f"folds {tf*1e3:6.1f} us ({nslabs} slabs, {mb:.0f} MB)   {flops/1e9:.0f} GF", flush=True)
    del data, probs, folds
