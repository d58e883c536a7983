"""All conv weight gradients of one backward pass (8 trainable layers x 2 view batches at the bench shapes): the grouped launch
(sw_conv3x3_wgrad_grouped + one fold per layer) for several K-tile targets against the per-layer launches it replaces."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
from sos_wsod_amd.backbone_vgg import _wgrad_grouped_splits, _wgrad_splitk
dt, dev = torch.bfloat16, "cuda"
LAYERS = [("conv3_1", 128, 128, 128, 256, 1), ("conv3_2", 128, 128, 256, 256, 1), ("conv3_3", 128, 128, 256, 256, 1),
          ("conv4_1", 64, 64, 256, 512, 1), ("conv4_2", 64, 64, 512, 512, 1), ("conv4_3", 64, 64, 512, 512, 1),
          ("conv5_1", 63, 63, 512, 512, 2), ("conv5_2", 63, 63, 512, 512, 2), ("conv5_3", 63, 63, 512, 512, 2)]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
data = []
for name, H, W, cin, cout, dil in LAYERS:
    for v in range(2):
        data.append((name, (torch.randn(2, H, W, cin, device=dev) * .5).to(dt), (torch.randn(2, H, W, cout, device=dev) * .5).to(dt), dil))
flops = sum(2.0 * x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] * 9 * dy.shape[3] for _, x, dy, _ in data)
def old():
    for i in range(0, len(data), 2):
        for v in range(2):
            _, x, dy, dil = data[i + v]
            cout, cin = dy.shape[3], x.shape[3]
            dw = torch.empty(cout, cin, 3, 3, device=dev)
            ops.conv3x3_wgrad(x, dy, dw, dil, splitk=_wgrad_splitk(cout, cin, x.shape[0] * x.shape[1] * x.shape[2]))
t = timeit(old)
print(f"per-layer launches: {t*1e3:.0f} us  ({flops/t/1e9:.0f} TFLOP/s)")
for T in [int(v) for v in os.environ.get("TS", "32,48,64,96,128").split(",")]:
    def grouped():
        probs, folds = [], []
        for i in range(0, len(data), 2):
            cout, cin = data[i][2].shape[3], data[i][1].shape[3]
            ns = [_wgrad_grouped_splits(d[1].shape[0] * d[1].shape[1] * d[1].shape[2], 64, T) for d in data[i:i + 2]]
            nsl = [ops.conv3x3_wgrad_nslab(d[1], cout, s) for d, s in zip(data[i:i + 2], ns)]
            ws = torch.empty(sum(nsl), cout * 9 * cin, device=dev)
            off = 0
            for d, s, k in zip(data[i:i + 2], ns, nsl):
                probs.append((d[1], d[2], ws[off:], d[3], s)); off += k
            folds.append((ws, sum(nsl), cout, cin))
        ops.conv3x3_wgrad_grouped(probs)
        for ws, n, cout, cin in folds:
            ops.conv3x3_wgrad_fold(ws, n, torch.empty(cout, cin, 3, 3, device=dev))
    t = timeit(grouped)
    print(f"grouped, ~{T} K-tiles per item: {t*1e3:.0f} us  ({flops/t/1e9:.0f} TFLOP/s)")
