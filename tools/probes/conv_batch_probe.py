"""direct conv forward at batch 2 vs batch 4 (= both scales of an image in ONE launch when their sizes agree) under the kernel's
development switches: which tile form wins when a layer's two view batches share a launch?  TAG / SW_CONV_DIRECT_KG / _TN from env."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for name, H, W, cin, cout, dil in [("conv5_3", 63, 63, 512, 512, 2), ("conv4_2", 64, 64, 512, 512, 1), ("conv4_1", 64, 64, 256, 512, 1),
                                   ("conv3_2", 128, 128, 256, 256, 1), ("conv2_2", 256, 256, 128, 128, 1), ("conv1_2", 512, 512, 64, 64, 1)]:
    row = []
    for n in (2, 4):
        x = rnd(n, H, W, cin); wk = rnd(cout, 9, cin); b = torch.zeros(cout, device=dev); out = torch.empty(n, H, W, cout, device=dev, dtype=dt)
        ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
        t = timeit(lambda: ops.conv3x3(x, wk, out, dil, ep))
        fl = 2.0 * n * H * W * cout * 9 * cin
        row.append(f"n={n}: {t*1e3:6.1f} us {fl/t/1e9:5.0f} TF/s")
    print(f"{os.environ.get('TAG', '-'):12s} {name}  " + "   ".join(row))
