"""conv5_3 / conv4_2 forward alone (20 back-to-back launches), for kernel experiments"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for name, n, H, W, cin, cout, dil in [("conv5_3", 2, 63, 63, 512, 512, 2), ("conv4_2", 2, 64, 64, 512, 512, 1), ("conv4_1", 2, 64, 64, 256, 512, 1),
                                      ("conv3_2", 2, 128, 128, 256, 256, 1), ("conv2_2", 2, 256, 256, 128, 128, 1),
                                      ("conv1_2", 2, 512, 512, 64, 64, 1)]:
    x = rnd(n, H, W, cin); wk = rnd(cout, 9, cin); b = torch.zeros(cout, device=dev); out = torch.empty(n, H, W, cout, device=dev, dtype=dt)
    ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
    t = timeit(lambda: ops.conv3x3(x, wk, out, dil, ep))
    fl = 2.0 * n * H * W * cout * 9 * cin
    print(f"{os.environ.get('TAG', '-'):10s} {name} {t*1e3:6.1f} us {fl/t/1e9:6.0f} TF/s")
