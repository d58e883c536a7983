"""conv3..conv5 weight gradient (implicit GEMM + slab reduce) against the split-K factor; SKS=3,7,14 selects the factors."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for name, n, H, W, cin, cout, dil in [("conv5_3", 2, 63, 63, 512, 512, 2), ("conv4_1", 2, 64, 64, 256, 512, 1), ("conv3_2", 2, 128, 128, 256, 256, 1), ("conv3_1", 2, 128, 128, 128, 256, 1)]:
    x = (torch.randn(n, H, W, cin, device=dev) * .5).to(dt); dy = (torch.randn(n, H, W, cout, device=dev) * .5).to(dt)
    dw = torch.empty(cout, cin, 3, 3, device=dev); ws = torch.empty(32 * cout * 9 * cin, device=dev)
    fl = 2.0 * n * H * W * cout * 9 * cin
    res = []
    for sk in [int(s) for s in os.environ.get("SKS", "1,2,3,4,6,8,12,16,24,32").split(",")]:
        t = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, dil, splitk=sk, workspace=ws))
        res.append(f"sk{sk}:{t*1e3:.0f}us")
    print(name, f"{fl/1e9:.1f}GF", " ".join(res))
