"""fc6 fwd / wgrad with a padded row pitch of the pooled-feature matrix."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
rnd = lambda *s: (torch.randn(*s, device=dev) * .5).to(dt)
M, D0, D1 = 8000, 25088, 4096
W1 = rnd(D1, D0 + 128)[:, :D0]; Y = torch.empty(M, D1 + 128, device=dev, dtype=dt)[:, :D1]
dZ = rnd(M, D1 + 128)[:, :D1]; dW = torch.empty(D1, D0, device=dev)
for pad in (0, 64, 128, 192):
    X = rnd(M, D0 + pad)[:, :D0]
    t1 = timeit(lambda: ops.gemm(X, W1, Y, M, D1, D0))
    t2 = timeit(lambda: ops.gemm(dZ, X, dW, D1, D0, M, a_kstrided=True, b_kstrided=True))
    print(f"pooled pitch {D0+pad}: fc6 fwd {t1*1e3:6.0f} us   wgrad {t2*1e3:6.0f} us")
    del X
