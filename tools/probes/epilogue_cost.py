"""How much of the fc6 weight-gradient GEMM is its f32 epilogue: the same output tiles with K = 64 (one K-tile)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
D0, D1 = 25088, 4096
dW = torch.empty(D1, D0, device=dev)
for K in (64, 128, 1024, 8000):
    dZ, X = rnd(K, D1), rnd(K, D0)
    t = timeit(lambda: ops.gemm(dZ, X, dW, D1, D0, K, a_kstrided=True, b_kstrided=True))
    print(f"wgrad-shaped f32 out, K={K:5d}: {t*1e3:7.1f} us")
Y = torch.empty(8000, D0, device=dev, dtype=dt)
for K in (64, 128, 1024, 4096):
    A, B = rnd(8000, K), rnd(D0, K)
    t = timeit(lambda: ops.gemm(A, B, Y, 8000, D0, K, ep=ops.make_epilogue(out_dtype=dt)))
    print(f"dgrad-shaped bf16 out, K={K:5d}: {t*1e3:7.1f} us")
