"""fc6 dgrad / wgrad as they are (NN / TN) vs as NT GEMMs on pre-transposed operands (what explicit transposes could buy)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
R, D0, D1 = 8000, 25088, 4096
rnd = lambda *s: (torch.randn(*s, device=dev) * .5).to(dt)
dz, X = rnd(R, D1), rnd(R, D0)
dW = torch.empty(D1, D0, device=dev)
t = timeit(lambda: ops.gemm(dz, X, dW, D1, D0, R, a_kstrided=True, b_kstrided=True))
print(f"wgrad TN (as is)            {t*1e3:7.0f} us")
dzT, XT = dz.t().contiguous(), X.t().contiguous()
t = timeit(lambda: ops.gemm(dzT, XT, dW, D1, D0, R))
print(f"wgrad NT (dz^T, pooled^T)   {t*1e3:7.0f} us")
t = timeit(lambda: ops.gemm(dzT, X, dW, D1, D0, R, b_kstrided=True))
print(f"wgrad NN (dz^T, pooled)     {t*1e3:7.0f} us")
t = timeit(lambda: X.t().contiguous())
print(f"torch transpose pooled      {t*1e3:7.0f} us")
del dW, XT, dzT
W1 = rnd(D1, D0); dX = torch.empty(R, D0, device=dev, dtype=dt)
t = timeit(lambda: ops.gemm(dz, W1, dX, R, D0, D1, b_kstrided=True, ep=ops.make_epilogue(out_dtype=dt)))
print(f"dgrad NN (as is)            {t*1e3:7.0f} us")
W1T = W1.t().contiguous()
t = timeit(lambda: ops.gemm(dz, W1T, dX, R, D0, D1, ep=ops.make_epilogue(out_dtype=dt)))
print(f"dgrad NT (W1^T)             {t*1e3:7.0f} us")
