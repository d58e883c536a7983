"""bf16 GEMM rate against K at the fc shapes (how much of a launch is prologue / epilogue)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
M, N = 8000, 25088
for K in (1024, 2048, 4096, 8192, 16384):
    A = (torch.randn(M, K, device=dev) * .5).to(dt); B = (torch.randn(K, N, device=dev) * .5).to(dt); C = torch.empty(M, N, device=dev, dtype=dt)
    t = timeit(lambda: ops.gemm(A, B, C, M, N, K, b_kstrided=True, ep=ops.make_epilogue(out_dtype=dt)))
    Bt = (torch.randn(N, K, device=dev) * .5).to(dt)
    t2 = timeit(lambda: ops.gemm(A, Bt, C, M, N, K, ep=ops.make_epilogue(out_dtype=dt)))
    print(f"K={K:6d}  NN {t*1e3:8.0f} us {2.0*M*N*K/t/1e9:7.0f} TF   NT {t2*1e3:8.0f} us {2.0*M*N*K/t2/1e9:7.0f} TF")
    del A, B, Bt, C
