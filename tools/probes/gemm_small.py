"""Plain NT GEMM at the conv4/conv5 implicit-GEMM shape (M=8192 px, N=512, K=4608) vs the conv kernel itself."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for (M, N, K) in ((8192, 512, 4608), (8192, 512, 512), (32768, 256, 2304), (131072, 128, 1152)):
    A = (torch.randn(M, K, device=dev) * .5).to(dt); B = (torch.randn(N, K, device=dev) * .5).to(dt); C = torch.empty(M, N, device=dev, dtype=dt)
    v = os.environ.get("SW_GEMM_V", "")
    if True:
        t = timeit(lambda: ops.gemm(A, B, C, M, N, K, ep=ops.make_epilogue(out_dtype=dt)))
        print(f"NT {M}x{N}x{K} V={v or 'auto'}: {t*1e3:7.1f} us  {2.0*M*N*K/t/1e9:6.0f} TF")
