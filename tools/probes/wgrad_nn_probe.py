"""What would a conv weight gradient reach as an NN GEMM (A = dy^T K-contiguous, B K-strided, no gather)?  M = Cout, N = 9 Cin,
K = pixels, split-K slabs as the grouped launch uses them.  Compare with TN (both K-strided), the form the grouped kernel runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for (M, N, K) in ((512, 4608, 8192), (512, 4608, 7936), (256, 2304, 32768), (512, 2304, 8192)):
    for sk in (1, 4, 7, 14):
        at = torch.randn(M, K, device=dev).to(torch.bfloat16)          # A K-contiguous [m][k]
        ak = at.t().contiguous()                                        # A K-strided [k][m]
        bk = torch.randn(K, N, device=dev).to(torch.bfloat16)          # B K-strided [k][n]
        c = torch.empty(M, N, device=dev)
        t_nn = t(lambda: ops.gemm(at, bk, c, M, N, K, b_kstrided=True, splitk=sk))
        t_tn = t(lambda: ops.gemm(ak, bk, c, M, N, K, a_kstrided=True, b_kstrided=True, splitk=sk))
        fl = 2.0 * M * N * K / 1e12
        print(f"M={M} N={N} K={K} splitk={sk}: NN {t_nn*1e3:.0f} us = {fl/t_nn*1e3:.0f} TF/s   TN {t_tn*1e3:.0f} us = {fl/t_tn*1e3:.0f} TF/s")
