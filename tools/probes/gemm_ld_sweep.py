"""fc6 dgrad shape (M=8000, N=25088, K=4096, B K-strided): does the row pitch of B matter?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
M, N, K = 8000, 25088, 4096
A = (torch.randn(M, K, device=dev) * .5).to(dt); C = torch.empty(M, N, device=dev, dtype=dt)
for ldb in (25088, 25088 + 64, 25088 + 128, 25088 + 512, 26624, 32768):
    Bf = (torch.randn(K, ldb, device=dev) * .5).to(dt)
    t = timeit(lambda: ops.gemm(A, Bf, C, M, N, K, b_kstrided=True, ldb=ldb, ep=ops.make_epilogue(out_dtype=dt)))
    print(f"NN ldb={ldb:6d} ({ldb*2} B)  {t*1e3:7.0f} us  {2.0*M*N*K/t/1e9:6.0f} TF")
    del Bf
# wgrad shape: M=4096, N=25088, K=8000, both strided
dZ = (torch.randn(8000, 4096, device=dev) * .5).to(dt)
dW = torch.empty(4096, 25088, device=dev, dtype=torch.float32)
for ldb in (25088, 25088 + 64, 26624):
    X = (torch.randn(8000, ldb, device=dev) * .5).to(dt)
    t = timeit(lambda: ops.gemm(dZ, X, dW, 4096, 25088, 8000, a_kstrided=True, b_kstrided=True, ldb=ldb))
    print(f"TN ldb={ldb:6d}  {t*1e3:7.0f} us  {2.0*M*N*K/t/1e9:6.0f} TF")
    del X
