"""fc6 weight-gradient tail (the 2 peeled tile columns: M=4096, N=512, K=8000, f32 atomics): split-K factor sweep"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
K, D1, N = 8000, 4096, 512
dZ, X = rnd(K, D1 + 128)[:, :D1], rnd(K, 25088 + 64)[:, 25088 - N:25088]
dW = torch.zeros(D1, 25088, device=dev)[:, 25088 - N:]
for sk in (1, 2, 3, 4, 6, 8, 12, 16):
    ep = ops.make_epilogue(out_dtype=torch.float32, atomic=(sk > 1))
    t = timeit(lambda: ops.gemm(dZ, X, dW, D1, N, K, a_kstrided=True, b_kstrided=True, ep=ep, splitk=sk))
    print(f"tail splitk {sk:2d}: {t*1e3:6.1f} us")
