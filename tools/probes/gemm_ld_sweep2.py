"""Row-pitch sensitivity of the fc7 GEMMs (power-of-two 8 KiB pitch) and of fc6 fwd with a padded W1 pitch."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
rnd = lambda *s: (torch.randn(*s, device=dev) * .5).to(dt)
M, D = 8000, 4096
for pad in (0, 64, 128):
    ld = D + pad
    A = rnd(M, ld)[:, :D]; W = rnd(D, ld)[:, :D]; C = torch.empty(M, ld, device=dev, dtype=dt)[:, :D]
    t1 = timeit(lambda: ops.gemm(A, W, C, M, D, D, ep=ops.make_epilogue(out_dtype=dt)))
    t2 = timeit(lambda: ops.gemm(A, W, C, M, D, D, b_kstrided=True, ep=ops.make_epilogue(out_dtype=dt)))
    dW = torch.empty(D, D, device=dev)
    t3 = timeit(lambda: ops.gemm(A, C, dW, D, D, M, a_kstrided=True, b_kstrided=True))
    print(f"fc7 pitch {ld}: fwd NT {t1*1e3:6.0f} us   dgrad NN {t2*1e3:6.0f} us   wgrad TN {t3*1e3:6.0f} us")
D0 = 25088
X = rnd(M, D0); Y = torch.empty(M, D, device=dev, dtype=dt)
for pad in (0, 128):
    W1 = rnd(D, D0 + pad)[:, :D0]
    t = timeit(lambda: ops.gemm(X, W1, Y, M, D, D0), n=10)
    print(f"fc6 fwd NT, W1 pitch {D0+pad}: {t*1e3:6.0f} us")
