"""fc6 forward / data gradient / weight gradient in the launch forms of the training step (roi_heads_oicrplus: the weight gradient as
transpose(dZ) + a GEMM with a K-strided B operand on the ping-pong loop), alone; TAG labels the line (A/B of library builds)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
rnd = lambda *s: torch.relu(torch.randn(*s, device=dev) * 0.5).to(dt)      # post-ReLU-like operands (half zeros), as in the step
M, D0, D1 = int(os.environ.get("M", 8000)), 25088, 4096
pad = lambda r, c, p=128: rnd(r, c + p)[:, :c]
X = pad(M, D0, 64); W1 = pad(D1, D0); dZ = pad(M, D1); W1T = pad(D0, D1)
Y = torch.empty(M, D1 + 128, device=dev, dtype=dt)[:, :D1]
dX = torch.empty(M, D0 + 64, device=dev, dtype=dt)[:, :D0]
dW = torch.empty(D1, D0, device=dev)
dZT = torch.empty(D1, M + 64, device=dev, dtype=dt)[:, :M]
ops.transpose_2d(dZ, dZT, M, D1)
fl = 2.0 * M * D0 * D1
for name, f in (("fwd", lambda: ops.gemm(X, W1, Y, M, D1, D0)),
                ("dgrad", lambda: ops.gemm(dZ, W1T, dX, M, D0, D1, ep=ops.make_epilogue(out_dtype=dt))),
                ("wgrad", lambda: ops.gemm(dZT, X, dW, D1, D0, M, b_kstrided=True))):
    t = timeit(f)
    print(f"{os.environ.get('TAG', '-'):8s} M={M} {name:6s} {t*1e3:7.0f} us  {fl/t/1e9:7.0f} TF/s  {fl/t/1e9/2500:.3f}", flush=True)
# ---- the weight gradient + fc1.weight's SGD update: gradient to memory then the optimizer's tiled kernel, vs the fused epilogue (round 6)
w = torch.randn(D1, D0, device=dev) * 0.01; mo = torch.zeros_like(w)
st0 = torch.zeros(D1, D0 + 128, device=dev, dtype=dt)[:, :D0]; st1 = torch.zeros(D0, D1 + 128, device=dev, dtype=dt)[:, :D1]
staging = dict(kind=3, dtype=dt, stage0=st0, stage1=st1, d0=D0, d1=0, d2=0, ld0=st0.stride(0), ld1=st1.stride(0))
ent = dict(param=w, buf=mo, lr=1e-3, weight_decay=5e-4, first=False, staging=staging, hyper=None)
def separate():
    ops.gemm(dZT, X, dW, D1, D0, M, b_kstrided=True)
    ops.sgd_multi([dict(ent, grad=dW)], 0.9, 1.0)
epf = ops.attach_sgd_fused(ops.make_epilogue(out_dtype=torch.float32), ent, 0.9, 1.0)
def fused():
    ops.gemm(dZT, X, dW, D1, D0, M, b_kstrided=True, ep=epf)
for name, f in (("wgrad then sgd_tile_t", separate), ("wgrad with the fused SGD epilogue", fused)):
    t = timeit(f)
    print(f"{os.environ.get('TAG', '-'):8s} M={M} {name:36s} {t*1e3:7.0f} us", flush=True)
