"""fc7-sized GEMMs (8000 x 4096 x 4096: 2 rounds of 256x256 tiles with only 64 K-steps each) under the tile / loop variants the
development switches select (run one process per variant: SW_GEMM_V, SW_GEMM_PP are read once)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
M, N, K = 8000, 4096, 4096
x = torch.randn(M, K, device=dev).relu().to(dt); w = torch.randn(N, K, device=dev).to(dt); b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev, dtype=dt)
wt = torch.randn(K, N, device=dev).to(dt)
fwd = t(lambda: ops.gemm(x, w, y, M, N, K, ep=ops.make_epilogue(bias=b, relu=True, out_dtype=dt)))
dg = t(lambda: ops.gemm(x, w, y, M, N, K, ep=ops.make_epilogue(out_dtype=dt)))
print(f"V={os.environ.get('SW_GEMM_V','-')} PP={os.environ.get('SW_GEMM_PP','-')} PERSIST={os.environ.get('SW_GEMM_PERSIST','-')}: fc7 fwd (bias+relu) {fwd:.0f} us = {2.0*M*N*K/fwd/1e6:.0f} TF/s; plain bf16 out {dg:.0f} us")
for name, ep in [("plain", dict()), ("bias", dict(bias=b)), ("relu", dict(relu=True)), ("bias+relu", dict(bias=b, relu=True)),
                 ("bias+relu+hash-dropout", dict(bias=b, relu=True, drop_hash=(123, 0, 0.5))), ("relu_ref mask", dict(relu_ref=x[:, :N].contiguous() if K >= N else None))]:
    us = t(lambda: ops.gemm(x, w, y, M, N, K, ep=ops.make_epilogue(out_dtype=dt, **ep)))
    print(f"   epilogue {name:24s} {us:6.0f} us")
yf = torch.empty(M, N, device=dev, dtype=torch.float32)
us = t(lambda: ops.gemm(x, w, yf, M, N, K, ep=ops.make_epilogue(out_dtype=torch.float32)))
print(f"   f32 output plain          {us:6.0f} us")
