"""1x1-convolution-shaped GEMMs (many pixels, few channels: memory bound) on the 256x256 ping-pong tile against the 128x128 one
(SW_GEMM_V=8 forces the small tile).  y (P, N) = x (P, K) @ W^T (N, K), bf16, bias + ReLU epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for P, N, K in [(121600, 256, 64), (121600, 256, 256), (60800, 256, 256), (30400, 512, 128), (30400, 512, 256), (30400, 256, 512), (7600, 1024, 256), (7600, 1024, 512), (182400, 256, 256), (1024, 1024, 12544), (1536, 1024, 12544)]:
    x = torch.randn(P, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt); b = torch.randn(N, device=dev)
    y = torch.empty(P, N, device=dev, dtype=dt)
    us = t(lambda: ops.gemm(x, w, y, P, N, K, ep=ops.make_epilogue(bias=b, relu=True, out_dtype=dt)))
    mb = (P * K + N * K + P * N) * 2 / 1e6
    print(f"P={P:7d} N={N:5d} K={K:6d}: {us:7.1f} us  {2.0*P*N*K/us/1e6:7.1f} TFLOP/s  {mb/us*1e-3*1e3:7.2f} GB/s x1e3 ({mb:.0f} MB)", flush=True)
