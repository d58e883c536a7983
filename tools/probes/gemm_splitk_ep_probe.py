"""bf16-output GEMMs with an epilogue (bias + residual + ReLU) as ONE launch against split-K slabs + the fold that applies the
epilogue (sw_gemm with splitk > 1 and a non-plain epilogue), on the few-tile shapes of the Stage-3 detector."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
shapes = [(3800, 256, 1024, 0), (7600, 256, 1024, 0), (950, 512, 2048, 0), (1900, 512, 2048, 0), (950, 2048, 1024, 0), (950, 2048, 512, 0),
          (3800, 1024, 256, 0), (3800, 256, 1024, 1), (7600, 256, 1024, 1), (1900, 512, 2048, 1), (950, 512, 2048, 1), (950, 1024, 2048, 1),
          (1024, 1024, 12544, 0), (512, 1024, 12544, 0), (1024, 1024, 1024, 0), (1024, 12544, 1024, 1)]
for M, N, K, bk in shapes:
    A = torch.randn(M, K, device=dev).to(dt); B = torch.randn((K, N) if bk else (N, K), device=dev).to(dt)
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).to(dt)
    C = torch.empty(M, N, device=dev, dtype=dt)
    ref = torch.relu(A.float() @ (B.float() if bk else B.float().t()) + bias + res.float())
    line = f"M={M:6d} N={N:6d} K={K:6d} N{'N' if bk else 'T'}:"
    for sk in (1, 2, 4, 8, 16):
        if K // sk < 128: continue
        us = t(lambda: ops.gemm(A, B, C, M, N, K, False, bool(bk), ep=ops.make_epilogue(bias=bias, relu=True, residual=res, out_dtype=dt), splitk=sk))
        err = float((C.float() - ref).abs().max() / ref.abs().max())
        line += f"  sk={sk}: {us:6.1f} us ({err:.0e})"
    print(line, flush=True)
