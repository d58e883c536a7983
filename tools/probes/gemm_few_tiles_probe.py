"""Few-tile, long-K GEMMs of the Stage-3 detector (res4 / res5 1x1 convolutions, their data and weight gradients, the box head's fc6
at 512-1024 ROIs): SW_GEMM_DEEP=0 (2-buffer ring) against the default (4-buffer ring when tiles x splits <= 256, K >= 512)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
# (M, N, K, a_kstrided, b_kstrided, f32 out, splitk)
shapes = [(3800, 256, 1024, 0, 0, 0, 1), (7600, 256, 1024, 0, 0, 0, 1), (950, 512, 2048, 0, 0, 0, 1), (1900, 512, 2048, 0, 0, 0, 1),
          (950, 2048, 1024, 0, 0, 0, 1), (950, 256, 2048, 0, 0, 0, 1), (3800, 256, 1024, 0, 1, 0, 1), (7600, 256, 1024, 0, 1, 0, 1),
          (1900, 512, 2048, 0, 1, 0, 1), (950, 512, 2048, 0, 1, 0, 1), (950, 1024, 2048, 0, 1, 0, 1), (1900, 1024, 2048, 0, 1, 0, 1),
          (1024, 1024, 12544, 0, 0, 0, 1), (512, 1024, 12544, 0, 0, 0, 1), (1024, 1024, 1024, 0, 0, 0, 1),
          (1024, 256, 7600, 1, 1, 1, 14), (256, 1024, 7600, 1, 1, 1, 14), (1024, 256, 3800, 1, 1, 1, 7), (2048, 512, 1900, 1, 1, 1, 3),
          (2048, 512, 950, 1, 1, 1, 1), (512, 2048, 950, 1, 1, 1, 1), (1024, 12544, 1024, 1, 1, 1, 1), (1024, 12544, 1024, 0, 1, 0, 1),
          (3800, 1024, 256, 0, 0, 0, 1), (7600, 1024, 256, 0, 0, 0, 1)]
for M, N, K, ak, bk, f32, sk in shapes:
    A = torch.randn((K, M) if ak else (M, K), device=dev).to(dt); B = torch.randn((K, N) if bk else (N, K), device=dev).to(dt)
    od = torch.float32 if f32 else dt
    C = torch.empty(M, N, device=dev, dtype=od)
    us = t(lambda: ops.gemm(A, B, C, M, N, K, bool(ak), bool(bk), ep=ops.make_epilogue(out_dtype=od), splitk=sk))
    ref = (A.float().t() if ak else A.float())[:128] @ (B.float() if bk else B.float().t())
    err = float((C[:128].float() - ref).abs().max() / ref.abs().max())
    print(f"M={M:6d} N={N:6d} K={K:6d} {'T' if ak else 'N'}{'N' if bk else 'T'} sk={sk:2d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  relerr {err:.1e}", flush=True)
