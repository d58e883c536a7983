"""Orientation only (not a product path): what the vendor BLAS behind torch.mm reaches at the fc6 / fc7 GEMM shapes of the step on
this GPU, next to sw_gemm's figures in the bench line.  bf16 in, bf16 / f32 out."""
import torch, time
dev = "cuda"
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for name, M, N, K, tb in [("fc6 fwd  X(8000,25088) @ W^T(4096,25088)", 8000, 4096, 25088, True),
                          ("fc6 dgrad dY(8000,4096) @ W(4096,25088)", 8000, 25088, 4096, False),
                          ("fc6 wgrad dY^T(4096,8000) @ X(8000,25088)", 4096, 25088, 8000, None),
                          ("fc7 fwd  (8000,4096) @ (4096,4096)^T", 8000, 4096, 4096, True)]:
    if tb is None:
        A = torch.randn(K, M, device=dev, dtype=torch.bfloat16); B = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: torch.mm(A.t(), B)
    elif tb:
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); B = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        fn = lambda: torch.mm(A, B.t())
    else:
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); B = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: torch.mm(A, B)
    ms = t(fn)
    print(f"{name:48s} {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
