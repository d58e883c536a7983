"""Which GEMMs does one Stage-3 iteration launch, and how fast is each?  ops.gemm is wrapped to log (M, N, K, operand forms,
epilogue) for ONE iteration; every distinct shape is then timed alone (20 launches between two events) with the same operand
forms.  Output: per shape count x time, bytes, GB/s, TFLOP/s, sorted by time per iteration.  (development tool, GPU only)"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
import stage3_step as S

dev = torch.device("cuda", 0)
step = S.make_step(torch.bfloat16, dev)
batches = S.make_batches(4, 800, 1216, dev)
for i in range(3):
    step.run_step(batches[i])
log = collections.Counter()
real = ops.gemm


def spy(A, B, C, M, N, K, a_kstrided=False, b_kstrided=False, lda=None, ldb=None, ldc=None, ep=None, splitk=1, tag=None):
    e = ep
    key = (M, N, K, bool(a_kstrided), bool(b_kstrided), str(A.dtype)[6:], str(C.dtype)[6:], splitk,
           bool(e is not None and e.bias), bool(e is not None and e.relu), bool(e is not None and e.residual), bool(e is not None and e.relu_ref))
    log[key] += 1
    return real(A, B, C, M, N, K, a_kstrided, b_kstrided, lda, ldb, ldc, ep, splitk, tag)


ops.gemm = spy
import sos_wsod_amd.frcnn as F
if hasattr(F, "ops"):
    F.ops.gemm = spy
step.run_step(batches[3])
torch.cuda.synchronize()
ops.gemm = real


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


rows = []
for key, cnt in log.items():
    M, N, K, ak, bk, adt, cdt, sk, bias, relu, res, ref = key
    dt = torch.bfloat16 if adt == "bfloat16" else torch.float32
    cd = torch.bfloat16 if cdt == "bfloat16" else torch.float32
    A = torch.randn((K, M) if ak else (M, K), device=dev).to(dt)
    B = torch.randn((K, N) if bk else (N, K), device=dev).to(dt)
    C = torch.empty(M, N, device=dev, dtype=cd)
    kw = dict(out_dtype=cd)
    if bias: kw["bias"] = torch.randn(N, device=dev)
    if relu: kw["relu"] = True
    if res: kw["residual"] = torch.randn(M, N, device=dev).to(cd)
    if ref: kw["relu_ref"] = torch.randn(M, N, device=dev).to(cd)
    us = timeit(lambda: real(A, B, C, M, N, K, ak, bk, ep=ops.make_epilogue(**kw), splitk=sk))
    es = 2 if dt == torch.bfloat16 else 4
    by = es * (M * K + N * K) + C.element_size() * M * N * (2 if res else 1) + (C.element_size() * M * N if ref else 0)
    rows.append((cnt * us, cnt, us, key, by))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{sum(r[1] for r in rows)} GEMM launches per iteration, {tot / 1e3:.2f} ms when each runs alone")
for t, cnt, us, key, by in rows[:60]:
    M, N, K, ak, bk, adt, cdt, sk, bias, relu, res, ref = key
    print(f"{t:8.1f} us = {cnt:3d} x {us:7.1f}  M={M:7d} N={N:5d} K={K:7d} {'T' if ak else 'N'}{'N' if bk else 'T'} {adt[:4]}->{cdt[:4]} sk={sk:2d}"
          f" {'b' if bias else '-'}{'r' if relu else '-'}{'+' if res else '-'}{'m' if ref else '-'}  {by / 1e6:7.1f} MB {by / us / 1e3:7.1f} GB/s {2.0 * M * N * K / us / 1e6:7.1f} TF")
