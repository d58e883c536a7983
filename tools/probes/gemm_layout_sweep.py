"""bf16 GEMM rate by operand layout at one neutral shape (M=N=K=8192): which operand's K-strided staging costs what."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
S = int(os.environ.get("S", 8192))
A = (torch.randn(S, S, device=dev) * .5).to(dt); B = (torch.randn(S, S, device=dev) * .5).to(dt)
for odt in (dt, torch.float32):
    C = torch.empty(S, S, device=dev, dtype=odt)
    for ak in (False, True):
        for bk in (False, True):
            if ak and not bk: continue
            t = timeit(lambda: ops.gemm(A, B, C, S, S, S, a_kstrided=ak, b_kstrided=bk, ep=ops.make_epilogue(out_dtype=odt)))
            print(f"out={str(odt)[6:]:8s} A {'K-strided' if ak else 'K-contig '}  B {'K-strided' if bk else 'K-contig '}  {t*1e3:7.0f} us  {2.0*S**3/t/1e9:6.0f} TF")
