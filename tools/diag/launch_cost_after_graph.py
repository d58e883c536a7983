"""debug: does the host cost of a launch / an allocation change after a torch.cuda.graph capture + replay in the process?"""
import os, sys, time, gc, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import sos_wsod_amd.ops as ops
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
def cost(tag):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(2000): ops.fill_zero(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(2000): y = torch.empty(4096, device=dev)
    t3 = time.perf_counter()
    for _ in range(2000): x.add_(1.0)
    t4 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{tag:28s} custom launch {(t1 - t) / 2000 * 1e6:5.1f} us (drain {(t2 - t1) * 1e3:.1f} ms)  torch.empty {(t3 - t2) / 2000 * 1e6:5.1f} us  torch add_ {(t4 - t3) / 2000 * 1e6:5.1f} us", flush=True)
cost("fresh process")
g = torch.cuda.CUDAGraph()
s = torch.zeros(1 << 20, device=dev)
with torch.cuda.graph(g):
    for _ in range(10): s.add_(1.0)
cost("after a capture")
for _ in range(5): g.replay()
torch.cuda.synchronize()
cost("after replays")
del g; gc.collect(); torch.cuda.empty_cache()
cost("graph deleted")
