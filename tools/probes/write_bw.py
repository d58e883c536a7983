"""pure HBM write / copy bandwidth of torch fill / copy kernels on 411 MB (the size of fc6's weight gradient)"""
import torch
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
x = torch.empty(4096, 25088, device=dev); y = torch.empty_like(x)
t = timeit(lambda: x.fill_(1.0)); print(f"fill 411 MB: {t*1e3:.0f} us = {x.numel()*4/t/1e9:.2f} TB/s")
t = timeit(lambda: x.zero_()); print(f"zero 411 MB: {t*1e3:.0f} us = {x.numel()*4/t/1e9:.2f} TB/s")
t = timeit(lambda: y.copy_(x)); print(f"copy 411 MB: {t*1e3:.0f} us = {2*x.numel()*4/t/1e9:.2f} TB/s (read+write)")
t = timeit(lambda: x.sum()); print(f"read 411 MB: {t*1e3:.0f} us = {x.numel()*4/t/1e9:.2f} TB/s")
