"""Does the steady-state training step still call hipMalloc / hipFree (caching-allocator misses)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer
dev = torch.device("cuda", 0)
model = bench.build(dev, torch.bfloat16); model.train()
opt = HipSGD([{"params": [p], "lr": 1e-3, "weight_decay": 5e-4} for p in model.parameters() if p.requires_grad], 1e-3, momentum=0.9)
tr = Trainer(model, opt)
data = [bench.make_inputs(dev, 1), bench.make_inputs(dev, 2)]
def stats():
    s = torch.cuda.memory_stats()
    return {k: s[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current", "allocated_bytes.all.peak")}
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.run_step(data[i % 2])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = stats()
    cur = torch.cuda.memory_allocated() / 2**30
    import gc
    print(f"step {i}: {dt*1e3:7.2f} ms  cur_alloc={cur:.2f} GiB gc={gc.get_count()} device_alloc={s['num_device_alloc']} device_free={s['num_device_free']} retries={s['num_alloc_retries']} reserved={s['reserved_bytes.all.current']/2**30:.2f} GiB peak_alloc={s['allocated_bytes.all.peak']/2**30:.2f} GiB")
