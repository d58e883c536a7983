"""Where do the memcpy / fill / at::native launches of one Stage-3 iteration come from?  torch.profiler over ONE iteration; every aten
op that launches a copy / fill / elementwise kernel is reported with the chain of profiler events that enclose it (autograd nodes,
module-level record_function ranges).  (development tool)"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stage3_step as S
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
step = S.make_step(torch.bfloat16, dev)
batches = S.make_batches(4, 800, 1216, dev)
for i in range(3):
    step.run_step(batches[i])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step.run_step(batches[3])
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::") or not ev.kernels:
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                                   # count the outermost aten op only
    chain, p = [], ev.cpu_parent
    while p is not None and len(chain) < 3:
        chain.append(p.name[:60]); p = p.cpu_parent
    shapes = ",".join(sorted(set(k.name[:40] for k in ev.kernels)))
    if ev.name == "aten::copy_" and "AccumulateGrad" in " ".join(chain):
        shapes += " " + str(ev.input_shapes[:1])
    site = next((f.strip()[-80:] for f in (ev.stack or []) if "sos-wsod_amd" in f or "sos_wsod_amd" in f), "")
    cnt[(ev.name, (" < ".join(chain) or "(top level)") + "  @ " + site, shapes)] += 1
for (name, chain, kern), n in cnt.most_common(70):
    print(f"{n:4d}  {name:16s} in {chain}   [{kern}]")
