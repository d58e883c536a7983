"""The data-parallel step on an RCCL process group of world size 1 (one GPU) at BASELINE config #2's shape: N eager DDP steps,
for a kernel trace (what does a rank launch per step beyond the single-process step?) and for timing.
    python tools/ddp_world1_step.py [steps] [ddp: 1|0]
    rocprofv3 --kernel-trace --stats -d gpurun_out/ddp -- python tools/ddp_world1_step.py 20 1
(no shebang on purpose: under rocprofv3 the interpreter itself must follow `--`; an env / shell hop would exec after the profiler's
preloaded library has initialised the GPU, which this pool forbids)"""
import os, sys, time, socket, datetime
import torch, torch.distributed as dist
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from sos_wsod_amd.solver import HipSGD
from sos_wsod_amd.trainer import Trainer

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ddp = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    if ddp:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    batches = [bench.make_inputs(dev, 100 + 2 * i) for i in range(2)]
    m = bench.build(dev, torch.bfloat16); m.train()
    gs = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
          for nm, p in m.named_parameters() if p.requires_grad]
    tr = Trainer(m, HipSGD(gs, 1e-3, momentum=0.9), ddp=ddp, use_graph=os.environ.get("USE_GRAPH", "0") == "1")
    for i in range(8):
        tr.run_step(batches[i % 2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        tr.run_step(batches[i % 2])
    torch.cuda.synchronize()
    print(f"{'DDP over RCCL world 1' if ddp else 'single process'}, {'graphs' if os.environ.get('USE_GRAPH', '0') == '1' else 'eager'}, upd_main={os.environ.get('SW_DDP_UPD_MAIN', '0')}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", file=sys.stderr)
    tr.finish()
    if ddp:
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
