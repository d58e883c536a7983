"""world-1 RCCL: the native data-parallel trainer (stage graphs) against the plain eager trainer, step by step"""
import os, sys, socket
import torch, torch.distributed as dist
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    import sos_wsod_amd  # noqa
    from helpers import build_model, load_params, to_batched_inputs
    from oracle import oicr_oracle as O
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.trainer import Trainer
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    K, R, H, W, dan = 20, 60, 96, 128, (256, 256)
    P = O.make_params(K, dan, tag="prccl", head_scale=20.0)
    _, gt0 = O.make_views(H, W, R, n_gt=2, K=K, scale2=1.25, tag="vrccl_0")
    def data_of(step):
        views, _ = O.make_views(H, W, R, n_gt=2, K=K, scale2=1.25, tag=f"vrccl_{step}")
        return to_batched_inputs(views, gt0, device=dev)
    def fresh():
        m = build_model(K, dan, torch.float32, device=dev); load_params(m, P); m.train(); m.roi_heads.seed = 4321; return m
    def groups(m):
        return [{"params": [p], "lr": 2e-4 if n.endswith(".bias") else 1e-4, "weight_decay": 0.0 if n.endswith(".bias") else 5e-4}
                for n, p in m.named_parameters() if p.requires_grad]
    ma, mb = fresh(), fresh()
    ta = Trainer(ma, HipSGD(groups(ma), 1e-2, momentum=0.9), ddp=True, use_graph=True, check_finite_every=0, metrics_period=0)
    tb = Trainer(mb, HipSGD(groups(mb), 1e-2, momentum=0.9), ddp=False, use_graph=False, check_finite_every=0, metrics_period=0)
    for step in range(8):
        la = ta.run_step(data_of(step)); lb = tb.run_step(data_of(step))
        torch.cuda.synchronize()
        worst = max((float((a.detach() - b.detach()).abs().max()), n) for (n, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()))
        gb = [float(b.flat.abs().max()) for b in ta._native.buckets]
        print(f"step {step}: replayed={ta._native.last_step_replayed} loss native {float(la.total()):.6f} plain {float(lb.total()):.6f} "
              f"worst param diff {worst[0]:.3e} ({worst[1]}) bucket |max| {gb}", flush=True)
    dist.destroy_process_group()

main()
