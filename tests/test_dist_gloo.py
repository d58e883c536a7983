"""CPU, world_size 2, gloo: the data-parallel step loop (Trainer + DistributedDataParallel, the reference's
train_net_multi.py:76-78,112-168 pattern).  The HIP model cannot run without a GPU, so a small torch module with the
same contract (dict of losses, custom autograd node with explicit backward) stands in; what is under test is the
distributed plumbing: env parsing, DDP wrap, gradient averaging, identical parameters on all ranks after the step,
ITER_SIZE accumulation and the max-over-ranks timing reduction bench.py uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _ExplicitLinear(torch.autograd.Function):
    """one autograd node with a hand-written backward, like the product's heads / backbone functions"""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return g @ w, g.t() @ x, g.sum(0)


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(7)
        self.w = torch.nn.Parameter(torch.randn(5, 8))
        self.b = torch.nn.Parameter(torch.zeros(5))
        self.frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)

    def forward(self, data):
        y = _ExplicitLinear.apply(data[0]["x"], self.w, self.b)
        return {"loss_cls": (y ** 2).mean(), "loss_box_reg_r0": y.abs().mean()}


def _worker(rank, world, port, out, compress=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sos_wsod_amd  # noqa: F401
    from sos_wsod_amd.trainer import Trainer, init_distributed
    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    model = _Toy()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    tr = Trainer(model, opt, iter_size=2, grad_compress=compress)
    assert isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
    tr.model.train()
    g = torch.Generator().manual_seed(100 + rank)
    for it in range(4):
        x = torch.randn(6, 8, generator=g)
        tr.run_step([{"x": x, "instances1": [0]}])
    # identical parameters everywhere
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # bench.py's max-over-ranks time reduction
    if rank == 0:
        torch.save({"same": same, "params": flat, "tmax": float(t.item()), "iters": tr.iter}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_step_loop_world2_gloo(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["same"] and res["tmax"] == 2.0 and res["iters"] == 4
    # single-process reference: mean of the two ranks' gradients each micro-step, step when iter % ITER_SIZE == 0 — the
    # reference's rule (train_net_multi.py:149: iterations 0 and 2 here) — DDP averages every micro-step like the reference,
    # which never uses no_sync
    model = _Toy()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    for it in range(4):
        for r in range(2):
            x = torch.randn(6, 8, generator=gens[r])
            losses = model([{"x": x}])
            (sum(losses.values()) / 2 / 2).backward()        # /ITER_SIZE and /world (DDP mean)
        if it % 2 == 0:
            opt.step(); opt.zero_grad()
    ref = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(res["params"], ref, rtol=1e-5, atol=1e-6)


def test_ddp_bf16_gradient_compression_opt_in(tmp_path):
    """opt-in bf16 all-reduce of the gradient buckets (Trainer(grad_compress="bf16") / SW_DDP_GRAD_COMPRESS): ranks stay in
    lock step, the result is the f32 run's up to bf16 rounding of the summed gradients"""
    out = str(tmp_path / "r0c.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, "bf16"), nprocs=2, join=True)
    res = torch.load(out)
    assert res["same"] and res["iters"] == 4
    out2 = str(tmp_path / "r0f.pt")
    mp.spawn(_worker, args=(2, _free_port(), out2), nprocs=2, join=True)
    ref = torch.load(out2)
    assert not torch.equal(res["params"], ref["params"])                      # the hook did run
    assert torch.allclose(res["params"], ref["params"], rtol=2e-2, atol=2e-3)
