"""The step loop of the reference's Trainer (uwsod/projects/WSL/tools/train_net_multi.py:66-168) with the same
control flow — skip empty images, forward, sum(losses)/ITER_SIZE, backward, step every ITER_SIZE — minus its
per-iteration host round trips: no .item() on the 9 losses, no gloo gather, no empty_cache().

Data parallelism = the reference's: one process per GPU, torch DistributedDataParallel; on ROCm the "nccl" backend IS
RCCL (xGMI).  Gradients leave the two autograd nodes of this model in two bursts (heads: 92 % of the bytes, first;
backbone afterwards), so DDP's bucketed all-reduce of the fc6/fc7 gradients overlaps the conv backward."""
import os

import torch
import torch.distributed as dist

from .events import EventStorage


def init_distributed(backend=None):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl" and "SW_BENCH_DEVICE" not in os.environ:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class Trainer:
    def __init__(self, model, optimizer, data_iter=None, iter_size=1, scheduler=None, ddp=None, find_unused=False,
                 check_finite_every=0, grad_compress=None):
        self.raw_model = model
        self.optimizer, self.scheduler = optimizer, scheduler
        self.iter_size = max(int(iter_size), 1)
        self.data_iter = data_iter
        self.iter = 0
        self.check_finite_every = check_finite_every
        self.storage = EventStorage(0)
        use_ddp = ddp if ddp is not None else (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if use_ddp:
            dev = next(model.parameters()).device
            ids = [dev.index] if dev.type == "cuda" else None
            # broadcast_buffers=False as the reference (train_net_multi.py:76-78); every trainable parameter is used
            # each step (REFINE_REG all True), so the unused-parameter scan is off
            self.model = torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False,
                                                                   find_unused_parameters=find_unused,
                                                                   gradient_as_bucket_view=True)
            # opt-in (NOT the reference's numerics): all-reduce the gradient buckets as bf16 — half the bytes on the xGMI
            # ring (the fc6 weight gradient alone is 411 MB per step); the sum is formed in bf16, the result returns as f32
            grad_compress = grad_compress or os.environ.get("SW_DDP_GRAD_COMPRESS")
            if grad_compress:
                assert grad_compress == "bf16", f"unknown gradient compression {grad_compress!r}"
                from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
                self.model.register_comm_hook(None, default_hooks.bf16_compress_hook)
        else:
            self.model = model

    def run_step(self, data=None):
        """train_net_multi.py:112-168"""
        assert self.model.training, "[Trainer] model was changed to eval mode!"
        if data is None:
            data = next(self.data_iter)
            while any(len(x["instances1"]) == 0 for x in data):           # :121-127 skip images without labels
                data = next(self.data_iter)
        with self.storage:
            loss_dict = self.model(data)
            # the 9 losses are views of one device vector (LossDict): one sum kernel forward, one expand backward instead
            # of 8 adds + 9 select-backward (zeros + copy + add each); a plain dict falls back to the reference's sum()
            total = getattr(loss_dict, "total", None)
            losses = total() if callable(total) else sum(loss_dict.values())
            if self.iter_size != 1:
                losses = losses / self.iter_size
            losses.backward()
        if (self.iter + 1) % self.iter_size == 0:
            self.optimizer.step()
            self.optimizer.zero_grad()
            if self.scheduler is not None:
                self.scheduler.step()
        if self.check_finite_every and (self.iter + 1) % self.check_finite_every == 0:
            if not torch.isfinite(losses).item():                          # train_loop.py:253-259, made periodic
                raise FloatingPointError("Loss became infinite or NaN at iteration={}!".format(self.iter))
        self.storage.step()
        self.iter += 1
        return loss_dict
