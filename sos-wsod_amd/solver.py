"""Solver of the OICR+ step loop (needed inside the measured step):
per-parameter SGD groups exactly as the reference builds them (uwsod/detectron2/solver/build.py:143-218:
bias lr x BIAS_LR_FACTOR, WEIGHT_DECAY_BIAS for biases), torch.optim.SGD update semantics executed by the
HIP kernel sw_sgd_momentum_step, and WarmupMultiStepLR (solver/lr_scheduler.py:13-58)."""
from bisect import bisect_right
from typing import List

import torch

from . import ops


class HipSGD(torch.optim.Optimizer):
    """SGD + momentum + weight decay; state and hyper-parameters follow torch.optim.SGD so that schedulers and
    checkpoints interoperate.  One fused elementwise launch per parameter tensor (no host sync)."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, device_hyper=False):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        # device_hyper: every group's (lr, weight_decay) lives in a small device buffer the update kernel reads, refreshed by
        # sync_hyper() when a scheduler changed them — a captured hipGraph of the step is then independent of the schedule
        self.device_hyper = bool(device_hyper)
        self._hyper_dev = None
        self._hyper_host = None
        self._done = set()                # ids of the parameters step_params() updated since the last step()
        self._group_of = None

    def sync_hyper(self):
        """bring the device copy of every group's (lr, weight_decay) up to date (one small async copy, only after a change);
        call it in stream order before a replay of a captured step"""
        if not self.device_hyper:
            return
        vals = [(float(g["lr"]), float(g["weight_decay"])) for g in self.param_groups]
        if self._hyper_dev is None:
            dev = next(p.device for g in self.param_groups for p in g["params"])
            self._hyper_dev = torch.zeros(len(vals), 2, dtype=torch.float32, device=dev)
            self._hyper_host = None
        if vals != self._hyper_host:
            host = torch.tensor(vals, dtype=torch.float32)
            if self._hyper_dev.is_cuda:
                host = host.pin_memory()
            self._hyper_dev.copy_(host, non_blocking=True)
            self._hyper_host = vals

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        """One launch (sw_sgd_multi) per 24 parameter tensors.  Parameters whose module registered compute-dtype staging
        copies (ops.STAGING) get those rewritten from the updated values in the same pass.  Parameters already updated in this
        iteration by step_params() (the data-parallel trainer's per-bucket update) are skipped."""
        done, self._done = self._done, set()
        self._update([(group, gi, p, p.grad) for gi, group in enumerate(self.param_groups) for p in group["params"]
                      if p.grad is not None and id(p) not in done], grad_scale)

    @torch.no_grad()
    def step_params(self, params, grads=None, grad_scale=1.0):
        """The same update for a SUBSET of the parameters, now — the data-parallel trainer calls it from DistributedDataParallel's
        communication hook as soon as a gradient bucket has been all-reduced, so that e.g. fc6's 411 MB are updated while the
        convolution backward still runs; the step() that follows the backward then skips them.  grads: the reduced gradients
        (default: each parameter's .grad)."""
        if self._group_of is None:
            self._group_of = {id(p): (group, gi) for gi, group in enumerate(self.param_groups) for p in group["params"]}
        items = []
        for i, p in enumerate(params):
            g = p.grad if grads is None else grads[i]
            ent = self._group_of.get(id(p))
            if ent is None or g is None or id(p) in self._done:
                continue
            items.append((ent[0], ent[1], p, g.view_as(p)))
            self._done.add(id(p))
        self._update(items, grad_scale)

    @torch.no_grad()
    def fused_update_entry(self, p):
        """For a weight-gradient GEMM that applies this parameter's update in its epilogue (ops.attach_sgd_fused; round 6, fc1.weight on
        the single-GPU path): -> (entry, momentum, finish) or None when the parameter cannot take it (no tiled staging registered).
        `entry` is what sgd_multi would get minus the gradient; call finish() after the launch (epoch / staging stamps, as _update does).
        The parameter must not receive a .grad this iteration (step() then has nothing to do for it)."""
        if self._group_of is None:
            self._group_of = {id(q): (group, gi) for gi, group in enumerate(self.param_groups) for q in group["params"]}
        ent = self._group_of.get(id(p))
        staging = ops.STAGING.get(id(p))
        if (ent is None or staging is None or staging["param"]() is not p or staging["kind"] != 3 or not self._staging_usable(staging, p)
                or p.dtype != torch.float32 or not p.is_contiguous()):
            return None
        group, gi = ent
        if self.device_hyper and (self._hyper_dev is None or not torch.cuda.is_current_stream_capturing()):
            self.sync_hyper()
        st = self.state[p]
        first = "momentum_buffer" not in st
        if first:
            st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
        entry = dict(param=p, buf=st["momentum_buffer"], lr=group["lr"], weight_decay=group["weight_decay"], first=first,
                     staging=staging, hyper=self._hyper_dev[gi] if self.device_hyper else None)

        def finish():
            ops.PARAM_EPOCH += 1
            ops.mark_updated(p)
            if staging["stamp"] is not None:
                staging["stamp"](ops.param_key(p))
        return entry, float(group["momentum"]), finish

    def _update(self, items, grad_scale):
        if not items:
            return
        ops.PARAM_EPOCH += 1
        by_mom = {}
        if self.device_hyper and (self._hyper_dev is None or not torch.cuda.is_current_stream_capturing()):
            self.sync_hyper()                  # (inside a capture the copy would freeze today's values into the graph)
        for group, gi, p, grad in items:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("HipSGD updates contiguous float32 master parameters")
            g = grad if grad.is_contiguous() else grad.contiguous()
            st = self.state[p]
            first = "momentum_buffer" not in st
            if first:
                st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
            staging = ops.STAGING.get(id(p))
            if staging is not None and (staging["param"]() is not p or not self._staging_usable(staging, p)):
                staging = None
            by_mom.setdefault(float(group["momentum"]), []).append(
                dict(param=p, grad=g, buf=st["momentum_buffer"], lr=group["lr"], weight_decay=group["weight_decay"],
                     first=first, staging=staging, hyper=self._hyper_dev[gi] if self.device_hyper else None))
        for mom, entries in by_mom.items():
            ops.sgd_multi(entries, mom, grad_scale)
            for e in entries:
                ops.mark_updated(e["param"])
            for e in entries:
                if e["staging"] is not None and e["staging"]["stamp"] is not None:
                    e["staging"]["stamp"](ops.param_key(e["param"]))

    @staticmethod
    def _staging_usable(st, p):
        for t in (st["stage0"], st["stage1"]):
            if t is not None and t.device != p.device:
                return False
        return True

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)


def build_optimizer(cfg, model) -> torch.optim.Optimizer:
    """solver/build.py:143-218: one group per parameter; biases get BASE_LR*BIAS_LR_FACTOR and WEIGHT_DECAY_BIAS; with
    SOLVER.REFINE_SCALE_ON every parameter whose name contains "refine" (the box_refinery_k heads) additionally gets
    lr x REFINE_LR_SCALE (:162-188).  VGG16 + fc heads hold no norm layers, so WEIGHT_DECAY_NORM never applies;
    NESTEROV must be False (the fused update is plain momentum SGD)."""
    assert not cfg.SOLVER.get("NESTEROV", False), "SOLVER.NESTEROV True is not implemented by HipSGD"
    refine_on = bool(cfg.SOLVER.get("REFINE_SCALE_ON", False))
    refine_scale = float(cfg.SOLVER.get("REFINE_LR_SCALE", 1.0))
    groups = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
        is_bias = ("bias" in name) if refine_on else name.endswith(".bias")     # :176-185 substring test / :204 key == "bias"
        if is_bias:
            lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
            wd = cfg.SOLVER.WEIGHT_DECAY_BIAS
        if refine_on and "refine" in name:
            lr = lr * refine_scale
        groups.append({"params": [p], "lr": lr, "weight_decay": wd})
    return HipSGD(groups, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


class WarmupMultiStepLR(torch.optim.lr_scheduler.LRScheduler):
    def __init__(self, optimizer, milestones: List[int], gamma=0.1, warmup_factor=0.001, warmup_iters=1000,
                 warmup_method="linear", last_epoch=-1):
        assert list(milestones) == sorted(milestones)
        self.milestones, self.gamma = list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def _warmup(self, it):
        if it >= self.warmup_iters:
            return 1.0
        if self.warmup_method == "constant":
            return self.warmup_factor
        alpha = it / self.warmup_iters
        return self.warmup_factor * (1 - alpha) + alpha

    def get_lr(self):
        w = self._warmup(self.last_epoch)
        return [b * w * self.gamma ** bisect_right(self.milestones, self.last_epoch) for b in self.base_lrs]


def build_lr_scheduler(cfg, optimizer):
    return WarmupMultiStepLR(optimizer, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA, warmup_factor=cfg.SOLVER.WARMUP_FACTOR,
                             warmup_iters=cfg.SOLVER.WARMUP_ITERS, warmup_method=cfg.SOLVER.WARMUP_METHOD)
