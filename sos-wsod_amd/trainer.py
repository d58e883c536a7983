"""The step loop of the reference's Trainer (uwsod/projects/WSL/tools/train_net_multi.py:66-168) with the same
control flow — skip empty images, forward, sum(losses)/ITER_SIZE, backward, step every ITER_SIZE — minus its
per-iteration host round trips: no .item() on the 9 losses, no gloo gather, no empty_cache().

Data parallelism = the reference's: one process per GPU, torch DistributedDataParallel; on ROCm the "nccl" backend IS
RCCL (xGMI).  Gradients leave the model's autograd nodes in the order they are produced: predictors + fc7 (68 MB), then fc6
(411 MB) right after its weight-gradient GEMM, then — after fc6's data gradient, the ROIPool backward and the conv backward
(~3.3 ms of compute the fc6 all-reduce hides behind) — the backbone's 59 MB (roi_heads_oicrplus._HeadsPoolFunction).

Step rule as the reference wrote it: optimizer.step() when `iter % ITER_SIZE == 0` (:149), the LR scheduler advances every
iteration (its hook), gradients are zeroed at the start iteration and after every step."""
import os
from collections import OrderedDict

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


def auto_scale_workers(cfg, num_workers: int):
    """train_net_multi.py:306-324, restated as written there: when fewer workers than SOLVER.REFERENCE_WORLD_SIZE run the
    recipe, the gradient is accumulated over ceil(ITER_SIZE / scale) iterations and BASE_LR is DIVIDED by scale =
    num_workers / REFERENCE_WORLD_SIZE (< 1, so the learning rate grows); more workers than the reference leave the
    config untouched."""
    import math
    old = cfg.SOLVER.REFERENCE_WORLD_SIZE
    if old == 0 or old == num_workers or old < num_workers:
        return cfg
    cfg = cfg.clone()
    scale = num_workers / old
    cfg.SOLVER.BASE_LR = cfg.SOLVER.BASE_LR / scale
    cfg.WSL.ITER_SIZE = math.ceil(cfg.WSL.ITER_SIZE / scale)
    return cfg


class Trainer:
    """`metrics_period`: every that many iterations the 9 losses are averaged over the ranks by ONE device-side all-reduce
    of the loss vector and stored (as device scalars) in the EventStorage — the reference gathers python floats over gloo
    every iteration (train_loop.py:274 -> comm.gather), i.e. one host sync per step."""

    def __init__(self, model, optimizer, data_iter=None, iter_size=1, scheduler=None, ddp=None, find_unused=False,
                 check_finite_every=20, grad_compress=None, metrics_period=20, bucket_cap_mb=None, start_iter=0,
                 use_graph=None, overlap_update=None):
        self.raw_model = model
        self.optimizer, self.scheduler = optimizer, scheduler
        self.iter_size = max(int(iter_size), 1)
        self.data_iter = data_iter
        self.iter = self.start_iter = int(start_iter)
        self.check_finite_every = check_finite_every
        self.metrics_period = metrics_period
        self.storage = EventStorage(self.iter)
        self._finite_flag = None            # (iteration, device bool) of the last queued check
        self._grad_seed = None
        # hipGraph replay of the whole step (forward + backward + optimizer) for recurring input signatures: see _StepGraphs
        if use_graph is None:
            use_graph = os.environ.get("SW_STEP_GRAPH", "0") == "1"
        self._graphs = None
        self._want_graph = bool(use_graph)
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        prepare = getattr(model, "prepare_for_training", None)
        if callable(prepare):
            prepare()                       # flat predictor master, compute-dtype weight copies: BEFORE DDP looks at the parameters
        use_ddp = ddp if ddp is not None else self.world > 1
        self._bucket_views = []
        self._native = None
        self.overlap_update = False
        # The data-parallel step WITHOUT torch's reducer (round 5; default for the OICR+ model on GPUs with HipSGD, SW_DDP_NATIVE=0
        # keeps DistributedDataParallel): flat gradient buckets the kernels write into, the backward cut into four stages at the
        # heads' node boundaries, one all-reduce per bucket issued between the stages, the update of a bucket queued behind its
        # all-reduce — and, because nothing of the collective lives inside the stages, each stage is a hipGraph.  Round 6: also under
        # gradient accumulation (ITER_SIZE > 1, what auto_scale_workers sets on 1-4 GPUs): micro-steps add into a second flat
        # buffer per bucket and only the stepping iteration all-reduces and updates (_NativeDDP docstring).
        native_ok = (use_ddp and os.environ.get("SW_DDP_NATIVE", "1") == "1" and not find_unused
                     and hasattr(optimizer, "step_params") and hasattr(model, "roi_heads") and hasattr(model, "backbone")
                     and next(model.parameters()).is_cuda and dist.is_available() and dist.is_initialized())
        if native_ok:
            self.model = model
            self._native = _NativeDDP(self, model, optimizer, grad_compress or os.environ.get("SW_DDP_GRAD_COMPRESS"),
                                      use_graph=self._want_graph)
            self.overlap_update = True
        elif use_ddp:
            if os.environ.get("SW_DDP_GRAD_IN_BUCKET", "1") == "1":
                # ops.grad_target users: fc / conv weights and biases (the 20 predictor tensors are row slices of ONE packed gradient
                # matrix; the reducer copies those into its bucket)
                self._bucket_views = [p for n, p in model.named_parameters()
                                      if p.requires_grad and ("backbone." in n or ".box_head." in n)]
            dev = next(model.parameters()).device
            ids = [dev.index] if dev.type == "cuda" else None
            # broadcast_buffers=False as the reference (train_net_multi.py:76-78); every trainable parameter is used
            # each step (REFINE_REG all True), so the unused-parameter scan is off.
            # Buckets: DDP fills them in REVERSE registration order = the order the gradients become ready here — the heads'
            # three backward nodes release predictors + fc7, then fc6, and the backbone node comes last.  A parameter is never
            # split, so fc1.weight (411 MB) is a bucket of its own whatever the cap; its all-reduce starts when the fc6
            # weight-gradient GEMM has finished and overlaps fc6's data gradient, the ROIPool backward and the whole conv
            # backward; bucket_cap_mb (default SW_DDP_BUCKET_MB or 128) only groups the small tensors (predictors + fc7
            # 68 MB, the convs 59 MB) into few launches.
            if bucket_cap_mb is None:
                bucket_cap_mb = float(os.environ.get("SW_DDP_BUCKET_MB", "128"))
            self.model = torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False,
                                                                   find_unused_parameters=find_unused,
                                                                   gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb)
            # opt-in (NOT the reference's numerics): all-reduce the gradient buckets as bf16 — half the bytes on the xGMI
            # ring (the fc6 weight gradient alone is 411 MB per step); the sum is formed in bf16, the result returns as f32
            grad_compress = grad_compress or os.environ.get("SW_DDP_GRAD_COMPRESS")
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            self._seed_scale = 1.0
            if grad_compress:
                assert grad_compress == "bf16", f"unknown gradient compression {grad_compress!r}"
                reduce_hook = default_hooks.bf16_compress_hook
            elif self.iter_size > 1:
                # gradient accumulation: a bucket holds the previous micro-steps' (already averaged) sum plus the new local gradient,
                # and only "divide, then sum over the ranks" leaves the first part unchanged — the stock hook
                reduce_hook = default_hooks.allreduce_hook
            else:
                # mean over the ranks without the averaging pass: DDP's default (and default_hooks.allreduce_hook) divides every
                # bucket by the world size before the all-reduce — a read + write of all 543 MB of gradients per step.  Here the
                # backward is seeded with 1 / world instead, so the gradients leave the kernels already divided (exact for the
                # power-of-two world sizes of a node) and the hook all-reduces with SUM.  (ITER_SIZE 1: a bucket holds this step's
                # gradients only.)
                self._seed_scale = 1.0 / self.world
                pg = dist.group.WORLD

                def reduce_hook(state, bucket, pg=pg):
                    return dist.all_reduce(bucket.buffer(), group=pg, async_op=True).get_future().then(lambda f: f.value()[0])
            # Update as the buckets return (MI355X-first; the reference steps after the whole backward, train_net_multi.py:157-164):
            # the optimizer's kernel for the parameters of a bucket is queued right behind that bucket's all-reduce — fc6's 411 MB
            # (76 % of the bytes, reduced 3.3 ms before the backward ends) are updated while the convolution backward still runs,
            # and only the backbone's 59 MB remain for after the backward.  Same arithmetic, same result bit for bit (the update
            # of a parameter depends on its own reduced gradient only); needs an optimizer with step_params (HipSGD) and
            # ITER_SIZE 1 (with accumulation the reference's rule steps every ITER_SIZE-th iteration only).
            if overlap_update is None:
                overlap_update = os.environ.get("SW_DDP_OVERLAP_UPDATE", "1") == "1"
            # find_unused_parameters: the reducer leaves .grad None for parameters no rank used and the reference's optimizer.step()
            # skips them; bucket.gradients() would still hand the hook zero views (weight decay + momentum applied) — so no overlap
            self.overlap_update = (bool(overlap_update) and hasattr(optimizer, "step_params") and self.iter_size == 1
                                   and not find_unused)
            if self.overlap_update:
                opt = self.optimizer

                def hook(state, bucket, reduce_hook=reduce_hook, opt=opt):
                    def update(fut):
                        v = fut.value()
                        opt.step_params(bucket.parameters(), bucket.gradients())     # views of the (now reduced) bucket buffer
                        return v[0] if isinstance(v, (list, tuple)) else v
                    return reduce_hook(state, bucket).then(update)
                self.model.register_comm_hook(None, hook)
            else:
                self.model.register_comm_hook(None, reduce_hook)
        else:
            self.model = model
        # fc1.weight's SGD update inside its weight-gradient GEMM's epilogue (round 6; sw_epilogue.sgd_fused): single process, ITER_SIZE 1
        # — a data-parallel step needs the all-reduced gradient, an accumulating one the sum of its micro-steps.  SW_FUSE_FC1_UPDATE=0: off
        heads = getattr(model, "roi_heads", None)
        if heads is not None and hasattr(heads, "_fused_fc1_plan"):
            fuse = (not use_ddp and self.iter_size == 1 and hasattr(optimizer, "fused_update_entry")
                    and os.environ.get("SW_FUSE_FC1_UPDATE", "1") == "1")
            heads.__dict__["_fused_opt"] = optimizer if fuse else None
        if self._want_graph and not use_ddp and self.iter_size == 1 and hasattr(model, "roi_heads") and \
                next(model.parameters()).is_cuda:
            self._graphs = _StepGraphs(self)

    def _seed_grad(self, total):
        """cotangent of the summed loss: 1 / ITER_SIZE as a cached device scalar (train_net_multi.py:146 `losses / iter_size`
        without the division kernel and without autograd's ones_like fill)"""
        if self._grad_seed is None or self._grad_seed.device != total.device or self._grad_seed.dtype != total.dtype:
            self._grad_seed = torch.full_like(total, getattr(self, "_seed_scale", 1.0) / self.iter_size)
        return self._grad_seed

    def _raise_if_nonfinite(self):
        if self._finite_flag is not None:
            it, flag = self._finite_flag
            self._finite_flag = None
            if not bool(flag.item()):                                          # train_loop.py:253-259 `_detect_anomaly`
                raise FloatingPointError("Loss became infinite or NaN at iteration={}!".format(it))

    def run_step(self, data=None):
        """train_net_multi.py:112-168"""
        assert self.model.training, "[Trainer] model was changed to eval mode!"
        if data is None:
            data = next(self.data_iter)
            while any(len(x["instances1"]) == 0 for x in data):           # :121-127 skip images without labels
                data = next(self.data_iter)
        if self.iter == self.start_iter:
            self.optimizer.zero_grad()                                     # :143-144
        if self._native is not None:
            loss_dict, losses = self._native.step(data)
        elif self._graphs is not None:
            # every step of a graph-enabled trainer — eager, capture or replay — runs on ONE side stream: a capture cannot use
            # the default stream, and autograd pins each parameter's gradient accumulation to the stream its first forward ran
            # on; an eager warm-up on the default stream followed by a capture elsewhere makes autograd synchronise the two
            # inside the capture (the default stream joins it and never leaves: an unjoined capture, a crash on ROCm)
            caller = torch.cuda.current_stream()
            self._graphs.stream.wait_stream(caller)
            with torch.cuda.stream(self._graphs.stream):
                replayed = self._graphs.step(data) if self.iter > self.start_iter else None
                loss_dict, losses = replayed if replayed is not None else self._forward_backward_update(data)
            caller.wait_stream(self._graphs.stream)
        else:
            loss_dict, losses = self._forward_backward_update(data)
        if self.scheduler is not None:
            self.scheduler.step()                                          # hooks.LRScheduler.after_step: every iteration
        # non-finite losses (train_loop.py:253-259): the reference syncs every iteration; here a device flag is queued every
        # `check_finite_every` iterations and read one check later, when it has long been computed (no pipeline bubble)
        if self.check_finite_every and (self.iter + 1) % self.check_finite_every == 0:
            self._raise_if_nonfinite()
            ff = getattr(loss_dict, "finite_flag", None)
            flag = ff() if callable(ff) else None
            # a SNAPSHOT: under hipGraph replay the flag is a static output of the graph that every later replay overwrites, and
            # the check reads it `check_finite_every` iterations from now
            self._finite_flag = (self.iter, torch.isfinite(losses.detach()).all() if flag is None else flag.detach().clone())
        if self.metrics_period and (self.iter + 1) % self.metrics_period == 0:
            self._write_metrics(loss_dict)
        self.storage.step()
        self.iter += 1
        return loss_dict

    def _forward_backward_update(self, data):
        with self.storage:
            loss_dict = self.model(data)
            # the 9 losses are views of one device vector (LossDict) whose sum is a second output of the heads node: no
            # torch arithmetic between the model and .backward(); a plain dict falls back to the reference's sum()
            total = getattr(loss_dict, "total", None)
            losses = total() if callable(total) else sum(loss_dict.values())
            losses.backward(gradient=self._seed_grad(losses))             # :146-147  (losses / iter_size).backward()
        if self.iter % self.iter_size == 0:                                # :149 — the reference's rule, first step at iter 0
            self.optimizer.step()
            if self._bucket_views:
                # the reducer has re-pointed every .grad at its bucket view: remember them, the next backward writes its weight
                # gradients there (ops.grad_target) and the reducer's copy into the bucket disappears
                for p in self._bucket_views:
                    if p.grad is not None:
                        p.__dict__["_sw_grad_view"] = p.grad
            self.optimizer.zero_grad()
        return loss_dict, losses

    def _write_metrics(self, loss_dict):
        """train_loop.py:261-297: rank-mean of every loss, on the device (one all-reduce of the 9-float vector)"""
        vec = getattr(loss_dict, "vector", None)
        if vec is None:
            vec = torch.stack([v.detach() for v in loss_dict.values()])
        # snapshots, not views: under hipGraph replay the loss vector and its sum are STATIC outputs of the captured graph (every
        # later replay of that graph overwrites them; an EventStorage holding the views would show the last step's values in every
        # history entry) — 11 floats per metrics_period iterations
        vec = vec.detach().clone()
        total = getattr(loss_dict, "total", None)
        tot = total().detach().clone() if callable(total) else None
        if self.world > 1:
            dist.all_reduce(vec)
            vec /= self.world
            tot = None
        with self.storage:
            for i, k in enumerate(loss_dict.keys()):
                self.storage.put_scalar(k, vec[i])
            self.storage.put_scalar("total_loss", vec.sum() if tot is None else tot)
            if ((self._graphs is not None and self._graphs.last_step_replayed)
                    or (self._native is not None and self._native.last_step_replayed)):
                # a replay does not run the heads' Python (which records these in eager steps): the counts are static graph outputs
                aux = getattr(self.raw_model.roi_heads, "last_aux", None) or {}
                for k, r in enumerate(aux.get("rounds", [])):
                    self.storage.put_scalar(f"roi_head/num_pgt_r{k}", r["pgt_count"].detach().clone())

    @staticmethod
    def _graphs_stage(static, data, owner):
        """copy a step's inputs into a captured graph's static input tensors (one launch) and stage the labels"""
        pairs, slow = [], []
        for s_, x in zip(static, data):
            for k in _StepGraphs.VIEW_KEYS:
                for src, dst in ((x["image" + k], s_["image" + k]),
                                 (x["proposals" + k].proposal_boxes.tensor, s_["proposals" + k].proposal_boxes.tensor),
                                 (x["proposals" + k].objectness_logits, s_["proposals" + k].objectness_logits)):
                    (pairs if src.dtype == dst.dtype and src.is_contiguous() else slow).append((src, dst))
        from . import ops
        ops.copy_multi(pairs)
        for src, dst in slow:
            dst.copy_(src, non_blocking=True)
        owner.heads.stage_labels([x["instances1"] for x in data], owner.labels)

    def finish(self):
        """call after the last step: surfaces a pending non-finite flag"""
        self._raise_if_nonfinite()


class _Bucket:
    """One flat float32 gradient buffer; parameter p's gradient is the view `views[i]`.  The weight-gradient kernels write straight
    into the views (ops.grad_target through p._sw_grad_view); gradients produced elsewhere (the predictor tensors: row slices of one
    packed matrix) are gathered with ONE copy launch."""

    def __init__(self, params, device):
        self.params = list(params)
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) & ~3                              # 16-byte aligned segments
        self.flat = torch.zeros(max(n, 4), device=device, dtype=torch.float32)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        for p, v in zip(self.params, self.views):
            p.__dict__["_sw_grad_view"] = v
        self.work = []
        self.acc = None                  # gradient accumulation (ITER_SIZE > 1): the micro-steps' sum so far, allocated on first use
        self.n_acc = 0

    def gather(self):
        from . import ops
        pairs = []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()                                           # (a parameter no kernel produced a gradient for: contributes 0)
            elif g.data_ptr() != v.data_ptr():
                pairs.append((g.contiguous(), v))
        if pairs:
            ops.copy_multi(pairs)
        for p, v in zip(self.params, self.views):
            p.grad = v


class _NativeDDP:
    """Data parallelism of the OICR+ step without torch's reducer (train_net_multi.py:76-78,143-164 is what it replaces).

    Buckets, in the order the backward produces them:  0 = predictors + fc7 (68 MB), 1 = fc6 (411 MB), 2 = backbone (59 MB).
    Stages of a step (main stream), with what is issued between them:
        S1  forward + loss + backward of the heads' top (predictors, fc7, dZ1)        -> all-reduce 0, update 0 behind it
        S2  fc6 weight gradient (optionally in row panels, each all-reduced as it finishes) -> all-reduce 1
        S3  fc6 data gradient + ROIPool backward  (the last readers of fc6's weight copies) -> update 1 behind all-reduce 1 AND S3
        S4  conv backward                                                               -> all-reduce 2, update 2
    The stages are cut with `torch.autograd.backward(..., inputs=...)` at the heads' handle tensors (roi_heads_oicrplus._cuts); the
    collectives are plain eager `dist.all_reduce(async_op=True)` calls BETWEEN stages, the updates run on a side stream — so with
    use_graph every stage is one captured hipGraph (RCCL inside a capture segfaults on this stack; here it never is inside one).
    Gradients are summed with the backward seeded by 1 / world (no averaging pass), as the torch-DDP path does.

    Gradient accumulation (ITER_SIZE > 1; train_net_multi.py:143-164 steps when iter % ITER_SIZE == 0 and never uses no_sync, so the
    reference all-reduces 543 MB on EVERY micro-step): a non-stepping iteration runs the same four stages and adds each
    bucket into its accumulation buffer — no collective, no update; the stepping iteration adds the accumulated sum to its own gradient,
    then all-reduces and updates as above.  Same value as the reference's (mean over ranks of every micro-step, summed) up to the
    order of the f32 additions; same collectives on every rank (the iteration counter decides, not the data)."""

    def __init__(self, trainer, model, optimizer, grad_compress=None, use_graph=False):
        self.tr, self.model, self.opt = trainer, model, optimizer
        self.heads = model.roi_heads
        self.dev = next(model.parameters()).device
        self.world = dist.get_world_size()
        self.group = dist.group.WORLD
        self.compress = grad_compress
        assert grad_compress in (None, "", "bf16"), f"unknown gradient compression {grad_compress!r}"
        trainer._seed_scale = 1.0 / self.world
        fc1 = self.heads.box_head.fc1
        top, fc6, bb = [], [], []
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            (fc6 if (p is fc1.weight or p is fc1.bias) else bb if n.startswith("backbone.") else top).append(p)
        self.buckets = [_Bucket(ps, self.dev) for ps in (top, fc6, bb)]
        self.fc1_weight = fc1.weight
        # DDP's constructor hands rank 0's parameters and buffers to every rank; so does this one
        if self.world > 1:
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, 0, group=self.group)
            from . import ops
            ops.invalidate_all_staged()
        from . import ops as _ops
        self.upd = _ops.worker_stream("upd", self.dev)
        self.main = _ops.worker_stream("main", self.dev) if use_graph else None
        self._dbg_sync = os.environ.get("SW_DDP_DEBUG_SYNC", "0") == "1"          # development switches
        self._dbg_upd_main = os.environ.get("SW_DDP_UPD_MAIN", "0") == "1"
        n_pan = int(os.environ.get("SW_DDP_FC1_PANELS", "0"))
        # row panels of fc1.weight's gradient are all-reduced from inside the eager S2; a rank that REPLAYS its stage graph would issue
        # one all-reduce of the whole bucket instead — ranks pick eager / replay by their own input signature, so with stage graphs
        # (or accumulation: a panel would leave before the accumulated sum is added) the collective layout stays the rank-invariant
        # one: whole buckets
        self.panels = n_pan if (n_pan > 1 and not use_graph and trainer.iter_size == 1) else 0
        self._panel_work = []
        self.use_graph = bool(use_graph)
        self.graphs = OrderedDict()
        self.seen = {}
        self.pool = None
        self.replays = self.captures = 0
        self.labels = torch.zeros(4096, dtype=torch.float32, device=self.dev) if use_graph else None
        if use_graph and hasattr(optimizer, "sync_hyper"):
            optimizer.device_hyper = True
        self._live = None
        self.last_step_replayed = False

    # ------------------------------------------------------------------ the four stages (eager, or inside a capture)
    def _s1(self, data):
        tr = self.tr
        self.heads._staged = True
        try:
            with tr.storage:
                loss_dict = self.model(data)
                total = loss_dict.total()
                total.backward(gradient=tr._seed_grad(total))       # stops at the leaf above the fc6 node
        finally:
            self.heads._staged = False
        self.buckets[0].gather()
        self._live = (loss_dict, total, self.heads._cuts)
        self.heads.__dict__["_cuts"] = None

    def _s2(self):
        h1, h1c = self._live[2][2]
        h1.backward(h1c.grad)                                       # fc6 weight / bias gradient
        self.buckets[1].gather()

    def _s3(self):
        h0, h0c = self._live[2][1]
        h0.backward(h0c.grad)                                       # fc6 data gradient + ROIPool backward: the feature leaves' .grad

    def _s4(self):
        pairs = [(f, c) for f, c in self._live[2][0] if c.grad is not None]
        if pairs:
            torch.autograd.backward([f for f, _ in pairs], [c.grad for _, c in pairs])
        self.buckets[2].gather()

    # ------------------------------------------------------------------ between the stages
    def _allreduce(self, t):
        if self.compress == "bf16":                                # opt-in, NOT the reference's numerics: half the bytes on the ring
            c = t.to(torch.bfloat16)
            c.record_stream(self.upd)                              # consumed by t.copy_(c) on the update stream (_wait)
            w = dist.all_reduce(c, group=self.group, async_op=True)
            return (w, c, t)
        return (dist.all_reduce(t, group=self.group, async_op=True), None, t)

    @staticmethod
    def _wait(job):
        w, c, t = job
        w.wait()                                                   # RCCL: the CURRENT stream waits; gloo: the host does
        if c is not None:
            t.copy_(c)

    def _reduce(self, i):
        b = self.buckets[i]
        if i == 1 and self._panel_work:
            b.work = self._panel_work + [self._allreduce(b.views[k]) for k, p in enumerate(b.params) if p is not self.fc1_weight]
            self._panel_work = []
        else:
            b.work = [self._allreduce(b.flat)]

    def _update(self, i, after=None):
        b = self.buckets[i]
        if self._dbg_sync:
            torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.current_stream() if self._dbg_upd_main else self.upd):
            for job in b.work:
                self._wait(job)
            if after is not None:
                self.upd.wait_event(after)
            self.opt.step_params(b.params, b.views)
        b.work = []

    def _panel_cb(self, i, rows):
        self._panel_work.append(self._allreduce(rows))

    def _finish_bucket(self, i, stepping, after=None):
        """bucket i's stage has run: stepping iteration -> (accumulated sum +) all-reduce, update behind it; else add into the accumulator"""
        b = self.buckets[i]
        if stepping:
            if b.n_acc:
                b.flat.add_(b.acc)
                b.n_acc = 0
            self._reduce(i)
            return True
        if b.acc is None:
            b.acc = torch.empty_like(b.flat)
        if b.n_acc:
            b.acc.add_(b.flat)
        else:
            b.acc.copy_(b.flat)
        b.n_acc += 1
        return False

    # ------------------------------------------------------------------ one step
    def step(self, data):
        caller = torch.cuda.current_stream()
        if self.main is not None:
            # graph mode: every step runs on one side stream (a capture cannot use the default stream; see Trainer.run_step)
            self.main.wait_stream(caller)
            with torch.cuda.stream(self.main):
                out = self._step_on(self.main, data)
            caller.wait_stream(self.main)
            return out
        return self._step_on(caller, data)

    def _step_on(self, main, data):
        tr = self.tr
        self.upd.wait_stream(main)                                  # the previous step's readers of the weights are queued
        hit = self._graph_for(data) if (self.use_graph and tr.iter > tr.start_iter) else None
        self.heads._fc6_panels = (self.panels, self._panel_cb) if (self.panels and hit is None) else None
        if hit is not None:
            g1, g2, g3, g4, static, loss_dict, total = hit
            tr._graphs_stage(static, data, self)
            if hasattr(self.opt, "sync_hyper"):
                self.opt.sync_hyper()
            stages = (g1.replay, g2.replay, g3.replay, g4.replay)
            self.replays += 1
        else:
            stages = (lambda: self._s1(data), self._s2, self._s3, self._s4)
        stepping = tr.iter % tr.iter_size == 0                      # train_net_multi.py:149 (first step at iteration 0)
        try:
            stages[0]()
            if self._finish_bucket(0, stepping):
                self._update(0)
            stages[1]()
            red1 = self._finish_bucket(1, stepping)
            stages[2]()
            if red1:
                e3 = torch.cuda.Event(); e3.record(main)
                self._update(1, after=e3)
            stages[3]()
            if self._finish_bucket(2, stepping):
                self._update(2)
        finally:
            self.heads._fc6_panels = None
            self.heads._prestaged_labels = None
        main.wait_stream(self.upd)
        if hit is None:
            loss_dict, total = self._live[0], self._live[1]
            self._live = None
        if stepping:
            self.opt.step()                                         # (nothing is left for it: every parameter sits in a bucket)
        self.opt.zero_grad()                                        # (.grad are views of the buckets: the next backward writes them anew)
        self.last_step_replayed = hit is not None
        return loss_dict, total

    # ------------------------------------------------------------------ stage graphs
    def _graph_for(self, data):
        sig = _StepGraphs._signature(self, data)
        if sig is None:
            return None
        hit = self.graphs.get(sig)
        if hit is not None:
            self.graphs.move_to_end(sig)
            return hit
        n = self.seen.get(sig, 0) + 1
        self.seen[sig] = n
        if n < 2:
            return None
        for p in self.model.parameters():                           # the update must not create state between the stages
            if p.requires_grad and "momentum_buffer" not in self.opt.state.get(p, {}):
                return None
        while len(self.graphs) >= _StepGraphs.MAX_GRAPHS:
            old, _ = self.graphs.popitem(last=False)
            self.seen[old] = 0
        static = _StepGraphs._clone_inputs(data)
        self.heads.stage_labels([x["instances1"] for x in data], self.labels)
        if hasattr(self.opt, "sync_hyper"):
            self.opt.sync_hyper()
        torch.cuda.synchronize()
        from . import ops
        gs = [torch.cuda.CUDAGraph() for _ in range(4)]
        fns = (lambda: self._s1(static), self._s2, self._s3, self._s4)
        stream = torch.cuda.current_stream()
        try:
            with ops.capture_guard():
                for g, fn in zip(gs, fns):
                    with torch.cuda.graph(g, pool=self.pool, stream=stream):
                        fn()
                    if self.pool is None:
                        self.pool = g.pool()
        finally:
            self.heads._prestaged_labels = None
        loss_dict, total = self._live[0], self._live[1]
        self._live = None
        for b in self.buckets:                                      # the captures only recorded the kernels: no gradient was produced
            for p in b.params:
                p.grad = None
        self.captures += 1
        hit = (gs[0], gs[1], gs[2], gs[3], static, loss_dict, total)
        self.graphs[sig] = hit
        return hit

    # attributes _StepGraphs._signature reads
    VIEW_KEYS = ("1", "1_flip", "2", "2_flip")
    lr_in_signature = False


class _StepGraphs:
    """hipGraph capture / replay of one whole training step (model forward, backward, HipSGD update of the masters and their
    compute-dtype copies): ~250 kernel launches on two streams become ONE graph launch, so the step no longer waits for the
    host (eager: 4-6 ms of Python / ctypes per step against 9-10 ms of GPU work, and the first ~1.5 ms of every step — the
    backbone forward's 40 short kernels — is issue bound).

    A graph is valid for one input SIGNATURE: the four view sizes and the proposal count of every image (buffer shapes), the
    number of image-level classes per image (a kernel argument of the mining kernel) and the optimizer's momentum (a kernel
    argument of the update).  Learning rates and weight decays are NOT part of it: a HipSGD under a graph-enabled trainer keeps
    them in a device buffer the update kernel reads (HipSGD.device_hyper, refreshed by sync_hyper() before a replay when the
    scheduler moved them), so warm-up and LR milestones neither invalidate the captured graphs nor fill the signature table.
    A signature is captured the second time it is seen; other steps run eagerly.  At MAX_GRAPHS the least recently replayed
    graph is dropped (its memory returns to the shared capture pool).  What varies from step to step travels through device memory the graph reads: the images and proposals are copied
    into the graph's static input tensors, the image-level labels into a static label buffer (OICRPlusHeads.stage_labels), and
    the dropout stream position is a device counter the graph itself advances (sw_counter_add).  One image-size bucket of a
    real training run = one graph; a run whose every image has its own size simply never replays."""

    MAX_GRAPHS = 16

    def __init__(self, trainer):
        self.tr = trainer
        self.model = trainer.raw_model
        self.heads = self.model.roi_heads
        self.dev = next(self.model.parameters()).device
        self.seen = {}
        self.graphs = OrderedDict()           # signature -> captured step, least recently used first
        self.pool = None
        self.last_step_replayed = False
        self.evictions = 0
        self.lr_in_signature = not hasattr(trainer.optimizer, "sync_hyper")
        if not self.lr_in_signature:
            trainer.optimizer.device_hyper = True
        self.labels = torch.zeros(4096, dtype=torch.float32, device=self.dev)
        from . import ops as _ops
        self.stream = _ops.worker_stream("main", self.dev)
        self.replays = self.captures = 0
        self.enabled = True                  # False: every step runs eagerly (on the same stream) — measurements, debugging

    VIEW_KEYS = ("1", "1_flip", "2", "2_flip")

    def _signature(self, data):
        sig = []
        for x in data:
            for k in self.VIEW_KEYS:
                im, p = x["image" + k], x["proposals" + k]
                if not (im.is_cuda and p.proposal_boxes.tensor.is_cuda and p.objectness_logits.is_cuda):
                    return None
                sig.append((tuple(im.shape), len(p)))
            g = x["instances1"].gt_classes
            sig.append(int(torch.unique(g.detach().cpu()).numel()))
        if self.lr_in_signature:             # a foreign optimizer: its hyper-parameters are frozen into the capture
            opt = tuple((float(g["lr"]), float(g["weight_decay"]), float(g.get("momentum", 0.0))) for g in self.tr.optimizer.param_groups)
        else:
            opt = tuple(float(g.get("momentum", 0.0)) for g in self.tr.optimizer.param_groups)
        return (tuple(sig), opt, self.model.training)

    @staticmethod
    def _clone_inputs(data):
        from .structures import Boxes, Instances
        out = []
        for x in data:
            d = {}
            for k, v in x.items():
                if k.startswith("image") and isinstance(v, torch.Tensor):
                    d[k] = v.clone()
                elif k.startswith("proposals"):
                    p = Instances(v.image_size)
                    p.proposal_boxes = Boxes(v.proposal_boxes.tensor.clone())
                    p.objectness_logits = v.objectness_logits.clone()
                    d[k] = p
                else:
                    d[k] = v                        # instances*: host-side labels, restaged per step (stage_labels)
            out.append(d)
        return out

    def _stage(self, static, data):
        pairs, slow = [], []
        for s, x in zip(static, data):
            for k in self.VIEW_KEYS:
                for src, dst in ((x["image" + k], s["image" + k]),
                                 (x["proposals" + k].proposal_boxes.tensor, s["proposals" + k].proposal_boxes.tensor),
                                 (x["proposals" + k].objectness_logits, s["proposals" + k].objectness_logits)):
                    (pairs if src.dtype == dst.dtype and src.is_contiguous() else slow).append((src, dst))
        from . import ops
        ops.copy_multi(pairs)                       # one launch for the whole batch (12 tensors per image)
        for src, dst in slow:
            dst.copy_(src, non_blocking=True)
        self.heads.stage_labels([x["instances1"] for x in data], self.labels)

    def step(self, data):
        """-> (loss_dict, total) after replaying (or capturing + replaying) this step's graph, or None: run eagerly"""
        self.last_step_replayed = False
        if not self.enabled:
            return None
        sig = self._signature(data)
        if sig is None:
            return None
        hit = self.graphs.get(sig)
        if hit is None:
            n = self.seen.get(sig, 0) + 1
            self.seen[sig] = n
            if n < 2:
                if len(self.seen) > 4096:
                    self.seen.clear()
                return None
            while len(self.graphs) >= self.MAX_GRAPHS:            # least recently replayed signature makes room
                old_sig, _ = self.graphs.popitem(last=False)
                self.seen[old_sig] = 0                            # it has to recur twice again: with more live size buckets than graphs
                self.evictions += 1                               # the steps run eagerly instead of re-capturing (and evicting) every time
            hit = self._capture(sig, data)
            if hit is None:
                return None
        self.graphs.move_to_end(sig)
        graph, static, loss_dict, losses = hit
        try:
            self._stage(static, data)
            if not self.lr_in_signature:
                self.tr.optimizer.sync_hyper()                    # in stream order before the replay reads the buffer
            graph.replay()
        finally:
            self.heads._prestaged_labels = None
        self.replays += 1
        self.last_step_replayed = True
        return loss_dict, losses

    def _capture(self, sig, data):
        tr = self.tr
        for p in self.model.parameters():                      # the captured optimizer must not create state
            if p.requires_grad and "momentum_buffer" not in tr.optimizer.state.get(p, {}):
                return None
        if not self.lr_in_signature:
            tr.optimizer.sync_hyper()                          # today's values, OUTSIDE the capture (the captured update only reads the buffer)
        static = self._clone_inputs(data)
        self.heads.stage_labels([x["instances1"] for x in data], self.labels)
        torch.cuda.synchronize()
        from . import ops
        graph = torch.cuda.CUDAGraph()
        try:
            with ops.capture_guard(), torch.cuda.graph(graph, pool=self.pool, stream=self.stream):
                loss_dict, losses = tr._forward_backward_update(static)
        finally:
            self.heads._prestaged_labels = None
        if self.pool is None:
            self.pool = graph.pool()
        self.captures += 1
        hit = (graph, static, loss_dict, losses)
        self.graphs[sig] = hit
        return hit
