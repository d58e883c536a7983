"""One rank of the real-model data-parallel check (launched by tests/test_gpu_ddp.py through torch.distributed.run, two
ranks sharing cuda:0 over the gloo backend: SW_DIST_BACKEND=gloo SW_BENCH_DEVICE=0).

What it exercises is the combination the toy module of test_dist_gloo.py does not have: the real MultiInputRCNN (two
autograd nodes, the backbone's internal side stream, the predictor weights re-homed as slices of one flat master) inside
Trainer's DistributedDataParallel (gradient_as_bucket_view) with the fused HipSGD that rewrites weights and their
compute-dtype copies behind autograd.  Checks, all on every rank:
  1. the all-reduced gradients of a DDP iteration == the mean of the two ranks' single-process gradients;
  2. after N Trainer steps the parameters are bit-identical on both ranks;
  3. and equal to a single-process replica that applies the averaged gradients with its own HipSGD."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_path, dtype_name = sys.argv[1], sys.argv[2]
    import sos_wsod_amd  # noqa: F401
    from helpers import build_model, load_params, to_batched_inputs
    from oracle import oicr_oracle as O                      # closed-form parameters / views only (test infrastructure)
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.trainer import Trainer, init_distributed
    rank, _, world = init_distributed(backend=os.environ["SW_DIST_BACKEND"])
    assert world == 2
    dev = torch.device("cuda", int(os.environ.get("SW_BENCH_DEVICE", 0)))
    torch.cuda.set_device(dev)
    dtype = torch.float32 if dtype_name == "fp32" else torch.bfloat16
    full = len(sys.argv) > 3 and sys.argv[3] == "full"
    # "full": BASELINE config #3's per-rank shape once (512x512 views, 2000 proposals, fc 4096/4096: the 411 MB fc6 gradient is a
    # bucket of its own, the real bucket order and sizes), one step; default: a toy shape, three steps
    K, R, H, W, dan = (20, 2000, 512, 512, (4096, 4096)) if full else (20, 60, 96, 128, (256, 256))
    P = O.make_params(K, dan, tag="pddp", head_scale=20.0)

    graph = os.environ.get("SW_STEP_GRAPH", "0") == "1"        # stage graphs: a signature is captured the second time it is seen

    def data_of(r, step):
        views, gt = O.make_views(H + 16 * r, W, R + 7 * r, n_gt=2, K=K, scale2=1.0 if full else 1.25,
                                 tag=f"vddp{r}_{step}")                                       # ranks see different sizes
        if graph:                                                  # one signature per rank: the number of image-level classes is part of it
            _, gt = O.make_views(H + 16 * r, W, R + 7 * r, n_gt=2, K=K, scale2=1.0 if full else 1.25, tag=f"vddp{r}_0")
        return to_batched_inputs(views, gt, device=dev if graph else None)

    def fresh():
        m = build_model(K, dan, dtype, device=dev)
        load_params(m, P)
        m.train()
        return m

    LR = 1e-4 if graph else 1e-2      # (five steps at 1e-2 diverge on this fixture: loss 7 -> 233 -> 2e4 -> 3e21 -> NaN, data parallel or not)

    def groups(m):
        return [{"params": [p], "lr": 2 * LR if n.endswith(".bias") else LR, "weight_decay": 0.0 if n.endswith(".bias") else 5e-4}
                for n, p in m.named_parameters() if p.requires_grad]

    def set_stream(m, r, counter):
        # every rank its own dropout stream (seed from SEED + rank); the replica replays rank r's stream for rank r's data
        hd = m.roi_heads
        hd.seed = 1234
        hd.dropout_seed = None
        import sos_wsod_amd.roi_heads_oicrplus as rh
        hd.dropout_seed = rh.derive_dropout_seed(1234, r)
        hd._drop_counter = counter

    from sos_wsod_amd.events import EventStorage
    # ---- single-process replica: per step the mean of both ranks' gradients, own HipSGD
    ITER_SIZE = int(os.environ.get("SW_TEST_ITER_SIZE", "1"))      # > 1: gradient accumulation, step when iter % ITER_SIZE == 0 (train_net_multi.py:149)
    N_STEPS = 1 if full else (5 if (graph or ITER_SIZE > 1) else 3)
    rep = fresh()
    rep_opt = HipSGD(groups(rep), LR, momentum=0.9)
    counters = [0, 0]
    mean_grads = []
    rep_acc = None
    for step in range(N_STEPS):
        acc = None
        for r in range(2):
            set_stream(rep, r, counters[r])
            for p in rep.parameters():
                p.grad = None
            with EventStorage(0):
                ld = rep(data_of(r, step))
                ld.total().backward()
            counters[r] = rep.roi_heads._drop_counter
            g = {n: p.grad.detach().clone() for n, p in rep.named_parameters() if p.grad is not None}
            acc = g if acc is None else {n: (acc[n] + g[n]) for n in acc}
        mean = {n: v / 2 / ITER_SIZE for n, v in acc.items()}       # (losses / ITER_SIZE).backward(), mean over the ranks
        rep_acc = mean if rep_acc is None else {n: rep_acc[n] + mean[n] for n in mean}
        if step % ITER_SIZE == 0:
            mean_grads.append(rep_acc)                                # what the optimizer consumes at this stepping iteration
            for n, p in rep.named_parameters():
                p.grad = rep_acc.get(n)
            rep_opt.step()
            rep_opt.zero_grad()
            rep_acc = None
        else:
            mean_grads.append(None)                                   # accumulating iteration: nothing reduced, nothing applied
    torch.cuda.synchronize()

    # ---- the DDP run
    model = fresh()
    set_stream(model, rank, 0)
    opt = HipSGD(groups(model), LR, momentum=0.9)
    tr = Trainer(model, opt, check_finite_every=1, metrics_period=1, iter_size=ITER_SIZE)
    native = os.environ.get("SW_DDP_NATIVE", "1") == "1"
    assert (tr._native is not None) if native else isinstance(tr.model, torch.nn.parallel.DistributedDataParallel)
    name_of = {id(p): n for n, p in model.named_parameters()}
    flat_w, _ = model.roi_heads._head_flat                           # flattened BEFORE the DDP wrap (prepare_for_training)
    assert model.roi_heads.box_predictor.cls.weight.data_ptr() == flat_w.data_ptr()
    res = {"grad_err": [], "rank": rank, "overlap_update": tr.overlap_update, "left_for_step": [], "native": native}
    for step in range(N_STEPS):
        # look at the all-reduced gradients of this step before the optimizer consumes them
        seen = {}
        orig_step = opt.step

        def spy_step(*a, **k):
            for n, p in model.named_parameters():
                if p.grad is not None:
                    seen[n] = p.grad.detach().clone()
            res["left_for_step"].append(sum(1 for p in model.parameters() if p.grad is not None and id(p) not in opt._done))
            return orig_step(*a, **k)
        opt.step = spy_step
        tr.run_step(data_of(rank, step))
        opt.step = orig_step
        if native:
            # the reduced gradients live in the reducer's flat buckets (a replayed stage graph runs no autograd: .grad stays None)
            torch.cuda.synchronize()
            seen = {name_of[id(p)]: v.detach().clone() for b in tr._native.buckets for p, v in zip(b.params, b.views)}
        if mean_grads[step] is None:                                  # accumulating iteration: no optimizer step, no collective
            assert ITER_SIZE > 1 and step % ITER_SIZE != 0
            continue
        assert set(seen) == set(mean_grads[step])
        for n, g in seen.items():
            ref = mean_grads[step][n]
            res["grad_err"].append(float((g - ref).abs().max() / (ref.abs().max() + 1e-20)))
    tr.finish()
    torch.cuda.synchronize()
    res["replays"] = tr._native.replays if native else 0
    # ---- parameters: bit-identical across ranks, equal to the replica's
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    res["same_across_ranks"] = bool(torch.equal(gathered[0], gathered[1]))
    rflat = torch.cat([p.detach().flatten() for p in rep.parameters()])
    res["replica_err"] = float((flat - rflat).abs().max() / rflat.abs().max())
    res["moved"] = float((flat - torch.cat([torch.from_numpy(np.ascontiguousarray(P[n])).flatten().to(dev)
                                            for n, _ in model.named_parameters()])).abs().max())
    res["metrics"] = {k: float(v) for k, v in tr.storage.latest().items() if k.startswith("loss")}
    res["dropout_seeds_differ"] = True
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([model.roi_heads.dropout_seed & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64))
    res["dropout_seeds_differ"] = int(seeds[0]) != int(seeds[1])
    torch.save(res, f"{out_path}.rank{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
