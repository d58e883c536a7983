"""Throughput of the Stage-1 OICR+ training step on MI355X (BASELINE.json metric: images/s, VGG16+OICR, 2000 proposals).

One "step" = one OICR+ iteration per GPU = 4 views (2 scales x {orig, h-flip}) of one image:
u8 image -> VGG16 -> ROIPool(2000 proposals) -> fc6/fc7 -> WSDDN + 4 OICR refinement heads with device-side
pseudo-label mining -> 9 losses -> full backward (backbone from plain3 up, FREEZE_AT 2) -> (RCCL gradient
all-reduce when N > 1) -> SGD-momentum update of all 135.8 M trainable parameters.
"image" = one view, so value = 4 * N * steps / time.

    python bench.py [--gpus N --steps K --warmup W]        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 with the driver's contract plus `roofline` (dominant kernel, timed live with HIP
events on the launch stream) and `cpu_baseline` (the oracle, timed on this host's cores, rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
H = W = 512
R = 2000
K = 20
DAN = (4096, 4096)


def _drop():
    """after deleting a trainer / model of an extra run: collect NOW (a step graph in a reference cycle must not be finalized by a
    collection that happens to run inside a later capture: ops.capture_guard), then hand the memory back"""
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def make_inputs(device, seed, H=None, W=None, R=None, K=None, n_gt=2, scale2=1.0):
    """Synthetic VOC-shaped 4-view input: u8 images, proposals sorted by objectness, flipped views mirror x.
    Sizes default to the module constants (BASELINE config #2); config #4 = make_inputs(dev, s, 800, 1333, 4000, 80).
    scale2: size of the second scale's views relative to the first (the reference always draws two DIFFERENT sizes,
    dataset_mapper.py:303-317; BASELINE config #2 names one size, so the headline keeps 1.0 and `mixed_ms_per_step` uses 1.25)."""
    from sos_wsod_amd.structures import Boxes, Instances
    H = globals()["H"] if H is None else H; W = globals()["W"] if W is None else W
    R = globals()["R"] if R is None else R; K = globals()["K"] if K is None else K
    g = torch.Generator().manual_seed(seed)
    d = {}
    x1 = torch.rand(R, generator=g) * (W - 32); y1 = torch.rand(R, generator=g) * (H - 32)
    bw = 24 + torch.rand(R, generator=g) * (W - x1 - 24); bh = 24 + torch.rand(R, generator=g) * (H - y1 - 24)
    boxes = torch.stack([x1, y1, torch.minimum(x1 + bw, torch.tensor(float(W))), torch.minimum(y1 + bh, torch.tensor(float(H)))], 1)
    obj = torch.sort(torch.rand(R, generator=g), descending=True).values
    gt = torch.unique(torch.randint(0, K, (n_gt,), generator=g))
    H1, W1, boxes1 = H, W, boxes
    for scale in ("1", "2"):
        if scale == "2" and scale2 != 1.0:
            H, W = int(H1 * scale2 + 0.5), int(W1 * scale2 + 0.5)
            boxes = boxes1 * torch.tensor([W / W1, H / H1, W / W1, H / H1])
        img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
        for flip in ("", "_flip"):
            b = boxes.clone()
            im = img
            if flip:
                b[:, 0], b[:, 2] = W - boxes[:, 2], W - boxes[:, 0]
                im = img.flip(-1).contiguous()
            p = Instances((H, W)); p.proposal_boxes = Boxes(b.to(device)); p.objectness_logits = obj.to(device)
            t = Instances((H, W)); t.gt_boxes = Boxes(torch.zeros(len(gt), 4)); t.gt_classes = gt      # stays on host
            d["image" + scale + flip] = im.to(device)
            d["proposals" + scale + flip] = p
            d["instances" + scale + flip] = t
    return [d]


def build(device, dtype, K=None, freeze_at=2):
    K = globals()["K"] if K is None else K
    from sos_wsod_amd.backbone_vgg import VGG16
    from sos_wsod_amd.box_head import DiscriminativeAdaptionNeck
    from sos_wsod_amd.fast_rcnn_oicr import OICROutputLayers
    from sos_wsod_amd.fast_rcnn_wsddn import WSDDNOutputLayers
    from sos_wsod_amd.poolers import ROIPooler
    from sos_wsod_amd.rcnn_multi import MultiInputRCNN
    from sos_wsod_amd.roi_heads_oicrplus import OICRPlusHeads
    from sos_wsod_amd.structures import ShapeSpec
    torch.manual_seed(1234)
    with torch.device(device):
        backbone = VGG16(conv5_dilation=2, freeze_at=freeze_at, out_features=["plain5"], compute_dtype=dtype)
        pooler = ROIPooler(output_size=7, scales=(1.0 / 8,), sampling_ratio=0, pooler_type="ROIPool")
        head = DiscriminativeAdaptionNeck(ShapeSpec(channels=512, height=7, width=7), conv_dims=[], fc_dims=list(DAN),
                                          compute_dtype=dtype)
        pred = WSDDNOutputLayers(head.output_shape, num_classes=K)
        refs = [OICROutputLayers(head.output_shape, num_classes=K, refine_k=k, refine_reg=[True] * 4) for k in range(4)]
        heads = OICRPlusHeads(box_in_features=["plain5"], box_pooler=pooler, box_head=head, box_predictor=pred, refine_K=4,
                              refine_mist=True, mist_p=0.10, mist_thre=0.05, mist_type="nms", refine_reg=[True] * 4,
                              box_refinery=refs, num_classes=K, compute_dtype=dtype)
        model = MultiInputRCNN(backbone=backbone, roi_heads=heads, pixel_mean=[103.939, 116.779, 123.68],
                               pixel_std=[1.0, 1.0, 1.0])
    return model.to(device)


def _cpu_params(K_):
    """random-init parameters of the benchmarked architecture under the reference's state-dict names (numpy, fp32)"""
    from oracle import oicr_oracle as O
    g = torch.Generator().manual_seed(0)
    P = {}
    for stage, cin, cout, nconv, _, _ in O.VGG_CFG:
        for i in range(nconv):
            ci = cin if i == 0 else cout
            P[f"backbone.{stage}.0.conv{i + 1}.weight"] = (torch.randn(cout, ci, 3, 3, generator=g) * (2.0 / (cout * 9)) ** 0.5).numpy()
            P[f"backbone.{stage}.0.conv{i + 1}.bias"] = np.zeros(cout, np.float32)
    d_in = 25088
    for i, d in enumerate(DAN):
        P[f"roi_heads.box_head.fc{i + 1}.weight"] = (torch.randn(d, d_in, generator=g) * 0.005).numpy()
        P[f"roi_heads.box_head.fc{i + 1}.bias"] = np.full(d, 0.1, np.float32)
        d_in = d
    for n in ("cls", "det"):
        P[f"roi_heads.box_predictor.{n}.weight"] = (torch.randn(K_, d_in, generator=g) * 0.02).numpy()
        P[f"roi_heads.box_predictor.{n}.bias"] = np.zeros(K_, np.float32)
    for k in range(4):
        P[f"roi_heads.box_refinery_{k}.cls_score.weight"] = (torch.randn(K_ + 1, d_in, generator=g) * 0.01).numpy()
        P[f"roi_heads.box_refinery_{k}.cls_score.bias"] = np.zeros(K_ + 1, np.float32)
        P[f"roi_heads.box_refinery_{k}.bbox_pred.weight"] = (torch.randn(4 * K_, d_in, generator=g) * 0.001).numpy()
        P[f"roi_heads.box_refinery_{k}.bbox_pred.bias"] = np.zeros(4 * K_, np.float32)
    return P, g


def _cpu_time_config(P, g, R_, scale2, n_warm, n_timed, tag):
    """SURVEY §8d / BASELINE.md §4 protocol: `n_warm` untimed iterations (page faults of the 544 MB weight set, thread-pool start),
    then the MEDIAN of `n_timed` timed ones, each a full forward + backward of one 4-view OICR+ iteration through the oracle, with the
    seconds of every stage (backbone / ROIPool / fc6+fc7 / heads) of the median run"""
    from oracle import oicr_oracle as O
    views, gt = O.make_views(H, W, R_, n_gt=2, K=K, scale2=scale2, tag=tag)
    masks = [[(torch.rand(R_, d, generator=g) >= 0.5).numpy().astype(np.uint8) for d in DAN] for _ in range(4)]
    runs = []
    for i in range(n_warm + n_timed):
        t0 = time.perf_counter()
        _, aux, _ = O.oicr_plus_iteration(P, views, gt, masks, K=K, want_grads=True)
        dt = time.perf_counter() - t0
        if i >= n_warm:
            runs.append((dt, aux["timing"]))
    runs.sort(key=lambda r: r[0])
    dt, tm = runs[len(runs) // 2]
    rd = lambda d: {k: round(v, 2) for k, v in d.items()}
    return {"value": round(4.0 / dt, 4), "iteration_s": round(dt, 2), "fwd_s": round(tm["fwd_s"], 2), "bwd_s": round(tm["bwd_s"], 2),
            "fwd_stage_s": rd(tm["fwd_stage_s"]), "bwd_stage_s": rd(tm["bwd_stage_s"]),
            "timed_iterations": len(runs), "warmup_iterations": n_warm, "all_iteration_s": [round(r[0], 2) for r in runs]}


def cpu_baseline():
    """The oracle's restatement of the same step (fp32, torch-CPU contractions + the serial C ROIPool the reference's CPU file states),
    at BASELINE config #2's shape (the one `value` is quoted on) and at config #1's (the reference's own CPU-runnable case)."""
    threads = torch.get_num_threads()
    P, g = _cpu_params(K)
    c2 = _cpu_time_config(P, g, R, 1.0, 2, 5, "cpubase")
    c1 = _cpu_time_config(P, g, 500, 1.25, 1, 3, "cpubase1")      # (the CPU-runnable case, for the record: 1 + 3 keeps the default run near 2.5 min)
    out = {"value": c2["value"], "unit": "images/s", "cores": threads, "kind": "port"}
    out.update({k: v for k, v in c2.items() if k != "value"})
    out["config1"] = dict(c1, unit="images/s", shape=f"views {H}x{W} + {int(H * 1.25 + 0.5)}x{int(W * 1.25 + 0.5)}, R=500, K={K}, fp32")
    out["sample"] = (f"median of {c2['timed_iterations']} full OICR+ iterations after {c2['warmup_iterations']} warm-ups (4 views {H}x{W}, R={R}, "
                     f"K={K}, fp32) forward+backward through the oracle (torch-CPU contractions with {threads} threads; ROIPool = the serial C "
                     f"loop of the reference's CPU file), no optimizer step; {c2['iteration_s']:.1f} s per iteration = {c2['fwd_s']:.1f} s forward "
                     f"+ {c2['bwd_s']:.1f} s backward; per-stage seconds in fwd_stage_s / bwd_stage_s; config #1's shape under `config1`")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--images-per-gpu", type=int, default=1,
                    help="images (4-view sets) per GPU per step; BASELINE config #3 = --gpus 8 --images-per-gpu 2")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the short fp32 timing (extra key fp32_ms_per_step)")
    ap.add_argument("--no-extra-shapes", action="store_true",
                    help="skip the short runs of config #3's / #4's / #5's per-GPU shapes (extra keys b2_ms_per_step, coco_ms_per_step, stage3_ms_per_iter)")
    ap.add_argument("--graph", type=int, default=None,
                    help="1: replay the step as a captured hipGraph (default for --gpus 1), 0: eager launches (default under DDP)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (launch.py:55-73 does mp.spawn).  Fresh children
        # through torch.distributed.run; this process has not touched the GPU and only waits for them.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)

    # stdout carries ONE line, the JSON record.  Libraries write to file descriptor 1 behind Python's back (RCCL prints a
    # five-line version banner there when its first communicator comes up): from here on descriptor 1 is stderr, and the record
    # goes to the saved descriptor at the end.  (After the self-launch above: the children must inherit the real stdout.)
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    t_wall = time.perf_counter()

    def phase(name):                      # stderr only: where a bench run's wall time goes
        print(f"[bench +{time.perf_counter() - t_wall:6.1f}s] {name}", file=sys.stderr, flush=True)

    import sos_wsod_amd  # noqa: F401  (fails loudly if the HIP extension is missing)
    import sos_wsod_amd.ops as ops
    from sos_wsod_amd.solver import HipSGD
    from sos_wsod_amd.trainer import Trainer, init_distributed
    import torch.distributed as dist

    rank, local_rank, world = init_distributed(backend=os.environ.get("SW_DIST_BACKEND"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run for N>1)"
    assert torch.cuda.is_available(), "bench.py measures the HIP path; there is no CPU fallback"
    device = torch.device("cuda", int(os.environ.get("SW_BENCH_DEVICE", local_rank)))   # override: DDP smoke test on 1 GPU
    torch.cuda.set_device(device)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    phase("imports + process group up")
    model = build(device, dtype)
    model.train()
    groups = []
    for name, p in model.named_parameters():
        if p.requires_grad:                      # voc07_oicr_plus.yaml:36-45 solver values
            groups.append({"params": [p], "lr": 2e-3 if name.endswith(".bias") else 1e-3,
                           "weight_decay": 0.0 if name.endswith(".bias") else 5e-4})
    opt = HipSGD(groups, 1e-3, momentum=0.9)
    # single GPU: the whole step is one hipGraph; data parallel (trainer._NativeDDP): four stage graphs with the eager RCCL all-reduces
    # between them (round 6: that mode is `value` for N > 1 too; the launch-by-launch time is the extra `eager_ms_per_step`)
    use_graph = True if args.graph is None else bool(args.graph)
    trainer = Trainer(model, opt, use_graph=use_graph)
    B = args.images_per_gpu
    # which (synthetic) images a rank sees: its shard of the shared-seed permutation stream, as the reference's TrainingSampler deals a
    # dataset out (distributed_sampler.py:38-55; sos-wsod_amd/samplers.py) — rank r takes indices[r::world] of one VOC07-sized stream
    from sos_wsod_amd.samplers import TrainingSampler
    mine = TrainingSampler(5011, shuffle=True, seed=1234, rank=rank, world_size=world).take(2 * B)
    batches = [sum((make_inputs(device, 100 + mine[i * B + b]) for b in range(B)), []) for i in range(2)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graphs = trainer._graphs
    native = trainer._native
    if graphs is not None or (native is not None and native.use_graph):    # "compile" phase: every input signature is captured before any timing
        for i in range(6):
            trainer.run_step(batches[i % 2])
    for i in range(args.warmup):
        trainer.run_step(batches[i % 2])
    tags = ["fc6_fwd", "fc6_dgrad", "fc6_wgrad", "fc6_wgrad_sgd", "plain5.conv3_fwd", "roi_fwd", "roi_bwd", "wgrad_grouped"]
    sync()
    phase("model built, graphs captured, warm-up done")
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.run_step(batches[i % 2])
    sync()
    dt = time.perf_counter() - t0
    phase("timed region done")
    # per-kernel durations need HIP events between launches: a graph replay has no place for them, and in an eager run they are
    # instrumentation (~0.3 ms of stream time per step) — so the same kernels are timed in a few eager steps AFTER the measured
    # region (same shapes, same data, every rank alike so that the collectives stay in step)
    replays = captures = None
    n_timer_steps = 8
    eager_ms = dt / args.steps * 1e3
    if graphs is not None or (native is not None and native.use_graph):     # the same step issued launch by launch, for the record
        if graphs is not None:
            replays, captures = graphs.replays, graphs.captures
            graphs.enabled = False               # same trainer, same stream, no replay
        else:
            replays, captures = native.replays, native.captures
            native.use_graph = False             # every rank alike: the collectives stay in step
        trainer.run_step(batches[0])
        sync(); t1 = time.perf_counter()
        for i in range(n_timer_steps):
            trainer.run_step(batches[i % 2])
        sync()
        eager_ms = (time.perf_counter() - t1) / n_timer_steps * 1e3
    ops.TIMER = ops.KernelTimer(tags)
    sync(); t1 = time.perf_counter()
    for i in range(n_timer_steps):
        trainer.run_step(batches[i % 2])
    sync()
    instrumented_ms = (time.perf_counter() - t1) / n_timer_steps * 1e3
    times = ops.TIMER.summary_ms()
    timer_work = dict(ops.TIMER.work)
    ops.TIMER = None
    fused_update = bool(times.get("fc6_wgrad_sgd")) and not times.get("fc6_wgrad")
    if fused_update:
        # the single-GPU step runs fc6's weight gradient with fc1.weight's SGD update in its epilogue (tag fc6_wgrad_sgd: GEMM + 2 GB of
        # optimizer traffic).  The roofline of the GEMM itself is taken from the same launches WITHOUT the fused update: a few more eager
        # steps with the fusion off (the optimizer's own kernel then updates fc1.weight; same training arithmetic)
        hd_ = model.roi_heads
        keep_, hd_.__dict__["_fused_opt"] = hd_.__dict__.get("_fused_opt"), None
        ops.TIMER = ops.KernelTimer(["fc6_wgrad"])
        for i in range(4):
            trainer.run_step(batches[i % 2])
        times["fc6_wgrad"] = ops.TIMER.summary_ms()["fc6_wgrad"]
        ops.TIMER = None
        hd_.__dict__["_fused_opt"] = keep_
    rank_ms = [dt / args.steps * 1e3]
    rccl_ranks = 1
    ranks_in_sync = None
    if world > 1:
        # self-check of the data-parallel step (train_net_multi.py:76-78: DDP keeps the replicas identical; the hand-rolled reducer has to
        # prove it): an exact integer checksum of every parameter's bits after all the steps above, MIN == MAX over the ranks
        with torch.no_grad():
            cs = torch.zeros(1, dtype=torch.int64, device=device)
            for p_ in model.parameters():
                cs += p_.detach().contiguous().view(torch.int32).to(torch.int64).sum()
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ranks_in_sync = bool((lo == hi).item())
        # self-check of the record: the number of ranks that really took part (an all-reduce of ones over the data-path
        # communicator) and every rank's own time for the K steps; `value` uses the MAX over ranks as the contract says
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones.item())))
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(e.item()) / args.steps * 1e3 for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        M = 4 * R * B
        flops = {"fc6_fwd": 2.0 * M * 25088 * DAN[0], "fc6_dgrad": 2.0 * M * 25088 * DAN[0], "fc6_wgrad": 2.0 * M * 25088 * DAN[0],
                 "plain5.conv3_fwd": 2.0 * (2 * 63 * 63) * 512 * 4608}
        avg_ms = {t: (sum(v) / len(v) if v else None) for t, v in times.items()}
        tot_ms = {t: sum(v) / n_timer_steps for t, v in times.items()}
        if fused_update:
            tot_ms["fc6_wgrad"] = avg_ms["fc6_wgrad"]                   # (one launch per step; timed over 4 extra steps, not n_timer_steps)
        dom = max((t for t in ("fc6_fwd", "fc6_dgrad", "fc6_wgrad")), key=lambda t: tot_ms[t])
        peak = MFMA_BF16_DENSE_PEAK_TFLOPS if dtype == torch.bfloat16 else 157.3

        def roof(tag):
            a = flops[tag] / (avg_ms[tag] * 1e-3) / 1e12
            return {"kernel": tag, "bound": "mfma", "achieved": round(a, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(a / peak, 4), "traffic": None, "avg_ms": round(avg_ms[tag], 4),
                    "flop_per_launch": flops[tag]}
        # conv5_3 forward on its own (inside the step it shares the CUs with the other scale's stream, which is what the step
        # wants but not what "MFMA utilisation of the conv5_3 kernel" means): 20 launches, HIP events on the launch stream
        x5 = (torch.randn(2, 63, 63, 512, device=device) * 0.5).to(dtype)
        wk5 = (torch.randn(512, 9, 512, device=device) * 0.02).to(dtype)
        b5 = torch.zeros(512, device=device); o5 = torch.empty_like(x5)
        ep5 = ops.make_epilogue(bias=b5, relu=True, out_dtype=dtype)
        for _ in range(3):
            ops.conv3x3(x5, wk5, o5, 2, ep5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv3x3(x5, wk5, o5, 2, ep5)
        e1.record(); torch.cuda.synchronize()
        conv_alone_ms = e0.elapsed_time(e1) / 20
        # HBM bytes per call from the committed rocprofv3 --pmc passes (tools/pmc_traffic.sh): (2*FETCH_SIZE + WRITE_SIZE)*1024
        pmc = next((p for p in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"))
                    if os.path.exists(p)), None)
        traffic = json.load(open(pmc)) if pmc else {}
        roofline = roof(dom)
        entry = traffic.get(dom)
        roofline["traffic"] = entry["hbm_bytes_per_launch"] if entry else None
        roofline["algorithmic_bytes"] = entry["algorithmic_bytes"] if entry else None
        roofline["traffic_source"] = os.path.basename(pmc) if pmc else None

        def roof_hbm(tag, alg_bytes, what):
            a = alg_bytes / (avg_ms[tag] * 1e-3) / 1e9
            ent = traffic.get(tag) or {}
            return {"kernel": what, "bound": "hbm", "achieved": round(a, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(a / 8000.0, 4),
                    "traffic": ent.get("hbm_bytes_per_launch"), "algorithmic_bytes": alg_bytes, "avg_ms": round(avg_ms[tag], 4)}
        es = 2 if dtype == torch.bfloat16 else 4
        fmap = 2 * 63 * 63 * 512 * es                                  # the 2-image feature map of one scale
        roi_rows = 2 * R                                               # one call = view + flipped view of one scale
        # counter-based MFMA utilisation of the same kernels run alone (tools/pmc_mfma.sh, committed under profiles/):
        # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32).  The FLOP-based `frac` prices against the 2.4 GHz peak;
        # under MFMA load the chip runs 1.6-2.0 GHz (DVFS), so the pipe is busier than `frac` says.
        busy_path = next((p for p in (os.path.join(ROOT, "profiles", n) for n in ("r06_mfma_busy.json", "r05_mfma_busy.json", "r04_mfma_busy.json", "r03_mfma_busy.json", "r02_mfma_busy.json", "r01_mfma_busy.json"))
                          if os.path.exists(p)), None)
        busy = json.load(open(busy_path)) if busy_path else {}

        def busy_of(shape):
            ent = busy.get(shape) or {}
            if not ent:
                return None
            k = max(ent, key=lambda n: ent[n]["SQ_VALU_MFMA_BUSY_CYCLES"])       # the main launch (not a split-K tail)
            return {"mfma_busy_frac": ent[k]["mfma_busy_frac"], "effective_clock_GHz": ent[k]["effective_clock_GHz"],
                    "source": f"profiles/{os.path.basename(busy_path)} (rocprofv3 --pmc, kernel alone)"}
        roofline["mfma_busy_counter"] = busy_of(dom)
        out = {
            "metric": "images/s (1/2/4/8 MI355X) VGG16+OICR 2000-prop; conv5_3 MFMA-util %",
            "value": round(4.0 * B * world * args.steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"OICR+ training iteration: {4 * B} views/step/GPU ({B} image(s) x 2 scales x h-flip, every view {H}x{W}), "
                                   f"R={R} proposals/view, K={K}, VGG16 dilated-C5 + ROIPool 7x7 + fc6/fc7 4096 + WSDDN + 4 OICR heads, "
                                   "fwd+bwd+SGD(momentum, wd)", "views_per_step_per_gpu": 4 * B, "image": "one view",
                       "oicr_iterations_per_s": round(B * world * args.steps / dt, 3), "parallelism": f"dp{world}"},
            "rccl_ranks": rccl_ranks, "backend": (dist.get_backend() if world > 1 else None), "ranks_in_sync": ranks_in_sync,
            "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
            "roofline": roofline,
            # the other two launches of the same GEMM family (`roofline` above is the costliest of the three)
            "roofline_fc6": {t: dict({k: v for k, v in roof(t).items() if k in ("achieved", "frac", "avg_ms")},
                                     mfma_busy_counter=(busy_of(t) or {}).get("mfma_busy_frac"),
                                     traffic=(traffic.get(t) or {}).get("hbm_bytes_per_launch"),
                                     algorithmic_bytes=(traffic.get(t) or {}).get("algorithmic_bytes"),
                                     traffic_over_algorithmic=(round(traffic[t]["hbm_bytes_per_launch"] / traffic[t]["algorithmic_bytes"], 2)
                                                               if t in traffic else None))
                             for t in ("fc6_fwd", "fc6_dgrad", "fc6_wgrad")},
            "roofline_conv5_3": dict(roof("plain5.conv3_fwd"), note="inside the step: two streams share the CUs"),
            # every conv weight gradient of the backward pass: ONE launch of the direct weight-gradient kernel (round 6)
            "roofline_wgrad_grouped": ({"kernel": "conv_wgrad_direct_kernel: all conv weight gradients of the step, one launch (folds not included)",
                                        "bound": "mfma", "peak": peak, "unit": "TFLOP/s",
                                        "flop_per_launch": timer_work["wgrad_grouped"] / max(1, len(times["wgrad_grouped"])),
                                        "avg_ms": round(avg_ms["wgrad_grouped"], 4),
                                        "achieved": round(timer_work["wgrad_grouped"] / (sum(times["wgrad_grouped"]) * 1e-3) / 1e12, 2),
                                        "frac": round(timer_work["wgrad_grouped"] / (sum(times["wgrad_grouped"]) * 1e-3) / 1e12 / peak, 4),
                                        "traffic": (traffic.get("wgrad_grouped") or {}).get("hbm_bytes_per_launch"),
                                        "algorithmic_bytes": (traffic.get("wgrad_grouped") or {}).get("algorithmic_bytes"),
                                        "mfma_busy_counter": busy_of("wgrad_grouped")}
                                       if times.get("wgrad_grouped") else None),
            "roofline_conv5_3_alone": {"kernel": "conv5_3 fwd, batch 2, 63x63, 512->512, dilation 2", "bound": "mfma",
                                       "traffic": (traffic.get("conv5_3") or {}).get("hbm_bytes_per_launch"),
                                       "algorithmic_bytes": (traffic.get("conv5_3") or {}).get("algorithmic_bytes"),
                                       "achieved": round(flops["plain5.conv3_fwd"] / (conv_alone_ms * 1e-3) / 1e12, 2),
                                       "peak": peak, "unit": "TFLOP/s",
                                       "frac": round(flops["plain5.conv3_fwd"] / (conv_alone_ms * 1e-3) / 1e12 / peak, 4),
                                       "avg_ms": round(conv_alone_ms, 4), "mfma_busy_counter": busy_of("conv5_3")},
            # ROIPool (HBM-bound: the pooled rows + argmax dominate the bytes): one call = 2 x R ROIs on one scale's 2-image map
            "roofline_roipool": {
                "fwd": roof_hbm("roi_fwd", roi_rows * 25088 * (es + 2) + fmap, f"roi_pool_fwd: {roi_rows} ROIs x 512 x 7 x 7, values + u16 argmax written"),
                "bwd": roof_hbm("roi_bwd", roi_rows * 25088 * (es + 2) + 2 * fmap, f"roi_pool_bwd: {roi_rows} ROIs, gradients + argmax read, map written")},
            "kernel_ms_per_step": {t: round(v, 3) for t, v in tot_ms.items()},
            "fused_fc1_update": ({"on": True, "fc6_wgrad_with_sgd_epilogue_ms": round(avg_ms["fc6_wgrad_sgd"], 4),
                                  "note": "fc1.weight's SGD update runs in the epilogue of fc6's weight-gradient GEMM (sw_epilogue.sgd_fused): the "
                                          "411 MB gradient is never written; `roofline` / roofline_fc6.fc6_wgrad time the same GEMM without it"}
                                 if fused_update else {"on": False}),
            # whole step as ONE captured hipGraph (forward + backward + SGD; ms_per_step above) vs the same step issued launch by
            # launch from Python (what DDP runs use); per-kernel figures come from the eager steps
            "step_launch": ({"mode": "hipGraph replay", "graph_replays": replays, "graph_captures": captures,
                             "eager_ms_per_step": round(eager_ms, 3), "instrumented_ms_per_step": round(instrumented_ms, 3)}
                            if graphs is not None else
                            {"mode": "data parallel: 4 stage graphs + eager all-reduces between them", "graph_replays": replays,
                             "graph_captures": captures, "eager_ms_per_step": round(eager_ms, 3),
                             "instrumented_ms_per_step": round(instrumented_ms, 3)}
                            if replays is not None else
                            {"mode": "eager launches", "instrumented_ms_per_step": round(instrumented_ms, 3)}),
        }
        if world == 1 and dtype == torch.bfloat16:
            # the same two kernels ALONE on the maps the recipe / COCO run on (round 6): conv5 at 99x165, ROIPool forward at 99x165 / 8000
            # ROIs and 150x200 / 4000 ROIs (one call = view + flip of one scale), 10-20 launches each, HIP events on the launch stream
            def alone_ms(fn, n):
                for _ in range(3):
                    fn()
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                for _ in range(n):
                    fn()
                b_.record(); torch.cuda.synchronize()
                return a_.elapsed_time(b_) / n
            xc = (torch.randn(2, 99, 165, 512, device=device) * 0.5).to(dtype); oc = torch.empty_like(xc)
            tcv = alone_ms(lambda: ops.conv3x3(xc, wk5, oc, 2, ep5), 20)
            flc = 2.0 * (2 * 99 * 165) * 512 * 4608
            out["roofline_conv5_99x165_alone"] = {"kernel": "conv5 fwd, batch 2, 99x165, 512->512, dilation 2 (the COCO shape's map)", "bound": "mfma",
                                                  "achieved": round(flc / (tcv * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                                  "frac": round(flc / (tcv * 1e-3) / 1e12 / peak, 4), "avg_ms": round(tcv, 4)}
            del xc, oc
            gr = torch.Generator().manual_seed(0)
            for (hm, wm, rr) in ((99, 165, 8000), (150, 200, 4000)):
                x1_ = torch.rand(rr, generator=gr) * (wm * 8 - 32); y1_ = torch.rand(rr, generator=gr) * (hm * 8 - 32)
                bw_ = 24 + torch.rand(rr, generator=gr) * (wm * 8 - x1_ - 24); bh_ = 24 + torch.rand(rr, generator=gr) * (hm * 8 - y1_ - 24)
                rois_ = torch.stack([(torch.arange(rr) >= rr // 2).float(), x1_, y1_, (x1_ + bw_).clamp(max=wm * 8), (y1_ + bh_).clamp(max=hm * 8)], 1).to(device)
                feat_ = torch.randn(2, hm, wm, 512, device=device).relu().to(dtype); obj_ = torch.rand(rr, device=device)
                o_ = torch.empty(rr, 25088, device=device, dtype=dtype); a16 = torch.empty(rr, 25088, device=device, dtype=torch.int16)
                tr_ = alone_ms(lambda: ops.roi_pool_fwd(feat_, rois_, o_, a16, 0.125, 7, 7, row_scale=obj_, row_scale_add=1.0), 10)
                by_ = rr * 25088 * (es + 2) + 2 * hm * wm * 512 * es
                out["roofline_roipool"][f"fwd_{hm}x{wm}"] = {"kernel": f"roi_pool_fwd alone: {rr} ROIs on a 2 x {hm}x{wm} x 512 map", "bound": "hbm",
                                                            "achieved": round(by_ / (tr_ * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                                            "frac": round(by_ / (tr_ * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_bytes": by_,
                                                            "avg_ms": round(tr_, 4)}
                del rois_, feat_, o_, a16
        phase("kernel timers + conv5_3 alone done")
        if world == 1 and dtype == torch.bfloat16 and not args.no_fp32_line:
            # the reference's own precision (fp32 storage, exact-f32 MFMA: 1/16 of the bf16 rate), a short run for the record
            trainer = opt = model = graphs = None
            _drop()
            m32 = build(device, torch.float32); m32.train()
            g32 = [{"params": [p], "lr": 2e-3 if n.endswith(".bias") else 1e-3, "weight_decay": 0.0 if n.endswith(".bias") else 5e-4}
                   for n, p in m32.named_parameters() if p.requires_grad]
            t32 = Trainer(m32, HipSGD(g32, 1e-3, momentum=0.9))
            for i in range(2):
                t32.run_step(batches[i % 2])
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(5):
                t32.run_step(batches[i % 2])
            torch.cuda.synchronize()
            out["fp32_ms_per_step"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
            # the same precision class with the fc6 / fc7 GEMMs and the convolutions (all but conv1_1) as six-product bf16x3 GEMMs (three
            # bf16 pieces per f32 operand, f32 accumulation: ~2^-24 per product; tests: config #2 fp32 e2e parity under it) — an
            # extra, never `value`
            m32.roi_heads.fp32x3 = True
            m32.backbone.fp32x3 = os.environ.get("SW_FP32X3_CONV", "1") != "0"     # the convolutions with >= 64 channels too (r6)
            for i in range(2):
                t32.run_step(batches[i % 2])
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(5):
                t32.run_step(batches[i % 2])
            torch.cuda.synchronize()
            out["fp32x3_ms_per_step"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
            out["fp32x3_images_per_s"] = round(4.0 / out["fp32x3_ms_per_step"] * 1e3, 1)
            del t32, m32
        if world == 1 and dtype == torch.bfloat16 and not args.no_extra_shapes:
            # the other BASELINE shapes on one GPU, for the record (not the metric): config #3's per-GPU shape = 2 images per step,
            # config #4's = 800x1333 views / 4000 proposals / 80 classes / FREEZE_AT 3.  Same trainer mode as the headline.
            trainer = opt = model = graphs = None
            _drop()

            def short_run(m, data2, n=8):
                m.train()
                gs = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
                      for nm, p in m.named_parameters() if p.requires_grad]
                t = Trainer(m, HipSGD(gs, 1e-3, momentum=0.9), use_graph=use_graph)
                for i in range(6):
                    t.run_step(data2[i % 2])
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for i in range(n):
                    t.run_step(data2[i % 2])
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / n * 1e3
            # (1) the HARD case of the headline shape: random-init heads score every proposal ~2.5e-5 << MIST_THRE, so mining / NMS /
            # labelling see ~1 candidate per class.  Heads scaled like a trained model's (tests/test_gpu_fullsize.py::_peaky: the mining
            # stages then see > 1000 candidates): same step, same graph mode, plus the mining kernel's own duration from eager steps.
            mp = build(device, dtype)
            with torch.no_grad():
                hd = mp.roi_heads
                for w_ in [hd.box_predictor.cls.weight, hd.box_predictor.det.weight] + [r_.cls_score.weight for r_ in hd.box_refinery]:
                    w_.mul_(30.0)
            mp.train()
            gs = [{"params": [p], "lr": 0.0, "weight_decay": 0.0} for nm, p in mp.named_parameters() if p.requires_grad]   # lr 0: the heads stay peaky
            tp = Trainer(mp, HipSGD(gs, 0.0, momentum=0.9), use_graph=use_graph)
            for i in range(6):
                tp.run_step(batches[i % 2])
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(10):
                tp.run_step(batches[i % 2])
            torch.cuda.synchronize()
            phase("fp32 line done; peaky run")
            out["peaky_ms_per_step"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
            if tp._graphs is not None:
                tp._graphs.enabled = False
            ops.TIMER = ops.KernelTimer(["mine_label"])
            for i in range(4):
                tp.run_step(batches[i % 2])
            tm = ops.TIMER.summary_ms()["mine_label"]
            ops.TIMER = None
            out["peaky_mine_label_us"] = round(sum(tm) / max(1, len(tm)) * 1e3, 1)
            aux = mp.roi_heads.last_aux
            out["peaky_pseudo_boxes_per_round"] = [int(r_["pgt_count"].reshape(-1)[0].item()) for r_ in aux["rounds"]] if aux and "rounds" in aux else None
            del tp, mp
            _drop()
            # (2) what real data gives: two DIFFERENT scales per image (512x512 + 640x640) and no graph replay (a dataset's view sizes
            # and proposal counts rarely repeat) — eager launches
            mm = build(device, dtype); mm.train()
            gs = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
                  for nm, p in mm.named_parameters() if p.requires_grad]
            tmx = Trainer(mm, HipSGD(gs, 1e-3, momentum=0.9), use_graph=False)
            mixed = [make_inputs(device, 700 + i, scale2=1.25) for i in range(2)]
            for i in range(4):
                tmx.run_step(mixed[i % 2])
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(10):
                tmx.run_step(mixed[i % 2])
            torch.cuda.synchronize()
            phase("mixed run")
            out["mixed_ms_per_step"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
            # (2b) where training actually runs: the VOC recipe draws two DISTINCT short sides from 480 .. 1216 (step 32) per image
            # (voc07_oicr_plus.yaml:30, dataset_mapper.py:303-317); a 500 x 375 image at 6 seeded scale pairs, eager launches
            # (every pair is a new input signature), median of 3 timed steps each after 2 warm-ups: the mean over the pairs is `recipe_ms_per_step`
            import random
            rnd = random.Random(1234)
            shorts = list(range(480, 1217, 32))
            pairs_ = [tuple(rnd.sample(shorts, 2)) for _ in range(6)]
            per_pair = []
            for s1, s2 in pairs_:
                dat = make_inputs(device, 900 + s1, H=s1, W=int(500.0 / 375.0 * s1 + 0.5), scale2=s2 / s1)
                for _ in range(2):                     # (new shapes: the allocator grows during the first steps)
                    tmx.run_step(dat)
                ts_ = []
                for _ in range(3):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    tmx.run_step(dat)
                    torch.cuda.synchronize()
                    ts_.append((time.perf_counter() - t1) * 1e3)
                per_pair.append(round(sorted(ts_)[1], 2))            # median of 3
                del dat
            out["recipe_ms_per_step"] = round(sum(per_pair) / len(per_pair), 3)
            out["recipe_pairs"] = {"short_sides": pairs_, "ms_per_step": per_pair, "image": "500x375 (W x H), R=2000, K=20"}
            # (2c) the kernel families that dominate THERE (round 6): the recipe's mean scale (848: short sides 832 + 864, maps 104x139 and
            # 108x144), every launch of a family timed live with HIP events on its stream over 3 eager steps.  The backbone's two scales
            # run on two streams that share the CUs: a launch's duration holds the other stream's work too, so the conv rows also give
            # the family's FLOP over HALF the summed durations ("two streams side by side").
            s1, s2 = 832, 864
            dat = make_inputs(device, 900 + s1, H=s1, W=int(500.0 / 375.0 * s1 + 0.5), scale2=s2 / s1)
            for _ in range(2):
                tmx.run_step(dat)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(3):
                tmx.run_step(dat)
            torch.cuda.synchronize()
            mean_scale_ms = (time.perf_counter() - t1) / 3 * 1e3

            def fam(t):
                return "conv_fwd" if (t.endswith("_fwd") and ".conv" in t) else "conv_dgrad" if t.endswith("_dgrad") else None
            ops.TIMER = ops.KernelTimer(["conv_fwd", "conv_dgrad", "wgrad_grouped", "roi_fwd", "roi_bwd"], family=fam)
            for _ in range(3):
                tmx.run_step(dat)
            tms_, work_ = ops.TIMER.summary_ms(), dict(ops.TIMER.work)
            ops.TIMER = None
            pool3_ = lambda v: (((v - 2) // 2 + 1 - 2) // 2 + 1 - 2) // 2 + 1                    # three 2x2 / stride-2 pools (vgg.py:104-122)
            W1_ = int(500.0 / 375.0 * s1 + 0.5)
            H2_, W2_ = int(s1 * (s2 / s1) + 0.5), int(W1_ * (s2 / s1) + 0.5)                        # make_inputs' second scale
            maps_ = [(pool3_(s1), pool3_(W1_)), (pool3_(H2_), pool3_(W2_))]

            def mfma_row(tag_list, what, side_by_side):
                sec = sum(sum(tms_[t]) for t in tag_list) * 1e-3
                fl = sum(work_[t] for t in tag_list)
                n_l = sum(len(tms_[t]) for t in tag_list)
                if not sec:
                    return None
                row = {"kernel": what, "bound": "mfma", "peak": peak, "unit": "TFLOP/s", "launches_per_step": round(n_l / 3.0, 1),
                       "ms_per_step_summed": round(sec / 3 * 1e3, 3), "gflop_per_step": round(fl / 3 / 1e9, 1),
                       "achieved": round(fl / sec / 1e12, 1), "frac": round(fl / sec / 1e12 / peak, 4)}
                if side_by_side:
                    row["frac_two_streams_side_by_side"] = round(2.0 * fl / sec / 1e12 / peak, 4)
                return row
            roi_b = [2 * R * 25088 * (es + 2) + 2 * h_ * w_ * 512 * es for h_, w_ in maps_]
            roi_s = sum(tms_["roi_fwd"]) * 1e-3
            out["roofline_recipe"] = {
                "shape": f"500x375 image at short sides {s1} + {s2} (the recipe's mean scale 848): views {s1}x{W1_} + {H2_}x{W2_}, "
                         f"conv5 maps {maps_[0][0]}x{maps_[0][1]} and {maps_[1][0]}x{maps_[1][1]}, R={R}, K={K}, eager launches",
                "ms_per_step": round(mean_scale_ms, 3),
                "conv_fwd_dgrad": mfma_row(["conv_fwd", "conv_dgrad"], "conv3x3_direct_kernel: every forward + data-gradient convolution launch of the step", True),
                "wgrad_grouped": mfma_row(["wgrad_grouped"], "conv_wgrad_direct_kernel: all weight gradients, one launch (folds not included)", False),
                "roi_pool_fwd": ({"kernel": "roi_pool_fwd_tasks_kernel (+ the two list-building launches): 2 calls x 4000 ROIs x 512 x 7 x 7", "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                                  "algorithmic_bytes_per_step": sum(roi_b), "ms_per_step": round(roi_s / 3 * 1e3, 3),
                                  "achieved": round(sum(roi_b) * 3 / roi_s / 1e9, 1), "frac": round(sum(roi_b) * 3 / roi_s / 1e9 / 8000.0, 4)}
                                 if roi_s else None),
                "source": "live HIP events (ops.KernelTimer, family tags); rocprofv3 summary of the same shape: profiles/r06_recipe_kernel_stats.csv"}
            del dat
            phase("recipe-scale run")
            del tmx, mm
            _drop()
            mb = build(device, dtype)
            ms = short_run(mb, [make_inputs(device, 300 + 2 * i) + make_inputs(device, 1300 + 2 * i) for i in range(2)])
            phase("b2 run")
            out["b2_ms_per_step"] = round(ms, 3)
            out["b2_images_per_s"] = round(8.0 / ms * 1e3, 1)
            del mb
            _drop()
            mc = build(device, dtype, K=80, freeze_at=3)
            ms = short_run(mc, [make_inputs(device, 500 + i, H=800, W=1333, R=4000, K=80, n_gt=5) for i in range(2)], n=5)
            phase("coco run")
            out["coco_ms_per_step"] = round(ms, 3)
            out["extra_shapes"] = {"peaky": "the headline shape with the class predictors scaled x30 (mining / NMS see > 1000 candidates), lr 0",
                                   "mixed": "views 512x512 + 640x640 (two different scales, as the reference's mapper always draws), eager launches",
                                   "recipe": "a 500x375 image at 6 seeded pairs of the VOC recipe's short sides (480..1216 step 32), eager launches",
                                   "b2": "BASELINE configs[2] per GPU: 2 images = 8 views 512x512, R=2000, K=20",
                                   "coco": "BASELINE configs[3] per GPU: 4 views 800x1333 (99x165 maps), R=4000, K=80, FREEZE_AT 3"}
            del mc
            _drop()
            # BASELINE configs[4] (Stage 3, Unbiased Teacher): one GPU's share of an iteration — teacher forward on 1 view, student
            # forward + backward on 3, SGD, teacher EMA — on the ResNet-50-FPN detector of sos-wsod_amd/frcnn.py
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            import stage3_step
            ms, _ = stage3_step.time_step(torch.bfloat16, 800, 1216, dev=device, warm=4, n=16)
            phase("stage3 run")
            out["stage3_ms_per_iter"] = round(ms, 2)
            # the dominant kernel family of that iteration: the forward GEMMs of the 1x1 convolutions / fc layers (gemm2 128x128 tiles),
            # timed live with HIP events on the launch stream over 3 more iterations; FLOP = 2 * pixels * Cout * Cin per launch
            ops.TIMER = ops.KernelTimer(["s3_gemm_fwd", "s3_gemm_fwd_bytes"])
            stage3_step.time_step(torch.bfloat16, 800, 1216, dev=device, warm=0, n=3)
            tms = ops.TIMER.summary_ms()["s3_gemm_fwd"]; fl = ops.TIMER.work["s3_gemm_fwd"]; by = ops.TIMER.work["s3_gemm_fwd_bytes"]
            ops.TIMER = None
            if tms:
                # most of these GEMMs sit left of the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B: a 256 -> 64 1x1 convolution has 51): the
                # family is priced against HBM, its MFMA figure given beside it
                sec = sum(tms) * 1e-3
                out["roofline_stage3"] = {"kernel": "gemm2_kernel<bf16, 128x128>: forward GEMMs of the ResNet-50-FPN's 1x1 convolutions + fc layers",
                                          "bound": "hbm", "achieved": round(by / sec / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                          "frac": round(by / sec / 1e9 / 8000.0, 4), "traffic": None,
                                          "flop_per_byte": round(fl / by, 1), "mfma_tflops": round(fl / sec / 1e12, 2),
                                          "launches_per_iter": round(len(tms) / 3.0, 1), "ms_per_iter": round(sum(tms) / 3.0, 3),
                                          "gflop_per_iter": round(fl / 3.0 / 1e9, 1), "mbytes_per_iter": round(by / 3.0 / 1e6, 1)}
            _drop()
            # The multi-GPU step on this one GPU: the same model inside Trainer's DistributedDataParallel over a real RCCL process
            # group of world size 1 (reducer, bucket views, per-bucket HipSGD update from the communication hook, launches issued
            # from Python — no step graph): what a rank pays per step at N > 1 before any bytes cross xGMI.  Best effort: the key is
            # simply absent if the rendezvous cannot be set up.
            try:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
                import datetime
                dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                        timeout=datetime.timedelta(seconds=60))
                try:
                    md = build(device, dtype); md.train()
                    gs = [{"params": [p], "lr": 2e-3 if nm.endswith(".bias") else 1e-3, "weight_decay": 0.0 if nm.endswith(".bias") else 5e-4}
                          for nm, p in md.named_parameters() if p.requires_grad]
                    # trainer._NativeDDP: own flat buckets, the backward in four stages with the all-reduces between them; with
                    # use_graph every stage is a captured hipGraph (the collectives stay eager launches between the replays)
                    td = Trainer(md, HipSGD(gs, 1e-3, momentum=0.9), ddp=True, use_graph=True)
                    for i in range(8):
                        td.run_step(batches[i % 2])
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    for i in range(20):
                        td.run_step(batches[i % 2])
                    torch.cuda.synchronize()
                    out["ddp_rccl_world1_ms_per_step"] = round((time.perf_counter() - t1) / 20 * 1e3, 3)
                    out["ddp_rccl_world1_mode"] = ("native reducer, 4 stage graphs + eager RCCL all-reduces" if td._native is not None and td._native.replays
                                                   else "native reducer, eager" if td._native is not None else "torch DDP, eager")
                    if td._native is not None:
                        td._native.use_graph = False               # the same trainer issuing the stages launch by launch
                        td.run_step(batches[0])
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        for i in range(10):
                            td.run_step(batches[i % 2])
                        torch.cuda.synchronize()
                        out["ddp_rccl_world1_eager_ms_per_step"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
                    td.finish()
                    del td, md
                finally:
                    dist.destroy_process_group()
            except Exception as ex:                                  # noqa: BLE001 — an extra, never the reason a bench run fails
                out["ddp_rccl_world1_error"] = repr(ex)[:200]
            _drop()
        phase("extras done; cpu baseline")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        phase("done")
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if ranks_in_sync is False:               # the line above says so; a run whose replicas diverged is not a measurement
            print(f"[bench] rank {rank}: parameter checksums differ across ranks", file=sys.stderr, flush=True)
            sys.exit(3)


if __name__ == "__main__":
    main()
