"""VGG16 conv backbone on gfx950 behind the reference's interface
(uwsod/projects/WSL/wsl/modeling/backbone/vgg.py: PlainBlock :20-122, VGG16 :125-231, build_vgg_backbone :234-245).

Same module tree / state-dict names (`plain{1..5}.0.conv{1..3}.{weight,bias}`, OIHW f32 master weights), same
freeze_at semantics, `forward(x) -> {"plain5": (N,512,h,w)}`.  Every conv is the implicit-GEMM MFMA kernel
(sw_conv3x3_igemm) on NHWC bf16/f32 activations with bias+ReLU fused; backward is explicit
(sw_conv3x3_wgrad, sw_conv3x3_igemm with flipped weights + fused ReLU mask, sw_maxpool2x2_bwd) inside ONE
torch.autograd.Function, so autograd sees a single node.  The returned feature is an NCHW *view* of NHWC storage."""
import os

import torch
import torch.nn as nn

from . import ops
from .registry import BACKBONE_REGISTRY
from .structures import ShapeSpec

# (name, cin, cout, n_conv, pool_stride, dilation) for conv5_dilation == 2 is patched in VGG16.__init__
_STAGES = [("plain1", 3, 64, 2), ("plain2", 64, 128, 2), ("plain3", 128, 256, 3), ("plain4", 256, 512, 3),
           ("plain5", 512, 512, 3)]


def _epc(dtype):
    return 8 if dtype == torch.bfloat16 else 4


class _Conv(nn.Module):
    """parameter holder with nn.Conv2d's names/shapes (weight OIHW, bias)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")   # c2_msra_fill (vgg.py:56)


class PlainBlock(nn.Module):
    def __init__(self, in_channels, out_channels, num_conv=3, dilation=1, stride=1, has_pool=False):
        super().__init__()
        assert num_conv < 5
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_conv, self.dilation, self.has_pool, self.pool_stride = num_conv, dilation, has_pool, stride
        for i in range(num_conv):
            self.add_module(f"conv{i + 1}", _Conv(in_channels if i == 0 else out_channels, out_channels))

    def convs(self):
        return [getattr(self, f"conv{i + 1}") for i in range(self.num_conv)]

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


def _wgrad_splitk(cout, cin, npix):
    """K-splits of a conv weight gradient: tiles x splits just under one resident wave of workgroups (256 CUs x 2 at 64 KiB
    LDS each = 512 slots) was the optimum for every conv3..conv5 shape (tools/probes/wgrad_sweep.py: 3 / 7 / 14 / 28 splits); one
    more split starts a second, mostly empty wave (+30 %)"""
    tiles = ((cout + 127) // 128) * ((9 * cin + 127) // 128)
    return max(1, min(32, 512 // tiles, max(1, npix // 1024)))


_WGRAD_PLAN_CACHE = {}


def _wgrad_grouped_target(shapes, bk, n_cu=256, candidates=(40, 48, 56, 64, 72, 80, 96, 112, 128, 160)):
    """K-tiles per work item of the grouped weight-gradient launch for this set of problems.  shapes: [(npix, cout, n_cols)].
    The launch deals the item list round-robin to n_cu resident workgroups, so its length is the busiest workgroup's sum of
    K-tiles: simulated here for a few targets (plus the extra slab traffic of more K-splits, priced at ~25 K-tile-times per
    extra slab of a 512 x 4608 gradient) and the cheapest kept.  Cached per shape set: the schedule of a training run's view
    sizes is computed once."""
    key = (tuple(shapes), bk, n_cu, candidates)
    hit = _WGRAD_PLAN_CACHE.get(key)
    if hit is not None:
        return hit
    best = None
    for T in candidates:
        load = [0.0] * n_cu
        t, slabs = 0, 0.0
        for npix, cout, ncols in shapes:
            ktiles = (npix + bk - 1) // bk
            ns = max(1, (ktiles + T // 2) // T)
            per = (ktiles + ns - 1) // ns
            ntile = ((cout + 255) // 256) * ((ncols + 255) // 256)
            for _ in range(ns * ntile):
                load[t % n_cu] += per
                t += 1
            slabs += (ns - 1) * cout * ncols / (512.0 * 4608.0)
        cost = max(load) + 25.0 * slabs / max(1, n_cu // 32)
        if best is None or cost < best[0]:
            best = (cost, T)
    if len(_WGRAD_PLAN_CACHE) > 512:
        _WGRAD_PLAN_CACHE.clear()
    _WGRAD_PLAN_CACHE[key] = best[1]
    return best[1]


def _wgrad_direct_covers(probs, dtype):
    """the shape conditions of the direct weight-gradient kernel (csrc/conv_wgrad_direct.hip, sw_conv3x3_wgrad_direct_try): every
    problem (n, H, W, cin, cout, dil) of a grouped launch must meet them, else the whole list runs as implicit GEMMs"""
    if dtype != torch.bfloat16 or os.environ.get("SW_WGRAD_DIRECT", "1") == "0":
        return False
    return all(cout % 64 == 0 and cin % 64 == 0 and dil in (1, 2) and H >= 8 and n * ((W + 31) // 32) * H >= 8
               for n, H, W, cin, cout, dil in probs)


def _wgrad_nslab(npix, nsplit, bk=64):
    """slabs sw_conv3x3_wgrad_workspace_floats(..., splitk = nsplit) stands for (the K range of a split is a multiple of bk pixels)"""
    kps = -(-(-(-npix // max(1, nsplit))) // bk) * bk
    return -(-npix // kps)


def _wgrad_direct_splits(probs, n_slots=512, candidates=(160, 192, 224, 256, 320, 384, 448, 512)):
    """pixel splits per problem for the direct weight-gradient kernel.  Its work items are (problem, split, 64 x 64 channel block), all of
    one problem equally long (steps = image rows of 32-pixel strips); the resident workgroups (two per CU) take the item list round-robin.
    For a few target item lengths: simulate that deal (plus ~6 steps of prologue / epilogue per item; a CU's two workgroups share its
    matrix pipes — ~1 us per step each side by side, ~0.6 us for one alone) and price the slabs (written by the kernel, read by the
    fold: ~0.25 us per MB) — keep the cheapest.  Measured (tools/wgrad_shapes.py, headline / recipe / COCO shape sets): 300-400 steps per
    item is the flat optimum, shorter items pay in slab traffic, one item per block loses the L2 sharing of a pixel range.  Cached per
    shape set."""
    key = ("direct", tuple(probs), n_slots, candidates)
    hit = _WGRAD_PLAN_CACHE.get(key)
    if hit is not None:
        return hit
    best = None
    for S in candidates:
        load = [0.0] * n_slots
        t, mb, ns_list = 0, 0.0, []
        for n, H, W, cin, cout, dil in probs:
            steps = n * ((W + 31) // 32) * H
            ns = max(1, min(int(steps / S + 0.5), steps // 8))
            eff = _wgrad_nslab(n * H * W, ns)
            while eff > 1 and -(-steps // eff) < 8:
                ns -= 1
                eff = _wgrad_nslab(n * H * W, ns)
            per = -(-steps // eff)
            ns_list.append(ns)
            for _ in range(eff * (cout // 64) * (cin // 64)):
                load[t % n_slots] += per + 6
                t += 1
            mb += eff * cout * 9 * cin * 4e-6
        half = n_slots // 2
        busiest = max(min(load[c], load[c + half]) + 0.6 * abs(load[c] - load[c + half]) for c in range(half))
        cost = busiest + 0.25 * mb
        if best is None or cost < best[0]:
            best = (cost, ns_list)
    if len(_WGRAD_PLAN_CACHE) > 512:
        _WGRAD_PLAN_CACHE.clear()
    _WGRAD_PLAN_CACHE[key] = best[1]
    return best[1]


def _wgrad_grouped_splits(npix, bk, target_ktiles):
    """K-splits of one (layer, view batch) problem of the grouped weight-gradient launch: work items of ~target_ktiles K-tiles
    each, so that the 256x256 items of all layers are of similar length (conv3 maps hold 4x the pixels of conv4 / conv5)"""
    ktiles = (npix + bk - 1) // bk
    return max(1, (ktiles + target_ktiles // 2) // target_ktiles)


class _VGGFunction(torch.autograd.Function):
    """(x_0 .. x_{n-1}, *params) -> (plain5_0 .. plain5_{n-1}); every x_i is an NHWC (N_i, H_i, W_i, cpad) compute-dtype
    batch without grad (one scale of one image: the view and its flipped copy), sizes may differ between the x_i.

    ONE autograd node for all view batches of an iteration: the batches alternate between the current stream and a side
    stream (their ~250-workgroup conv4 / conv5 launches share the CUs: 64 KiB LDS per workgroup -> two per CU) in forward
    AND backward, and the weight / bias gradients of all batches meet in shared split-K workspaces that ONE ordered fold
    per parameter sums — the sum over views never becomes an autograd accumulation (18 ATen adds per step before), the
    result is deterministic, and autograd sees a single-stream node."""

    @staticmethod
    def forward(ctx, module, n_in, *args):
        xs, params = args[:n_in], args[n_in:]
        main = torch.cuda.current_stream()
        side = module.side_stream() if n_in > 1 else None
        if side is not None:
            side.wait_stream(main)
        infos, outs = [], []
        for i, x in enumerate(xs):
            st = side if (side is not None and i % 2 == 1) else main
            with torch.cuda.stream(st):
                if st is not main:
                    x.record_stream(st)
                out, info = _VGGFunction._forward_one(module, x, params)
                if st is not main:
                    out.record_stream(main)                 # allocated on the side stream's pool, consumed on the main one
            infos.append(info); outs.append(out)
        if side is not None:
            main.wait_stream(side)
        ctx.module, ctx.infos, ctx.params = module, infos, params
        return tuple(outs)

    @staticmethod
    def _forward_one(module, x, params):
        dtype = x.dtype
        cur = x
        pi = 0
        stage_info = []     # per stage: ([(input act, output act) per conv], pre-pool act or None)
        ft = module.first_trainable_conv()
        for bi, blk in enumerate(module.blocks):
            conv_io = []
            for ci in range(blk.num_conv):
                w, b = params[pi], params[pi + 1]
                pi += 2
                cin = cur.shape[3]
                # a stage nothing differentiates through (below FREEZE_AT): its last convolution pools inside its epilogue — the
                # unpooled map (67 MB per 512x512 view pair at conv1_2) is never written, read back or kept (sw_conv3x3_relu_pool2)
                if (ci == blk.num_conv - 1 and blk.has_pool and blk.pool_stride == 2 and blk.dilation == 1 and dtype == torch.bfloat16
                        and (ft is None or bi < ft[0]) and cur.shape[1] >= 2 and cur.shape[2] >= 2):
                    oh, ow = (cur.shape[1] - 2) // 2 + 1, (cur.shape[2] - 2) // 2 + 1
                    pooled = torch.empty(cur.shape[0], oh, ow, blk.out_channels, device=x.device, dtype=dtype)
                    if ops.conv3x3_relu_pool2(cur, module.staged_weight(w, 0, cin, dtype), b, pooled, tag=f"{blk.tag}.conv{ci + 1}_fwd"):
                        conv_io.append((cur, None))
                        cur = pooled
                        continue
                out = torch.empty(cur.shape[0], cur.shape[1], cur.shape[2], blk.out_channels, device=x.device, dtype=dtype)
                ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dtype)
                if module.x3_layer(dtype, cin, blk.out_channels):
                    # reference precision on the bf16 MFMA: the f32 input as three bf16 pieces, K-concatenated along the channels
                    # ([a1|a1|a2|a1|a2|a3] against the weights' [b1|b2|b1|b3|b2|b1]): ONE bf16 convolution over 6 Cin, f32 accumulation
                    # and output (sw_split_bf16x3; the fc layers' form, ops.gemm_f32x3)
                    n_, h_, w_ = cur.shape[:3]
                    x3 = ops.split_bf16x3(cur.view(n_ * h_ * w_, cin), 0,
                                          out=torch.empty(n_ * h_ * w_, 6 * cin, device=x.device, dtype=torch.bfloat16))
                    ops.conv3x3(x3.view(n_, h_, w_, 6 * cin), module.staged_weight_x3(w, 0, cin), out, blk.dilation, ep,
                                tag=f"{blk.tag}.conv{ci + 1}_fwd")
                elif not (module.winograd_ok(cin, blk.out_channels) and
                        ops.conv3x3_winograd(cur, module.winograd_weight(w, 0), out, blk.dilation, ep, tag=f"{blk.tag}.conv{ci + 1}_fwd")):
                    wk = module.staged_weight(w, 0, cin, dtype)
                    ops.conv3x3(cur, wk, out, blk.dilation, ep, tag=f"{blk.tag}.conv{ci + 1}_fwd")
                conv_io.append((cur, out))
                cur = out
            pre_pool = None
            if blk.has_pool and conv_io[-1][1] is not None:
                pre_pool = cur
                s = blk.pool_stride
                oh, ow = (cur.shape[1] - 2) // s + 1, (cur.shape[2] - 2) // s + 1
                pooled = torch.empty(cur.shape[0], oh, ow, cur.shape[3], device=x.device, dtype=dtype)
                ops.maxpool_fwd(cur, pooled, s)
                cur = pooled
            stage_info.append((conv_io, pre_pool))
        return cur, stage_info

    @staticmethod
    def backward(ctx, *gs):
        module, infos, params = ctx.module, ctx.infos, ctx.params
        # the saved activations include this node's own outputs (ReLU mask of the last conv): node -> ctx -> tensor -> grad_fn
        # is a reference cycle that only Python's cyclic GC would free, hundreds of MB per step later.  Drop it now.
        ctx.infos = None
        n_in = len(infos)
        grads = [None] * len(params)
        first_trainable = module.first_trainable_conv()      # (stage idx, conv idx) or None
        live = [i for i in range(n_in) if gs[i] is not None]
        if first_trainable is None or not live:
            return (None, None) + (None,) * n_in + tuple(grads)
        dev = gs[live[0]].device
        dtype = infos[0][0][0][0][0].dtype
        main = torch.cuda.current_stream()
        side = module.side_stream() if len(live) > 1 else None
        # ---- plan: per trainable conv one slab workspace / one partial-row workspace shared by all view batches.
        # grouped: the weight gradients of ALL layers and view batches run as ONE launch after the data-gradient chains
        # (sw_conv3x3_wgrad_grouped: 256x256 tiles, every CU busy); else one 128x128-tile launch per layer and view inside the chains
        grouped = module.grouped_wgrad
        # fp32 mode with bf16x3: the weight gradients as bf16 problems over SIX stacked copies of the batch (the three-piece splits of dY
        # and X along the image dimension, sw_split_bf16x3 along_rows: [a1;a1;a2;a1;a2;a3] against [b1;b2;b1;b3;b2;b1] — the sum over
        # images is the sum of the six products) when every trainable convolution qualifies
        x3w = False
        if module.fp32x3 and dtype == torch.float32 and grouped:
            x3w, pj = True, len(params)
            for sj in range(len(module.blocks) - 1, -1, -1):
                for cj in range(module.blocks[sj].num_conv - 1, -1, -1):
                    pj -= 2
                    if params[pj].requires_grad and not module.x3_layer(dtype, infos[live[0]][sj][0][cj][0].shape[3], module.blocks[sj].out_channels):
                        x3w = False
        wdt, nmul = (torch.bfloat16, 6) if x3w else (dtype, 1)
        bk = 64 if wdt == torch.bfloat16 else 32
        deferred = []
        module._colsum_deferred = []
        target = module.wgrad_target_ktiles
        direct_ns = None                        # (pidx, view batch) -> splits, when the direct weight-gradient kernel takes the list
        if grouped and target <= 0:             # pick the K-tiles per item for THIS set of (layer, view batch) problems
            shapes, probs6, keys = [], [], []
            pj = len(params)
            for sj in range(len(module.blocks) - 1, -1, -1):
                bj = module.blocks[sj]
                for cj in range(bj.num_conv - 1, -1, -1):
                    pj -= 2
                    if params[pj].requires_grad:
                        for i in live:
                            xin = infos[i][sj][0][cj][0]
                            shapes.append((nmul * xin.shape[0] * xin.shape[1] * xin.shape[2], bj.out_channels, 9 * xin.shape[3]))
                            probs6.append((nmul * xin.shape[0], xin.shape[1], xin.shape[2], xin.shape[3], bj.out_channels, bj.dilation))
                            keys.append((pj, i))
            if _wgrad_direct_covers(probs6, wdt):
                direct_ns = dict(zip(keys, _wgrad_direct_splits(probs6)))
            target = _wgrad_grouped_target(shapes, bk)
        plan = {}            # pidx -> dict(ws, rows, per-batch offsets, totals, dw, db)
        pidx = len(params)
        for si in range(len(module.blocks) - 1, -1, -1):
            blk = module.blocks[si]
            for ci in range(blk.num_conv - 1, -1, -1):
                pidx -= 2
                w = params[pidx]
                if w.requires_grad:
                    cout = blk.out_channels
                    slab_off, row_off, nslab, nrow, splits = {}, {}, 0, 0, {}
                    for i in live:
                        x_in = infos[i][si][0][ci][0]
                        n, H, W, cin = x_in.shape
                        splits[i] = (direct_ns[(pidx, i)] if direct_ns is not None
                                     else _wgrad_grouped_splits(nmul * n * H * W, bk, target) if grouped
                                     else _wgrad_splitk(cout, cin, n * H * W))
                        slab_off[i], row_off[i] = nslab, nrow
                        nslab += ops.conv3x3_wgrad_nslab_shape(wdt, nmul * n, H, W, cin, cout, splits[i])
                        nrow += ops.colsum_nrows(dtype, n * H * W, cout)
                    cin = infos[live[0]][si][0][ci][0].shape[3]
                    plan[pidx] = dict(ws=torch.empty(nslab, cout * 9 * cin, device=dev, dtype=torch.float32),
                                      rows=torch.empty(nrow, cout, device=dev, dtype=torch.float32), slab_off=slab_off,
                                      row_off=row_off, nslab=nslab, nrow=nrow, splits=splits, cin=cin,
                                      dw=ops.grad_target(w, (cout, cin, 3, 3), dev),
                                      db=ops.grad_target(params[pidx + 1], (cout,), dev))
                if (si, ci) == first_trainable:
                    break
            if si == first_trainable[0]:
                break
        if side is not None:
            side.wait_stream(main)
        for k, i in enumerate(live):
            st = side if (side is not None and k % 2 == 1) else main
            with torch.cuda.stream(st):
                g = gs[i]
                if st is not main:
                    g.record_stream(st)
                _VGGFunction._backward_one(module, infos[i], params, g, dtype, plan, i, first_trainable,
                                           deferred if grouped else None, None if st is main else main, x3w)
            infos[i] = None
        if side is not None:
            main.wait_stream(side)
        if deferred:
            ops.conv3x3_wgrad_grouped(deferred, tag="wgrad_grouped")
            ops.colsum_partial_multi(module._colsum_deferred)
            deferred.clear()
        module._colsum_deferred = []
        # ---- one ordered fold per parameter over the slabs / partial rows of every view batch: all of them in two launches
        ops.conv3x3_wgrad_fold_multi([(pl["ws"], pl["nslab"], pl["dw"]) for pl in plan.values()])
        ops.colsum_fold_multi([(pl["rows"], pl["nrow"], pl["db"]) for pl in plan.values()])
        for pidx, pl in plan.items():
            w = params[pidx]
            grads[pidx] = pl["dw"][:, : w.shape[1]].contiguous() if pl["cin"] != w.shape[1] else pl["dw"]
            grads[pidx + 1] = pl["db"]
        return (None, None) + (None,) * n_in + tuple(grads)

    @staticmethod
    def _backward_one(module, stage_info, params, g, dtype, plan, i, first_trainable, deferred, consumer_stream, x3w=False):
        g = g.contiguous()
        if g.dtype != dtype:
            g = g.to(dtype)
        # dz of the last conv: ReLU backward of the output feature (idempotent if the producer already masked)
        last_out = stage_info[-1][0][-1][1]
        assert stage_info[-1][1] is None, "backward expects the last stage to have no pool (vgg.py:197)"
        dz = ops.relu_bwd(last_out, g, out=torch.empty_like(g))      # autograd owns g: masked copy, not in place
        pidx = len(params)
        for si in range(len(stage_info) - 1, -1, -1):
            blk = module.blocks[si]
            conv_io, _ = stage_info[si]
            for ci in range(blk.num_conv - 1, -1, -1):
                pidx -= 2
                x_in, _ = conv_io[ci]
                w = params[pidx]
                n, H, W, cin = x_in.shape
                pl = plan.get(pidx)
                if pl is not None:
                    npix = n * H * W
                    if deferred is None:
                        ops.conv3x3_wgrad_slabs(x_in, dz, pl["ws"][pl["slab_off"][i]:], blk.dilation, splitk=pl["splits"][i])
                    else:                       # the grouped launch reads (x_in, dz) later, on the main stream
                        xw, dw_ = x_in, dz
                        if x3w:                 # six stacked bf16 copies of the batch (see backward)
                            cout = blk.out_channels
                            xw = ops.split_bf16x3(x_in.view(npix, cin), 1, along_rows=True,
                                                  out=torch.empty(6 * npix, cin, device=g.device, dtype=torch.bfloat16)).view(6 * n, H, W, cin)
                            dw_ = ops.split_bf16x3(dz.view(npix, cout), 0, along_rows=True,
                                                   out=torch.empty(6 * npix, cout, device=g.device, dtype=torch.bfloat16)).view(6 * n, H, W, cout)
                        if consumer_stream is not None:
                            xw.record_stream(consumer_stream); dw_.record_stream(consumer_stream)
                            dz.record_stream(consumer_stream)
                        deferred.append((xw, dw_, pl["ws"][pl["slab_off"][i]:], blk.dilation, pl["splits"][i]))
                    if deferred is None:
                        ops.colsum_partial(dz.view(npix, blk.out_channels), npix, blk.out_channels, pl["rows"][pl["row_off"][i]:])
                    else:       # bias partial rows of every layer x view batch: ONE launch behind the chains (18 launches before, each
                        #             in front of a data-gradient convolution on its stream's critical path)
                        module._colsum_deferred.append((dz.view(npix, blk.out_channels), pl["rows"][pl["row_off"][i]:]))
                if (si, ci) == first_trainable:
                    return
                # data gradient: conv with flipped/transposed weights; ReLU mask of the producer fused when the
                # input is a direct conv output (ci > 0); stage inputs go through the pool backward instead
                dx = torch.empty(n, H, W, cin, device=g.device, dtype=dtype)
                ref = x_in.view(n * H * W, cin) if ci > 0 else None
                epd = ops.make_epilogue(relu_ref=ref, out_dtype=dtype)
                if module.x3_layer(dtype, blk.out_channels, cin):
                    cout = blk.out_channels
                    dz3 = ops.split_bf16x3(dz.view(n * H * W, cout), 0,
                                           out=torch.empty(n * H * W, 6 * cout, device=g.device, dtype=torch.bfloat16))
                    ops.conv3x3(dz3.view(n, H, W, 6 * cout), module.staged_weight_x3(w, 1, cin), dx, blk.dilation, epd,
                                tag=f"{blk.tag}.conv{ci + 1}_dgrad")
                elif not (module.winograd_ok(blk.out_channels, cin) and
                        ops.conv3x3_winograd(dz, module.winograd_weight(w, 1), dx, blk.dilation, epd)):
                    wkd = module.staged_weight(w, 1, cin, dtype)
                    ops.conv3x3(dz, wkd, dx, blk.dilation, epd, tag=f"{blk.tag}.conv{ci + 1}_dgrad")
                dz = dx
            # dz is now the gradient wrt this stage's input = previous stage's pooled output
            prev_pre_pool = stage_info[si - 1][1]
            pblk = module.blocks[si - 1]
            din = torch.empty_like(prev_pre_pool)
            ops.maxpool_bwd(prev_pre_pool, dz, din, pblk.pool_stride, relu_mask=True)
            dz = din


class VGG16(nn.Module):
    """vgg.py:125-231.  `compute_dtype`: torch.bfloat16 (MFMA bf16) or torch.float32 (exact f32 MFMA)."""

    def __init__(self, conv5_dilation, freeze_at, num_classes=None, out_features=None, compute_dtype=torch.bfloat16):
        super().__init__()
        self.num_classes = num_classes
        self.compute_dtype = compute_dtype
        self.dual_stream = True           # alternate the view batches of forward_views between two HIP streams
        self.grouped_wgrad = os.environ.get("SW_WGRAD_GROUPED", "1") != "0"      # development switch: per-layer launches
        self.wgrad_target_ktiles = int(os.environ.get("SW_WGRAD_KTILES", "0"))   # 0: chosen per shape set (_wgrad_grouped_target)
        self._side = None
        self._wk_cache = {}
        self._wk3_cache = {}
        # fp32 mode: the convolutions with >= 64 channels on both sides as six-product bf16x3 convolutions (the heads' fc GEMMs: roi_heads_oicrplus
        # fp32x3).  SW_FP32X3_CONV=0 keeps them on the exact-f32 MFMA (A/B timing)
        self.fp32x3 = (compute_dtype == torch.float32 and os.environ.get("SW_FP32X3", "0") == "1"
                       and os.environ.get("SW_FP32X3_CONV", "1") != "0")
        # Winograd F(2x2, 3x3) for the forward / data gradient of the wide bf16 layers (csrc/conv_winograd.hip): 2.25x fewer MFMA cycles
        # than the direct kernel — built, bit-validated (relative L2 3.8e-3 against float64, the direct form 2.4e-3) and MEASURED SLOWER
        # on this part: conv5_3 42.4 vs 38.2 us, conv3_2 49.4 vs 39.3 us (profiles/r05_winograd_experiment.txt: the f32 input transform
        # and three barriers per 32-channel chunk cost more than the MFMAs saved).  So it is OFF by default; SW_CONV_WINOGRAD=1 routes
        # the layers with Cin >= SW_WINOGRAD_MIN_CIN (default 256: conv3_2 .. conv5_3) through it.  fp32 mode never uses it.
        self.winograd = os.environ.get("SW_CONV_WINOGRAD", "0") == "1" and compute_dtype == torch.bfloat16
        self.winograd_min_cin = int(os.environ.get("SW_WINOGRAD_MIN_CIN", "256"))
        self._wino_cache = {}
        self._out_feature_strides, self._out_feature_channels = {}, {}
        self.stages_and_names = []
        strides = {"plain1": 2, "plain2": 4, "plain3": 8, "plain4": 8 if conv5_dilation == 2 else 16,
                   "plain5": 8 if conv5_dilation == 2 else 16}
        self.blocks = []
        for i, (name, cin, cout, nconv) in enumerate(_STAGES):
            if name == "plain4":
                blk = PlainBlock(cin, cout, num_conv=nconv, stride=1 if conv5_dilation == 2 else 2, has_pool=True)
            elif name == "plain5":
                blk = PlainBlock(cin, cout, num_conv=nconv, stride=1, dilation=conv5_dilation, has_pool=False)
            else:
                blk = PlainBlock(cin, cout, num_conv=nconv, stride=2, has_pool=True)
            stage = nn.Sequential(blk)               # keeps the reference's "plainK.0.convJ" names
            self.add_module(name, stage)
            self.stages_and_names.append((stage, name))
            blk.tag = name
            self.blocks.append(blk)
            self._out_feature_strides[name] = strides[name]
            self._out_feature_channels[name] = cout
            if freeze_at >= i + 1:
                blk.freeze()
        if out_features is None:
            out_features = ["plain5"]
        self._out_features = out_features
        assert self._out_features == ["plain5"], "this build exposes the hot path's feature only (IN_FEATURES ['plain5'])"

    @property
    def size_divisibility(self):
        return 0

    def staged_weight(self, w, mode, cin_pad, dtype):
        """compute-dtype kernel-layout copy of an OIHW master weight (mode 0: forward [co][tap][ci], mode 1: data
        gradient [ci][8-tap][co]); rebuilt only when the parameter changed (the two backbone calls of an iteration and
        the backward share one copy).  The buffers are persistent and registered in ops.STAGING, so that HipSGD's fused
        step rewrites them from the updated weights and this method finds them current (no staging kernels per step)."""
        key = (ops.param_key(w), mode, cin_pad, dtype)
        slot = (id(w), mode)
        hit = self._wk_cache.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        cout, cin = w.shape[:2]
        shape = (cout, 9, cin_pad) if mode == 0 else (cin, 9, cout)
        if hit is not None and tuple(hit[1].shape) == shape and hit[1].dtype == dtype and hit[1].device == w.device:
            wk = hit[1]
        else:
            wk = torch.zeros(shape, device=w.device, dtype=dtype)
        ops.conv_weight_prep(w.detach(), wk, mode, cin_pad if mode == 0 else None)
        self._wk_cache[slot] = (key, wk)
        if w.requires_grad:
            self._register_staging(w, dtype)
        return wk

    def x3_layer(self, dtype, cin, cout):
        """fp32 mode with MODEL.AMD.FP32_GEMM "bf16x3" (SW_FP32X3=1): this convolution runs as a six-product bf16 convolution"""
        return self.fp32x3 and dtype == torch.float32 and cin % 64 == 0 and cout % 64 == 0

    def staged_weight_x3(self, w, mode, cin_pad):
        """the three-piece bf16 copy ([b1|b2|b1|b3|b2|b1] along the reduction channels) of the f32 kernel-layout weight staged_weight
        holds: mode 0 [co][tap][6 ci], mode 1 [ci][tap][6 co]; rebuilt when the parameter changed"""
        key = (ops.param_key(w), mode, cin_pad)
        slot = (id(w), mode)
        hit = self._wk3_cache.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        wk = self.staged_weight(w, mode, cin_pad, torch.float32)
        rows, cols = wk.shape[0] * 9, wk.shape[2]
        buf = hit[1] if hit is not None and tuple(hit[1].shape) == (wk.shape[0], 9, 6 * cols) else \
            torch.empty(wk.shape[0], 9, 6 * cols, device=wk.device, dtype=torch.bfloat16)
        ops.split_bf16x3(wk.view(rows, cols), 1, out=buf.view(rows, 6 * cols))
        self._wk3_cache[slot] = (key, buf)
        return buf

    def winograd_ok(self, cin, cout):
        return self.winograd and cin >= self.winograd_min_cin and cin % 32 == 0 and cout % 2 == 0

    def winograd_weight(self, w, mode):
        """the transformed filters U = G g G^T (bf16, (16, n_out, n_in)) of an OIHW master for the Winograd kernel — mode 0 forward,
        1 data gradient — current for the parameter's present value (stage_all_weights builds every stale one in ONE launch; a
        miss here builds its own)"""
        hit = self._wino_cache.get((id(w), mode))
        if hit is None or hit[0] != ops.param_key(w):
            self._winograd_refresh([(w, mode)])
            hit = self._wino_cache[(id(w), mode)]
        return hit[1]

    def _winograd_refresh(self, wanted):
        items = []
        for w, mode in wanted:
            key = ops.param_key(w)
            hit = self._wino_cache.get((id(w), mode))
            if hit is not None and hit[0] == key:
                continue
            cout, cin = w.shape[:2]
            shape = (16, cout, cin) if mode == 0 else (16, cin, cout)
            U = hit[1] if (hit is not None and tuple(hit[1].shape) == shape and hit[1].device == w.device) else \
                torch.empty(shape, device=w.device, dtype=torch.bfloat16)
            items.append((w.detach(), U, mode))
            self._wino_cache[(id(w), mode)] = (key, U)
        if items:
            ops.winograd_weight_prep(items)

    def _register_staging(self, w, dtype):
        cout, cin = w.shape[:2]
        s0, s1 = self._wk_cache.get((id(w), 0)), self._wk_cache.get((id(w), 1))
        s0 = s0 if s0 is not None and s0[0][3] == dtype else None
        s1 = s1 if s1 is not None and s1[0][3] == dtype else None

        def stamp(pk, wid=id(w), cache=self._wk_cache):               # (captures the cache dict, not the module: ops.register_staging)
            for mode in (0, 1):
                h = cache.get((wid, mode))
                if h is not None and h[0][3] == dtype:
                    cache[(wid, mode)] = ((pk, mode, h[0][2], dtype), h[1])

        ops.register_staging(w, 2, dtype, stage0=None if s0 is None else s0[1], stage1=None if s1 is None else s1[1],
                             d0=cout, d1=cin, d2=(s0[0][2] if s0 is not None else cin), stamp=stamp)

    def stage_all_weights(self, with_dgrad):
        """build every compute-dtype weight copy on the CURRENT stream (call before forking side streams)"""
        dtype = self.compute_dtype
        epc = _epc(dtype)
        ft = self.first_trainable_conv()
        wino = []
        for si, blk in enumerate(self.blocks):
            for ci, c in enumerate(blk.convs()):
                cout, cin = c.weight.shape[:2]
                if self.winograd_ok(cin, cout):                      # the Winograd layers take transformed filters instead
                    wino.append((c.weight, 0))
                    if with_dgrad and ft is not None and (si, ci) > ft and self.winograd_ok(cout, cin):
                        wino.append((c.weight, 1))
                    elif with_dgrad and ft is not None and (si, ci) > ft:
                        self.staged_weight(c.weight, 1, cin, dtype)
                    continue
                cin_pad = (cin + epc - 1) // epc * epc
                self.staged_weight(c.weight, 0, cin_pad, dtype)
                if self.x3_layer(dtype, cin_pad, cout):              # the three-piece copies too: both streams read them
                    self.staged_weight_x3(c.weight, 0, cin_pad)
                if with_dgrad and ft is not None and (si, ci) > ft:
                    if self.winograd_ok(cout, cin):
                        wino.append((c.weight, 1))
                    else:
                        self.staged_weight(c.weight, 1, cin, dtype)
                        if self.x3_layer(dtype, cout, cin):
                            self.staged_weight_x3(c.weight, 1, cin)
        self._winograd_refresh(wino)

    def first_trainable_conv(self):
        for si, blk in enumerate(self.blocks):
            for ci, c in enumerate(blk.convs()):
                if c.weight.requires_grad:
                    return (si, ci)
        return None

    def _flat_params(self):
        out = []
        for blk in self.blocks:
            for c in blk.convs():
                out += [c.weight, c.bias]
        return out

    def side_stream(self):
        if not self.dual_stream:
            return None
        if self._side is None:
            self._side = ops.worker_stream("side")
        return self._side

    def forward_nhwc(self, x_nhwc):
        """x_nhwc: (N,H,W,cpad) compute-dtype, channels >= 3 zero.  Returns NHWC plain5."""
        return _VGGFunction.apply(self, 1, x_nhwc, *self._flat_params())[0]

    def forward_views(self, xs):
        """xs: list of NHWC view batches of possibly different sizes (rcnn_multi.py:153-154,174-175 calls the backbone once per
        scale).  One autograd node; the batches alternate between two HIP streams.  Returns the list of NHWC plain5 maps."""
        self.stage_all_weights(with_dgrad=torch.is_grad_enabled())      # on the current stream, before the fork
        return list(_VGGFunction.apply(self, len(xs), *xs, *self._flat_params()))

    def forward(self, x):
        """x: (N,3,H,W) float32 normalised image batch (reference call shape) -> {"plain5": (N,512,h,w) view}"""
        assert x.dim() == 4 and x.shape[1] == 3
        n, _, H, W = x.shape
        xin = torch.empty(n, H, W, _epc(self.compute_dtype), device=x.device, dtype=self.compute_dtype)
        ops.nchw_to_nhwc(x.contiguous().float(), xin)
        f = self.forward_nhwc(xin)
        return {"plain5": f.permute(0, 3, 1, 2)}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


def _dtype_from_cfg(cfg):
    s = str(cfg.MODEL.get("AMD", {}).get("COMPUTE_DTYPE", "bf16")).lower()
    return torch.float32 if s in ("fp32", "f32", "float32") else torch.bfloat16


@BACKBONE_REGISTRY.register()
def build_vgg_backbone(cfg, input_shape=None):
    depth = cfg.MODEL.VGG.DEPTH
    if depth != 16:
        raise NotImplementedError("only VGG16 is on the OICR+ hot path (voc07_oicr_plus.yaml:7-12)")
    return VGG16(cfg.MODEL.VGG.CONV5_DILATION, cfg.MODEL.BACKBONE.FREEZE_AT, out_features=cfg.MODEL.VGG.OUT_FEATURES,
                 compute_dtype=_dtype_from_cfg(cfg))
