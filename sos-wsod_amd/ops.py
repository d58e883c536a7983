"""Thin torch-tensor wrappers over the C-ABI kernels (include/soswsod_hip.h).

torch is used here only for device memory and the current HIP stream; every computation is a
hand-written gfx950 kernel.  All tensors must live on the GPU and be contiguous where stated.
"""
import ctypes
import os

import torch

from ._lib import SW_BF16, SW_F32, Epilogue, check, lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current stream's hipStream_t (the raw-handle query: ~0.3 us; building a torch.cuda.Stream object per launch cost
    10 us x ~55 launches per step of host time)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt(t_or_dtype):
    d = t_or_dtype if isinstance(t_or_dtype, torch.dtype) else t_or_dtype.dtype
    if d == torch.float32:
        return SW_F32
    if d == torch.bfloat16:
        return SW_BF16
    raise TypeError(f"unsupported dtype {d}")


class KernelTimer:
    """Optional in-process timing of tagged launches with HIP events recorded on the launch stream (used by
    bench.py to price the dominant kernels live; off by default => zero overhead)."""

    def __init__(self, tags, family=None):
        """family: optional callable tag -> one of `tags` (or None): several call sites timed under one key, e.g. every backbone
        convolution's forward launch (tags "plain3.conv2_fwd", ...) as "conv_fwd" """
        self.tags = set(tags)
        self.family = family
        self.pairs = {t: [] for t in tags}
        self.work = {t: 0.0 for t in tags}        # algorithmic work (FLOP or bytes) noted by the call sites of a tag

    def _key(self, tag):
        if tag in self.tags:
            return tag
        return self.family(tag) if self.family is not None else None

    def note(self, tag, amount):
        tag = self._key(tag)
        if tag is not None:
            self.work[tag] += float(amount)

    def wrap(self, tag, fn):
        tag = self._key(tag)
        if tag is None:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.pairs[tag].append((a, b))
        return r

    def summary_ms(self):
        torch.cuda.synchronize()
        return {t: [a.elapsed_time(b) for a, b in ps] for t, ps in self.pairs.items()}


TIMER = None
PARAM_EPOCH = 0        # bumped whenever a kernel writes parameters behind torch's version counters ("something changed": HipSGD, EMA, ...)
BUFFER_EPOCH = 0       # bumped by kernels that write module BUFFERS behind torch's version counters (the Stage-3 teacher EMA)
INVALIDATE_EPOCH = 0   # bumped when EVERY cached compute-dtype copy is stale (checkpoint load, teacher EMA, re-homed parameter storage)


def invalidate_all_staged():
    """every cached compute-dtype weight copy is stale (parameters were written wholesale behind the version counters)"""
    global PARAM_EPOCH, INVALIDATE_EPOCH
    PARAM_EPOCH += 1
    INVALIDATE_EPOCH += 1


def mark_updated(p):
    """a kernel (HipSGD) has just rewritten THIS parameter behind torch's version counter: its own cached copies are stale unless
    the kernel rewrote them too and stamps them with the new key.  Per parameter: under the data-parallel trainer the update runs
    bucket by bucket, several calls per step — a global counter made every call invalidate the stamps of the buckets before it, and
    all but the last bucket's weights were re-staged in the next forward (12 staging launches and 0.3 ms per step)."""
    p.__dict__["_sw_epoch"] = PARAM_EPOCH


def param_key(p):
    """cache key of a parameter's current value (compute-dtype weight copies are rebuilt only when it changes): storage, torch's
    version counter, the epoch of the last wholesale invalidation and of the parameter's own last kernel update"""
    return (p.data_ptr(), p._version, INVALIDATE_EPOCH, p.__dict__.get("_sw_epoch", 0))


GRAD_SCOPE = None      # the active grad_scope (or None)


_WORKER_STREAMS = {}
_WORKER_ROLES = ("main", "side", "upd", "chk")


def worker_stream(role, device=None):
    """The process's ONE stream per role and device — "main" (a step graph's capture / replay stream), "side" (the second backbone
    scale, the Stage-3 teacher), "upd" (the data-parallel bucket updates), "chk" (the speculation ledger's comparison).  All four are
    created together, in this order, the first time any is asked for.  Why not a fresh torch.cuda.Stream() per model / trainer: the
    runtime spreads streams over a handful of hardware queues in creation order, and two streams that are meant to run side by side
    but landed on one queue serialise — or stall each other at their event waits.  With every trainer of a process creating its own
    streams, the SAME code measured 12.8 or 18.9 ms per Stage-3 iteration and 8.5 or 11.3 ms per data-parallel step depending on how
    many models had lived in the process before (bench.py's extras).  Sharing a role's stream between models only adds ordering."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if os.environ.get("SW_FRESH_STREAMS") == "1":                   # development switch: a new pool stream per call (the old behaviour)
        return torch.cuda.Stream(device=torch.device("cuda", idx))
    got = _WORKER_STREAMS.get(idx)
    if got is None:
        got = _WORKER_STREAMS[idx] = {r: torch.cuda.Stream(device=torch.device("cuda", idx)) for r in _WORKER_ROLES}
    return got[role]


class capture_guard:
    """Around a hipGraph capture: Python's cycle collector stays off.  A collection that happens to run inside the capture — it can,
    on the autograd thread too, whenever the allocation counter says so — may finalize an older torch.cuda.CUDAGraph that was
    waiting in a reference cycle (a deleted Trainer and its graphs), and destroying a graph while ANY stream of the process
    captures is an error of the runtime ("operation not permitted when stream is capturing" from ~CUDAGraph: the whole process
    dies; seen in bench.py between two of its short runs).  Collect first, then disable until the capture is over."""

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


class grad_scope:
    """Wrap the forward passes AND the one `backward()` call of an iteration whose graph uses parameters more than once (the Stage-3
    student: two forward passes per iteration, unbias/ubteacher/engine/trainer.py:527-538; the RPN head's convolution on five FPN
    levels in each).  Autograd would hand every use's weight gradient to the parameter's accumulator and add them with a torch
    kernel.  Inside the scope
      * the first weight-gradient node of a parameter registers the buffer it returns (`note_grad`) and every later node of the same
        parameter ADDS to that buffer inside its own fold / GEMM epilogue (`pending_grad`) and returns no gradient;
      * a forward pass counts the uses of a 3x3 weight (`count_use`); its weight-gradient nodes then only QUEUE their (x, dy)
        pair and the last of them runs all pairs as one grouped launch + one fold (frcnn._wgrad_3x3).
    The accumulator node of a parameter runs after all of its incoming edges (every use), on the same stream, so it sees the
    finished sum; the data-parallel reducer's hooks hang on that node and are therefore not affected.  A counted use whose
    backward node never runs would leave a queued gradient unfinished: __exit__ raises in that case (opt out: do not open a
    scope).  Outside a scope every helper is inert: plain autograd behaviour."""

    def __enter__(self):
        global GRAD_SCOPE
        self._prev, GRAD_SCOPE = GRAD_SCOPE, self
        self.bufs, self.uses, self.queued = {}, {}, {}
        return self

    def finish(self):
        """Call between `backward()` and `optimizer.step()`: every queued weight gradient must have been run by its last use.  A
        counted use whose backward node never ran leaves the buffer autograd was handed unfinished (uninitialised memory): raise
        BEFORE the optimizer applies it."""
        left = [k for k, q in self.queued.items() if q["probs"]]
        if left:
            self.bufs, self.uses, self.queued = {}, {}, {}
            raise RuntimeError(f"grad_scope: {len(left)} queued weight gradient(s) were never finished — a use counted in the forward pass "
                               "did not take part in backward(); run this graph without ops.grad_scope")

    def __exit__(self, exc_type, *exc):
        global GRAD_SCOPE
        GRAD_SCOPE = self._prev
        left = [k for k, q in self.queued.items() if q["probs"]]
        self.bufs, self.uses, self.queued = {}, {}, {}
        if left and exc_type is None:
            raise RuntimeError(f"grad_scope: {len(left)} queued weight gradient(s) were never finished — a use counted in the forward pass "
                               "did not take part in backward(); run this graph without ops.grad_scope")
        return False


CALLER_GRAD_ENABLED = True      # grad mode of the code that called the running CountedFunction.apply (see there)


class CountedFunction(torch.autograd.Function):
    """torch.autograd.Function whose forward may call `count_use`.  Inside `forward` autograd has ALREADY switched grad mode off —
    whether the caller ran under torch.no_grad() or not — and ctx.needs_input_grad stays True for a parameter that requires grad:
    the only place that still sees the caller's mode is `apply` itself, so it is recorded there.  (Round 5 first tested
    torch.is_grad_enabled() inside count_use: always False in a forward, so nothing was counted and every Stage-3 weight gradient
    silently took the one-by-one path — 16.2 -> 17.3 ms; found in the kernel statistics, where the grouped launches had disappeared.)"""

    @classmethod
    def apply(cls, *args, **kwargs):
        global CALLER_GRAD_ENABLED
        prev, CALLER_GRAD_ENABLED = CALLER_GRAD_ENABLED, torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            CALLER_GRAD_ENABLED = prev


def count_use(key):
    """forward side (inside a CountedFunction.forward): this pass will contribute one weight-gradient node for `key`.  A pass under
    torch.no_grad() builds no node: not counted (a counted use that never reaches backward would leave a queued gradient unfinished)."""
    if GRAD_SCOPE is not None and key is not None and CALLER_GRAD_ENABLED:
        GRAD_SCOPE.uses[key] = GRAD_SCOPE.uses.get(key, 0) + 1


def pending_grad(key, shape):
    """the f32 buffer an earlier node of this backward pass registered for `key` (viewed as `shape`), or None"""
    if GRAD_SCOPE is None or key is None:
        return None
    buf = GRAD_SCOPE.bufs.get(key)
    if buf is None or buf.numel() != int(torch.Size(shape).numel()) or not buf.is_contiguous():
        return None
    return buf.view(shape)


def note_grad(key, buf):
    if GRAD_SCOPE is not None and key is not None:
        # an ALIAS (its own tensor object on the same storage): autograd's accumulator adopts a gradient without copying it only
        # when nobody else holds the tensor object it was handed (18 device-to-device copies per Stage-3 iteration otherwise)
        GRAD_SCOPE.bufs[key] = buf.view(buf.shape)


def grad_target(param, shape, dev):
    """Where a weight-gradient GEMM writes.  Under DistributedDataParallel(gradient_as_bucket_view=True) the Trainer remembers the
    bucket view each parameter's gradient lived in last step (`param._sw_grad_view`); writing the new gradient THERE lets the
    reducer find `grad.is_alias_of(bucket_view)` and skip its copy of the tensor into the bucket (fc6: 411 MB read + written per
    step).  Only when the parameter holds no gradient (no accumulation in flight) and the view still fits; else a fresh tensor."""
    view = param.__dict__.get("_sw_grad_view")
    if (view is not None and param.grad is None and tuple(view.shape) == tuple(shape) and view.device == dev
            and view.dtype == torch.float32 and view.is_contiguous()):
        return view.detach()          # a tensor object of its own on the bucket's storage: autograd adopts a gradient only if nobody else holds it
    return torch.empty(*shape, device=dev, dtype=torch.float32)


# Weight staging registry: id(parameter) -> dict(kind, dtype, stage0, stage1, d0, d1, d2, ld0, stamp).  Modules register the
# persistent compute-dtype copies of their weights here; HipSGD's fused step (sw_sgd_multi) then rewrites those copies from
# the freshly updated values and calls stamp(key) so that the module's cache sees them as current.
STAGING = {}


def register_staging(p, kind, dtype, stage0=None, stage1=None, d0=0, d1=0, d2=0, ld0=0, ld1=0, stamp=None):
    """The registry must not keep a model alive: the parameter is held weakly and its entry (with the staged copies) leaves when the
    parameter dies — a strong reference here leaked every deleted model's fc6 weight and its two bf16 copies (0.8 GB per VGG16
    detector; bench.py builds and drops six).  `stamp` closures must not capture their module either."""
    import weakref
    key = id(p)
    fresh = key not in STAGING or STAGING[key]["param"]() is not p
    STAGING[key] = dict(param=weakref.ref(p), kind=kind, dtype=dtype, stage0=stage0, stage1=stage1, d0=d0, d1=d1, d2=d2, ld0=ld0,
                        ld1=ld1, stamp=stamp)
    if fresh:
        weakref.finalize(p, _drop_staging, key)


def _drop_staging(key):
    ent = STAGING.get(key)
    if ent is not None and ent["param"]() is None:             # (an id reused by a live parameter keeps its entry)
        STAGING.pop(key, None)


def _launch(tag, fn):
    if TIMER is None or tag is None:
        return fn()
    return TIMER.wrap(tag, fn)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sos-wsod_amd kernels need GPU tensors (there is no CPU fallback)")


def make_epilogue(bias=None, relu=False, drop_mask=None, drop_scale=2.0, relu_ref=None, ref_scale=1.0,
                  out_dtype=torch.float32, atomic=False, absmax_out=None, drop_hash=None, splitk_workspace=None, residual=None,
                  row_scale=None):
    """drop_hash=(seed, offset, p[, device counter]): dropout decided in the epilogue by the hash that sw_dropout_mask uses (no
    mask tensor); with a device counter (uint64 scalar tensor) the stream position is offset + *counter at run time"""
    ep = Epilogue()
    ep.bias = None if bias is None else bias.data_ptr()
    ep.relu = int(relu)
    ep.drop_mask = None if drop_mask is None else drop_mask.data_ptr()
    ep.ld_drop = 0 if drop_mask is None else drop_mask.stride(0)
    ep.drop_scale = float(drop_scale)
    if relu_ref is not None and relu_ref.dim() != 2:
        raise ValueError("relu_ref: a (rows, columns) matrix (its row pitch is the reference's leading dimension)")
    ep.relu_ref = None if relu_ref is None else relu_ref.data_ptr()
    ep.ld_ref = 0 if relu_ref is None else relu_ref.stride(0)
    ep.ref_scale = float(ref_scale)
    ep.ref_dtype = SW_F32 if relu_ref is None else dt(relu_ref)
    ep.out_dtype = dt(out_dtype)
    ep.accumulate_atomic = int(atomic)
    ep.absmax_out = None if absmax_out is None else absmax_out.data_ptr()
    ep.splitk_workspace = None if splitk_workspace is None else splitk_workspace.data_ptr()
    if drop_hash is not None and drop_mask is None:
        ep.drop_seed, ep.drop_offset, ep.drop_hash_p = int(drop_hash[0]) & (2 ** 64 - 1), int(drop_hash[1]), float(drop_hash[2])
        if len(drop_hash) > 3 and drop_hash[3] is not None:
            ep.drop_offset_dev = drop_hash[3].data_ptr()
    if residual is not None and residual.dim() != 2:
        raise ValueError("residual: a (rows, columns) matrix")
    ep.residual = None if residual is None else residual.data_ptr()
    ep.ld_res = 0 if residual is None else residual.stride(0)
    ep.res_dtype = SW_F32 if residual is None else dt(residual)
    ep.fold_row_scale = None if row_scale is None else row_scale.data_ptr()        # plain f32-output GEMMs: C[m][:] *= row_scale[m]
    ep._keepalive = (bias, drop_mask, relu_ref, absmax_out, splitk_workspace, drop_hash, residual, row_scale)     # the struct holds raw pointers only
    return ep


def gemm(A, B, C, M, N, K, a_kstrided=False, b_kstrided=False, lda=None, ldb=None, ldc=None, ep=None, splitk=1, tag=None):
    """C[m][n] = sum_k A(m,k) B(k,n); see sw_gemm.  A and B share one dtype (f32 / bf16)."""
    _need_gpu(A, B, C)
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    ldc = C.stride(0) if ldc is None else ldc
    if ep is None:
        ep = make_epilogue(out_dtype=C.dtype)
    if (C.dtype == torch.float32 or splitk > 1) and not ep.accumulate_atomic and not ep.splitk_workspace:
        # split-K (explicit, or the tail peel of the large weight-gradient GEMMs) through slabs + an ordered fold: no atomics,
        # bitwise reproducible; with an epilogue (bias / residual / ReLU / mask, bf16 C) the fold applies it
        need = int(lib.sw_gemm_splitk_workspace_floats(M, N, K, splitk))
        if need > 0:
            ws = torch.empty(need, device=C.device, dtype=torch.float32)
            ep.splitk_workspace = ws.data_ptr()
            ep._keepalive = ep._keepalive + (ws,)
    check(_launch(tag, lambda: lib.sw_gemm(dt(A), int(a_kstrided), int(b_kstrided), M, N, K, _p(A), lda, _p(B), ldb, _p(C),
                                           ldc, ctypes.byref(ep), splitk, _stream())), "sw_gemm")
    return C


def conv3x3_winograd(x, U, out, dilation, ep, tag=None):
    """x [n][H][W][Cin], U [16][Cout][Cin] (winograd_weight_prep), out [n][H][W][Cout], bf16.  -> True if the Winograd kernel ran"""
    _need_gpu(x, U, out)
    n, H, W, Cin = x.shape
    Cout = out.shape[3]
    rc = _launch(tag, lambda: lib.sw_conv3x3_winograd(dt(x), n, H, W, Cin, Cout, dilation, _p(x), _p(U), _p(out), ctypes.byref(ep), _stream()))
    if rc < 0:
        check(-rc, "sw_conv3x3_winograd")
    return rc == 1


def winograd_weight_prep(items):
    """items: [(w f32 OIHW master, U bf16 (16, n_out, n_in), mode)]: every layer's transformed filters in one launch"""
    from ._lib import WinogradPrep
    n = len(items)
    if n == 0:
        return
    arr = (WinogradPrep * n)()
    for i, (w, U, mode) in enumerate(items):
        _need_gpu(w, U)
        assert w.dtype == torch.float32 and w.is_contiguous() and U.dtype == torch.bfloat16 and U.is_contiguous() and U.numel() == 16 * w.shape[0] * w.shape[1]
        arr[i].w, arr[i].U, arr[i].Cout, arr[i].Cin, arr[i].mode = w.data_ptr(), U.data_ptr(), w.shape[0], w.shape[1], int(mode)
    check(lib.sw_winograd_weight_prep(n, arr, _stream()), "sw_winograd_weight_prep")


def conv3x3(x, wk, out, dilation, ep, tag=None):
    """x [n][H][W][Cin], wk [Cout][9][Cin], out [n][H][W][Cout]"""
    _need_gpu(x, wk, out)
    n, H, W, Cin = x.shape
    Cout = out.shape[3]
    if TIMER is not None and tag is not None:
        TIMER.note(tag, 2.0 * n * H * W * Cout * 9 * Cin)
    check(_launch(tag, lambda: lib.sw_conv3x3_igemm(dt(x), n, H, W, Cin, Cout, dilation, _p(x), _p(wk), _p(out),
                                                    ctypes.byref(ep), _stream())), "sw_conv3x3_igemm")
    return out


def conv3x3_relu_pool2(x, wk, bias, out_pooled, tag=None):
    """x [n][H][W][Cin] -> out_pooled [n][(H-2)//2+1][(W-2)//2+1][Cout] = maxpool2x2/2(relu(conv3x3(x) + bias)) in one launch
    (sw_conv3x3_relu_pool2); False when the shape is not covered (the caller then runs conv3x3 + maxpool_fwd)"""
    _need_gpu(x, wk, out_pooled)
    n, H, W, Cin = x.shape
    Cout = out_pooled.shape[3]
    if TIMER is not None and tag is not None:
        TIMER.note(tag, 2.0 * n * H * W * Cout * 9 * Cin)
    rc = _launch(tag, lambda: lib.sw_conv3x3_relu_pool2(dt(x), n, H, W, Cin, Cout, _p(x), _p(wk), _p(bias), _p(out_pooled), _stream()))
    if rc < 0:
        check(rc, "sw_conv3x3_relu_pool2")
    return rc == 1


def conv3x3_multi(problems):
    """problems: list of (x NHWC, wk [Cout][9][Cin], out NHWC, epilogue) — stride-1, dilation-1 convolutions of different maps (the FPN
    levels of a detector) in ONE launch of the direct kernel (sw_conv3x3_multi); shapes it does not cover run one by one"""
    from ._lib import ConvProblem
    n = len(problems)
    if n == 0:
        return
    if 1 < n <= 8 and problems[0][0].dtype == torch.bfloat16:
        arr = (ConvProblem * n)()
        for i, (x, wk, out, ep) in enumerate(problems):
            _need_gpu(x, wk, out)
            q = arr[i]
            q.nimg, q.H, q.W, q.Cin = x.shape
            q.Cout = out.shape[3]
            q.in_, q.wk, q.out, q.ep = x.data_ptr(), wk.data_ptr(), out.data_ptr(), ctypes.pointer(ep)
        rc = int(lib.sw_conv3x3_multi(SW_BF16, n, arr, _stream()))
        if rc == 1:
            return
        if rc < 0:
            check(rc, "sw_conv3x3_multi")
    for x, wk, out, ep in problems:
        conv3x3(x, wk, out, 1, ep)


def conv3x3_wgrad(x, dy, dw_oihw, dilation, splitk=1, workspace=None, tag=None, cout_scale=None, accumulate=False):
    """dw_oihw (Cout, Cin, 3, 3) f32 is overwritten (accumulate: added to); workspace: Cout*9*Cin floats (allocated here if None);
    cout_scale (Cout,) f32: dw[co] *= cout_scale[co] inside the slab fold (FrozenBN fold of the gradient)"""
    _need_gpu(x, dy, dw_oihw)
    n, H, W, Cin = x.shape
    Cout = dy.shape[3]
    need = int(lib.sw_conv3x3_wgrad_workspace_floats(dt(x), n, H, W, Cin, Cout, splitk))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x.device, dtype=torch.float32)
    check(_launch(tag, lambda: lib.sw_conv3x3_wgrad_acc(dt(x), n, H, W, Cin, Cout, dilation, _p(x), _p(dy), _p(dw_oihw),
                                                        _p(workspace), splitk, _p(cout_scale), int(accumulate), _stream())), "sw_conv3x3_wgrad")
    return dw_oihw


def conv3x3_wgrad_small(x, dy, dw_oihw, cout_scale=None, accumulate=False):
    """3x3 weight gradient of a map of a few pixels (sw_conv3x3_wgrad_small): x (n, H, W, Cin), dy (n, H, W, Cout) -> dw (Cout, Cin, 3, 3) f32
    (accumulate: added to dw)"""
    _need_gpu(x, dy, dw_oihw)
    n, H, W, Cin = x.shape
    check(lib.sw_conv3x3_wgrad_small_acc(dt(x), n, H, W, Cin, dy.shape[3], _p(x), _p(dy), _p(cout_scale), _p(dw_oihw), int(accumulate),
                                         _stream()), "sw_conv3x3_wgrad_small")
    return dw_oihw


def conv3x3_wgrad_nslab(x, cout, splitk):
    """slabs that conv3x3_wgrad_slabs(x, dy -> cout channels, splitk) writes"""
    n, H, W, Cin = x.shape
    return int(lib.sw_conv3x3_wgrad_workspace_floats(dt(x), n, H, W, Cin, cout, splitk)) // (cout * 9 * Cin)


def conv3x3_wgrad_nslab_shape(dtype, n, H, W, Cin, cout, splitk):
    """conv3x3_wgrad_nslab for a batch given by its shape and torch dtype"""
    code = SW_BF16 if dtype == torch.bfloat16 else SW_F32
    return int(lib.sw_conv3x3_wgrad_workspace_floats(code, n, H, W, Cin, cout, splitk)) // (cout * 9 * Cin)


def conv3x3_wgrad_slabs(x, dy, workspace, dilation, splitk=1):
    """split-K partial weight gradients of one (x, dy) pair into `workspace` (no fold): see sw_conv3x3_wgrad_slabs"""
    _need_gpu(x, dy, workspace)
    n, H, W, Cin = x.shape
    check(lib.sw_conv3x3_wgrad_slabs(dt(x), n, H, W, Cin, dy.shape[3], dilation, _p(x), _p(dy), _p(workspace), splitk,
                                     _stream()), "sw_conv3x3_wgrad_slabs")


def conv3x3_wgrad_grouped(problems, tag=None):
    """problems: list of (x NHWC, dy NHWC, slabs f32 tensor, dilation, nsplit) — every weight gradient of a backward pass in one
    launch (sw_conv3x3_wgrad_grouped); each writes conv3x3_wgrad_nslab(x, cout, nsplit) slabs at `slabs`"""
    from ._lib import WgradProblem
    n = len(problems)
    if n == 0:
        return
    arr = (WgradProblem * n)()
    for i, (x, dy, slabs, dil, nsplit) in enumerate(problems):
        _need_gpu(x, dy, slabs)
        q = arr[i]
        q.nimg, q.H, q.W, q.Cin = x.shape
        q.Cout, q.dilation, q.nsplit = dy.shape[3], int(dil), int(nsplit)
        q.x, q.dy, q.slabs = x.data_ptr(), dy.data_ptr(), slabs.data_ptr()
    if TIMER is not None and tag is not None:
        TIMER.note(tag, sum(2.0 * x.numel() * 9 * dy.shape[3] for x, dy, _, _, _ in problems))
    check(_launch(tag, lambda: lib.sw_conv3x3_wgrad_grouped(dt(problems[0][0]), n, arr, _stream())), "sw_conv3x3_wgrad_grouped")


def conv3x3_wgrad_fold(workspace, nslab, dw_oihw, cout_scale=None, accumulate=False):
    """dw_oihw (Cout, Cin, 3, 3) f32 = (+=, accumulate) cout_scale[co] * ordered sum of `nslab` consecutive [co][tap][ci] slabs"""
    Cout, Cin = dw_oihw.shape[:2]
    check(lib.sw_conv3x3_wgrad_fold_acc(Cin, Cout, nslab, _p(workspace), _p(dw_oihw), _p(cout_scale), int(accumulate), _stream()),
          "sw_conv3x3_wgrad_fold")
    return dw_oihw


def conv3x3_wgrad_fold_multi(folds):
    """folds: list of (workspace, nslab, dw_oihw): every weight-gradient fold of a backward pass in one launch"""
    from ._lib import WgradFold
    n = len(folds)
    if n == 0:
        return
    arr = (WgradFold * n)()
    for i, (ws, nslab, dw) in enumerate(folds):
        arr[i].Cout, arr[i].Cin, arr[i].nslab = dw.shape[0], dw.shape[1], int(nslab)
        arr[i].workspace, arr[i].dw_oihw = ws.data_ptr(), dw.data_ptr()
    check(lib.sw_conv3x3_wgrad_fold_multi(n, arr, _stream()), "sw_conv3x3_wgrad_fold_multi")


def gemm_kk_nslab(dtype, K, nsplit):
    """slabs a problem of gemm_kk_grouped with this K and split count writes"""
    return int(lib.sw_gemm_kk_grouped_slabs(dt(dtype), int(K), int(nsplit)))


def gemm_kk_grouped(problems):
    """problems: list of (A (K, M) rows of pitch lda, B (K, N) rows of pitch ldb, slabs f32, nsplit): slabs[z] = A_z^T B_z for all of them
    in ONE launch (sw_gemm_kk_grouped: resident 256x256-tile grid over every problem's K-splits)"""
    from ._lib import GemmKKProblem
    n = len(problems)
    if n == 0:
        return
    arr = (GemmKKProblem * n)()
    for i, (A, B, slabs, nsplit) in enumerate(problems):
        _need_gpu(A, B, slabs)
        q = arr[i]
        q.A, q.B, q.slabs = A.data_ptr(), B.data_ptr(), slabs.data_ptr()
        q.K, q.M, q.N, q.nsplit = A.shape[0], A.shape[1], B.shape[1], int(nsplit)
        q.lda, q.ldb = A.stride(0), B.stride(0)
    check(lib.sw_gemm_kk_grouped(dt(problems[0][0]), n, arr, _stream()), "sw_gemm_kk_grouped")


def splitk_fold_multi(folds):
    """folds: list of (workspace, nslab, C (M, N) f32, row_scale or None, accumulate): C = [C +] row_scale[:, None] * sum of the slabs,
    all in ONE launch (sw_splitk_fold_multi)"""
    from ._lib import SplitkFold
    n = len(folds)
    if n == 0:
        return
    arr = (SplitkFold * n)()
    for i, (ws, nslab, C, rs, acc) in enumerate(folds):
        _need_gpu(ws, C)
        q = arr[i]
        q.M, q.N, q.nslab, q.accumulate = C.shape[0], C.shape[1], int(nslab), int(bool(acc))
        q.workspace, q.C, q.ldc, q.row_scale = ws.data_ptr(), C.data_ptr(), C.stride(0), (None if rs is None else rs.data_ptr())
    check(lib.sw_splitk_fold_multi(n, arr, _stream()), "sw_splitk_fold_multi")


def colsum_fold_multi(folds):
    """folds: list of (workspace, n_rows, out): every bias-gradient fold of a backward pass in one launch"""
    from ._lib import ColsumFold
    n = len(folds)
    if n == 0:
        return
    arr = (ColsumFold * n)()
    for i, (ws, rows, out) in enumerate(folds):
        arr[i].N, arr[i].n_partial_rows, arr[i].workspace, arr[i].out = out.numel(), int(rows), ws.data_ptr(), out.data_ptr()
    check(lib.sw_colsum_fold_multi(n, arr, _stream()), "sw_colsum_fold_multi")


def colsum_nrows(X_dtype, M, N):
    """partial rows that colsum_partial writes for an M x N matrix"""
    return int(lib.sw_colsum_workspace_floats(dt(X_dtype), M, N)) // N


def colsum_partial(X, M, N, workspace, ld=None):
    check(lib.sw_colsum_partial(dt(X), M, N, _p(X), X.stride(0) if ld is None else ld, _p(workspace), _stream()),
          "sw_colsum_partial")


def colsum_partial_multi(parts):
    """parts: list of (X 2-D view (M, N) with unit inner stride, workspace f32) — every colsum_partial of a backward pass in ONE launch"""
    from ._lib import ColsumPart
    n = len(parts)
    if n == 0:
        return
    arr = (ColsumPart * n)()
    for i, (X, ws) in enumerate(parts):
        _need_gpu(X, ws)
        q = arr[i]
        q.M, q.N = X.shape
        q.X, q.ld, q.workspace = X.data_ptr(), X.stride(0), ws.data_ptr()
    check(lib.sw_colsum_partial_multi(dt(parts[0][0]), n, arr, _stream()), "sw_colsum_partial_multi")


def colsum_fold(workspace, n_rows, out):
    check(lib.sw_colsum_fold(out.numel(), n_rows, _p(workspace), _p(out), _stream()), "sw_colsum_fold")
    return out


def conv_weight_prep(w_oihw, wk, mode, cin_pad=None):
    Cout, Cin = w_oihw.shape[:2]
    cin_pad = Cin if cin_pad is None else cin_pad
    check(lib.sw_conv_weight_prep(dt(wk), mode, Cout, Cin, cin_pad, _p(w_oihw), _p(wk), _stream()), "sw_conv_weight_prep")
    return wk


def maxpool_fwd(x, out, stride):
    n, H, W, C = x.shape
    check(lib.sw_maxpool2x2_fwd(dt(x), n, H, W, C, stride, _p(x), _p(out), _stream()), "sw_maxpool2x2_fwd")
    return out


def maxpool_bwd(x, dout, din, stride, relu_mask):
    n, H, W, C = x.shape
    check(lib.sw_maxpool2x2_bwd(dt(x), n, H, W, C, stride, _p(x), _p(dout), _p(din), int(relu_mask), _stream()),
          "sw_maxpool2x2_bwd")
    return din


def preprocess(img_u8_chw, out_hwc, mean, std):
    _need_gpu(img_u8_chw, out_hwc)
    H, W, cpad = out_hwc.shape
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(lib.sw_preprocess(dt(out_hwc), H, W, cpad, _p(img_u8_chw), m, s, _p(out_hwc), _stream()), "sw_preprocess")
    return out_hwc


def preprocess_multi(imgs_u8_chw, out_nhwc, mean, std):
    """n u8 CHW images of one size -> out [n][H][W][cpad], one launch"""
    _need_gpu(out_nhwc, *imgs_u8_chw)
    n, H, W, cpad = out_nhwc.shape
    assert len(imgs_u8_chw) == n and all(im.is_contiguous() and tuple(im.shape) == (3, H, W) for im in imgs_u8_chw)
    ptrs = (ctypes.c_void_p * n)(*[im.data_ptr() for im in imgs_u8_chw])
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(lib.sw_preprocess_multi(dt(out_nhwc), n, H, W, cpad, ptrs, m, s, _p(out_nhwc), _stream()), "sw_preprocess_multi")
    return out_nhwc


def roi_pool_fwd_workspace(n, R, PH, PW, device):
    """scratch for roi_pool_fwd's prepared-task form (sw_roi_pool_fwd_ws); uint8, 256-byte aligned by the allocator"""
    return torch.empty(max(16, lib.sw_roi_pool_fwd_workspace_bytes(n, R, PH, PW)), device=device, dtype=torch.uint8)


def roi_pool_fwd(feat, rois, out, argmax, spatial_scale, PH, PW, row_scale=None, row_scale_add=0.0, tag=None, workspace="auto"):
    """feat [n][H][W][C], rois [R][5] f32, out [R][C*PH*PW]; argmax same shape, int32 (h*W+w or -1) or int16/uint16
    storage holding uint16 (h*W+w or 0xFFFF; see argmax_to_int32).  workspace: roi_pool_fwd_workspace(...) tensor, "auto" = allocate
    one here (caching allocator: graph-safe), None = the entry without workspace (sw_roi_pool_fwd)"""
    _need_gpu(feat, rois, out, argmax)
    n, H, W, C = feat.shape
    R = rois.shape[0]
    if isinstance(workspace, str):
        workspace = roi_pool_fwd_workspace(n, R, PH, PW, feat.device) if R > 0 else None
    if workspace is None:
        check(_launch(tag, lambda: lib.sw_roi_pool_fwd(dt(feat), n, H, W, C, PH, PW, float(spatial_scale), _p(feat), _p(rois), R,
                                                       _p(row_scale), float(row_scale_add), _p(out), _p(argmax), _argmax_bits(argmax),
                                                       _roi_pitch(out, argmax), _stream())), "sw_roi_pool_fwd")
    else:
        check(_launch(tag, lambda: lib.sw_roi_pool_fwd_ws(dt(feat), n, H, W, C, PH, PW, float(spatial_scale), _p(feat), _p(rois), R,
                                                          _p(row_scale), float(row_scale_add), _p(out), _p(argmax), _argmax_bits(argmax),
                                                          _roi_pitch(out, argmax), _p(workspace), workspace.numel(), _stream())),
              "sw_roi_pool_fwd_ws")
    return out, argmax


def _roi_pitch(vals, argmax):
    if vals.stride(0) != argmax.stride(0) or vals.stride(-1) != 1 or argmax.stride(-1) != 1:
        raise ValueError("ROIPool values and argmax must share one row pitch (unit inner stride)")
    return vals.stride(0)


def _argmax_bits(argmax):
    if argmax.dtype == torch.int32:
        return 32
    if argmax.dtype in (torch.int16, torch.uint16):
        return 16
    raise TypeError(f"ROIPool argmax must be int32 or (u)int16, got {argmax.dtype}")


def roi_argmax_dtype(H, W):
    """compact index type for an H x W map: uint16 (int16 storage) below 65535 pixels, else int32"""
    return torch.int16 if H * W < 65535 else torch.int32


def argmax_to_int32(argmax):
    """decode either storage to the reference's int32 convention (-1 = empty bin)"""
    if argmax.dtype == torch.int32:
        return argmax
    a = argmax.view(torch.int16).to(torch.int32) & 0xFFFF
    return torch.where(a == 0xFFFF, torch.full_like(a, -1), a)


def absmax(x, out=None):
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float32)
    check(lib.sw_absmax(dt(x), x.numel(), _p(x), _p(out), _stream()), "sw_absmax")
    return out


def roi_pool_bwd(dout, argmax, rois, dfeat, PH, PW, row_scale=None, row_scale_add=0.0, relu_ref=None, dout_absmax="auto",
                 tag=None, spatial_scale=0.0):
    """dout_absmax: device scalar >= max|dout| (selects the fixed-point accumulation), "auto" = compute it here,
    None = LDS float atomics.  spatial_scale: the forward's scale (lets large maps skip ROIs outside a workgroup's pixel range)."""
    _need_gpu(dout, argmax, rois, dfeat)
    n, H, W, C = dfeat.shape
    R = rois.shape[0]
    if isinstance(dout_absmax, str):
        dout_absmax = absmax(dout)
    check(_launch(tag, lambda: lib.sw_roi_pool_bwd(dt(dfeat), n, H, W, C, PH, PW, _p(dout), _p(argmax), _argmax_bits(argmax),
                                                   _roi_pitch(dout, argmax), _p(rois), R, _p(row_scale), float(row_scale_add),
                                                   _p(relu_ref), _p(dout_absmax), _p(dfeat), float(spatial_scale), _stream())),
          "sw_roi_pool_bwd")
    return dfeat


def wsddn_mil(logits, V, R, K, cls_col, det_col, gt_onehot, scores, loss_view, dlogits=None, grad_scale=None,
              mean_scores=None, workspace=None):
    """mean_scores: optional [R][>=K] f32 (row pitch = its stride) receiving the view-mean scores"""
    _need_gpu(logits, gt_onehot, scores, loss_view)
    if workspace is None:
        workspace = torch.empty(int(lib.sw_wsddn_workspace_floats(V, R, K)), device=logits.device, dtype=torch.float32)
    check(lib.sw_wsddn_mil(V, R, K, _p(logits), logits.stride(0), cls_col, det_col, _p(gt_onehot), _p(scores),
                           _p(loss_view), _p(dlogits), 0 if dlogits is None else dlogits.stride(0), _p(grad_scale),
                           _p(mean_scores), 0 if mean_scores is None else mean_scores.stride(-2), _p(workspace),
                           _stream()), "sw_wsddn_mil")


def sgd_multi(entries, momentum, grad_scale=1.0):
    """entries: list of dict(param, grad, buf, lr, weight_decay, first, staging=None|STAGING entry, hyper=None|device tensor
    {lr, weight_decay} read by the kernel instead of the two floats).  One launch per 24."""
    from ._lib import SgdTensor
    n = len(entries)
    if n == 0:
        return
    arr = (SgdTensor * n)()
    for i, e in enumerate(entries):
        _fill_sgd_tensor(arr[i], e)
    check(lib.sw_sgd_multi(n, arr, float(momentum), float(grad_scale), _stream()), "sw_sgd_multi")


def _fill_sgd_tensor(d, e):
    p = e["param"]
    g = e.get("grad")
    d.param, d.grad, d.momentum_buf, d.n = p.data_ptr(), (None if g is None else g.data_ptr()), e["buf"].data_ptr(), p.numel()
    d.lr, d.weight_decay, d.first_step = float(e["lr"]), float(e["weight_decay"]), int(e["first"])
    hy = e.get("hyper")
    d.hyper_dev = None if hy is None else hy.data_ptr()
    st = e.get("staging")
    if st is None:
        d.stage_kind = 0
    else:
        d.stage_kind, d.stage_dtype = st["kind"], dt(st["dtype"])
        d.stage0 = None if st["stage0"] is None else st["stage0"].data_ptr()
        d.stage1 = None if st["stage1"] is None else st["stage1"].data_ptr()
        d.d0, d.d1, d.d2, d.ld0, d.ld1 = st["d0"], st["d1"], st["d2"], st["ld0"], st.get("ld1", 0)


def gemm_sgd_fused_supported(dtype, M, N, K, a_kstrided=False, b_kstrided=False):
    """can sw_gemm take an epilogue with `sgd_fused` for this shape (the ping-pong 256x256 form; see sw_epilogue.sgd_fused)?"""
    return bool(lib.sw_gemm_sgd_fused_supported(dt(dtype), int(a_kstrided), int(b_kstrided), int(M), int(N), int(K)))


def attach_sgd_fused(ep, entry, momentum, grad_scale=1.0):
    """ep: an Epilogue for the weight-gradient GEMM of a 2D parameter; entry: the optimizer's record for that parameter (as for sgd_multi,
    without a gradient).  The GEMM then applies the SGD update in its epilogue instead of writing the gradient (sw_epilogue.sgd_fused)."""
    from ._lib import SgdTensor
    d = SgdTensor()
    _fill_sgd_tensor(d, entry)
    ep.sgd_fused = ctypes.addressof(d)
    ep.sgd_momentum, ep.sgd_grad_scale = float(momentum), float(grad_scale)
    ep._keepalive = ep._keepalive + (d, entry["param"], entry["buf"], entry.get("hyper"))
    return ep


def mean_views(x, out):
    V = x.shape[0]
    check(lib.sw_mean_views(V, out.numel(), _p(x), _p(out), _stream()), "sw_mean_views")
    return out


def oicr_mean_probs(logits, V, R, K, n_rounds, cls_col0, col_stride, out):
    """out [n_rounds][R][K+1] = view-mean softmax of each refinement head's class logits"""
    _need_gpu(logits, out)
    check(lib.sw_oicr_mean_probs(V, R, K, n_rounds, _p(logits), logits.stride(0), cls_col0, col_stride, _p(out), _stream()),
          "sw_oicr_mean_probs")
    return out


def mine_workspace_bytes(R, top_k, G, n_rounds=1):
    return int(lib.sw_mine_workspace_bytes(R, top_k, G)) * n_rounds


def oicr_mine_label(scores, gt_classes_i32, boxes, K, top_k, thresh, nms_thresh, iou_bg, iou_fg, lab_class, lab_weight,
                    lab_index, pgt_count, pgt_index, pgt_class, pgt_score, workspace):
    """scores [R][ncol] (one round) or [n_rounds][R][ncol]; the outputs carry the same leading dimension"""
    _need_gpu(scores, gt_classes_i32, boxes)
    n_rounds = 1 if scores.dim() == 2 else scores.shape[0]
    R, ncol = scores.shape[-2:]
    G = gt_classes_i32.numel()
    _launch("mine_label", lambda: check(
        lib.sw_oicr_mine_label(R, ncol, K, n_rounds, _p(scores), _p(gt_classes_i32), G, _p(boxes), int(top_k),
                               float(thresh), float(nms_thresh), float(iou_bg), float(iou_fg), _p(lab_class),
                               _p(lab_weight), _p(lab_index), _p(pgt_count), _p(pgt_index), _p(pgt_class), _p(pgt_score),
                               _p(workspace), _stream()), "sw_oicr_mine_label"))


def oicr_refine_loss(logits, V, R, K, cls_col, box_col, boxes, lab_class, lab_weight, lab_index, pred_view, reg_weights,
                     loss_view, dlogits=None, grad_scale=None, workspace=None, n_rounds=1, col_stride=0):
    """n_rounds heads at columns cls_col/box_col + k*col_stride; lab_* [n_rounds][R]; loss_view [n_rounds][2][V];
    grad_scale device float[2*n_rounds]"""
    rw = (ctypes.c_float * 4)(*[float(v) for v in reg_weights])
    if workspace is None:
        workspace = torch.empty(n_rounds * 2 * V * R, device=logits.device, dtype=torch.float32)
    check(lib.sw_oicr_refine_loss(V, R, K, n_rounds, _p(logits), logits.stride(0), cls_col, box_col, col_stride, _p(boxes),
                                  _p(lab_class), _p(lab_weight), _p(lab_index), _p(pred_view), rw, _p(loss_view),
                                  _p(dlogits), 0 if dlogits is None else dlogits.stride(0), _p(grad_scale), _p(workspace),
                                  _stream()), "sw_oicr_refine_loss")


def colsum(X, M, N, out, ld=None, accumulate=False):
    """out[n] = sum_m X[m, n] (f32), deterministic: partial rows per row chunk + an ordered fold (no zero fill, no atomics);
    accumulate: out[n] += the sum"""
    need = int(lib.sw_colsum_workspace_floats(dt(X), M, N))
    ws = torch.empty(max(need, 4), device=X.device, dtype=torch.float32)
    check(lib.sw_colsum_acc(dt(X), M, N, _p(X), X.stride(0) if ld is None else ld, _p(out), _p(ws), int(accumulate), _stream()), "sw_colsum")
    return out


def convert_2d(src_f32, dst, rows, cols, ld_src=None, ld_dst=None):
    check(lib.sw_convert_2d(dt(dst), rows, cols, _p(src_f32), src_f32.stride(0) if ld_src is None else ld_src, _p(dst),
                            dst.stride(0) if ld_dst is None else ld_dst, _stream()), "sw_convert_2d")
    return dst


def split_bf16x3(src_f32, side, along_rows=False, out=None):
    """the K-concatenated three-piece bf16 operand of an f32 matrix (sw_split_bf16x3): (rows, cols) f32 -> (rows, 6 cols) bf16, or
    (6 rows, cols) with along_rows.  side 0 = the A operand's block pattern, 1 = the B operand's.  Row pitch padded by 128 elements."""
    _need_gpu(src_f32)
    rows, cols = src_f32.shape
    assert src_f32.dtype == torch.float32 and src_f32.stride(1) == 1
    shape = (6 * rows, cols) if along_rows else (rows, 6 * cols)
    if out is None or tuple(out.shape) != shape:
        out = torch.empty(shape[0], shape[1] + 128, device=src_f32.device, dtype=torch.bfloat16)[:, :shape[1]]
    check(lib.sw_split_bf16x3(rows, cols, _p(src_f32), src_f32.stride(0), _p(out), out.stride(0), int(side), int(bool(along_rows)), _stream()),
          "sw_split_bf16x3")
    return out


def gemm_f32x3(A, B, C, M, N, K, a_kstrided=False, b_kstrided=False, ep=None, tag=None, A3=None, B3=None):
    """C = A . B for f32 operands at (about) f32 accuracy on the bf16 MFMA: both operands split into three bf16 pieces, six products
    accumulated in f32 as ONE bf16 GEMM over 6 K (sw_split_bf16x3).  A3 / B3: an operand already split (weights, cached per update).
    Operand forms as ops.gemm: A (M, K) or, a_kstrided, (K, M); B (N, K) or, b_kstrided, (K, N)."""
    if A3 is None:
        A3 = split_bf16x3(A, 0, along_rows=a_kstrided)
    if B3 is None:
        B3 = split_bf16x3(B, 1, along_rows=b_kstrided)
    return gemm(A3, B3, C, M, N, 6 * K, a_kstrided=a_kstrided, b_kstrided=b_kstrided, ep=ep, tag=tag)


def convert_2d_t(src_f32, dst, rows, cols):
    """dst[c][r] = src[r][c] in dst's dtype (rows, cols multiples of 64)"""
    check(lib.sw_convert_2d_t(dt(dst), rows, cols, _p(src_f32), src_f32.stride(0), _p(dst), dst.stride(0), _stream()),
          "sw_convert_2d_t")
    return dst


def to_f32(src, dst):
    check(lib.sw_to_f32(dt(src), src.numel(), _p(src), _p(dst), _stream()), "sw_to_f32")
    return dst


def fill_zero(t):
    check(lib.sw_fill_zero(_p(t), t.numel() * t.element_size(), _stream()), "sw_fill_zero")
    return t


def dropout_mask(keep_u8, seed, offset, p=0.5):
    check(lib.sw_dropout_mask(_p(keep_u8), keep_u8.numel(), int(seed) & (2 ** 64 - 1), int(offset), float(p), _stream()),
          "sw_dropout_mask")
    return keep_u8


def sgd_momentum_step(param, grad, buf, lr, momentum, weight_decay, first_step, grad_scale=1.0):
    check(lib.sw_sgd_momentum_step(_p(param), _p(grad), _p(buf), param.numel(), float(lr), float(momentum),
                                   float(weight_decay), int(first_step), float(grad_scale), _stream()),
          "sw_sgd_momentum_step")


def loss_finalize(loss_view, out, total=None):
    """loss_view [n_losses][V] or [n_images][n_losses][V] -> out[n_losses] (mean over views and images), total[1] = sum"""
    if loss_view.dim() == 2:
        loss_view = loss_view[None]
    B, n, V = loss_view.shape
    check(lib.sw_loss_finalize(n, V, B, _p(loss_view), _p(out), _p(total), _stream()), "sw_loss_finalize")
    return out


def scale_cols_loss(src_f32, g_losses, g_total, col_to_loss_i32, mul, dst, M, N, n_valid):
    check(lib.sw_scale_cols_loss(dt(dst), M, N, n_valid, _p(src_f32), src_f32.stride(0), _p(g_losses), _p(g_total),
                                 _p(col_to_loss_i32), float(mul), _p(dst), dst.stride(0), _stream()), "sw_scale_cols_loss")
    return dst


def resize_pass_u8(src, dst, bounds_i32, kk_i32, ksize, horizontal, dst_flip=None):
    """one pass of Pillow's 8-bit resize (sw_resize_pass_u8): src (C,H,W) u8 -> dst (C,H,out) / (C,out,W).  `src` may be a
    window of a larger planar image (unit pixel stride, the parent's row / plane strides): a crop costs no copy"""
    _need_gpu(src, dst, bounds_i32, kk_i32)
    C, H, W = src.shape
    assert src.stride(2) == 1 and dst.is_contiguous() and (dst_flip is None or dst_flip.is_contiguous())
    out_size = dst.shape[2] if horizontal else dst.shape[1]
    check(lib.sw_resize_pass_u8(C, H, W, src.stride(1), src.stride(0), out_size, int(horizontal), _p(src), _p(bounds_i32),
                                _p(kk_i32), int(ksize), _p(dst), _p(dst_flip), _stream()), "sw_resize_pass_u8")
    return dst


def color_jitter_u8(src, w_bright=None, w_sat=None, with_flip=False):
    """RandomBrightness / RandomSaturation blends on a planar (3,H,W) u8 image (sw_color_jitter_u8); weights None = that blend
    is skipped.  -> out, or (out, x-mirrored out)"""
    import numpy as np
    _need_gpu(src)
    assert src.dtype == torch.uint8 and src.dim() == 3 and src.shape[0] == 3 and src.is_contiguous()
    out = torch.empty_like(src)
    flip = torch.empty_like(src) if with_flip else None
    mode = (1 if w_bright is not None else 0) | (2 if w_sat is not None else 0)
    wb = float(np.float32(w_bright)) if w_bright is not None else 1.0        # numpy scales the float32 image by float32(w)
    ws = float(np.float32(w_sat)) if w_sat is not None else 1.0
    src_w = (1 - float(w_sat)) if w_sat is not None else 0.0                   # `1 - w`: a python double in the reference
    check(lib.sw_color_jitter_u8(src.shape[1], src.shape[2], mode, _p(src), wb, src_w, ws, _p(out), _p(flip), _stream()),
          "sw_color_jitter_u8")
    return (out, flip) if with_flip else out


def transpose_2d(src, dst, rows, cols):
    """dst[c][r] = src[r][c]; src (rows, cols), dst (cols, rows), same dtype, unit inner stride"""
    _need_gpu(src, dst)
    check(lib.sw_transpose_2d(dt(src), rows, cols, _p(src), src.stride(0), _p(dst), dst.stride(0), _stream()), "sw_transpose_2d")
    return dst


class StagePlan:
    """The device-resident entry table of sw_stage_weights_multi for one model and compute dtype: built once (the tensors it points
    at must stay where they are: parameters / buffers are updated in place, the destinations are owned by the plan's entries),
    run() = one launch.  entries: dicts(kind, w, dst, bn=None | (weight, bias, mean, var), scale=None, shift=None)."""

    def __init__(self, entries, compute_dtype, eps=1e-5):
        from ._lib import StageDesc
        n = len(entries)
        arr = (StageDesc * max(n, 1))()
        blocks = 0
        keep = []
        for i, e in enumerate(entries):
            w = e["w"]
            _need_gpu(w, e["dst"])
            if w.dtype != torch.float32 or not w.is_contiguous() or not e["dst"].is_contiguous():
                raise TypeError("stage plan takes contiguous float32 sources and contiguous destinations")
            kind = int(e["kind"])
            rows, cols = (w.shape[0], w.numel() // w.shape[0]) if kind in (0, 3) else (w.shape[0], w.shape[1])
            want = torch.float32 if kind == 3 else compute_dtype
            if e["dst"].dtype != want or e["dst"].numel() < w.numel():
                raise TypeError("stage plan destination has the wrong dtype or size")
            d = arr[i]
            d.w, d.dst, d.kind, d.rows, d.cols, d.block_start = w.data_ptr(), e["dst"].data_ptr(), kind, rows, cols, blocks
            bn = e.get("bn")
            if bn is not None:
                for t in bn:
                    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != rows:
                        raise TypeError("FrozenBN buffers must be contiguous float32 of Cout elements")
                d.bn_weight, d.bn_bias, d.bn_mean, d.bn_var = (t.data_ptr() for t in bn)
                d.scale = None if e.get("scale") is None else e["scale"].data_ptr()
                d.shift = None if e.get("shift") is None else e["shift"].data_ptr()
            blocks += int(lib.sw_stage_blocks(kind, rows, cols))
            keep.append((w, e["dst"], bn, e.get("scale"), e.get("shift")))
        self.n, self.blocks, self.eps, self.dtype = n, blocks, float(eps), compute_dtype
        self._keep = keep
        self.ptrs = tuple(w.data_ptr() for w, *_ in keep)
        raw = bytes(arr) if n else b""
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(entries[0]["w"].device) if n else None

    def run(self):
        if self.n:
            check(lib.sw_stage_weights_multi(dt(self.dtype), self.n, self.table.data_ptr(), self.blocks, self.eps, _stream()),
                  "sw_stage_weights_multi")


class EmaPlan:
    """the host-side argument arrays of sw_ema_multi for fixed lists of tensors (built once: ~600 state-dict entries of a detector)"""

    def __init__(self, teacher, student):
        n = len(teacher)
        assert n == len(student)
        for t, s_ in zip(teacher, student):
            _need_gpu(t, s_)
            if t.dtype != torch.float32 or s_.dtype != torch.float32 or not t.is_contiguous() or not s_.is_contiguous() or t.numel() != s_.numel():
                raise TypeError("ema_multi takes matching contiguous float32 tensors")
        self.n = n
        self.tp = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in teacher])
        self.sp = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in student])
        self.ne = (ctypes.c_long * max(n, 1))(*[t.numel() for t in teacher])
        self._keep = (list(teacher), list(student))

    def run(self, keep_rate):
        if self.n:
            check(lib.sw_ema_multi(self.n, self.tp, self.sp, self.ne, float(keep_rate), _stream()), "sw_ema_multi")


def ema_multi(teacher, student, keep_rate):
    """teacher[i] <- student[i] * (1 - keep_rate) + teacher[i] * keep_rate over lists of contiguous f32 tensors (sw_ema_multi)"""
    EmaPlan(teacher, student).run(keep_rate)


def weighted_sum(values, weights, out):
    """out[i] = values[i] * weights[i] (f32 scalars anywhere on the device), out[n] = their sum in index order (sw_weighted_sum)"""
    n = len(values)
    _need_gpu(out, *values)
    assert all(v.dtype == torch.float32 and v.numel() == 1 for v in values) and out.dtype == torch.float32 and out.numel() >= n + 1
    ptrs = (ctypes.c_void_p * n)(*[v.data_ptr() for v in values])
    ws = (ctypes.c_float * n)(*[float(w) for w in weights])
    check(lib.sw_weighted_sum(n, ptrs, ws, _p(out), _stream()), "sw_weighted_sum")
    return out


def scale_scalars(g, weights, out):
    """out[i] = g * weights[i] (sw_scale_scalars)"""
    n = len(weights)
    _need_gpu(g, out)
    ws = (ctypes.c_float * n)(*[float(w) for w in weights])
    check(lib.sw_scale_scalars(n, _p(g), ws, _p(out), _stream()), "sw_scale_scalars")
    return out


def threshold_select(scores, classes, boxes, thres, allowed=None):
    """-> (count[1] i32, boxes [n,4], classes [n] i32 or None, scores [n], index [n] i32): entries with score > thres (and class in
    `allowed`, an int32 device tensor, when given), compacted in input order; the first count rows are valid"""
    _need_gpu(scores, boxes)
    n = scores.shape[0]
    dev = scores.device
    cnt = torch.empty(1, device=dev, dtype=torch.int32)
    ob = torch.empty(max(n, 1), 4, device=dev); osc = torch.empty(max(n, 1), device=dev)
    oc = torch.empty(max(n, 1), device=dev, dtype=torch.int32) if classes is not None else None
    oi = torch.empty(max(n, 1), device=dev, dtype=torch.int32)
    check(lib.sw_threshold_select(n, _p(scores), _p(classes), _p(boxes), float(thres), _p(allowed),
                                  0 if allowed is None else allowed.numel(), _p(cnt), _p(ob), _p(oc), _p(osc), _p(oi), _stream()),
          "sw_threshold_select")
    return cnt, ob, oc, osc, oi


def copy_multi(pairs):
    """[(src, dst)] contiguous device tensors of equal dtype and shape: all copied by one launch"""
    from ._lib import CopyDesc
    pairs = [(s, d) for s, d in pairs if d.numel()]
    if not pairs:
        return
    arr = (CopyDesc * len(pairs))()
    for i, (s, d) in enumerate(pairs):
        assert s.is_cuda and d.is_cuda and s.dtype == d.dtype and s.shape == d.shape and s.is_contiguous() and d.is_contiguous()
        arr[i].src, arr[i].dst, arr[i].bytes = s.data_ptr(), d.data_ptr(), d.numel() * d.element_size()
    check(lib.sw_copy_multi(len(pairs), arr, _stream()), "sw_copy_multi")


def focal_loss(logits, targets_i32, gamma, loss, dlogits=None):
    """loss[0] = sum_r (1 - p_r)^gamma CE_r / N (sw_focal_loss); dlogits (N, C) f32 optional"""
    _need_gpu(logits, targets_i32, loss)
    N, C = logits.shape
    ws = torch.empty(N, device=logits.device, dtype=torch.float32)
    check(lib.sw_focal_loss(N, C, _p(logits), logits.stride(0), _p(targets_i32), float(gamma), _p(loss), _p(dlogits),
                            0 if dlogits is None else dlogits.stride(0), _p(ws), _stream()), "sw_focal_loss")
    return loss


def counter_add(counter_u64, increment):
    """*counter += increment in stream order (the device-resident dropout stream position)"""
    check(lib.sw_counter_add(_p(counter_u64), int(increment), _stream()), "sw_counter_add")


def pack_views(box_list, obj_list, boxes, obj, rois):
    """4 x (R,4) f32 boxes + 4 x (R,) f32 objectness (contiguous device tensors) -> boxes (4,R,4), obj (4,R), rois (2,2R,5)"""
    _need_gpu(*box_list, *obj_list, boxes, obj, rois)
    R = box_list[0].shape[0]
    bp = (ctypes.c_void_p * 4)(*[b.data_ptr() for b in box_list])
    op = (ctypes.c_void_p * 4)(*[o.data_ptr() for o in obj_list])
    check(lib.sw_pack_views(R, bp, op, _p(boxes), _p(obj), _p(rois), _stream()), "sw_pack_views")


def nchw_to_nhwc(x_nchw_f32, out_nhwc):
    N, C, H, W = x_nchw_f32.shape
    check(lib.sw_nchw_to_nhwc(dt(out_nhwc), N, C, H, W, out_nhwc.shape[3], _p(x_nchw_f32), _p(out_nhwc), _stream()),
          "sw_nchw_to_nhwc")
    return out_nhwc


def relu_bwd(ref, grad, out=None):
    """out = ref > 0 ? grad : 0; in place on `grad` when no `out` is given"""
    out = grad if out is None else out
    check(lib.sw_relu_bwd_out(dt(grad), grad.numel(), _p(ref), _p(grad), _p(out), _stream()), "sw_relu_bwd_out")
    return out


def scale_cols(src_f32, colscale, dst, M, N):
    check(lib.sw_scale_cols(dt(dst), M, N, _p(src_f32), src_f32.stride(0), _p(colscale), _p(dst), dst.stride(0), _stream()),
          "sw_scale_cols")
    return dst


def oicr_predict(logits, R, K, refine_k, base_col, round_stride, boxes, reg_weights, scale_clamp, all_scores, all_boxes):
    rw = (ctypes.c_float * 4)(*[float(v) for v in reg_weights])
    check(lib.sw_oicr_predict(R, K, refine_k, _p(logits), logits.stride(0), base_col, round_stride, _p(boxes), rw,
                              float(scale_clamp), _p(all_scores), _p(all_boxes), _stream()), "sw_oicr_predict")


def detect_postprocess(all_scores, all_boxes, img_h, img_w, score_thresh, nms_thresh, topk, out=None):
    """-> (count[1] i32, boxes [topk,4], scores [topk], classes [topk] i32, rows [topk] i32) device tensors.  out: optional
    preallocated (count, boxes, scores, classes, rows) — only the first count[0] rows are written (without `out` the rest is zero)"""
    R, K1 = all_scores.shape
    K = K1 - 1
    dev = all_scores.device
    if out is not None:
        cnt, boxes, scores, classes, rows = out
    else:
        cnt = torch.zeros(1, device=dev, dtype=torch.int32)
        boxes = torch.zeros(topk, 4, device=dev); scores = torch.zeros(topk, device=dev)
        classes = torch.zeros(topk, device=dev, dtype=torch.int32); rows = torch.zeros(topk, device=dev, dtype=torch.int32)
    nbytes = int(lib.sw_detect_workspace_bytes2(R, K, topk))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    check(lib.sw_detect_postprocess2(R, K, _p(all_scores), _p(all_boxes), int(img_h), int(img_w), float(score_thresh),
                                     float(nms_thresh), int(topk), _p(cnt), _p(boxes), _p(scores), _p(classes), _p(rows), _p(ws),
                                     nbytes, _stream()), "sw_detect_postprocess2")
    return cnt, boxes, scores, classes, rows


# ------------------------------------------------------------------------------------------------ Stage-3 detector (csrc/detector.hip)
def preprocess_pad(img_u8_chw, out_hw4, mean, std):
    """u8 (3, h, w) -> out (H, W, 4) = (img - mean) / std, zero padding (bottom / right) and zero 4th channel"""
    _need_gpu(img_u8_chw, out_hw4)
    h, w = img_u8_chw.shape[1:]
    H, W = out_hw4.shape[:2]
    m = (ctypes.c_float * 3)(*[float(v) for v in mean]); s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(lib.sw_preprocess_pad(dt(out_hw4), h, w, H, W, _p(img_u8_chw), m, s, _p(out_hw4), _stream()), "sw_preprocess_pad")
    return out_hw4


def stem_conv7x7(x_nhwc4, w_oihw, scale, bias, out):
    n, H, W, _ = x_nhwc4.shape
    check(lib.sw_stem_conv7x7(dt(x_nhwc4), n, H, W, _p(x_nhwc4), _p(w_oihw), _p(scale), _p(bias), _p(out), _stream()), "sw_stem_conv7x7")
    return out


def maxpool3x3s2(x, out):
    n, H, W, C = x.shape
    check(lib.sw_maxpool3x3s2(dt(x), n, H, W, C, _p(x), _p(out), _stream()), "sw_maxpool3x3s2")
    return out


def subsample2x(x, out):
    n, H, W, C = x.shape
    check(lib.sw_subsample2x(dt(x), n, H, W, C, _p(x), _p(out), _stream()), "sw_subsample2x")
    return out


def scatter2x(g, out):
    """out (n, H, W, C) fully written: g (n, ceil(H/2), ceil(W/2), C) at the even pixels, 0 elsewhere"""
    n, H, W, C = out.shape
    check(lib.sw_scatter2x(dt(out), n, H, W, C, _p(g), _p(out), _stream()), "sw_scatter2x")
    return out


def add_relu(a, b, out, relu=True):
    check(lib.sw_add_relu(dt(a), a.numel(), _p(a), _p(b), _p(out), int(relu), _stream()), "sw_add_relu")
    return out


def upsample2x_add(lateral, top, out):
    n, h, w, C = top.shape
    assert tuple(lateral.shape) == (n, 2 * h, 2 * w, C), (tuple(lateral.shape), tuple(top.shape))
    check(lib.sw_upsample2x_add(dt(top), n, h, w, C, _p(lateral), _p(top), _p(out), _stream()), "sw_upsample2x_add")
    return out


def downsample2x_sum(g, out):
    n, h, w, C = out.shape
    check(lib.sw_downsample2x_sum(dt(g), n, h, w, C, _p(g), _p(out), _stream()), "sw_downsample2x_sum")
    return out


def roi_align_fwd(feat, rois, sel_i32, out, scale, PH=7, PW=7, sampling_ratio=0, n_sel_dev=None):
    """feat (N, H, W, C) of one level; rois (R, 5) f32; sel int32 rows of this level; out (R, C*PH*PW).  n_sel_dev: device int32 [1]
    holding the real length of `sel` (then sel.numel() is only its bound: no host round trip for the level lists)"""
    _need_gpu(feat, rois, sel_i32, out)
    n, H, W, C = feat.shape
    check(lib.sw_roi_align_fwd(dt(feat), H, W, C, PH, PW, float(scale), sampling_ratio, _p(feat), _p(rois), _p(sel_i32),
                               sel_i32.numel(), _p(n_sel_dev), _p(out), out.stride(0), _stream()), "sw_roi_align_fwd")
    return out


def roi_align_bwd(gout, rois, sel_i32, dfeat_f32, scale, PH=7, PW=7, sampling_ratio=0, n_sel_dev=None):
    _need_gpu(gout, rois, sel_i32, dfeat_f32)
    n, H, W, C = dfeat_f32.shape
    check(lib.sw_roi_align_bwd(dt(gout), H, W, C, PH, PW, float(scale), sampling_ratio, _p(gout), gout.stride(0), _p(rois),
                               _p(sel_i32), sel_i32.numel(), _p(n_sel_dev), _p(dfeat_f32), _stream()), "sw_roi_align_bwd")
    return dfeat_f32


def roi_align_bwd_fx(gout, rois, sel_i32, acc_i64, scale, gout_absmax, PH=7, PW=7, sampling_ratio=0, n_sel_dev=None):
    """the deterministic ROIAlign backward (sw_roi_align_bwd_fx): acc_i64 (N, H, W, C) int64, zero-filled, collects the level's
    contributions as 64-bit fixed-point integers scaled by 2^40 / gout_absmax (device scalar from ops.absmax(gout)); fx_to_float
    turns it into the gradient map"""
    _need_gpu(gout, rois, sel_i32, acc_i64, gout_absmax)
    assert acc_i64.dtype == torch.int64 and acc_i64.is_contiguous()
    n, H, W, C = acc_i64.shape
    check(lib.sw_roi_align_bwd_fx(dt(gout), H, W, C, PH, PW, float(scale), sampling_ratio, _p(gout), gout.stride(0), _p(rois),
                                  _p(sel_i32), sel_i32.numel(), _p(n_sel_dev), _p(gout_absmax), _p(acc_i64), _stream()), "sw_roi_align_bwd_fx")


def fx_to_float(acc_i64, absmax, out):
    """out[i] = acc_i64[i] * absmax / 2^40 (sw_fx_to_float); out f32 or bf16, same element count"""
    _need_gpu(acc_i64, absmax, out)
    assert acc_i64.dtype == torch.int64 and out.numel() == acc_i64.numel() and out.is_contiguous()
    check(lib.sw_fx_to_float(dt(out), acc_i64.numel(), _p(acc_i64), _p(absmax), _p(out), _stream()), "sw_fx_to_float")
    return out


def wsddn_scores_bwd(logits, K, g_scores, dlogits):
    """analytic backward of scores = softmax(C, 1) * softmax(D, 0) for ONE image's rows (sw_wsddn_scores_bwd): logits (R, >= 2K) f32,
    g_scores (R, K) f32 -> dlogits[:, :2K]"""
    _need_gpu(logits, g_scores, dlogits)
    R = logits.shape[0]
    ws = torch.empty(int(lib.sw_wsddn_scores_bwd_workspace_floats(R, K)), device=logits.device, dtype=torch.float32)
    check(lib.sw_wsddn_scores_bwd(R, K, _p(logits), logits.stride(0), _p(g_scores), g_scores.stride(0), _p(dlogits), dlogits.stride(0),
                                  _p(ws), _stream()), "sw_wsddn_scores_bwd")
    return dlogits


def scale_col_blocks(src, dst, split, g0, g1):
    """dst[:, :split] = src[:, :split] * g0, dst[:, split:] = src[:, split:] * g1 (device scalars); src / dst (M, N) f32 views of
    buffers with one row pitch, padding columns of dst zeroed"""
    _need_gpu(src, dst, g0, g1)
    M, N = src.shape
    assert src.stride(0) == dst.stride(0) and src.dtype == dst.dtype == torch.float32
    check(lib.sw_scale_col_blocks(M, N, int(split), _p(src), src.stride(0), _p(g0), _p(g1), _p(dst), _stream()), "sw_scale_col_blocks")
    return dst


def convert_flat(src_f32, dtype):
    """a float32 tensor in another compute dtype (same shape), through sw_convert_2d"""
    _need_gpu(src_f32)
    out = torch.empty(src_f32.shape, device=src_f32.device, dtype=dtype)
    c = src_f32.shape[-1]
    r = src_f32.numel() // c
    check(lib.sw_convert_2d(dt(out), r, c, _p(src_f32), c, _p(out), c, _stream()), "sw_convert_2d")
    return out


def rpn_unpack(y, N, A, hw_per_level):
    """y (rows, ld) f32, the RPN head's packed GEMM output -> logits (N, At), deltas (N, At, 4) in anchor order (sw_rpn_unpack)"""
    _need_gpu(y)
    L = len(hw_per_level)
    At = A * int(sum(hw_per_level))
    hw = (ctypes.c_int * L)(*[int(v) for v in hw_per_level])
    logits = torch.empty(N, At, device=y.device); deltas = torch.empty(N, At, 4, device=y.device)
    check(lib.sw_rpn_unpack(N, L, A, hw, _p(y), y.stride(0), _p(logits), _p(deltas), _stream()), "sw_rpn_unpack")
    return logits, deltas


def rpn_unpack_bwd(dlogits, ddeltas, N, A, hw_per_level, rows, ld, device, g_logits=None, g_deltas=None):
    """gradient of rpn_unpack's input: dy (rows, ld) f32 from dlogits (N, At) / ddeltas (N, At, 4), each times a device scalar"""
    L = len(hw_per_level)
    hw = (ctypes.c_int * L)(*[int(v) for v in hw_per_level])
    dy = torch.empty(rows, ld, device=device)
    check(lib.sw_rpn_unpack_bwd(N, L, A, hw, _p(dlogits), _p(ddeltas), _p(g_logits), _p(g_deltas), _p(dy), ld, _stream()),
          "sw_rpn_unpack_bwd")
    return dy


def roi_assign_levels(boxes_base, row_cnt, box_off_floats):
    """dense ROI rows + FPN levels (sw_roi_assign_levels).  boxes_base: a f32 tensor the images' box blocks live in; image i contributes
    row_cnt[i] boxes starting box_off_floats[i] floats into it.  -> rois (R, 5), level_of (R,), sel (4, R) int32, sel_cnt (4,) int32"""
    _need_gpu(boxes_base)
    n = len(row_cnt)
    R = int(sum(row_cnt))
    if n > 64 or R > (1 << 20):
        raise ValueError(f"sw_roi_assign_levels takes at most 64 images and 2^20 boxes per call (got {n} images, {R} boxes): its per-level "
                         "row lists run over all images of the call; pool larger batches in several calls")
    dev = boxes_base.device
    rois = torch.empty(R, 5, device=dev); lv = torch.empty(R, device=dev, dtype=torch.int32)
    sel = torch.empty(4, max(R, 1), device=dev, dtype=torch.int32); cnt = torch.empty(4, device=dev, dtype=torch.int32)
    rc = (ctypes.c_int * n)(*[int(v) for v in row_cnt]); bo = (ctypes.c_long * n)(*[int(v) for v in box_off_floats])
    check(lib.sw_roi_assign_levels(n, rc, bo, _p(boxes_base), _p(rois), _p(lv), _p(sel), _p(cnt), _stream()), "sw_roi_assign_levels")
    return rois, lv, sel, cnt


def decode_boxes(deltas, boxes, weights, scale_clamp, out):
    """deltas (n, >=4) f32 rows; boxes (n_boxes, 4), row i uses boxes[i % n_boxes]; out (n, 4)"""
    _need_gpu(deltas, boxes, out)
    rw = (ctypes.c_float * 4)(*[float(v) for v in weights])
    check(lib.sw_decode_boxes(out.shape[0], boxes.shape[0], _p(deltas), deltas.stride(0), _p(boxes), rw, float(scale_clamp), _p(out),
                              _stream()), "sw_decode_boxes")
    return out


def rpn_select_pack(logits, deltas, anchors, pre_topk, weights, scale_clamp, img_hw_dev, ints_out=None):
    """RPN proposal selection up to the NMS (sw_rpn_select_pack).  anchors: per level (n_l, 4); img_hw_dev (N, 2) int32 device;
    logits / deltas: either per-level lists ((N, n_l) / (N, n_l, 4) f32, each dense) or ONE pair (N, At) / (N, At, 4) in anchor order
    (ops.rpn_unpack) whose column ranges are the levels.  -> cand_scores (N, L * pre_topk, L + 1), cand_boxes (N, L * pre_topk, 4 L) in
    sw_detect_postprocess2's form (class = level), finite (N,) int32 (a slice of ints_out when given)"""
    L = len(anchors)
    dev = anchors[0].device
    n_list = [int(a.shape[0]) for a in anchors]
    if isinstance(logits, torch.Tensor):
        N, At = logits.shape
        assert At == sum(n_list) and logits.is_contiguous() and deltas.is_contiguous() and logits.dtype == deltas.dtype == torch.float32
        _need_gpu(logits, deltas)
        off, lp, dp = 0, [], []
        for n in n_list:
            lp.append(logits.data_ptr() + 4 * off); dp.append(deltas.data_ptr() + 16 * off); off += n
        stride = At
    else:
        N = logits[0].shape[0]
        for t in list(logits) + list(deltas):
            _need_gpu(t)
            assert t.dtype == torch.float32 and t.is_contiguous()
        lp, dp, stride = [t.data_ptr() for t in logits], [t.data_ptr() for t in deltas], 0
    for a in anchors:
        assert a.dtype == torch.float32 and a.is_contiguous()
    n_l = (ctypes.c_int * L)(*n_list)
    arr = lambda ps: (ctypes.c_void_p * L)(*ps)
    nbytes = int(lib.sw_rpn_select_workspace_bytes(N, L, n_l))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    rows = L * pre_topk
    sc = torch.empty(N, rows, L + 1, device=dev); bx = torch.empty(N, rows, 4 * L, device=dev)
    fin = ints_out if ints_out is not None else torch.empty(N, device=dev, dtype=torch.int32)
    sel = torch.empty(N * L, pre_topk, device=dev, dtype=torch.int32)
    rw = (ctypes.c_float * 4)(*[float(v) for v in weights])
    check(lib.sw_rpn_select_pack(N, L, arr(lp), arr(dp), arr([a.data_ptr() for a in anchors]), n_l, stride, int(pre_topk), rw,
                                 float(scale_clamp), _p(img_hw_dev), _p(sc), _p(bx), _p(fin), _p(sel), _p(ws), nbytes, _stream()),
          "sw_rpn_select_pack")
    return sc, bx, fin


def rpn_label_anchors(anchors, gt_boxes, gt_counts, seeds, batch_size, max_pos, thr_lo=0.3, thr_hi=0.7):
    """RPN anchor labels + matched boxes (sw_rpn_label_anchors).  anchors (A, 4); gt_boxes (sum G_i, 4) the images' boxes back to back;
    gt_counts: host ints per image; seeds: host ints, (positives, negatives) per image.  -> labels int8 (N, A), matched (N, A, 4)"""
    _need_gpu(anchors)
    N, A = len(gt_counts), anchors.shape[0]
    dev = anchors.device
    tot = int(sum(gt_counts))
    assert anchors.dtype == torch.float32 and anchors.is_contiguous() and len(seeds) == 2 * N
    assert tot == 0 or (gt_boxes.is_cuda and gt_boxes.dtype == torch.float32 and gt_boxes.is_contiguous() and gt_boxes.shape[0] == tot)
    nbytes = int(lib.sw_rpn_label_workspace_bytes(N, A, tot))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    labels = torch.empty(N, A, device=dev, dtype=torch.int8); matched = torch.empty(N, A, 4, device=dev)
    cnt = (ctypes.c_int * N)(*[int(c) for c in gt_counts])
    sd = (ctypes.c_uint64 * (2 * N))(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds])
    check(lib.sw_rpn_label_anchors(N, A, _p(anchors), _p(gt_boxes) if tot else None, cnt, float(thr_lo), float(thr_hi), int(batch_size),
                                   int(max_pos), sd, _p(labels), _p(matched), _p(ws), nbytes, _stream()), "sw_rpn_label_anchors")
    return labels, matched


def roi_label_sample(p_cnt_dev, proposals, gt_boxes, gt_classes_i32, gt_counts, seeds, append_gt, iou_thresh, num_classes, batch_size, max_pos):
    """ROI-head matching + sampling (sw_roi_label_sample).  proposals (N, p_stride, 4) with device counts p_cnt_dev (N,) int32; gt_boxes
    (sum G_i, 4) / gt_classes_i32 back to back, gt_counts host ints.  -> count (N,) i32, index / classes (N, batch) i32, both (2, N, batch, 4)
    = (sampled boxes, their matched gt boxes): image i's sampled rows are the first count[i] (foreground list, then background, each in
    random-key order)"""
    _need_gpu(p_cnt_dev, proposals)
    N, ps = proposals.shape[0], proposals.shape[1]
    dev = proposals.device
    tot = int(sum(gt_counts))
    assert proposals.dtype == torch.float32 and proposals.is_contiguous() and p_cnt_dev.dtype == torch.int32 and len(seeds) == 2 * N
    offs, r = [], 0
    for c in gt_counts:
        offs.append(r); r += int(c)
    g_off = (ctypes.c_int * N)(*offs); g_cnt = (ctypes.c_int * N)(*[int(c) for c in gt_counts])
    sd = (ctypes.c_uint64 * (2 * N))(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds])
    cnt = torch.empty(N, device=dev, dtype=torch.int32)
    idx = torch.empty(N, batch_size, device=dev, dtype=torch.int32); cls = torch.empty(N, batch_size, device=dev, dtype=torch.int32)
    both = torch.empty(2, N, batch_size, 4, device=dev)                 # [0] the sampled boxes, [1] their matched gt boxes
    check(lib.sw_roi_label_sample(N, _p(p_cnt_dev), ps, _p(proposals), g_off, g_cnt, _p(gt_boxes) if tot else None,
                                  _p(gt_classes_i32) if tot else None, int(bool(append_gt)), float(iou_thresh), int(num_classes),
                                  int(batch_size), int(max_pos), sd, batch_size, _p(cnt), _p(idx), _p(cls), _p(both[0]), _p(both[1]),
                                  _stream()), "sw_roi_label_sample")
    return cnt, idx, cls, both


def rpn_loss(logits, deltas, labels_i8, anchors, matched_gt, weights, inv_norm, losses2, dlogits=None, ddeltas=None):
    """logits (n,), deltas (n, 4), labels int8 (n,), anchors (A, 4) repeating over the images, matched_gt (n, 4) -> losses2 (2,)"""
    _need_gpu(logits, deltas, labels_i8, anchors, matched_gt, losses2)
    rw = (ctypes.c_float * 4)(*[float(v) for v in weights])
    ws = torch.empty(int(lib.sw_rpn_loss_workspace_floats()), device=logits.device, dtype=torch.float32)
    check(lib.sw_rpn_loss(logits.numel(), anchors.shape[0], _p(logits), _p(deltas), _p(labels_i8), _p(anchors), _p(matched_gt), rw,
                          float(inv_norm), _p(losses2), _p(dlogits), _p(ddeltas), _p(ws), _stream()), "sw_rpn_loss")
    return losses2
