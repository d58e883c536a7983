"""GPU parity of every C-ABI kernel against the CPU oracle / an fp64 torch reference.
Integer outputs (ROI argmax, mined indices, labels) must be bit exact; floating point within the
tolerance written next to each check."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import oicr_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def ops():
    import sos_wsod_amd  # noqa: F401
    import sos_wsod_amd.ops as ops
    assert torch.cuda.is_available()
    return ops


def _rand(shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-5}     # inputs are pre-rounded; accumulation is f32 in both


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_modes(ops, dtype, mode):
    M, N, K = 300, 448, 520 if mode == "tn" else 512
    if mode == "tn":
        M = 296                                            # K-strided operands need M, N multiples of 8
    a = _rand((M, K), 1, dtype); b = _rand((K, N), 2, dtype)
    ref = (a.double() @ b.double())
    A = (a.t().contiguous() if mode == "tn" else a).cuda()             # tn: A stored [K][M]
    B = (b.contiguous() if mode in ("nn", "tn") else b.t().contiguous()).cuda()   # nt: B stored [N][K]
    C = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm(A, B, C, M, N, K, a_kstrided=(mode == "tn"), b_kstrided=(mode != "nt"))
    err = (C.cpu().double() - ref).abs().max() / ref.abs().max()
    assert err < TOL[dtype], err


@pytest.mark.parametrize("dtype", DT)
def test_gemm_epilogue_bias_relu_dropout_and_splitk_atomic(ops, dtype):
    M, N, K = 200, 256, 1024
    a = _rand((M, K), 3, dtype); w = _rand((N, K), 4, dtype); bias = _rand((N,), 5)
    keep = (torch.rand(M, N, generator=torch.Generator().manual_seed(6)) > 0.5).to(torch.uint8)
    ref = F.relu(a.double() @ w.double().t() + bias.double()) * keep.double() * 2.0
    C = torch.empty((M, N), device="cuda", dtype=dtype)
    ep = ops.make_epilogue(bias=bias.cuda(), relu=True, drop_mask=keep.cuda(), drop_scale=2.0, out_dtype=dtype)
    ops.gemm(a.cuda(), w.cuda(), C, M, N, K, ep=ep)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-5      # bf16 output rounding
    assert ((C.cpu().double() - ref).abs().max() / ref.abs().max()) < tol
    # relu-backward mask epilogue + split-K atomic accumulation
    refm = _rand((M, N), 7, dtype)
    C2 = torch.zeros((M, N), device="cuda")
    ep2 = ops.make_epilogue(relu_ref=refm.cuda(), ref_scale=2.0, out_dtype=torch.float32, atomic=True)
    ops.gemm(a.cuda(), w.cuda(), C2, M, N, K, ep=ep2, splitk=4)
    ref2 = (a.double() @ w.double().t()) * (refm.double() > 0) * 2.0
    assert ((C2.cpu().double() - ref2).abs().max() / ref2.abs().max()) < 2e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K,pad,sk", [(200, 256, 1024, 0, 4), (136, 104, 777 * 8, 10, 7), (256, 4096, 8000, 0, 4)])
def test_gemm_splitk_deterministic_slabs(ops, dtype, M, N, K, pad, sk):
    """split-K through per-split slabs + an ordered fold (sw_epilogue.splitk_workspace; ops.gemm attaches it): the result
    overwrites C (no zero fill), equals the contraction, and is bitwise identical run to run (weight-gradient form, TN)"""
    a = _rand((K, M), 31, dtype).cuda(); b = _rand((K, N), 32, dtype).cuda()
    want = a.double().t() @ b.double()
    outs = []
    for _ in range(3):
        C = torch.full((M, N + pad), float("nan"), device="cuda")[:, :N]
        ops.gemm(a, b, C, M, N, K, a_kstrided=True, b_kstrided=True, splitk=sk)
        assert torch.isfinite(C).all()
        assert ((C.double() - want).abs().max() / want.abs().max()) < (2e-5 if dtype == torch.float32 else 2e-5)
        outs.append(C.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("plain", [True, False])
def test_gemm_tail_peel_with_column_local_epilogue(ops, plain):
    """32 x 18 tiles of 256 x 256 = 2.25 rounds of 256 CUs: sw_gemm peels the last two tile columns of the plain f32 form (fc
    weight gradients) into a second launch; the bias / ReLU-mask / bf16-output / absmax form of the same shape runs whole.  Both
    have to equal the contraction, column for column."""
    M, N, K = 8000, 4600, 128
    dt = torch.bfloat16
    a = _rand((M, K), 41, dt).cuda(); w = _rand((N, K), 42, dt).cuda()
    want = a.double() @ w.double().t()
    if plain:
        C = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(a, w, C, M, N, K)
        assert ((C.double() - want).abs().max() / want.abs().max()) < 2e-5
        return
    bias = _rand((N,), 43).cuda(); refm = _rand((M, N), 44, dt).cuda()
    amax = torch.zeros(1, device="cuda")
    C = torch.full((M, N + 8), float("nan"), device="cuda", dtype=dt)[:, :N]
    ep = ops.make_epilogue(bias=bias, relu_ref=refm, ref_scale=1.0, out_dtype=dt, absmax_out=amax)
    ops.gemm(a, w, C, M, N, K, ep=ep)
    ref = (want + bias.double()) * (refm.double() > 0)
    assert torch.isfinite(C.float()).all()
    assert ((C.double() - ref).abs().max() / ref.abs().max()) < 1e-2          # bf16 output rounding
    assert abs(amax.item() - C.float().abs().max().item()) <= 1e-2 * amax.item()
    for c in (0, 4095, 4096, 4599):                                          # either side of the peel boundary
        assert ((C[:, c].double() - ref[:, c]).abs().max() / ref.abs().max()) < 1e-2


# ------------------------------------------------------------------------------------------ conv
def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cin,cout,dil", [(3, 64, 1), (64, 128, 1), (128, 64, 2)])
def test_conv3x3_fwd_dgrad_wgrad(ops, dtype, cin, cout, dil):
    n, H, W = 2, 19, 23
    epc = 8 if dtype == torch.bfloat16 else 4
    cpad = (cin + epc - 1) // epc * epc
    x = _rand((n, cin, H, W), 10, dtype); w = _rand((cout, cin, 3, 3), 11, dtype, 0.1); b = _rand((cout,), 12)
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
    y = F.relu(F.conv2d(xd, wd, b.double(), padding=dil, dilation=dil))
    gy = _rand((n, cout, H, W), 13, dtype)
    (y * gy.double()).sum().backward()
    # forward
    xp = torch.zeros(n, H, W, cpad, dtype=dtype); xp[..., :cin] = _nhwc(x)
    wk = torch.empty(cout, 9, cpad, device="cuda", dtype=dtype)
    ops.conv_weight_prep(w.float().cuda(), wk, 0, cpad)
    out = torch.empty(n, H, W, cout, device="cuda", dtype=dtype)
    ops.conv3x3(xp.cuda(), wk, out, dil, ops.make_epilogue(bias=b.cuda(), relu=True, out_dtype=dtype))
    tol = 1e-2 if dtype == torch.bfloat16 else 3e-5
    ref = _nhwc(y.detach())
    assert ((out.cpu().double() - ref).abs().max() / ref.abs().max()) < tol
    if cin % epc:
        return
    # dz = gy * relu'(y)  (pre-activation gradient), as the product path forms it
    dz = (_nhwc(gy).double() * (ref > 0)).to(dtype)
    # weight gradient (f32, OIHW, atomic)
    dw = torch.zeros(cout, cin, 3, 3, device="cuda")
    ops.conv3x3_wgrad(xp.cuda(), dz.cuda(), dw, dil, splitk=3)
    dz_ref = dz.double().permute(0, 3, 1, 2)
    gw_ref = torch.autograd.grad(F.conv2d(xd, wd, None, padding=dil, dilation=dil), wd, dz_ref)[0]
    assert ((dw.cpu().double() - gw_ref).abs().max() / gw_ref.abs().max()) < 3e-5
    # data gradient = conv with flipped / transposed weights, masked by (x > 0)
    wkd = torch.empty(cin, 9, cout, device="cuda", dtype=dtype)
    ops.conv_weight_prep(w.float().cuda(), wkd, 1)
    dx = torch.empty(n, H, W, cin, device="cuda", dtype=torch.float32)
    ops.conv3x3(dz.cuda(), wkd, dx, dil, ops.make_epilogue(relu_ref=xp.cuda().view(n * H * W, cin), out_dtype=torch.float32))
    gx_ref = torch.autograd.grad(F.conv2d(xd, wd, None, padding=dil, dilation=dil), xd, dz_ref)[0]
    gx_ref = _nhwc(gx_ref) * (_nhwc(x).double() > 0)
    assert ((dx.cpu().double() - gx_ref).abs().max() / gx_ref.abs().max()) < 3e-5


@pytest.mark.parametrize("n,H,W,cin,cout,dil", [(2, 19, 40, 64, 64, 1), (1, 26, 77, 64, 128, 2), (2, 100, 166, 128, 64, 2), (2, 106, 141, 64, 128, 1),
                                                  (1, 9, 16, 128, 64, 1), (1, 70, 200, 64, 72, 1)])
def test_conv3x3_direct_ragged_edge_forms(ops, n, H, W, cin, cout, dil):
    """the direct kernel's edge forms (conv_direct.hip, round 6): tiles whose right 16 columns lie outside the map skip those sub-tiles,
    waves whose rows lie below it only stage — forward (bias + ReLU) and the data-gradient form (ReLU mask from a reference map) against a
    float64 convolution of the same bf16 operands; few-tile shapes take the two-K-group form, the others the four-wave form"""
    dt = torch.bfloat16
    g = torch.Generator(device="cuda"); g.manual_seed(H * 1000 + W + dil)
    x = (torch.randn(n, H, W, cin, device="cuda", generator=g) * 0.7).to(dt)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05).to(dt)
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    wk = torch.empty(cout, 9, cin, device="cuda", dtype=dt)
    ops.conv_weight_prep(w.float(), wk, 0, cin)
    out = torch.full((n, H, W, cout), float("nan"), device="cuda", dtype=dt)
    ops.conv3x3(x, wk, out, dil, ops.make_epilogue(bias=b, relu=True, out_dtype=dt))
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=dil, dilation=dil)
    ref = F.relu(y).permute(0, 2, 3, 1)
    assert torch.isfinite(out.float()).all()
    assert ((out.double() - ref).abs().max() / ref.abs().max()) < 1e-2                     # bf16 output rounding
    mask = (torch.randn(n, H, W, cout, device="cuda", generator=g) * 0.5).to(dt)
    out2 = torch.full((n, H, W, cout), float("nan"), device="cuda", dtype=dt)
    ops.conv3x3(x, wk, out2, dil, ops.make_epilogue(relu_ref=mask.view(n * H * W, cout), out_dtype=dt))
    ref2 = (y - b.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1) * (mask.double() > 0)
    assert torch.isfinite(out2.float()).all()
    assert ((out2.double() - ref2).abs().max() / ref2.abs().max()) < 1e-2


@pytest.mark.parametrize("n,H,W,cin,cout,dil", [(2, 19, 40, 64, 64, 1), (2, 63, 63, 128, 128, 2), (1, 70, 100, 64, 192, 1), (1, 9, 16, 128, 64, 1)])
def test_conv3x3_bf16x3_convolution_is_f32_accurate(ops, n, H, W, cin, cout, dil):
    """the fp32 mode's bf16x3 convolution (backbone_vgg.x3_layer): the f32 input and weights as three bf16 pieces each, K-concatenated
    along the channels (sw_split_bf16x3 sides 0 / 1), ONE bf16 convolution over 6 Cin with f32 output — the direct kernel's f32 epilogue
    (conv_direct.hip out_f32; few-tile shapes: the two-K-group form) — forward (bias + ReLU) and data-gradient form (f32 ReLU reference)
    against a float64 convolution of the f32 operands: f32-class error (a plain bf16 convolution of these operands is ~3e-3)"""
    g = torch.Generator(device="cuda"); g.manual_seed(H * 1000 + W + dil)
    x = torch.randn(n, H, W, cin, device="cuda", generator=g) * torch.exp(torch.randn(n, H, W, cin, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    wk = torch.empty(cout, 9, cin, device="cuda", dtype=torch.float32)
    ops.conv_weight_prep(w, wk, 0, cin)
    wk3 = ops.split_bf16x3(wk.view(cout * 9, cin), 1, out=torch.empty(cout * 9, 6 * cin, device="cuda", dtype=torch.bfloat16)).view(cout, 9, 6 * cin)
    x3 = ops.split_bf16x3(x.view(n * H * W, cin), 0, out=torch.empty(n * H * W, 6 * cin, device="cuda", dtype=torch.bfloat16)).view(n, H, W, 6 * cin)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=dil, dilation=dil)
    ref = F.relu(y).permute(0, 2, 3, 1)
    out = torch.full((n, H, W, cout), float("nan"), device="cuda")
    ops.conv3x3(x3, wk3, out, dil, ops.make_epilogue(bias=b, relu=True, out_dtype=torch.float32))
    assert torch.isfinite(out).all()
    assert ((out.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    plain = torch.empty(n, H, W, cout, device="cuda")
    wkb = wk.to(torch.bfloat16)
    ops.conv3x3(x.to(torch.bfloat16), wkb, plain, dil, ops.make_epilogue(bias=b, relu=True, out_dtype=torch.float32))
    assert ((plain.double() - ref).abs().max() / ref.abs().max()) > 2e-4              # the pieces matter on these operands
    mask = torch.randn(n, H, W, cout, device="cuda", generator=g)
    out2 = torch.full((n, H, W, cout), float("nan"), device="cuda")
    ops.conv3x3(x3, wk3, out2, dil, ops.make_epilogue(relu_ref=mask.view(n * H * W, cout), out_dtype=torch.float32))
    ref2 = (y - b.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1) * (mask.double() > 0)
    assert torch.isfinite(out2).all()
    assert ((out2.double() - ref2).abs().max() / ref2.abs().max()) < 2e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,cols", [(8000, 4096), (130, 72), (64, 64), (37, 200)])
def test_transpose_2d(ops, dtype, rows, cols):
    vec = 8 if dtype == torch.bfloat16 else 4
    pr, pc = (rows + vec - 1) // vec * vec + vec, (cols + vec - 1) // vec * vec + vec          # padded pitches
    src = torch.empty(rows, pc, dtype=dtype, device="cuda")[:, :cols]
    src.copy_(_rand((rows, cols), 5, dtype))
    dst = torch.full((cols, pr), 7.0, dtype=dtype, device="cuda")
    ops.transpose_2d(src, dst[:, :rows], rows, cols)
    assert torch.equal(dst[:, :rows], src.t())
    assert (dst[:, rows:] == 7.0).all()                                     # nothing written past the row


@pytest.mark.parametrize("dtype", DT)
def test_conv3x3_wgrad_grouped_launch(ops, dtype):
    """all weight gradients of a backward pass in ONE launch (sw_conv3x3_wgrad_grouped): problems of different map sizes,
    channel counts, dilations and K-splits (two view batches share a parameter = consecutive slabs, one fold) against an fp64
    convolution gradient; more problems than one kernel-argument block holds (40) exercise the chunked launch"""
    specs = [(2, 19, 23, 64, 128, 1, 2), (2, 25, 17, 64, 128, 1, 3),          # one parameter, two view batches
             (2, 31, 33, 128, 64, 2, 1), (1, 40, 37, 32, 264, 1, 4), (2, 16, 16, 8, 8, 1, 1)]
    specs = specs + [(1, 9, 11, 16, 16, 1, 1)] * 38                             # 43 problems > GROUPED_MAX
    probs, checks = [], []
    seed = 100
    for (n, H, W, cin, cout, dil, ns) in specs:
        x = _rand((n, H, W, cin), seed, dtype); dz = _rand((n, H, W, cout), seed + 1, dtype); seed += 2
        nslab = ops.conv3x3_wgrad_nslab(x.cuda(), cout, ns)
        slabs = torch.full((nslab, cout * 9 * cin), float("nan"), device="cuda")
        probs.append((x.cuda(), dz.cuda(), slabs, dil, ns))
        checks.append((x, dz, slabs, nslab, cin, cout, dil))
    ops.conv3x3_wgrad_grouped(probs)
    torch.cuda.synchronize()

    def ref_grad(x, dz, cin, cout, dil):
        wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x.double().permute(0, 3, 1, 2), wd, None, padding=dil, dilation=dil)
        return torch.autograd.grad(y, wd, dz.double().permute(0, 3, 1, 2))[0]
    for k, (x, dz, slabs, nslab, cin, cout, dil) in enumerate(checks[:6]):
        dw = torch.empty(cout, cin, 3, 3, device="cuda")
        ops.conv3x3_wgrad_fold(slabs, nslab, dw)
        want = ref_grad(x, dz, cin, cout, dil)
        assert ((dw.cpu().double() - want).abs().max() / want.abs().max()) < 3e-5, k
        one = torch.empty(cout, cin, 3, 3, device="cuda")                        # the per-launch path on the same problem
        ops.conv3x3_wgrad(x.cuda(), dz.cuda(), one, dil, splitk=1)
        assert ((dw - one).abs().max() / one.abs().max()) < 1e-5, k
    # two view batches of one parameter: their slabs back to back, ONE fold = the sum of both gradients
    (x0, dz0, s0, n0, cin, cout, dil), (x1, dz1, s1, n1, _, _, _) = checks[0], checks[1]
    both = torch.cat([s0, s1], 0).contiguous()
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    ops.conv3x3_wgrad_fold(both, n0 + n1, dw)
    want = ref_grad(x0, dz0, cin, cout, dil) + ref_grad(x1, dz1, cin, cout, dil)
    assert ((dw.cpu().double() - want).abs().max() / want.abs().max()) < 3e-5
    last = checks[-1]
    dw = torch.empty(last[5], last[4], 3, 3, device="cuda")
    ops.conv3x3_wgrad_fold(last[2], last[3], dw)
    want = ref_grad(last[0], last[1], last[4], last[5], last[6])
    assert ((dw.cpu().double() - want).abs().max() / want.abs().max()) < 3e-5


@pytest.mark.parametrize("spec", [(2, 19, 40, 64, 128, 1, 2), (1, 26, 77, 128, 256, 2, 3), (2, 33, 33, 64, 128, 2, 1), (2, 64, 64, 128, 128, 1, 4),
                                  (2, 10, 330, 64, 128, 1, 7), (2, 12, 70, 64, 128, 2, 4), (1, 8, 32, 64, 128, 1, 1), (2, 63, 63, 128, 256, 2, 5)])
def test_conv3x3_wgrad_direct_kernel(ops, spec):
    """conv_wgrad_direct.hip (round 6: input rows staged once for the nine taps, transposed LDS reads, 128 x 64 x 9 register blocks) behind
    sw_conv3x3_wgrad_grouped: ragged strips (W % 32 != 0), both dilations, split boundaries that get moved off a strip's first / last rows,
    one- and many-split problems — against a float64 convolution gradient of the same bf16 operands, and against the implicit-GEMM path
    (SW_WGRAD_DIRECT is read once per process, so that path is reached through a shape the direct kernel declines: the per-launch entry)"""
    n, H, W, cin, cout, dil, ns = spec
    dt = torch.bfloat16
    g = torch.Generator(device="cuda"); g.manual_seed(H * 977 + W)
    x = (torch.randn(n, H, W, cin, device="cuda", generator=g) * 0.7).to(dt)
    dz = (torch.randn(n, H, W, cout, device="cuda", generator=g) * 0.5).to(dt)
    nslab = ops.conv3x3_wgrad_nslab(x, cout, ns)
    slabs = torch.full((nslab, cout * 9 * cin), float("nan"), device="cuda")
    ops.conv3x3_wgrad_grouped([(x, dz, slabs, dil, ns)])
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    ops.conv3x3_wgrad_fold(slabs, nslab, dw)
    assert torch.isfinite(slabs).all()
    wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), wd, None, padding=dil, dilation=dil)
    want = torch.autograd.grad(y, wd, dz.double().permute(0, 3, 1, 2))[0]
    assert ((dw.double() - want).abs().max() / want.abs().max()) < 3e-5
    one = torch.empty(cout, cin, 3, 3, device="cuda")                        # the implicit-GEMM path on the same problem
    ops.conv3x3_wgrad(x, dz, one, dil, splitk=1)
    assert ((dw - one).abs().max() / one.abs().max()) < 2e-5


def test_conv3x3_wgrad_direct_kernel_many_problems_in_one_call(ops):
    """more problems than one kernel-argument block of the direct weight-gradient kernel holds (32): the entry point launches in chunks;
    mixed dilations, map sizes and split counts in one list, every gradient against float64"""
    dt = torch.bfloat16
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    probs, refs = [], []
    for i in range(37):
        n, H, W = 1 + (i % 2), 16 + (i % 5) * 3, 20 + (i % 7) * 9
        cin, cout, dil, ns = 64 * (1 + i % 2), 64 * (1 + (i // 2) % 2), 1 + (i % 3 == 0), 1 + (i % 2)
        steps, npx = n * ((W + 31) // 32) * H, n * H * W
        eff = -(-npx // (-(-(-(-npx // ns)) // 64) * 64))
        assert -(-steps // eff) >= 8                                   # (every problem qualifies: the list runs on the direct kernel)
        x = (torch.randn(n, H, W, cin, device="cuda", generator=g) * 0.7).to(dt)
        dz = (torch.randn(n, H, W, cout, device="cuda", generator=g) * 0.5).to(dt)
        nslab = ops.conv3x3_wgrad_nslab(x, cout, ns)
        slabs = torch.full((nslab, cout * 9 * cin), float("nan"), device="cuda")
        probs.append((x, dz, slabs, dil, ns)); refs.append((x, dz, slabs, nslab, cin, cout, dil))
    ops.conv3x3_wgrad_grouped(probs)
    for k, (x, dz, slabs, nslab, cin, cout, dil) in enumerate(refs):
        dw = torch.empty(cout, cin, 3, 3, device="cuda")
        ops.conv3x3_wgrad_fold(slabs, nslab, dw)
        wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
        y = F.conv2d(x.double().permute(0, 3, 1, 2), wd, None, padding=dil, dilation=dil)
        want = torch.autograd.grad(y, wd, dz.double().permute(0, 3, 1, 2))[0]
        assert ((dw.double() - want).abs().max() / want.abs().max()) < 3e-5, k


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C", [24, 6])
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("H,W", [(11, 14), (9, 13), (8, 8)])       # odd sizes: the last row / column belongs to no stride-2 window
def test_maxpool_fwd_bwd(ops, dtype, stride, C, H, W):
    n = 2                                                        # C=24: 16-byte channel vectors, C=6: scalar form
    x = F.relu(_rand((n, C, H, W), 20, dtype)).float()          # post-ReLU like the backbone (ties at 0)
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 2, stride)
    gy = _rand(tuple(y.shape), 21, dtype).float()
    y.backward(gy)
    xin = _nhwc(x).to(dtype).cuda()
    out = torch.empty(n, y.shape[2], y.shape[3], C, device="cuda", dtype=dtype)
    ops.maxpool_fwd(xin, out, stride)
    assert torch.equal(out.cpu().float(), _nhwc(y.detach()))
    din = torch.full_like(xin, float("nan"))
    ops.maxpool_bwd(xin, _nhwc(gy).to(dtype).cuda(), din, stride, relu_mask=True)
    ref = _nhwc(xr.grad) * (_nhwc(x) > 0)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-6
    assert (din.cpu().float() - ref).abs().max() <= tol * ref.abs().max()


@pytest.mark.parametrize("n,H,W,Cin,Cout,dil,mode", [(1, 37, 50, 64, 96, 1, 0), (2, 21, 30, 64, 64, 2, 0), (1, 37, 50, 96, 64, 2, 1),
                                                    (2, 63, 63, 256, 128, 2, 0), (1, 64, 64, 128, 256, 1, 1)])
def test_conv3x3_winograd_form_against_float64_and_the_direct_kernel(ops, n, H, W, Cin, Cout, dil, mode):
    """csrc/conv_winograd.hip (F(2x2, 3x3), bf16; an opt-in form: SW_CONV_WINOGRAD=1): forward (bias + ReLU) and data gradient (flipped
    filters, ReLU mask of the producer) on ragged maps, a channel count that is not a multiple of the 64-channel block, dilation 2
    (four parity classes).  Against a float64 convolution of the same bf16 inputs and f32 master filters: relative L2 <= 6e-3 (measured
    3.7e-3 - 4.3e-3; the direct kernel 2.3e-3; one bf16 rounding of the output alone is 1.7e-3), and within 1e-2 of the direct kernel."""
    import math
    g = torch.Generator().manual_seed(H * 7 + Cin + mode)
    x = (torch.randn(n, H, W, Cin, generator=g).relu() * 1.5).to(torch.bfloat16).cuda()
    wm = (torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))).cuda()
    bias = (torch.randn(Cout, generator=g) * 0.1).cuda()
    if mode == 0:
        n_out, n_in, xin = Cout, Cin, x
        ep = ops.make_epilogue(bias=bias, relu=True, out_dtype=torch.bfloat16)
        wk = torch.zeros(Cout, 9, Cin, device="cuda", dtype=torch.bfloat16); ops.conv_weight_prep(wm, wk, 0, Cin)
        ref = F.conv2d(xin.double().cpu().permute(0, 3, 1, 2), wm.double().cpu(), bias.double().cpu(), padding=dil, dilation=dil).relu()
    else:
        n_out, n_in = Cin, Cout
        xin = (torch.randn(n, H, W, Cout, generator=g) * 0.1).to(torch.bfloat16).cuda()
        ep = ops.make_epilogue(relu_ref=x.view(n * H * W, Cin), out_dtype=torch.bfloat16)
        wk = torch.zeros(Cin, 9, Cout, device="cuda", dtype=torch.bfloat16); ops.conv_weight_prep(wm, wk, 1, None)
        wt = wm.double().cpu().flip(2, 3).permute(1, 0, 2, 3)
        ref = F.conv2d(xin.double().cpu().permute(0, 3, 1, 2), wt, None, padding=dil, dilation=dil) * (x.double().cpu().permute(0, 3, 1, 2) > 0)
    ref = ref.permute(0, 2, 3, 1)
    U = torch.empty(16, n_out, n_in, device="cuda", dtype=torch.bfloat16)
    ops.winograd_weight_prep([(wm, U, mode)])
    out_w = torch.full((n, H, W, n_out), 7.0, device="cuda", dtype=torch.bfloat16); out_d = torch.empty_like(out_w)
    assert ops.conv3x3_winograd(xin, U, out_w, dil, ep)
    ops.conv3x3(xin, wk, out_d, dil, ep)
    torch.cuda.synchronize()
    ew = float((out_w.double().cpu() - ref).norm() / ref.norm())
    assert ew <= 6e-3, ew
    assert float((out_w.double() - out_d.double()).norm() / out_d.double().norm()) <= 1e-2
    if mode == 1:                                                 # masked positions are exact zeros in both forms
        assert torch.equal(out_w == 0, out_d == 0) or float(((out_w == 0) != (out_d == 0)).float().mean()) < 1e-3


@pytest.mark.parametrize("a_ks,b_ks", [(False, False), (False, True), (True, True)])
def test_gemm_f32x3_six_product_bf16_form_against_float64(ops, a_ks, b_ks):
    """ops.gemm_f32x3: an f32 GEMM as three bf16 pieces per operand and six products on the bf16 MFMA (sw_split_bf16x3 + one sw_gemm
    over 6 K).  Against a float64 contraction: no worse than 2x the exact-f32 MFMA GEMM's error (both are summation-order noise of an
    f32 accumulator over K = 1536: ~1e-6 of the row scale); operands with a wide dynamic range so that the second and third
    pieces matter (dropping them — a plain bf16 GEMM — is 2e-3 here).  The three operand forms the fc layers use."""
    M, N, K = 320, 448, 1536
    g = torch.Generator().manual_seed(9)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    B = (torch.randn(N, K, generator=g) * 0.05).cuda()
    ref = A.double() @ B.double().t()
    Ain = A.t().contiguous() if a_ks else A
    Bin = B.t().contiguous() if b_ks else B
    C3 = torch.empty(M, N, device="cuda"); C1 = torch.empty(M, N, device="cuda")
    ops.gemm_f32x3(Ain, Bin, C3, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks)
    ops.gemm(Ain, Bin, C1, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks)
    Cb = torch.empty(M, N, device="cuda")
    ops.gemm(Ain.to(torch.bfloat16), Bin.to(torch.bfloat16), Cb, M, N, K, a_kstrided=a_ks, b_kstrided=b_ks)
    torch.cuda.synchronize()
    scale = ref.abs().amax(dim=1, keepdim=True)
    e3 = float(((C3.double() - ref).abs() / scale).max()); e1 = float(((C1.double() - ref).abs() / scale).max())
    eb = float(((Cb.double() - ref).abs() / scale).max())
    print(f"gemm_f32x3 vs float64: max error / row scale {e3:.2e} (exact-f32 MFMA {e1:.2e}, plain bf16 {eb:.2e})")
    assert e3 <= max(2 * e1, 2e-6) and eb > 100 * e3


# ------------------------------------------------------------------------------------------ ROIPool
@pytest.mark.parametrize("adt", [torch.int32, torch.int16])
@pytest.mark.parametrize("dtype", DT)
def test_roi_pool_bit_exact_and_backward(ops, dtype, adt):
    n, C, H, W, R = 2, 96, 31, 40, 300
    feat = _rand((n, C, H, W), 30, dtype).float()
    views, _ = O.make_views(H * 8, W * 8, R, tag="rp")
    boxes = views[0]["boxes"].copy()
    boxes[:5] = [[0, 0, 1e4, 1e4], [-50, -50, -10, -10], [100, 100, 90, 90], [8 * W - 1, 8 * H - 1, 8 * W + 40, 8 * H + 40],
                 [3.9, 3.9, 4.1, 4.1]]                           # clipped / outside / malformed / edge / tiny
    rois = np.concatenate([(np.arange(R) % n)[:, None].astype(np.float32), boxes], 1).astype(np.float32)
    obj = views[0]["obj"]
    ref_out, ref_arg = O.roi_pool_fwd(feat.numpy(), rois, 1.0 / 8)
    ref_scaled = torch.from_numpy(ref_out) * (torch.from_numpy(obj) + 1).view(-1, 1, 1, 1)
    f = _nhwc(feat).to(dtype).cuda()
    pad = 8 if adt == torch.int16 else 0                          # padded row pitch (shared by values and argmax)
    out = torch.full((R, C * 49 + pad), 7.0, device="cuda", dtype=dtype)[:, :C * 49]
    arg = torch.empty(R, C * 49 + pad, device="cuda", dtype=adt)[:, :C * 49]
    ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7, row_scale=torch.from_numpy(obj).cuda(),
                     row_scale_add=1.0)
    assert np.array_equal(ops.argmax_to_int32(arg.contiguous()).cpu().numpy().reshape(ref_arg.shape), ref_arg)   # bit exact bins/argmax
    assert torch.equal(out.cpu().float().reshape(ref_scaled.shape), ref_scaled.to(dtype).float())   # values copied
    if pad:
        assert torch.all(out.as_strided((R, pad), (C * 49 + pad, 1), C * 49) == 7.0)       # the pitch gap is not written
    g = _rand((R, C, 7, 7), 31, dtype).float()
    ref_g = O.roi_pool_bwd((g * (torch.from_numpy(obj) + 1).view(-1, 1, 1, 1)).numpy(), ref_arg, rois, feat.shape)
    ref_g = _nhwc(torch.from_numpy(ref_g)) * (_nhwc(feat) > 0)
    for amax in ("auto", None):                                   # fixed-point and float-atomic accumulation
        dfeat = torch.empty(n, H, W, C, device="cuda", dtype=dtype)
        gg = torch.zeros(R, C * 49 + pad, device="cuda", dtype=dtype)[:, :C * 49]
        gg.copy_(g.to(dtype).view(R, -1))
        ops.roi_pool_bwd(gg, arg, torch.from_numpy(rois).cuda(), dfeat, 7, 7,
                         row_scale=torch.from_numpy(obj).cuda(), row_scale_add=1.0, relu_ref=f, dout_absmax=amax)
        tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
        assert (dfeat.cpu().float() - ref_g).abs().max() <= tol * ref_g.abs().max()


@pytest.mark.parametrize("dtype", DT)
def test_roi_pool_backward_tiny_rois_and_poisoned_gradients(ops, dtype):
    """(1) 2000 ROIs no wider than one feature pixel, all on the SAME pixel with gradients of one sign: every one of the 49 bins
    of every ROI scatters onto one accumulator — the fixed-point sizing must count PH*PW bins per ROI, not 4 (a 4-per-ROI
    bound wraps the 32-bit accumulator here).  (2) a NaN / Inf anywhere in the incoming gradient reaches the feature gradient
    (the integer conversion would turn NaN into 0)."""
    n, C, H, W, R = 2, 16, 20, 24, 2000
    feat = torch.rand(n, C, H, W, generator=torch.Generator().manual_seed(4)).add_(0.5).to(dtype).float()   # the values both sides see
    boxes = np.tile(np.array([[40.0, 48.0, 44.0, 52.0]], np.float32), (R, 1))           # round(x / 8) = 5 .. 6: a 1-2 pixel ROI
    rois = np.concatenate([np.zeros((R, 1), np.float32), boxes], 1)
    ref_out, ref_arg = O.roi_pool_fwd(feat.numpy(), rois, 1.0 / 8)
    f = _nhwc(feat).to(dtype).cuda()
    out = torch.empty(R, C * 49, device="cuda", dtype=dtype); arg = torch.empty(R, C * 49, device="cuda", dtype=torch.int16)
    ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7)
    g = torch.full((R, C, 7, 7), 0.75)                                                   # same sign, near the absmax: worst case
    ref_g = _nhwc(torch.from_numpy(O.roi_pool_bwd(g.numpy(), ref_arg, rois, feat.shape)))
    assert ref_g.max() > 4 * R * 0.75                                                    # more than 4 bins per ROI did hit one pixel
    dfeat = torch.empty(n, H, W, C, device="cuda", dtype=dtype)
    ops.roi_pool_bwd(g.to(dtype).view(R, -1).cuda().contiguous(), arg, torch.from_numpy(rois).cuda(), dfeat, 7, 7)
    assert (dfeat.cpu().float() - ref_g).abs().max() <= (1e-2 if dtype == torch.bfloat16 else 1e-5) * ref_g.abs().max()
    for bad in (float("nan"), float("inf")):
        gb = g.clone(); gb[17, 3, 2, 2] = bad
        ops.roi_pool_bwd(gb.to(dtype).view(R, -1).cuda().contiguous(), arg, torch.from_numpy(rois).cuda(), dfeat, 7, 7)
        assert torch.isnan(dfeat[0]).all()                                               # image 0 received the poisoned ROI


@pytest.mark.parametrize("shape", [(63, 63, 2000), (99, 165, 4000)])
def test_roi_pool_backward_heavy_tailed_gradients_bf16(ops, shape):
    """The benchmarked (bf16) mode's ROI scatter on the real map sizes with the gradient a training step hands it: magnitudes
    spread over many decades (log-normal, sigma = 3) and half of the proposal rows scaled by 1e-6, as ignored / background rows
    are (their weights reach 1e-29, tests/test_gpu_e2e.py).  A fixed-point accumulation sized from the GLOBAL maximum with too
    few bits per term has a dead zone there: round 2's single 32-bit word kept 13 / 12 bits (every term below max/16384 became
    0).  Bars (VERDICT r2 #1b): relative L2 against the oracle's scatter-add (ROILoopPool_cpu.cpp:98-123) <= 2^-8 and relative
    error <= 1e-2 on EVERY pixel-channel above 1e-4 of the largest; the result stays bitwise reproducible."""
    H, W, R = shape
    n, C = 2, 64
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(H * 1000 + R)
    feat = _rand((n, C, H, W), 77, dtype).float()
    views, _ = O.make_views(H * 8 + 8, W * 8 + 8, n * R, tag=f"ht{H}")
    rois = np.concatenate([(np.arange(n * R) % n)[:, None].astype(np.float32), views[0]["boxes"]], 1).astype(np.float32)
    obj = views[0]["obj"]
    _, ref_arg = O.roi_pool_fwd(feat.numpy(), rois, 1.0 / 8)
    f = _nhwc(feat).to(dtype).cuda()
    out = torch.empty(n * R, C * 49, device="cuda", dtype=dtype)
    arg = torch.empty(n * R, C * 49, device="cuda", dtype=torch.int16)
    ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7)
    assert np.array_equal(ops.argmax_to_int32(arg).cpu().numpy().reshape(ref_arg.shape), ref_arg)
    g = torch.exp(3.0 * torch.randn(n * R, C, 7, 7, generator=gen)) * torch.sign(torch.randn(n * R, C, 7, 7, generator=gen))
    row = torch.where(torch.rand(n * R, generator=gen) < 0.5, 1e-6, 1.0)
    g = (g * row.view(-1, 1, 1, 1)).to(dtype).float()                       # the values both sides see
    ref = O.roi_pool_bwd((g * (torch.from_numpy(obj) + 1).view(-1, 1, 1, 1)).numpy(), ref_arg, rois, feat.shape)
    ref = _nhwc(torch.from_numpy(ref)).double()
    got = []
    for sc in (0.0, 0.0, 1.0 / 8):            # with the forward's scale a workgroup lists only the ROIs that reach its pixel range
        dfeat = torch.empty(n, H, W, C, device="cuda", dtype=dtype)
        ops.roi_pool_bwd(g.to(dtype).view(n * R, -1).cuda().contiguous(), arg, torch.from_numpy(rois).cuda(), dfeat, 7, 7,
                         row_scale=torch.from_numpy(obj).cuda(), row_scale_add=1.0, spatial_scale=sc)
        got.append(dfeat.cpu())
    assert torch.equal(got[0].view(torch.int16), got[1].view(torch.int16))   # integer accumulation: bitwise reproducible
    assert torch.equal(got[0].view(torch.int16), got[2].view(torch.int16))   # ... and the ROI filter changes nothing
    d = got[0].double()
    rel_l2 = float((d - ref).norm() / ref.norm())
    assert rel_l2 <= 2.0 ** -8, rel_l2
    big = ref.abs() > 1e-4 * ref.abs().max()
    rel = ((d - ref).abs() / ref.abs().clamp_min(1e-300))[big]
    assert float(rel.max()) <= 1e-2, (float(rel.max()), int(big.sum()))
    # the decades below: nothing above 1e-6 of the largest may vanish (a dead zone shows up as exact zeros)
    small = (ref.abs() > 1e-6 * ref.abs().max()) & ~big
    assert int(((d == 0) & small).sum()) == 0


@pytest.mark.parametrize("dtype", DT)
def test_roi_pool_ties_and_special_values(ops, dtype):
    """first maximum in row-major order on plateaus; -0.0/+0.0, -inf, +-NaN and the most negative finite value follow
    the reference's strict '>' from -FLT_MAX (ROILoopPool_cpu.cpp:60-72)"""
    n, C, H, W, R = 1, 16, 24, 24, 120
    rng = np.random.default_rng(5)
    feat = rng.integers(-2, 3, size=(n, C, H, W)).astype(np.float32)      # many exact ties
    feat[0, 0, :, :] = -np.inf
    feat[0, 1, ::2, :] = np.nan
    feat[0, 2, :, :] = np.where(rng.random((H, W)) < 0.5, -0.0, 0.0)
    feat[0, 3, :, :] = float(torch.finfo(dtype).min)
    feat[0, 4, :, ::3] = -np.nan
    feat[0, 5, :, :] = np.inf
    views, _ = O.make_views(H * 8, W * 8, R, tag="ties")
    rois = np.concatenate([np.zeros((R, 1), np.float32), views[0]["boxes"]], 1).astype(np.float32)
    ref_out, ref_arg = O.roi_pool_fwd(feat, rois, 1.0 / 8)
    f = _nhwc(torch.from_numpy(feat)).to(dtype).cuda()
    for adt in (torch.int32, torch.int16):
        out = torch.empty(R, C * 49, device="cuda", dtype=dtype)
        arg = torch.empty(R, C * 49, device="cuda", dtype=adt)
        ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7)
        got_arg = ops.argmax_to_int32(arg).cpu().numpy().reshape(ref_arg.shape)
        assert np.array_equal(got_arg, ref_arg)
        got = out.cpu().float().view(ref_out.shape).numpy()
        want = torch.from_numpy(ref_out).to(dtype).float().numpy()
        assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("H,W", [(100, 120), (125, 167)])
def test_roi_pool_forward_slab_widths_of_large_maps(ops, dtype, H, W):
    """maps whose whole plane does not fit LDS with 16-byte pixels: bf16 runs the band form (row bands of 8-channel slabs; ROIs
    reaching far outside the image exercise its beyond-the-band rows), f32 the narrower slabs — bins, argmax and values still
    bit-exact vs the C oracle"""
    n, C, R = 2, 16, 120
    feat = _rand((n, C, H, W), 61, dtype).float()
    views, _ = O.make_views(H * 8, W * 8, R, tag="slab")
    boxes = views[0]["boxes"].copy()
    boxes[:6] = [[0, 0, 1e4, 1e4], [-900, -700, 8 * W + 500, 8 * H + 900], [40, -3000, 300, 5000], [-50, -50, -10, -10],
                 [8 * W - 9, 8 * H - 9, 8 * W + 40, 8 * H + 40], [0, 8 * H - 200, 8 * W, 8 * H - 1]]   # bins taller than a band's overlap
    rois = np.concatenate([(np.arange(R) % n)[:, None].astype(np.float32), boxes], 1).astype(np.float32)
    ref_out, ref_arg = O.roi_pool_fwd(feat.numpy(), rois, 1.0 / 8)
    f = _nhwc(feat).to(dtype).cuda()
    out = torch.empty(R, C * 49, device="cuda", dtype=dtype)
    arg = torch.empty(R, C * 49, device="cuda", dtype=ops.roi_argmax_dtype(H, W))
    ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7)
    assert np.array_equal(ops.argmax_to_int32(arg).cpu().numpy().reshape(ref_arg.shape), ref_arg)
    assert torch.equal(out.cpu().float().reshape(ref_out.shape), torch.from_numpy(ref_out).to(dtype).float())


@pytest.mark.parametrize("ws", ["auto", None], ids=["prepared_tasks", "no_workspace"])
@pytest.mark.parametrize("H,W,R", [(63, 63, 1500), (21, 30, 300), (60, 75, 900), (76, 114, 1200), (99, 165, 1600), (150, 200, 700), (9, 200, 200)])
def test_roi_pool_forward_row_sparse_table_form_bit_exact(ops, H, W, R, ws):
    """bf16 forward through the row sparse table kernels (csrc/roipool.hip: roi_pool_fwd_tasks_kernel over the task list
    roi_pool_tasks_kernel prepares — sw_roi_pool_fwd_ws, the hot path — and roi_pool_fwd_sparse_kernel, the entry without workspace;
    whole-map form up to ~4600 pixels, row-band form beyond) against the C oracle on EVERY bin of every ROI of one 8-channel slab group: values, argmax and the objectness
    prior.  The ROI set holds what the level rule has to survive: ROIs sticking out of the map on every side (windows the right edge
    clips below the ROI's span read one span; on the left they take the pixel loop), ROIs far outside, malformed (end < start), one
    pixel wide, the whole map, half-integer edges, widths right at a power of two, and (band form) ROIs reaching far below their
    band.  Maps with plateaus (quantised values): ties must go to the first pixel in row-major order through the span maxima."""
    n, C = 2, 16
    rng = np.random.RandomState(H * 1000 + W)
    feat = (np.round(rng.randn(n, C, H, W) * 2) / 2).astype(np.float32)          # many exact ties, negatives, zeros
    feat[0, 3] = -np.inf; feat[1, 5, ::2] = np.nan; feat[0, 7] = -0.0
    IH, IW = H * 8 + 8, W * 8 + 8                                                    # the image the map came from (map = H/8 - 1)
    x1 = rng.rand(R) * (IW - 32); y1 = rng.rand(R) * (IH - 32)
    bw = 16 + rng.rand(R) * (IW - x1 - 16); bh = 16 + rng.rand(R) * (IH - y1 - 16)
    small = rng.rand(R) < 0.35                                                       # VOC-like: many small boxes
    bw[small] = 16 + rng.rand(small.sum()) * 120; bh[small] = 16 + rng.rand(small.sum()) * 120
    boxes = np.stack([x1, y1, np.minimum(x1 + bw, IW), np.minimum(y1 + bh, IH)], 1).astype(np.float32)
    k = R // 10
    boxes[:k, 0] -= rng.rand(k) * 200; boxes[:k, 1] -= rng.rand(k) * 200            # out on the left / top
    boxes[k:2 * k, 2] += rng.rand(k) * 300; boxes[k:2 * k, 3] += rng.rand(k) * 300  # out on the right / bottom
    pow2 = np.array([[8.0 * a, 8.0 * b, 8.0 * a + 8 * 7 * w_ - 8, 8.0 * b + 8 * 7 * w_ - 8] for a, b, w_ in
                     [(1, 1, 1), (2, 3, 2), (0, 0, 4), (5, 1, 8), (W - 9, 2, 1), (W - 17, H - 17, 2), (W - 30, 1, 4)] if a >= 0 and b >= 0], np.float32)
    special = np.array([[0, 0, 1e4, 1e4], [-500, -500, -100, -100], [100, 100, 90, 90], [IW - 9, IH - 9, IW + 40, IH + 40], [3.9, 3.9, 4.1, 4.1],
                        [-40, 10, 8 * W + 50, 30], [20, -30, 40, 8 * H + 90], [4, 4, 12, 8 * H], [-1e4, -1e4, 1e4, 1e4], [60, 60, 67.9, 68.1]], np.float32)
    boxes[2 * k:2 * k + len(special)] = special
    boxes[2 * k + len(special):2 * k + len(special) + len(pow2)] = pow2
    rois = np.concatenate([(np.arange(R) % n)[:, None].astype(np.float32), boxes], 1).astype(np.float32)
    obj = rng.rand(R).astype(np.float32)
    ref_out, ref_arg = O.roi_pool_fwd(feat, rois, 1.0 / 8)
    ref_scaled = torch.from_numpy(ref_out) * (torch.from_numpy(obj) + 1).view(-1, 1, 1, 1)
    f = _nhwc(torch.from_numpy(feat)).to(torch.bfloat16).cuda()
    assert torch.equal(f.float().cpu().nan_to_num(7.0), _nhwc(torch.from_numpy(feat)).nan_to_num(7.0))       # the fixture is bf16-exact
    for adt in (torch.int16, torch.int32):
        out = torch.full((R, C * 49 + 8), 7.0, device="cuda", dtype=torch.bfloat16)[:, :C * 49]
        arg = torch.empty(R, C * 49 + 8, device="cuda", dtype=adt)[:, :C * 49]
        ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7, row_scale=torch.from_numpy(obj).cuda(), row_scale_add=1.0,
                         workspace=ws)
        got_arg = ops.argmax_to_int32(arg.contiguous()).cpu().numpy().reshape(ref_arg.shape)
        bad = np.argwhere(got_arg != ref_arg)
        assert len(bad) == 0, (adt, len(bad), bad[:4].tolist(), [rois[b[0]].tolist() for b in bad[:2]])
        got = out.cpu().float().reshape(ref_scaled.shape)
        want = ref_scaled.to(torch.bfloat16).float()
        assert torch.equal(got.nan_to_num(123.0), want.nan_to_num(123.0))
        assert torch.all(out.as_strided((R, 8), (C * 49 + 8, 1), C * 49) == 7.0)


@pytest.mark.parametrize("ws", ["auto", None], ids=["prepared_tasks", "no_workspace"])
def test_roi_pool_forward_row_sparse_table_form_other_pooled_sizes(ops, ws):
    """the sparse-table kernels with a pooled size other than the hot path's compile-time 7 x 7 (runtime PH x PW form): 6 x 5 and 3 x 8"""
    n, C, H, W, R = 2, 8, 40, 52, 500
    rng = np.random.RandomState(12)
    feat = (np.round(rng.randn(n, C, H, W) * 2) / 2).astype(np.float32)
    x1 = rng.rand(R) * (W * 8 - 32); y1 = rng.rand(R) * (H * 8 - 32)
    boxes = np.stack([x1 - 20, y1 - 20, x1 + 16 + rng.rand(R) * 300, y1 + 16 + rng.rand(R) * 300], 1).astype(np.float32)
    rois = np.concatenate([(np.arange(R) % n)[:, None].astype(np.float32), boxes], 1).astype(np.float32)
    f = _nhwc(torch.from_numpy(feat)).to(torch.bfloat16).cuda()
    for ph, pw in ((6, 5), (3, 8)):
        ref_out, ref_arg = O.roi_pool_fwd(feat, rois, 1.0 / 8, ph, pw)
        out = torch.empty(R, C * ph * pw, device="cuda", dtype=torch.bfloat16); arg = torch.empty(R, C * ph * pw, device="cuda", dtype=torch.int32)
        ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, ph, pw, workspace=ws)
        assert np.array_equal(arg.cpu().numpy().reshape(ref_arg.shape), ref_arg), (ph, pw)
        assert torch.equal(out.cpu().float().reshape(ref_out.shape), torch.from_numpy(ref_out))


def test_roi_pool_large_map_uses_gather_form(ops):
    """a map whose H*W plane does not fit LDS falls back to the gather kernel; > 65534 pixels needs int32 indices"""
    n, C, H, W, R = 1, 8, 260, 256, 64
    feat = _rand((n, C, H, W), 33, torch.float32)
    views, _ = O.make_views(H * 8, W * 8, R, tag="big")
    rois = np.concatenate([np.zeros((R, 1), np.float32), views[0]["boxes"]], 1).astype(np.float32)
    ref_out, ref_arg = O.roi_pool_fwd(feat.numpy(), rois, 1.0 / 8)
    f = _nhwc(feat).cuda()
    out = torch.empty(R, C * 49, device="cuda"); arg = torch.empty(R, C * 49, device="cuda", dtype=torch.int32)
    ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7)
    assert np.array_equal(arg.cpu().numpy().reshape(ref_arg.shape), ref_arg)
    assert np.array_equal(out.cpu().numpy().reshape(ref_out.shape), ref_out)
    assert ops.roi_argmax_dtype(H, W) == torch.int32
    with pytest.raises(Exception):
        ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, torch.empty(R, C * 49, device="cuda", dtype=torch.int16),
                         1.0 / 8, 7, 7)


@pytest.mark.parametrize("dtype", DT)
def test_roi_pool_backward_on_a_map_larger_than_lds(ops, dtype):
    """130x130 map: the fixed-point backward owns the plane in 5 pixel ranges (one workgroup each per channel slab)"""
    n, C, H, W, R = 2, 8, 130, 130, 96
    feat = _rand((n, C, H, W), 34, dtype).float()
    views, _ = O.make_views(H * 8, W * 8, R, tag="bigbwd")
    rois = np.concatenate([(np.arange(R) % n)[:, None].astype(np.float32), views[0]["boxes"]], 1).astype(np.float32)
    ref_out, ref_arg = O.roi_pool_fwd(feat.numpy(), rois, 1.0 / 8)
    f = _nhwc(feat).to(dtype).cuda()
    out = torch.empty(R, C * 49, device="cuda", dtype=dtype); arg = torch.empty(R, C * 49, device="cuda", dtype=torch.int16)
    ops.roi_pool_fwd(f, torch.from_numpy(rois).cuda(), out, arg, 1.0 / 8, 7, 7)
    assert np.array_equal(ops.argmax_to_int32(arg).cpu().numpy().reshape(ref_arg.shape), ref_arg)
    g = _rand((R, C, 7, 7), 35, dtype).float()
    ref_g = _nhwc(torch.from_numpy(O.roi_pool_bwd(g.numpy(), ref_arg, rois, feat.shape)))
    dfeat = torch.empty(n, H, W, C, device="cuda", dtype=dtype)
    ops.roi_pool_bwd(g.to(dtype).view(R, -1).cuda(), arg, torch.from_numpy(rois).cuda(), dfeat, 7, 7)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert (dfeat.cpu().float() - ref_g).abs().max() <= tol * ref_g.abs().max()


# ------------------------------------------------------------------------------------------ heads
@pytest.mark.parametrize("R,K", [(37, 20), (2000, 20), (700, 80)])
def test_wsddn_mil_loss_and_grad(ops, R, K):
    V, LD = 4, 448
    lg = _rand((V * R, LD), 40, scale=3.0)
    gt = torch.zeros(1, K); gt[0, [1, K // 2]] = 1
    gs = torch.tensor([0.7])
    x = lg.clone().requires_grad_(True)
    losses, sc = [], []
    for v in range(V):
        blk = x[v * R:(v + 1) * R]
        s = F.softmax(blk[:, 3:3 + K], 1) * F.softmax(blk[:, 200:200 + K], 0)
        sc.append(s); losses.append(O.wsddn_loss(s, gt))
    (sum(losses) / V * gs[0]).backward()
    scores = torch.empty(V, R, K, device="cuda"); lv = torch.empty(V, device="cuda")
    dl = torch.zeros(V * R, LD, device="cuda")
    ops.wsddn_mil(lg.cuda(), V, R, K, 3, 200, gt.cuda().view(-1), scores, lv, dl, gs.cuda())
    np.testing.assert_allclose(lv.cpu().numpy(), torch.stack(losses).detach().numpy(), rtol=2e-5)
    np.testing.assert_allclose(scores.cpu().numpy(), torch.stack(sc).detach().numpy(), rtol=2e-5, atol=1e-10)
    ref = x.grad
    assert (dl.cpu() - ref).abs().max() <= 2e-5 * ref.abs().max()


@pytest.mark.parametrize("case", ["a", "b"])
def test_mining_and_labels_bit_exact_vs_golden(ops, case, golden_dir):
    g = np.load(os.path.join(golden_dir, f"mining_{case}.npz"))
    R, K = int(g["R"]), int(g["K"])
    views, _ = O.make_views(256, 320, R, n_gt=len(g["gt"]), K=K, tag=str(g["boxes_tag"]))
    boxes = torch.from_numpy(views[0]["boxes"]).cuda()
    gt = torch.from_numpy(g["gt"].astype(np.int32)).cuda()
    G = gt.numel(); top_k = max(int(R * 0.10), 1)
    for variant in ("wsddn", "refine"):
        sc = torch.from_numpy(g[f"{variant}/scores"]).cuda()
        lab_c = torch.empty(R, dtype=torch.int32, device="cuda"); lab_w = torch.empty(R, device="cuda")
        lab_i = torch.empty(R, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        pi = torch.empty(top_k * G, dtype=torch.int32, device="cuda"); pc = torch.empty_like(pi)
        ps = torch.empty(top_k * G, device="cuda")
        ws = torch.empty(ops.mine_workspace_bytes(R, top_k, G), dtype=torch.uint8, device="cuda")
        ops.oicr_mine_label(sc, gt, boxes, K, top_k, 0.05, 0.01, 0.5, 0.6, lab_c, lab_w, lab_i, cnt, pi, pc, ps, ws)
        n = int(cnt.item())
        assert np.array_equal(pi[:n].cpu().numpy(), g[f"{variant}/pgt_index"])
        assert np.array_equal(pc[:n].cpu().numpy(), g[f"{variant}/pgt_classes"])
        assert np.array_equal(ps[:n].cpu().numpy(), g[f"{variant}/pgt_scores"])
        assert np.array_equal(lab_c.cpu().numpy(), g[f"{variant}/gt_classes"])
        assert np.array_equal(lab_i.cpu().numpy(), g[f"{variant}/gt_index"])
        assert np.array_equal(lab_w.cpu().numpy(), g[f"{variant}/gt_weights"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_mining_rounds_side_by_side(ops, golden_dir, case):
    """n_rounds problems in one launch (one workgroup each) give exactly the single-round results"""
    g = np.load(os.path.join(golden_dir, f"mining_{case}.npz"))
    R, K = int(g["R"]), int(g["K"])
    views, _ = O.make_views(256, 320, R, n_gt=len(g["gt"]), K=K, tag=str(g["boxes_tag"]))
    boxes = torch.from_numpy(views[0]["boxes"]).cuda()
    gt = torch.from_numpy(g["gt"].astype(np.int32)).cuda()
    G = gt.numel(); top_k = max(int(R * 0.10), 1)
    names = ["wsddn", "refine", "refine"]
    sc = torch.zeros(3, R, K + 1)
    sc[0, :, :K] = torch.from_numpy(g["wsddn/scores"]); sc[1] = torch.from_numpy(g["refine/scores"]); sc[2] = sc[1]
    NR = 3
    lab_c = torch.empty(NR, R, dtype=torch.int32, device="cuda"); lab_w = torch.empty(NR, R, device="cuda")
    lab_i = torch.empty(NR, R, dtype=torch.int32, device="cuda"); cnt = torch.zeros(NR, dtype=torch.int32, device="cuda")
    pi = torch.empty(NR, top_k * G, dtype=torch.int32, device="cuda"); pc = torch.empty_like(pi)
    ps = torch.empty(NR, top_k * G, device="cuda")
    ws = torch.empty(ops.mine_workspace_bytes(R, top_k, G, NR), dtype=torch.uint8, device="cuda")
    ops.oicr_mine_label(sc.cuda(), gt, boxes, K, top_k, 0.05, 0.01, 0.5, 0.6, lab_c, lab_w, lab_i, cnt, pi, pc, ps, ws)
    for k, variant in enumerate(names):
        n = int(cnt[k].item())
        assert np.array_equal(pi[k, :n].cpu().numpy(), g[f"{variant}/pgt_index"])
        assert np.array_equal(pc[k, :n].cpu().numpy(), g[f"{variant}/pgt_classes"])
        assert np.array_equal(ps[k, :n].cpu().numpy(), g[f"{variant}/pgt_scores"])
        assert np.array_equal(lab_c[k].cpu().numpy(), g[f"{variant}/gt_classes"])
        assert np.array_equal(lab_i[k].cpu().numpy(), g[f"{variant}/gt_index"])
        assert np.array_equal(lab_w[k].cpu().numpy(), g[f"{variant}/gt_weights"])


def test_refine_loss_rounds_side_by_side(ops):
    """the batched launch (n_rounds heads at a column stride) equals n_rounds single launches bit for bit"""
    V, R, K, NR = 4, 300, 20, 3
    stride = 5 * K + 1
    LD = (8 + NR * stride + 7) // 8 * 8
    lg = _rand((V * R, LD), 52, scale=2.0).cuda()
    views, _ = O.make_views(256, 320, R, tag="rl2")
    boxes = torch.from_numpy(np.stack([v["boxes"] for v in views])).cuda()
    gen = torch.Generator().manual_seed(53)
    lab_c = torch.randint(-1, K + 1, (NR, R), generator=gen).to(torch.int32).cuda()
    lab_i = torch.randint(0, R, (NR, R), generator=gen).to(torch.int32).cuda()
    lab_w = torch.rand(NR, R, generator=gen).cuda()
    gs = torch.rand(2 * NR, generator=gen).cuda() + 0.5
    pvw = torch.tensor([0, 1, 2, 2], dtype=torch.int32).cuda()
    lv = torch.empty(NR, 2, V, device="cuda"); dl = torch.zeros(V * R, LD, device="cuda")
    ops.oicr_refine_loss(lg, V, R, K, 8, 8 + K + 1, boxes, lab_c, lab_w, lab_i, pvw, (10.0, 10.0, 5.0, 5.0), lv, dl, gs,
                         n_rounds=NR, col_stride=stride)
    lv1 = torch.empty(NR, 2, V, device="cuda"); dl1 = torch.zeros(V * R, LD, device="cuda")
    for k in range(NR):
        ops.oicr_refine_loss(lg, V, R, K, 8 + k * stride, 8 + K + 1 + k * stride, boxes, lab_c[k], lab_w[k], lab_i[k], pvw,
                             (10.0, 10.0, 5.0, 5.0), lv1[k], dl1, gs[2 * k:2 * k + 2])
    assert torch.equal(lv, lv1) and torch.equal(dl, dl1)


def test_mining_ties_and_single_proposal(ops):
    """ties -> ascending index (the oracle's documented rule); R=1 edge case"""
    for R in (1, 70):
        K = 20
        sc = np.full((R, K), 0.25, np.float32)
        views, _ = O.make_views(200, 200, R, tag=f"tie{R}")
        gt = np.array([2, 5], np.int64)
        o = O.get_pgt_mist(sc, views[0]["boxes"], gt); l = O.label_proposals(o, views[0]["boxes"], K)
        top_k = max(int(R * 0.1), 1); G = 2
        lab_c = torch.empty(R, dtype=torch.int32, device="cuda"); lab_w = torch.empty(R, device="cuda")
        lab_i = torch.empty(R, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        pi = torch.empty(top_k * G, dtype=torch.int32, device="cuda"); pc = torch.empty_like(pi); ps = torch.empty(top_k * G, device="cuda")
        ws = torch.empty(ops.mine_workspace_bytes(R, top_k, G), dtype=torch.uint8, device="cuda")
        ops.oicr_mine_label(torch.from_numpy(sc).cuda(), torch.from_numpy(gt.astype(np.int32)).cuda(),
                            torch.from_numpy(views[0]["boxes"]).cuda(), K, top_k, 0.05, 0.01, 0.5, 0.6, lab_c, lab_w, lab_i,
                            cnt, pi, pc, ps, ws)
        n = int(cnt.item())
        assert np.array_equal(pi[:n].cpu().numpy(), o["index"]) and np.array_equal(lab_c.cpu().numpy(), l["gt_classes"])
        assert np.array_equal(lab_i.cpu().numpy(), l["gt_index"])


@pytest.mark.parametrize("R,K", [(500, 20), (300, 80)])
def test_refine_loss_and_grad(ops, R, K):
    V, LD = 4, 4 + (K + 1) + 4 * K + 7
    LD = (LD + 7) // 8 * 8
    cls_col, box_col = 4, 4 + K + 1
    lg = _rand((V * R, LD), 50, scale=2.0)
    views, _ = O.make_views(256, 320, R, tag="rl")
    gen = torch.Generator().manual_seed(51)
    lab_class = torch.randint(-1, K + 1, (R,), generator=gen)
    lab_index = torch.randint(0, R, (R,), generator=gen)
    lab_weight = torch.rand(R, generator=gen)
    boxes = np.stack([v["boxes"] for v in views])
    gs = torch.tensor([0.9, 1.3])
    x = lg.clone().requires_grad_(True)
    lc, lb, probs = [], [], []
    for v in range(V):
        pv = 2 if v == 3 else v
        blk = x[pv * R:(pv + 1) * R]
        a, b = O.oicr_losses(blk[:, cls_col:cls_col + K + 1], blk[:, box_col:box_col + 4 * K], boxes[v],
                             boxes[v][lab_index.numpy()], lab_class.numpy(), lab_weight.numpy(), K)
        lc.append(a); lb.append(b)
        probs.append(F.softmax(x[v * R:(v + 1) * R, cls_col:cls_col + K + 1].detach(), -1))
    (sum(lc) / V * gs[0] + sum(lb) / V * gs[1]).backward()
    lv = torch.empty(2, V, device="cuda"); pr = torch.empty(1, R, K + 1, device="cuda")
    dl = torch.full((V * R, LD), 7.0, device="cuda")
    ops.oicr_refine_loss(lg.cuda(), V, R, K, cls_col, box_col, torch.from_numpy(boxes).cuda(),
                         lab_class.to(torch.int32).cuda(), lab_weight.cuda(), lab_index.to(torch.int32).cuda(),
                         torch.tensor([0, 1, 2, 2], dtype=torch.int32).cuda(), (10.0, 10.0, 5.0, 5.0), lv, dl, gs.cuda())
    np.testing.assert_allclose(lv[0].cpu().numpy(), torch.stack(lc).detach().numpy(), rtol=2e-5)
    np.testing.assert_allclose(lv[1].cpu().numpy(), torch.stack(lb).detach().numpy(), rtol=2e-5)
    ops.oicr_mean_probs(lg.cuda(), V, R, K, 1, cls_col, 0, pr)          # next round's mining scores: view-mean softmax
    np.testing.assert_allclose(pr[0].cpu().numpy(), torch.stack(probs).mean(0).numpy(), rtol=2e-5, atol=1e-9)
    ref = x.grad[:, cls_col:box_col + 4 * K]
    got = dl.cpu()[:, cls_col:box_col + 4 * K]
    assert (got - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert torch.all(dl.cpu()[:, :cls_col] == 7.0)             # other columns untouched


@pytest.mark.parametrize("first", [True, False])
@pytest.mark.parametrize("M,N,K", [(4096, 25088, 2048), (512, 25600, 1024)])
def test_gemm_fused_sgd_epilogue_equals_gemm_then_optimizer(ops, M, N, K, first):
    """sw_epilogue.sgd_fused (round 6): the weight-gradient GEMM applies the SGD update in its epilogue — parameter, momentum buffer, the
    row-major and the transposed bf16 copies come out BIT-identical to the GEMM writing the gradient followed by the optimizer's tiled
    kernel (sw_sgd_multi, stage_kind 3), whole rounds through the epilogue and the peeled tail columns through the tiled kernel
    ((4096, 25088): fc6's shape = 6 rounds + 2 peeled tile columns; (512, 25600): 200 tiles, no peel), first step (no momentum yet) or not,
    learning rate / weight decay from a device buffer."""
    dt = torch.bfloat16
    g = torch.Generator(device="cuda"); g.manual_seed(M + N + K + int(first))
    A = (torch.randn(M, K + 64, device="cuda", generator=g) * 0.05).to(dt)[:, :K]          # dZ^T (K-contiguous)
    B = (torch.randn(K, N + 64, device="cuda", generator=g).clamp_(min=0)).to(dt)[:, :N]  # pooled (K-strided)
    w0 = torch.randn(M, N, device="cuda", generator=g) * 0.01
    m0 = torch.randn(M, N, device="cuda", generator=g) * 0.001
    hyper = torch.tensor([1e-3, 5e-4], device="cuda")
    assert ops.gemm_sgd_fused_supported(dt, M, N, K, False, True)

    def state():
        w, mo = w0.clone(), m0.clone()
        st0 = torch.zeros(M, N + 128, device="cuda", dtype=dt)[:, :N]; st1 = torch.zeros(N, M + 128, device="cuda", dtype=dt)[:, :M]
        staging = dict(kind=3, dtype=dt, stage0=st0, stage1=st1, d0=N, d1=0, d2=0, ld0=st0.stride(0), ld1=st1.stride(0))
        return w, mo, st0, st1, staging
    # reference: gradient to memory, then the optimizer's kernel
    w, mo, st0, st1, staging = state()
    dW = torch.empty(M, N, device="cuda")
    ops.gemm(A, B, dW, M, N, K, b_kstrided=True)
    ops.sgd_multi([dict(param=w, grad=dW, buf=mo, lr=123.0, weight_decay=456.0, first=first, staging=staging, hyper=hyper)], 0.9, 1.0)
    # fused
    w2, mo2, st02, st12, staging2 = state()
    dW2 = torch.full((M, N), float("nan"), device="cuda")
    ep = ops.make_epilogue(out_dtype=torch.float32)
    ops.attach_sgd_fused(ep, dict(param=w2, buf=mo2, lr=123.0, weight_decay=456.0, first=first, staging=staging2, hyper=hyper), 0.9, 1.0)
    ops.gemm(A, B, dW2, M, N, K, b_kstrided=True, ep=ep)
    torch.cuda.synchronize()
    assert torch.equal(w2, w) and torch.equal(mo2, mo)
    assert torch.equal(st02.view(torch.int16), st0.view(torch.int16)) and torch.equal(st12.view(torch.int16), st1.view(torch.int16))
    assert not torch.equal(w, w0)
    n_written = int(torch.isfinite(dW2).any(dim=0).sum())                     # only the peeled tail columns carry a gradient
    assert n_written in (0, 512), n_written


def test_gemm_hash_dropout_equals_mask_dropout(ops):
    """dropout decided inside the epilogue (seed, offset, p) == the same GEMM with the keep mask sw_dropout_mask writes"""
    dt = torch.bfloat16
    M, N, K = 300, 264, 128
    A = _rand((M, K), 80, dt).cuda(); B = _rand((N, K), 81, dt).cuda(); bias = _rand((N,), 82).cuda()
    seed, off = 0x1234567, 777
    mask = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    ops.dropout_mask(mask, seed, off, 0.5)
    assert 0.45 < mask.float().mean().item() < 0.55
    C1 = torch.empty(M, N, device="cuda", dtype=dt); C2 = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm(A, B, C1, M, N, K, ep=ops.make_epilogue(bias=bias, relu=True, drop_mask=mask, out_dtype=dt))
    ops.gemm(A, B, C2, M, N, K, ep=ops.make_epilogue(bias=bias, relu=True, drop_hash=(seed, off, 0.5), out_dtype=dt))
    assert torch.equal(C1, C2)


# ------------------------------------------------------------------------------------------ optimizer
@pytest.mark.parametrize("sdt", DT)
def test_sgd_multi_matches_single_tensor_steps_and_staging(ops, sdt):
    """sw_sgd_multi == per-tensor sw_sgd_momentum_step + sw_conv_weight_prep / sw_convert_2d on the updated weights,
    bit for bit (28 tensors -> two launches; odd sizes, unaligned views, first step and later steps)"""
    gen = torch.Generator().manual_seed(70)
    shapes = [(64, 32, 3, 3), (32,), (40, 24, 3, 3), (130, 72), (7,), (3, 1001), (128, 192)] + [(5 + i,) for i in range(21)]
    ps = [torch.randn(*sh, generator=gen).cuda() for sh in shapes]
    flat = torch.randn(11 * 72 + 3, generator=gen).cuda()
    ps.append(flat[3:3 + 11 * 72].view(11, 72))                   # a row-slice style view at a 12-byte offset
    for first in (True, False):
        gs = [torch.randn(p.shape, generator=gen).cuda() for p in ps]
        bufs = [torch.randn(p.shape, generator=gen).cuda() for p in ps]
        ref_p = [p.clone() for p in ps]; ref_b = [b.clone() for b in bufs]
        for i, (p, g, b) in enumerate(zip(ref_p, gs, ref_b)):
            ops.sgd_momentum_step(p, g, b, 0.01 * (1 + i % 3), 0.9, 5e-4 * (i % 2), first, 0.5)
        st = {}
        wk0 = torch.zeros(64, 9, 32, device="cuda", dtype=sdt); wk1 = torch.zeros(32, 9, 64, device="cuda", dtype=sdt)
        st[0] = dict(kind=2, dtype=sdt, stage0=wk0, stage1=wk1, d0=64, d1=32, d2=32, ld0=0)
        wk2 = torch.zeros(40, 9, 24, device="cuda", dtype=sdt)
        st[2] = dict(kind=2, dtype=sdt, stage0=wk2, stage1=None, d0=40, d1=24, d2=24, ld0=0)
        m3 = torch.zeros(130, 72 + 8, device="cuda", dtype=sdt)[:, :72]
        st[3] = dict(kind=1, dtype=sdt, stage0=m3, stage1=None, d0=72, d1=0, d2=0, ld0=m3.stride(0))
        m5 = torch.zeros(3, 1001 + 3, device="cuda", dtype=sdt)[:, :1001]
        st[5] = dict(kind=1, dtype=sdt, stage0=m5, stage1=None, d0=1001, d1=0, d2=0, ld0=m5.stride(0))
        m6 = torch.zeros(128, 192 + 8, device="cuda", dtype=sdt)[:, :192]; m6t = torch.zeros(192, 128 + 8, device="cuda", dtype=sdt)[:, :128]
        st[6] = dict(kind=3, dtype=sdt, stage0=m6, stage1=m6t, d0=192, d1=0, d2=0, ld0=m6.stride(0), ld1=m6t.stride(0))
        mv = torch.zeros(11, 72, device="cuda", dtype=sdt)
        st[len(ps) - 1] = dict(kind=1, dtype=sdt, stage0=mv, stage1=None, d0=72, d1=0, d2=0, ld0=72)
        entries = [dict(param=p, grad=g, buf=b, lr=0.01 * (1 + i % 3), weight_decay=5e-4 * (i % 2), first=first,
                        staging=st.get(i)) for i, (p, g, b) in enumerate(zip(ps, gs, bufs))]
        ops.sgd_multi(entries, 0.9, 0.5)
        for p, rp, b, rb in zip(ps, ref_p, bufs, ref_b):
            assert torch.equal(p, rp) and torch.equal(b, rb)
        e0 = torch.empty_like(wk0); e1 = torch.empty_like(wk1); e2 = torch.empty_like(wk2)
        ops.conv_weight_prep(ps[0], e0, 0, 32); ops.conv_weight_prep(ps[0], e1, 1, None); ops.conv_weight_prep(ps[2], e2, 0, 24)
        assert torch.equal(wk0, e0) and torch.equal(wk1, e1) and torch.equal(wk2, e2)
        assert torch.equal(m3, ps[3].to(sdt)) and torch.equal(m5, ps[5].to(sdt)) and torch.equal(mv, ps[-1].to(sdt))
        assert torch.equal(m6, ps[6].to(sdt)) and torch.equal(m6t, ps[6].to(sdt).t())
        e6 = torch.zeros(192, 128 + 8, device="cuda", dtype=sdt)[:, :128]
        ops.convert_2d_t(ps[6], e6, 128, 192)                      # the eager form of the transposed copy
        assert torch.equal(e6, m6t)


# ------------------------------------------------------------------------------------------ utilities
def test_preprocess_colsum_sgd_dropout(ops):
    img = torch.randint(0, 256, (3, 37, 53), dtype=torch.uint8)
    out = torch.empty(37, 53, 4, device="cuda")
    ops.preprocess(img.cuda(), out, O.PIXEL_MEAN, O.PIXEL_STD)
    ref = O.preprocess(img).permute(1, 2, 0)
    assert torch.equal(out.cpu()[..., :3], ref) and torch.all(out.cpu()[..., 3] == 0)
    X = _rand((1001, 130), 60)
    cs = torch.empty(130, device="cuda")
    ops.colsum(X.cuda(), 1001, 130, cs)
    np.testing.assert_allclose(cs.cpu().numpy(), X.double().sum(0).numpy(), rtol=1e-5, atol=1e-4)
    p = _rand((5000,), 61); g = _rand((5000,), 62); buf = torch.zeros(5000)
    opt_p = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([opt_p], lr=0.01, momentum=0.9, weight_decay=5e-4)
    pc, bc = p.cuda(), buf.cuda()
    for step in range(3):
        opt_p.grad = g.clone(); opt.step()
        ops.sgd_momentum_step(pc, g.cuda(), bc, 0.01, 0.9, 5e-4, first_step=(step == 0))
    np.testing.assert_allclose(pc.cpu().numpy(), opt_p.detach().numpy(), rtol=1e-5, atol=1e-7)
    keep = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    ops.dropout_mask(keep, seed=1234, offset=0, p=0.5)
    frac = keep.float().mean().item()
    assert abs(frac - 0.5) < 5e-3


@pytest.mark.parametrize("R,K", [(600, 20), (300, 80)])
def test_detect_postprocess_vs_oracle(ops, R, K):
    """per-class NMS on class-offset clipped boxes + global top-100 (fast_rcnn_oicr.py:86-148) — order bit exact"""
    g = torch.Generator().manual_seed(70)
    sc = torch.softmax(torch.randn(R, K + 1, generator=g) * 3, 1)
    ctr = torch.rand(R, K, 2, generator=g) * torch.tensor([320.0, 240.0]); wh = torch.rand(R, K, 2, generator=g) * 120 + 4
    bx = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).reshape(R, 4 * K)          # partly outside the image -> clipping
    H, W, thr, nms_thr, topk = 240, 320, 0.02, 0.3, 100
    # oracle: same steps as oracle.oicr_plus_inference's tail
    s = sc[:, :-1].numpy(); pb = bx.numpy().reshape(R, K, 4).copy()
    pb[..., 0::2] = pb[..., 0::2].clip(0, W); pb[..., 1::2] = pb[..., 1::2].clip(0, H)
    r_idx, c_idx = np.nonzero(s > np.float32(thr))
    bsel, ssel = pb[r_idx, c_idx], s[r_idx, c_idx]
    off = c_idx.astype(np.float32) * np.float32(bsel.max() + 1)
    keep = O.nms_keep((bsel + off[:, None]).astype(np.float32), ssel, nms_thr)[:topk]
    cnt, dboxes, dscores, dclasses, drows = ops.detect_postprocess(sc.cuda(), bx.cuda(), H, W, thr, nms_thr, topk)
    n = int(cnt.item())
    assert n == len(keep)
    assert np.array_equal(dclasses[:n].cpu().numpy(), c_idx[keep]) and np.array_equal(drows[:n].cpu().numpy(), r_idx[keep])
    assert np.array_equal(dscores[:n].cpu().numpy(), ssel[keep]) and np.array_equal(dboxes[:n].cpu().numpy(), bsel[keep])


@pytest.mark.parametrize("case", ["rpn", "classes", "one_class_beyond_lds", "ragged"])
def test_detect_postprocess_mask_form_equals_single_workgroup_form(ops, case):
    """sw_detect_postprocess2 (mask form: prep / 64x64 IoU tiles over the chip / one resolving wave per class) against
    sw_detect_postprocess (one workgroup per class, itself pinned by the oracle above and by tools/fuzz_parity.py): every output bit
    for bit — the RPN's use (5 levels as classes, 8741 candidates, best 1000 at IoU 0.7), 20 classes x 2000 proposals at the
    detector's thresholds, ONE class of 5000 candidates (rows beyond the resolver's 4096-row LDS window), classes of 0 / 1 / 65
    candidates."""
    import ctypes
    from sos_wsod_amd._lib import lib
    g = torch.Generator().manual_seed(71)
    if case == "rpn":
        counts, sizes, H, W, thr, nms_thr, topk = [2000, 2000, 2000, 2000, 741], [32, 64, 128, 256, 512], 800, 1216, -3.0e38, 0.7, 1000
    elif case == "classes":
        counts, sizes, H, W, thr, nms_thr, topk = None, None, 375, 500, 1e-5, 0.3, 100
    elif case == "one_class_beyond_lds":
        counts, sizes, H, W, thr, nms_thr, topk = [5000], [40], 600, 900, -3.0e38, 0.5, 2000
    else:
        counts, sizes, H, W, thr, nms_thr, topk = [0, 1, 65, 300, 0, 64], [16, 16, 30, 60, 10, 20], 300, 400, -3.0e38, 0.5, 50
    if counts is not None:
        L = len(counts)
        sc, bx, lv = [], [], []
        for i, (n, s) in enumerate(zip(counts, sizes)):
            cx = torch.rand(n, generator=g) * W; cy = torch.rand(n, generator=g) * H
            w = s * (0.5 + torch.rand(n, generator=g)); h = s * (0.5 + torch.rand(n, generator=g))
            bx.append(torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1))
            v = torch.randn(n, generator=g)
            if n > 10:
                v[n // 2] = v[n // 3]                                                        # a tied score: proposal order decides
            sc.append(v); lv.append(torch.full((n,), i, dtype=torch.int64))
        sc, bx, lv = torch.cat(sc), torch.cat(bx), torch.cat(lv)
        R = sc.numel()
        scores = torch.full((R, L + 1), -float("inf")); scores[torch.arange(R), lv] = sc
        boxes = bx[:, None, :].expand(R, L, 4).reshape(R, 4 * L).contiguous()
        K = L
    else:
        R, K = 2000, 20
        scores = torch.softmax(torch.randn(R, K + 1, generator=g) * 3, 1)
        ctr = torch.rand(R, K, 2, generator=g) * torch.tensor([float(W), float(H)]); wh = torch.rand(R, K, 2, generator=g) * 150 + 4
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).reshape(R, 4 * K)
    scores, boxes = scores.cuda().contiguous(), boxes.cuda().contiguous()
    assert int(lib.sw_detect_workspace_bytes2(R, K, topk)) > int(lib.sw_detect_workspace_bytes(K, topk)) + 1024       # the mask form is on offer
    got = ops.detect_postprocess(scores, boxes, H, W, thr, nms_thr, topk)
    dev = scores.device
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    b = torch.zeros(topk, 4, device=dev); s_ = torch.zeros(topk, device=dev)
    c_ = torch.zeros(topk, device=dev, dtype=torch.int32); r_ = torch.zeros(topk, device=dev, dtype=torch.int32)
    ws = torch.empty(int(lib.sw_detect_workspace_bytes(K, topk)), device=dev, dtype=torch.uint8)
    rc = lib.sw_detect_postprocess(R, K, scores.data_ptr(), boxes.data_ptr(), H, W, thr, nms_thr, topk, cnt.data_ptr(), b.data_ptr(),
                                   s_.data_ptr(), c_.data_ptr(), r_.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    n = int(cnt.item())
    assert n == int(got[0].item()) and n > 0
    for x, y in zip(got[1:], (b, s_, c_, r_)):
        assert torch.equal(x[:n], y[:n])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(7600, 1024, 256), (8000, 4096, 1024), (300, 72, 64)])
def test_gemm_epilogue_residual_add(ops, dtype, M, N, K):
    """sw_epilogue.residual: y = relu(x @ W^T + bias + residual) — the bottleneck's shortcut add inside conv3's GEMM (resnet.py:205-212);
    the 128x128 tile (K < 1024), the ping-pong 256x256 tile and a ragged small shape"""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, K, generator=g).cuda().to(dtype); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(dtype)
    b = torch.randn(N, generator=g).cuda(); res = torch.randn(M, N, generator=g).cuda().to(dtype)
    y = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(x, w, y, M, N, K, ep=ops.make_epilogue(bias=b, relu=True, out_dtype=dtype, residual=res))
    ref = torch.relu(x.double() @ w.double().t() + b.double() + res.double())
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert float((y.double() - ref).abs().max()) <= tol * float(ref.abs().max())
    y2 = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(x, w, y2, M, N, K, ep=ops.make_epilogue(bias=b, relu=False, out_dtype=dtype, residual=res))
    ref2 = x.double() @ w.double().t() + b.double() + res.double()
    assert float((y2.double() - ref2).abs().max()) <= tol * float(ref2.abs().max())


def test_detect_postprocess_mask_form_randomised(ops):
    """12 random shapes (R 256 .. 3000, K 1 .. 24, thresholds, box scales, top-k): sw_detect_postprocess2 == sw_detect_postprocess"""
    from sos_wsod_amd._lib import lib
    rng = np.random.RandomState(5)
    used_mask = 0
    for case in range(12):
        R = int(rng.randint(256, 3000)); K = int(rng.choice([1, 2, 5, 20, 24]))
        H, W = int(rng.randint(100, 900)), int(rng.randint(100, 1300))
        topk = int(rng.choice([10, 100, 600])); topk = min(topk, 16384 // K)
        thr = float(rng.choice([-3.0e38, 1e-5, 0.05])); nms_thr = float(rng.choice([0.3, 0.5, 0.7]))
        g = torch.Generator().manual_seed(1000 + case)
        scores = torch.softmax(torch.randn(R, K + 1, generator=g) * float(rng.choice([1, 4])), 1)
        scale = float(rng.choice([20, 80, 300]))
        ctr = torch.rand(R, K, 2, generator=g) * torch.tensor([float(W), float(H)]); wh = torch.rand(R, K, 2, generator=g) * scale + 2
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).reshape(R, 4 * K)
        scores, boxes = scores.cuda().contiguous(), boxes.cuda().contiguous()
        used_mask += int(lib.sw_detect_workspace_bytes2(R, K, topk)) > int(lib.sw_detect_workspace_bytes(K, topk)) + 1024
        got = ops.detect_postprocess(scores, boxes, H, W, thr, nms_thr, topk)
        dev = scores.device
        cnt = torch.zeros(1, device=dev, dtype=torch.int32)
        b = torch.zeros(topk, 4, device=dev); s_ = torch.zeros(topk, device=dev)
        c_ = torch.zeros(topk, device=dev, dtype=torch.int32); r_ = torch.zeros(topk, device=dev, dtype=torch.int32)
        ws = torch.empty(int(lib.sw_detect_workspace_bytes(K, topk)), device=dev, dtype=torch.uint8)
        assert lib.sw_detect_postprocess(R, K, scores.data_ptr(), boxes.data_ptr(), H, W, thr, nms_thr, topk, cnt.data_ptr(), b.data_ptr(),
                                         s_.data_ptr(), c_.data_ptr(), r_.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        n = int(cnt.item())
        assert n == int(got[0].item()), (case, R, K, n, int(got[0].item()))
        for x, y in zip(got[1:], (b, s_, c_, r_)):
            assert torch.equal(x[:n], y[:n]), (case, R, K)
    assert used_mask == 12


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,pad", [(8192, 512, 0), (8000, 4096, 128), (32768, 256, 0), (1, 64, 0), (33, 264, 8), (1001, 130, 0),
                                     (700, 24, 4), (5000, 20000, 0)])
def test_colsum_workspace_form(ops, dtype, M, N, pad):
    """bias gradients: deterministic partial-row form (16-byte loads, ordered fold) and the atomic fallback for shapes it
    does not cover (N or the row pitch not a multiple of 16 bytes), repeated and on a second stream"""
    X = (torch.randn(M, N + pad, generator=torch.Generator().manual_seed(M + N)) * 2).to(dtype).cuda()[:, :N]
    want = X.double().sum(0).cpu().numpy()
    tol = dict(rtol=2e-5, atol=2e-5 * float(np.sqrt(M)) * 4)
    outs = []
    for rep in range(3):
        cs = torch.full((N,), float("nan"), device="cuda")
        ops.colsum(X, M, N, cs)
        np.testing.assert_allclose(cs.cpu().numpy(), want, **tol)
        outs.append(cs)
    vec = 8 if dtype == torch.bfloat16 else 4
    if N % vec == 0 and (N + pad) % vec == 0:
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])          # deterministic
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    cs_a = torch.empty(N, device="cuda"); cs_b = torch.empty(N, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(4):
            ops.colsum(X, M, N, cs_b)
    for _ in range(4):
        ops.colsum(X, M, N, cs_a)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    np.testing.assert_allclose(cs_a.cpu().numpy(), want, **tol)
    np.testing.assert_allclose(cs_b.cpu().numpy(), want, **tol)


def test_copy_multi_and_relu_bwd_out(ops):
    """sw_copy_multi: aligned / unaligned / odd-sized / empty tensors in one launch; sw_relu_bwd_out leaves its gradient input alone"""
    g = torch.Generator().manual_seed(3)
    srcs = [torch.randint(0, 256, (3, 512, 512), generator=g, dtype=torch.uint8), torch.randn(2000, 4, generator=g),
            torch.randn(2000, generator=g), torch.randn(1037, generator=g), torch.randint(0, 256, (77,), generator=g, dtype=torch.uint8),
            torch.empty(0)] + [torch.randn(33 + i, generator=g) for i in range(14)]           # 20 tensors: two launches
    srcs = [s.cuda() for s in srcs]
    srcs[3] = srcs[3][1:]                                       # 4-byte aligned source: the byte-wise branch
    dsts = [torch.zeros_like(s) for s in srcs]
    assert (srcs[3].data_ptr() & 15) != 0
    ops.copy_multi(list(zip(srcs, dsts)))
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
    for dt_ in DT:
        ref = _rand((1000, 64), 5, dt_).cuda(); gr = _rand((1000, 64), 6, dt_).cuda(); keep = gr.clone()
        out = ops.relu_bwd(ref, gr, out=torch.full_like(gr, float("nan")))
        assert torch.equal(gr, keep) and torch.equal(out, torch.where(ref > 0, gr, torch.zeros_like(gr)))
        assert ops.relu_bwd(ref, gr) is gr and torch.equal(gr, out)


@pytest.mark.parametrize("dtype", DT)
def test_preprocess_multi_equals_single_image_launches(ops, dtype):
    """sw_preprocess_multi (a view batch in one launch, also more images than one launch carries) == sw_preprocess per image == the oracle"""
    g = torch.Generator().manual_seed(8)
    H, W, n = 37, 53, 19
    cpad = 8 if dtype == torch.bfloat16 else 4
    imgs = [torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).cuda() for _ in range(n)]
    mean, std = [103.939, 116.779, 123.68], [1.0, 57.375, 1.0]
    out = torch.full((n, H, W, cpad), float("nan"), device="cuda", dtype=dtype)
    ops.preprocess_multi(imgs, out, mean, std)
    for i, im in enumerate(imgs):
        one = torch.empty(H, W, cpad, device="cuda", dtype=dtype)
        ops.preprocess(im, one, mean, std)
        assert torch.equal(out[i], one)
        want = ((im.float().cpu() - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)).permute(1, 2, 0).to(dtype)
        assert torch.equal(out[i, :, :, :3].cpu(), want) and float(out[i, :, :, 3:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("out_bf16", [True, False])
@pytest.mark.parametrize("bk", [False, True])
def test_gemm_split_k_with_an_epilogue_folds_like_the_single_launch(ops, out_bf16, bk):
    """sw_gemm with splitk > 1 AND a non-plain epilogue: plain f32 slabs, then the fold applies row scale, bias, residual, ReLU, the
    ReLU-mask reference and the output conversion in the GEMM epilogue's order.  Against torch in f64 and against the unsplit
    launch (same epilogue in the GEMM itself); an f32 residual that IS the output accumulates in place."""
    torch.manual_seed(5)
    dev = "cuda"
    M, N, K = 950, 512, 2048
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(K, N, device=dev) if bk else torch.randn(N, K, device=dev)).bfloat16()
    bias = torch.randn(N, device=dev)
    od = torch.bfloat16 if out_bf16 else torch.float32
    res = torch.randn(M, N, device=dev).to(od)
    ref_t = torch.randn(M, N, device=dev).bfloat16()
    prod = A.double() @ (B.double() if bk else B.double().t())
    want = torch.relu(prod + bias.double() + res.double())
    want = torch.where(ref_t.double() > 0, want, torch.zeros_like(want))
    outs = {}
    for sk in (1, 4, 8):
        C = torch.full((M, N), float("nan"), device=dev, dtype=od)
        ops.gemm(A, B, C, M, N, K, False, bk, ep=ops.make_epilogue(bias=bias, relu=True, residual=res, relu_ref=ref_t, out_dtype=od), splitk=sk)
        torch.cuda.synchronize()
        err = float((C.double() - want).abs().max() / want.abs().max())
        assert err <= (6e-3 if out_bf16 else 2e-5), (sk, err)
        outs[sk] = C
    assert float((outs[4].float() - outs[1].float()).abs().max()) <= (0.5 if out_bf16 else 1e-3)
    if not out_bf16:                                     # gradient accumulation: C += A B^T with C as its own residual
        C = res.clone()
        ops.gemm(A, B, C, M, N, K, False, bk, ep=ops.make_epilogue(residual=C, out_dtype=torch.float32), splitk=4)
        torch.cuda.synchronize()
        err = float((C.double() - (prod + res.double())).abs().max() / prod.abs().max())
        assert err <= 2e-5, err


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_grouped_plain_weight_gradient_gemms_and_multi_fold(ops, dtype):
    """sw_gemm_kk_grouped + sw_splitk_fold_multi: several C = A^T B problems (both operands K-strided, K = pixels) of different
    sizes, pitches and split counts in one launch, folded — with a row scale, and added to an existing C — in one more launch.
    Against torch in f64; two problems of the same weight share one fold over all of their slabs."""
    torch.manual_seed(3)
    dev = "cuda"
    spec = [(128, 512, 3800, 3, None), (512, 128, 950, 2, None), (2048, 512, 1900, 4, "scale"), (64, 256, 7000, 5, "acc"),
            (256, 1024, 333, 1, "scale")]
    probs, folds, refs, outs = [], [], [], []
    for M, N, K, ns, opt in spec:
        A = torch.randn(K, M + 8, device=dev).to(dtype)[:, :M]            # pitch > M
        B = torch.randn(K, N, device=dev).to(dtype)
        A2 = torch.randn(K // 2 + 7, M, device=dev).to(dtype)             # a second use of the same weight (another pass)
        B2 = torch.randn(K // 2 + 7, N, device=dev).to(dtype)
        n1, n2 = ops.gemm_kk_nslab(dtype, K, ns), ops.gemm_kk_nslab(dtype, A2.shape[0], 2)
        ws = torch.full((n1 + n2, M * N), float("nan"), device=dev)
        probs += [(A, B, ws, ns), (A2, B2, ws[n1:], 2)]
        ref = A.double().t() @ B.double() + A2.double().t() @ B2.double()
        rs = (torch.rand(M, device=dev) + 0.5) if opt == "scale" else None
        C = torch.randn(M, N, device=dev) if opt == "acc" else torch.full((M, N), float("nan"), device=dev)
        if rs is not None:
            ref = ref * rs.double()[:, None]
        if opt == "acc":
            ref = ref + C.double()
        folds.append((ws, n1 + n2, C, rs, opt == "acc"))
        refs.append(ref); outs.append(C)
    ops.gemm_kk_grouped(probs)
    ops.splitk_fold_multi(folds)
    torch.cuda.synchronize()
    for (M, N, K, ns, opt), C, ref in zip(spec, outs, refs):
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-5, (M, N, K, ns, opt, err)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 256, 256, 64, 64), (1, 375, 501, 64, 128), (2, 187, 250, 128, 128), (1, 9, 66, 64, 64),
                                   (2, 131, 270, 64, 128)])
def test_conv3x3_relu_pool2_fused_equals_conv_then_pool(shape):
    """sw_conv3x3_relu_pool2 (the frozen conv1_2 / conv2_2 of the backbone: 2x2 / stride-2 max pool inside the convolution's epilogue,
    vgg.py:104-122) against sw_conv3x3_igemm + sw_maxpool2x2_fwd on the same operands: identical bits, odd heights / widths (the last
    row / column belongs to no window) and partial tiles included."""
    import sos_wsod_amd.ops as ops
    n, H, W, cin, cout = shape
    g = torch.Generator(device="cuda"); g.manual_seed(H * 1000 + W)
    x = (torch.randn(n, H, W, cin, device="cuda", generator=g) * 0.7).to(torch.bfloat16)
    wk = (torch.randn(cout, 9, cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    full = torch.empty(n, H, W, cout, device="cuda", dtype=torch.bfloat16)
    ops.conv3x3(x, wk, full, 1, ops.make_epilogue(bias=b, relu=True, out_dtype=torch.bfloat16))
    oh, ow = (H - 2) // 2 + 1, (W - 2) // 2 + 1
    want = torch.empty(n, oh, ow, cout, device="cuda", dtype=torch.bfloat16)
    ops.maxpool_fwd(full, want, 2)
    got = torch.full((n, oh, ow, cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    took = ops.conv3x3_relu_pool2(x, wk, b, got)
    if H * W < 4096:                 # few tiles: the unfused convolution runs another kernel form (K groups) — the fused entry declines
        assert not took
        return
    assert took
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert float(want.float().abs().max()) > 0
