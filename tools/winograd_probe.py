"""Winograd F(2x2,3x3) conv kernel against the direct kernel and a float64 convolution of the same bf16 operands: error and time.
    python tools/winograd_probe.py"""
import os, sys, math, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
dev, bf = "cuda", torch.bfloat16


def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def case(n, H, W, Cin, Cout, dil, mode, check=True):
    g = torch.Generator().manual_seed(H * 7 + Cin + mode)
    x = (torch.randn(n, H, W, Cin, generator=g).relu() * 1.5).to(bf).to(dev)
    wm = (torch.randn(Cout, Cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * Cin))).to(dev)            # f32 OIHW master
    bias = torch.randn(Cout, generator=g).to(dev) * 0.1
    if mode == 0:
        n_out, n_in = Cout, Cin
        ep = ops.make_epilogue(bias=bias, relu=True, out_dtype=bf)
        wk = torch.zeros(Cout, 9, Cin, device=dev, dtype=bf); ops.conv_weight_prep(wm, wk, 0, Cin)
        xin = x
    else:                                      # data gradient: input = dy (n, H, W, Cout), output (n, H, W, Cin), ReLU mask of the producer
        n_out, n_in = Cin, Cout
        xin = (torch.randn(n, H, W, Cout, generator=g) * 0.1).to(bf).to(dev)
        ref = x.view(n * H * W, Cin)
        ep = ops.make_epilogue(relu_ref=ref, out_dtype=bf)
        wk = torch.zeros(Cin, 9, Cout, device=dev, dtype=bf); ops.conv_weight_prep(wm, wk, 1, None)
    U = torch.empty(16, n_out, n_in, device=dev, dtype=bf)
    ops.winograd_weight_prep([(wm, U, mode)])
    out_w = torch.full((n, H, W, n_out), 7.0, device=dev, dtype=bf); out_d = torch.empty_like(out_w)
    took = ops.conv3x3_winograd(xin, U, out_w, dil, ep)
    ops.conv3x3(xin, wk, out_d, dil, ep)
    torch.cuda.synchronize()
    line = f"n={n} {H}x{W} {n_in}->{n_out} dil={dil} mode={mode}: winograd launched={took}"
    if check:
        wd = wm.double().cpu()
        if mode == 0:
            r = F.conv2d(xin.double().cpu().permute(0, 3, 1, 2), wd, bias.double().cpu(), padding=dil, dilation=dil).relu()
        else:
            wt = wd.flip(2, 3).permute(1, 0, 2, 3)
            r = F.conv2d(xin.double().cpu().permute(0, 3, 1, 2), wt, None, padding=dil, dilation=dil) * (x.double().cpu().permute(0, 3, 1, 2) > 0)
        r = r.permute(0, 2, 3, 1)
        ew = float((out_w.double().cpu() - r).norm() / r.norm()); ed = float((out_d.double().cpu() - r).norm() / r.norm())
        mw = float((out_w.double().cpu() - r).abs().max() / r.abs().max())
        line += f"  relL2 vs f64: winograd {ew:.2e} (max {mw:.2e})  direct {ed:.2e}"
    tw = t_us(lambda: ops.conv3x3_winograd(xin, U, out_w, dil, ep)); td = t_us(lambda: ops.conv3x3(xin, wk, out_d, dil, ep))
    fl = 2.0 * n * H * W * n_out * 9 * n_in
    print(line + f"  time: winograd {tw:.1f} us ({fl / tw / 1e6:.0f} TF/s direct-equivalent)  direct {td:.1f} us ({fl / td / 1e6:.0f} TF/s)", flush=True)


if __name__ == "__main__":
    case(1, 37, 50, 64, 96, 1, 0)
    case(2, 21, 30, 64, 64, 2, 0)
    case(1, 37, 50, 96, 64, 2, 1)
    for mode in (0, 1):
        case(2, 63, 63, 512, 512, 2, mode)
        case(2, 64, 64, 512, 512, 1, mode)
        case(2, 128, 128, 256, 256, 1, mode, check=(mode == 0))
    case(2, 64, 64, 256, 512, 1, 0, check=False)
    case(2, 99, 165, 512, 512, 2, 0, check=False)
    t = torch.empty(16, 512, 512, device=dev, dtype=bf); w = torch.randn(512, 512, 3, 3, device=dev)
    print(f"weight prep 512x512 (one layer, one mode): {t_us(lambda: ops.winograd_weight_prep([(w, t, 0)])):.1f} us; "
          f"dgrad mode {t_us(lambda: ops.winograd_weight_prep([(w, t, 1)])):.1f} us")
