"""Every direct-convolution layer of the VGG16 backbone, forward, ALONE (20 back-to-back launches, HIP events), at a given view size:
   python tools/conv_shapes.py [H W]...      (default: the headline 512x512, the recipe's mean scale 848x1131, the COCO shape 800x1333)
Prints per layer: us per launch, TFLOP/s by the layer's true FLOP, fraction of the 2.5 PFLOP/s dense bf16 peak, workgroup rounds.
TAG=... labels the lines (kernel experiments: build variants / development switches in the environment)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops

dt, dev = torch.bfloat16, "cuda"
PEAK = 2500.0


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def layers(H, W):
    """(name, h, w, cin, cout, dilation) of the direct-kernel layers of a view of H x W pixels (vgg.py:104-122: pools after conv1..3 only)"""
    p = lambda v: (v - 2) // 2 + 1
    h1, w1 = H, W
    h2, w2 = p(h1), p(w1)
    h3, w3 = p(h2), p(w2)
    h4, w4 = p(h3), p(w3)
    return [("conv1_2", h1, w1, 64, 64, 1), ("conv2_1", h2, w2, 64, 128, 1), ("conv2_2", h2, w2, 128, 128, 1),
            ("conv3_1", h3, w3, 128, 256, 1), ("conv3_2", h3, w3, 256, 256, 1),
            ("conv4_1", h4, w4, 256, 512, 1), ("conv4_2", h4, w4, 512, 512, 1), ("conv5_x", h4, w4, 512, 512, 2)]


def main():
    a = [int(v) for v in sys.argv[1:]]
    sizes = list(zip(a[0::2], a[1::2])) or [(512, 512), (848, 1131), (800, 1333)]
    tag = os.environ.get("TAG", "-")
    only = os.environ.get("ONLY")
    for H, W in sizes:
        tot_t = tot_f = 0.0
        for name, h, w, cin, cout, dil in layers(H, W):
            if only and only not in name:
                continue
            x = rnd(2, h, w, cin); wk = rnd(cout, 9, cin) * 0.05
            b = torch.zeros(cout, device=dev); out = torch.empty(2, h, w, cout, device=dev, dtype=dt)
            ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
            t = timeit(lambda: ops.conv3x3(x, wk, out, dil, ep))
            fl = 2.0 * 2 * h * w * cout * 9 * cin
            tiles = 2 * ((h + 7) // 8) * ((w + 31) // 32) * ((cout + 63) // 64)
            waste = tiles * 256 * 64 / (2.0 * h * w * cout)
            tot_t += t; tot_f += fl
            print(f"{tag:10s} {H}x{W} {name} {h}x{w} {cin}->{cout} d{dil}: {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s {fl/t/1e9/PEAK:5.3f}  "
                  f"tiles {tiles} ({tiles/256:.2f}/CU) issued/true {waste:.3f}", flush=True)
            del x, wk, out
        if tot_t:
            print(f"{tag:10s} {H}x{W} all layers once: {tot_t*1e3:7.1f} us {tot_f/tot_t/1e9:6.0f} TF/s {tot_f/tot_t/1e9/PEAK:5.3f}", flush=True)


if __name__ == "__main__":
    main()
