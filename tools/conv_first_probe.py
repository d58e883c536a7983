"""conv1_1 (3 -> 64 channels, input padded to 8) alone at the headline / recipe / COCO view sizes, as a fraction of the HBM time of its
input + output bytes (round 6: 0.33-0.41 of 8 TB/s alone — the 179 us per call inside the recipe step is the two scale streams sharing
the chip, not the kernel; an LDS-staged rewrite was started on the in-step number and dropped on this one).  DUMP=path saves the outputs,
CMP=path compares them bit for bit with a saved set (A/B of two library builds)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
g = torch.Generator(device=dev); g.manual_seed(5)
outs = {}
for H, W in [(512, 512), (848, 1131), (800, 1333), (37, 70), (5, 3)]:
    x = torch.zeros(2, H, W, 8, device=dev, dtype=dt); x[..., :3] = (torch.randn(2, H, W, 3, device=dev, generator=g) * 60).to(dt)
    wk = (torch.randn(64, 9, 8, device=dev, generator=g) * 0.05).to(dt); wk[..., 3:] = 0
    b = torch.randn(64, device=dev, generator=g) * 0.1
    out = torch.full((2, H, W, 64), float("nan"), device=dev, dtype=dt)
    ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
    t = timeit(lambda: ops.conv3x3(x, wk, out, 1, ep))
    outs[(H, W)] = out.cpu()
    by = 2 * H * W * (64 + 8) * 2
    print(f"{os.environ.get('TAG', '-'):8s} conv1_1 2x{H}x{W}: {t*1e3:7.1f} us  {by/t/1e6:6.0f} GB/s of in+out bytes ({by/t/1e6/8000:.3f} of 8 TB/s)", flush=True)
if os.environ.get("DUMP"):
    torch.save(outs, os.environ["DUMP"])
if os.environ.get("CMP"):
    ref = torch.load(os.environ["CMP"])
    for k, v in outs.items():
        same = torch.equal(v.view(torch.int16), ref[k].view(torch.int16))
        print(f"   {k}: bit-equal to the other form: {same}, finite: {bool(torch.isfinite(v.float()).all())}")
