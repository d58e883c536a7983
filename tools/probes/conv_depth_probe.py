"""DEPTH 2 vs 3 weight ring of the KG = 2 direct kernel (dilation 1): equality (same accumulation order) + time, each in its own process"""
import os, sys, subprocess, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if len(sys.argv) > 1:
    import sos_wsod_amd.ops as ops
    dt, dev = torch.bfloat16, "cuda"
    g = torch.Generator(device=dev); g.manual_seed(5)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dt)
    def timeit(fn, n=40):
        for _ in range(5): fn()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
    res = {}
    for name, n, H, W, cin, cout, dil in [("conv4_2", 2, 64, 64, 512, 512, 1), ("conv4_1", 2, 64, 64, 256, 512, 1), ("conv4_2o", 2, 63, 61, 512, 512, 1),
                                          ("conv4_2c", 2, 64, 64, 64, 512, 1), ("conv5_3", 2, 63, 63, 512, 512, 2)]:
        x = rnd(n, H, W, cin); wk = rnd(cout, 9, cin); b = rnd(cout).float(); out = torch.empty(n, H, W, cout, device=dev, dtype=dt)
        ep = ops.make_epilogue(bias=b, relu=True, out_dtype=dt)
        t = min(timeit(lambda: ops.conv3x3(x, wk, out, dil, ep)) for _ in range(3))
        fl = 2.0 * n * H * W * cout * 9 * cin
        print(f"depth={os.environ.get('SW_CONV_DIRECT_DEPTH', '3(default)'):11s} {name:9s} {t*1e3:6.1f} us {fl/t/1e9:6.0f} TF/s", flush=True)
        res[name] = out.cpu()
    torch.save(res, sys.argv[1])
else:
    for d, f in (("2", "/tmp/cd2.pt"), ("", "/tmp/cd3.pt")):
        env = dict(os.environ)
        if d: env["SW_CONV_DIRECT_DEPTH"] = d
        subprocess.run([sys.executable, __file__, f], env=env, check=True, timeout=300)
    a, b = torch.load("/tmp/cd2.pt"), torch.load("/tmp/cd3.pt")
    for k in a:
        print(k, "bit-equal" if torch.equal(a[k], b[k]) else f"DIFF max {float((a[k].float()-b[k].float()).abs().max())}")
