"""ROIPool forward: the prepared-task form (sw_roi_pool_fwd_ws) against the form without workspace, and its development switches.
    python tools/roi_tasks_forms.py                     # all shapes: no workspace | prepared tasks (default geometry)
    python tools/roi_tasks_forms.py sweep               # + SW_ROI_TASKS_CB / _NT / _HALO / SW_ROI_FWD_WGS variants on the large maps
Each variant runs in its own child process (the switches are read once per process); prints the time and checks that values and
argmax equal the no-workspace form's bit for bit."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SHAPES = [(63, 63, 4000), (76, 114, 4000), (99, 165, 8000), (104, 139, 4000), (125, 167, 4000), (150, 200, 4000), (47, 62, 4000)]


def one(H, W, R):
    import torch
    sys.path.insert(0, ROOT)
    import sos_wsod_amd.ops as ops
    dt, dev, C = torch.bfloat16, "cuda", 512
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(R, generator=g) * (W * 8 - 32); y1 = torch.rand(R, generator=g) * (H * 8 - 32)
    bw = 24 + torch.rand(R, generator=g) * (W * 8 - x1 - 24); bh = 24 + torch.rand(R, generator=g) * (H * 8 - y1 - 24)
    if os.environ.get("ROI_MAXH"):                                   # experiment: ROIs no taller than this many map rows
        bh = bh.clamp(max=8.0 * float(os.environ["ROI_MAXH"]))
    rois = torch.stack([(torch.arange(R) >= R // 2).float(), x1, y1, (x1 + bw).clamp(max=W * 8), (y1 + bh).clamp(max=H * 8)], 1).cuda()
    feat = torch.randn(2, H, W, C, device=dev).relu().to(dt); obj = torch.rand(R, device=dev)
    mk = lambda d: torch.empty(R, C * 49 + 64, device=dev, dtype=d)[:, :C * 49]
    out0, arg0, out, arg = mk(dt), mk(torch.int16), mk(dt), mk(torch.int16)
    ops.roi_pool_fwd(feat, rois, out0, arg0, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0, workspace=None)
    ws = ops.roi_pool_fwd_workspace(2, R, 7, 7, dev)
    res = []
    for w in (None, ws):
        fn = lambda: ops.roi_pool_fwd(feat, rois, out, arg, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0, workspace=w)
        for _ in range(3): fn()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); [fn() for _ in range(20)]; b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 20 * 1e3)
    same = torch.equal(out.view(torch.int16), out0.view(torch.int16)) and torch.equal(arg, arg0)
    alg = R * 25088 * 4 + 2 * H * W * C * 2
    sw = " ".join(f"{k[7:]}={v}" for k, v in os.environ.items() if k.startswith("SW_ROI_"))
    print(f"{H}x{W} map, {R} ROIs [{sw}]: no workspace {res[0]:7.1f} us | prepared tasks {res[1]:7.1f} us = {alg / res[1] / 1e6 / 8:.3f} of 8 TB/s"
          f"  {'identical' if same else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        sweep = len(sys.argv) > 1 and sys.argv[1] == "sweep"
        variants = [{}]
        if sweep:
            variants += [{"SW_ROI_TASKS_CB": "2"}, {"SW_ROI_TASKS_CB": "4"}, {"SW_ROI_TASKS_CB": "8"}, {"SW_ROI_TASKS_CB": "2", "SW_ROI_TASKS_NT": "512"},
                         {"SW_ROI_TASKS_CB": "2", "SW_ROI_FWD_WGS": "512"}]
        for H, W, R in SHAPES:
            for v in variants:
                subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(H), str(W), str(R)], env=dict(os.environ, **v))
