"""ROIPool forward: the scan form (SW_ROI_FWD_SPARSE=0) against the row sparse table form at the map sizes the recipe produces.
Each form runs in its own child process (the switch is read once per process).
    python tools/roi_fwd_forms.py            # all shapes, both forms
    python tools/roi_fwd_forms.py one H W R  # one measurement in this process (used by the driver loop)"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SHAPES = [(63, 63, 4000), (76, 114, 4000), (99, 165, 8000), (125, 167, 4000), (150, 200, 4000), (47, 62, 4000)]


def one(H, W, R):
    import torch
    sys.path.insert(0, ROOT)
    import sos_wsod_amd.ops as ops
    dt, dev, C = torch.bfloat16, "cuda", 512
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(R, generator=g) * (W * 8 - 32); y1 = torch.rand(R, generator=g) * (H * 8 - 32)
    bw = 24 + torch.rand(R, generator=g) * (W * 8 - x1 - 24); bh = 24 + torch.rand(R, generator=g) * (H * 8 - y1 - 24)
    rois = torch.stack([(torch.arange(R) >= R // 2).float(), x1, y1, (x1 + bw).clamp(max=W * 8), (y1 + bh).clamp(max=H * 8)], 1).cuda()
    feat = torch.randn(2, H, W, C, device=dev).relu().to(dt); obj = torch.rand(R, device=dev)
    out = torch.empty(R, C * 49 + 64, device=dev, dtype=dt)[:, :C * 49]; arg = torch.empty(R, C * 49 + 64, device=dev, dtype=torch.int16)[:, :C * 49]
    fn = lambda: ops.roi_pool_fwd(feat, rois, out, arg, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0)
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(20)]; b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    alg = R * 25088 * 4 + 2 * H * W * C * 2
    print(f"{H}x{W} map, {R} ROIs, sparse={os.environ.get('SW_ROI_FWD_SPARSE', '1')}: {us:7.1f} us  = {alg / us / 1e6:.2f} TB/s algorithmic "
          f"({alg / us / 1e6 / 8:.3f} of 8 TB/s)  checksum {int(arg.to(torch.int32).sum())} {float(out.float().sum()):.6g}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        for H, W, R in SHAPES:
            for sp in ("0", "1"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(H), str(W), str(R)], env=dict(os.environ, SW_ROI_FWD_SPARSE=sp))
