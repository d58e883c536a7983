"""ROIPool forward / backward at a given map size (env RH, RW, RR; default the VOC-sized 76 x 114 map, 4000 ROIs on 2 images): which
slab width is faster when the plane exceeds 76 KiB?  SW_ROI_FWD_PXB=16|8|4|0 forces the forward's slab bytes per pixel (0 = gather)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
R, C, H, W = int(os.environ.get('RR', 4000)), 512, int(os.environ.get('RH', 76)), int(os.environ.get('RW', 114))
g = torch.Generator().manual_seed(0)
x1 = torch.rand(R, generator=g) * (W * 8 - 32); y1 = torch.rand(R, generator=g) * (H * 8 - 32)
bw = 24 + torch.rand(R, generator=g) * (W * 8 - x1 - 24); bh = 24 + torch.rand(R, generator=g) * (H * 8 - y1 - 24)
rois = torch.stack([(torch.arange(R) >= R // 2).float(), x1, y1, (x1 + bw).clamp(max=W * 8), (y1 + bh).clamp(max=H * 8)], 1).cuda()
feat = torch.randn(2, H, W, C, device=dev).relu().to(dt); obj = torch.rand(R, device=dev)
out = torch.empty(R, C * 49, device=dev, dtype=dt); arg = torch.empty(R, C * 49, device=dev, dtype=torch.int16)
dout = torch.randn(R, C * 49, device=dev).to(dt); dfeat = torch.empty_like(feat)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tf = t(lambda: ops.roi_pool_fwd(feat, rois, out, arg, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0))
amax = ops.absmax(dout)
tb = t(lambda: ops.roi_pool_bwd(dout, arg, rois, dfeat, 7, 7, row_scale=obj, row_scale_add=1.0, relu_ref=feat, dout_absmax=amax, spatial_scale=0.0 if os.environ.get('NOSCALE') else 0.125))
print(f"{H}x{W} map, {R} ROIs, pxb={os.environ.get('SW_ROI_FWD_PXB', 'auto')}: roi_pool fwd {tf*1e3:.0f} us   bwd {tb*1e3:.0f} us")
