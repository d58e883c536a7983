"""ROIPool forward / backward alone at the benchmark shape (2 x 63 x 63 x 512 map, 4000 ROIs), bytes of output + argmax per second."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
dt, dev = torch.bfloat16, "cuda"
R, C, H, W = 4000, 512, 63, 63
g = torch.Generator().manual_seed(0)
x1 = torch.rand(R, generator=g) * 480; y1 = torch.rand(R, generator=g) * 480
bw = 24 + torch.rand(R, generator=g) * (512 - x1 - 24); bh = 24 + torch.rand(R, generator=g) * (512 - y1 - 24)
rois = torch.stack([(torch.arange(R) >= R // 2).float(), x1, y1, (x1 + bw).clamp(max=512), (y1 + bh).clamp(max=512)], 1).cuda()
feat = torch.randn(2, H, W, C, device=dev).relu().to(dt); obj = torch.rand(R, device=dev)
out = torch.empty(R, C * 49, device=dev, dtype=dt); arg = torch.empty(R, C * 49, device=dev, dtype=torch.int16 if os.environ.get("A16", "1") == "1" else torch.int32)
dout = torch.randn(R, C * 49, device=dev).to(dt); dfeat = torch.empty_like(feat)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
tf = t(lambda: ops.roi_pool_fwd(feat, rois, out, arg, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0))
amax = ops.absmax(dout)
tb = t(lambda: ops.roi_pool_bwd(dout, arg, rois, dfeat, 7, 7, row_scale=obj, row_scale_add=1.0, relu_ref=feat, dout_absmax=amax, spatial_scale=0.0 if os.environ.get('NOSCALE') else 0.125))
gb = R * C * 49 * (2 + arg.element_size()) / 1e9
print(f"roi_pool fwd {tf*1e3:.0f} us ({gb/tf:.2f} TB/s of out+argmax)   bwd {tb*1e3:.0f} us ({gb/tb:.2f} TB/s)")
