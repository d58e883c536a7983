"""Where the row-sparse-table ROIPool forward spends its time: per-phase shader-clock cycles of each workgroup's thread 0 (variant build
with -DSW_ROI_PHASES: tools/build_variant.sh phases SRC=roipool -DSW_ROI_PHASES; run with SW_LIB_PATH=.../libsoswsod_hip_phases.so)."""
import ctypes, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import sos_wsod_amd.ops as ops
from sos_wsod_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.sw_debug_roi_phases.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
NAMES = ["sort (sparse kernel)", "level-0 table", "level advances", "chunk tables + task list | first records of a level", "task scan | item loop", "barrier wait behind a scan / at a level's start", "whole kernel", "workgroups"]
for H, W, R in [(63, 63, 4000), (99, 165, 8000), (150, 200, 4000)]:
    dt, dev, C = torch.bfloat16, "cuda", 512
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(R, generator=g) * (W * 8 - 32); y1 = torch.rand(R, generator=g) * (H * 8 - 32)
    bw = 24 + torch.rand(R, generator=g) * (W * 8 - x1 - 24); bh = 24 + torch.rand(R, generator=g) * (H * 8 - y1 - 24)
    rois = torch.stack([(torch.arange(R) >= R // 2).float(), x1, y1, (x1 + bw).clamp(max=W * 8), (y1 + bh).clamp(max=H * 8)], 1).cuda()
    feat = torch.randn(2, H, W, C, device=dev).relu().to(dt); obj = torch.rand(R, device=dev)
    out = torch.empty(R, C * 49 + 64, device=dev, dtype=dt)[:, :C * 49]; arg = torch.empty(R, C * 49 + 64, device=dev, dtype=torch.int16)[:, :C * 49]
    fn = lambda: ops.roi_pool_fwd(feat, rois, out, arg, 0.125, 7, 7, row_scale=obj, row_scale_add=1.0)
    for _ in range(3): fn()
    buf = (ctypes.c_uint64 * 8)()
    assert lib.sw_debug_roi_phases(buf, 1) == 0
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(10)]; b.record(); torch.cuda.synchronize()
    assert lib.sw_debug_roi_phases(buf, 1) == 0
    v = [x / 10 for x in buf]
    print(f"{H}x{W} map, {R} ROIs: {a.elapsed_time(b) / 10 * 1e3:.1f} us per call, {v[7]:.0f} workgroups, {v[6] / v[7]:.0f} cycles per workgroup")
    for i in range(6):
        print(f"   {NAMES[i]:28s} {v[i] / v[7]:9.0f} cycles per workgroup  {100 * v[i] / v[6]:5.1f} %")
