"""ROIPooler, single-level `ROIPool` branch of the reference
(uwsod/projects/WSL/wsl/modeling/poolers.py:81-108 convert_boxes_to_pooler_format, :111-306 ROIPooler;
the level pooler is torchvision.ops.RoIPool there, sw_roi_pool_fwd/bwd here)."""
import torch
import torch.nn as nn

from . import ops
from .structures import Boxes


def convert_boxes_to_pooler_format(box_lists):
    """list[Boxes] (one per image) -> (M,5) f32 rows (batch index, x0, y0, x1, y1)."""
    out = []
    for i, b in enumerate(box_lists):
        t = b.tensor if isinstance(b, Boxes) else b
        out.append(torch.cat([torch.full((len(t), 1), float(i), dtype=t.dtype, device=t.device), t], dim=1))
    return torch.cat(out, dim=0)


class _RoIPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_nhwc, rois, out_size, scale):
        R, C = rois.shape[0], feat_nhwc.shape[3]
        out = torch.empty(R, C * out_size * out_size, device=feat_nhwc.device, dtype=feat_nhwc.dtype)
        arg = torch.empty(R, C * out_size * out_size, device=feat_nhwc.device,
                          dtype=ops.roi_argmax_dtype(feat_nhwc.shape[1], feat_nhwc.shape[2]))
        ops.roi_pool_fwd(feat_nhwc, rois, out, arg, scale, out_size, out_size)
        ctx.save_for_backward(arg, rois)
        ctx.shape, ctx.out_size, ctx.scale = feat_nhwc.shape, out_size, scale
        return out.view(R, C, out_size, out_size)

    @staticmethod
    def backward(ctx, g):
        arg, rois = ctx.saved_tensors
        dfeat = torch.empty(ctx.shape, device=g.device, dtype=g.dtype)
        ops.roi_pool_bwd(g.contiguous().view(g.shape[0], -1), arg, rois, dfeat, ctx.out_size, ctx.out_size, spatial_scale=ctx.scale)
        return dfeat, None, None, None


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio=0, pooler_type="ROIPool", canonical_box_size=224,
                 canonical_level=4):
        super().__init__()
        if isinstance(output_size, (tuple, list)):
            assert output_size[0] == output_size[1]
            output_size = output_size[0]
        assert pooler_type == "ROIPool", "POOLER_TYPE ROIPool is the hot path (Base-RCNN-DilatedC5.yaml / voc07_oicr_plus.yaml:25)"
        assert len(scales) == 1, "single feature level (IN_FEATURES ['plain5'])"
        self.output_size, self.scale = output_size, float(scales[0])

    def forward(self, x, box_lists):
        """x: [feature (N,C,h,w)] (an NCHW view of NHWC storage is consumed without a copy);
        box_lists: list[Boxes] per image -> (M, C, P, P)."""
        assert isinstance(x, list) and isinstance(box_lists, list), "Arguments to pooler must be lists"
        assert len(x) == 1 and len(box_lists) == x[0].size(0)
        feat = x[0].permute(0, 2, 3, 1).contiguous()
        rois = convert_boxes_to_pooler_format(box_lists).to(torch.float32)
        return _RoIPoolFunction.apply(feat, rois, self.output_size, self.scale)
