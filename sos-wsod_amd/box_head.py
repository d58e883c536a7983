"""fc6/fc7 neck with the reference's interface and state-dict names
(uwsod/projects/WSL/wsl/modeling/roi_heads/box_head.py:17-103: `fc1`, `fc2`, init N(0,0.005) / bias 0.1,
forward = flatten -> [Linear -> ReLU -> dropout(0.5)] x len(fc_dims))."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .registry import ROI_BOX_HEAD_REGISTRY
from .structures import ShapeSpec


class _Linear(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d_out, d_in))
        self.bias = nn.Parameter(torch.empty(d_out))


@ROI_BOX_HEAD_REGISTRY.register()
class DiscriminativeAdaptionNeck(nn.Module):
    def __init__(self, input_shape: ShapeSpec, *, conv_dims=(), fc_dims=(4096, 4096), conv_norm="",
                 compute_dtype=torch.bfloat16):
        super().__init__()
        assert len(conv_dims) == 0, "NUM_CONV 0 on this path (voc07_oicr_plus.yaml:26)"
        assert len(fc_dims) == 2, "the fused head kernels are laid out for fc6+fc7 (DAN_DIM of length 2)"
        self.compute_dtype = compute_dtype
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        d_in = int(np.prod(self._output_size))
        self.fcs = []
        for k, d in enumerate(fc_dims):
            fc = _Linear(d_in, d)
            torch.nn.init.normal_(fc.weight, std=0.005)      # box_head.py:64-67
            torch.nn.init.constant_(fc.bias, 0.1)
            self.add_module("fc{}".format(k + 1), fc)
            self.fcs.append(fc)
            d_in = d
        self._output_size = d_in

    @classmethod
    def from_config(cls, cfg, input_shape):
        from .backbone_vgg import _dtype_from_cfg
        return dict(input_shape=input_shape, conv_dims=[cfg.MODEL.ROI_BOX_HEAD.CONV_DIM] * cfg.MODEL.ROI_BOX_HEAD.NUM_CONV,
                    fc_dims=cfg.MODEL.ROI_BOX_HEAD.DAN_DIM, conv_norm=cfg.MODEL.ROI_BOX_HEAD.NORM,
                    compute_dtype=_dtype_from_cfg(cfg))

    @property
    def output_shape(self):
        return ShapeSpec(channels=self._output_size)

    @torch.no_grad()
    def forward(self, x, drop_masks=None):
        """Inference-style forward (no autograd; training runs through OICRPlusHeads' fused function).
        x: (R, C, 7, 7) or (R, D) in compute dtype.  drop_masks: optional (m1, m2) uint8 keep masks."""
        x = x.reshape(x.shape[0], -1)
        if x.dtype != self.compute_dtype:
            x = x.to(self.compute_dtype)
        for i, fc in enumerate(self.fcs):
            d_out, d_in = fc.weight.shape
            # the compute-dtype copy is kept until the parameter changes (it was rebuilt on every call: 616 MB of traffic per view
            # at inference for fc6 + fc7)
            key = (ops.param_key(fc.weight), self.compute_dtype, x.device)
            hit = self.__dict__.setdefault("_fwd_stage", {}).get(i)
            if hit is None or hit[0] != key:
                w = torch.empty(d_out, d_in, device=x.device, dtype=self.compute_dtype)
                ops.convert_2d(fc.weight.detach(), w, d_out, d_in)
                self.__dict__["_fwd_stage"][i] = hit = (key, w)
            w = hit[1]
            out = torch.empty(x.shape[0], d_out, device=x.device, dtype=self.compute_dtype)
            m = None if drop_masks is None else drop_masks[i]
            ops.gemm(x, w, out, x.shape[0], d_out, d_in,
                     ep=ops.make_epilogue(bias=fc.bias.detach(), relu=True, drop_mask=m, drop_scale=2.0,
                                          out_dtype=self.compute_dtype))
            x = out
        return x
