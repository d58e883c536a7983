"""Pillow's 8-bit bilinear resize on the device (the pixel side of `DatasetMapperMultiInput`,
uwsod/detectron2/data/dataset_mapper.py:303-352, and of `DatasetMapperTTAAVG`,
uwsod/projects/WSL/wsl/modeling/test_time_augmentation_avg.py:199-310 — both go through `ResizeTransform.apply_image`
= `PIL.Image.resize((w, h), Image.BILINEAR)` on the CPU in the reference).

Pillow resamples separably: a horizontal pass into an 8-bit intermediate, then a vertical pass, each
`clip8((2^21 + sum in * k) >> 22)` with coefficients normalised in double precision and rounded to 22 fractional bits
(libImaging/Resample.c).  The coefficient tables depend only on (input size, output size): they are computed here on the host
exactly as Pillow computes them (plain IEEE double arithmetic, same operation order), cached, and uploaded once; the two passes
are the HIP kernel `sw_resize_pass_u8`.  Pixels are bit-identical to Pillow's (tests/golden/resize_*.npz, generated from
Pillow itself)."""
import math

import numpy as np
import torch

from . import ops

_PRECISION_BITS = 32 - 8 - 2
_TABLES = {}


def _coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0)"""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale > 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    one = 1 << _PRECISION_BITS
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = 0 if xmin < 0 else xmin
        xmax = int(center + support + 0.5)
        xmax = in_size if xmax > in_size else xmax
        n = xmax - xmin
        k = [0.0] * n
        ww = 0.0
        for x in range(n):
            a = (x + xmin - center + 0.5) * inv
            a = -a if a < 0.0 else a
            w = (1.0 - a) if a < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(n):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (xmin, n)
    return bounds, kk


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    hit = _TABLES.get(key)
    if hit is None:
        b, k = _coeffs(in_size, out_size)
        hit = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1])
        if len(_TABLES) > 256:
            _TABLES.clear()
        _TABLES[key] = hit
    return hit


def resize_bilinear_u8(img: torch.Tensor, out_hw, with_flip=False):
    """img (C, H, W) uint8 on the GPU -> (C, h, w) uint8, bit-identical to PIL.Image.resize((w, h), BILINEAR) per band.
    with_flip: also return the x-mirrored result (written by the same launch)."""
    assert img.dtype == torch.uint8 and img.dim() == 3
    ops._need_gpu(img)
    if img.stride(2) != 1:
        img = img.contiguous()
    C, H, W = img.shape                                 # a crop window keeps its parent's strides: the kernel takes them
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = img.device
    cur = img
    passes = []
    if ow != W:
        passes.append((1, W, ow))
    if oh != H:
        passes.append((0, H, oh))
    flip = None
    if not passes:                                      # Pillow returns a copy
        cur = img.contiguous().clone() if not img.is_contiguous() else img.clone()
        return (cur, cur.flip(-1).contiguous()) if with_flip else cur
    for i, (horizontal, n_in, n_out) in enumerate(passes):
        bounds, kk, ksize = _tables(n_in, n_out, dev)
        h_in, w_in = cur.shape[1], cur.shape[2]
        out = torch.empty((C, h_in, n_out) if horizontal else (C, n_out, w_in), device=dev, dtype=torch.uint8)
        last = i == len(passes) - 1
        if last and with_flip:
            flip = torch.empty_like(out)
        ops.resize_pass_u8(cur, out, bounds, kk, ksize, horizontal, flip if last else None)
        cur = out
    return (cur, flip) if with_flip else cur
