"""GPU: the HIP restatement of Pillow's 8-bit bilinear resize (sw_resize_pass_u8 through sos_wsod_amd.resize) — bit exact
against the fixtures generated from Pillow (tests/golden/resize_*.npz) and, at the recipe's real sizes, against the oracle
restatement (itself pinned to Pillow in tests/test_resize_cpu.py); then the two mappers that use it."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.resize_oracle import resize_bilinear_u8 as oracle_resize  # noqa: E402  (checker only)


def test_resize_equals_pillow_fixtures(golden_dir):
    from sos_wsod_amd.resize import resize_bilinear_u8
    files = sorted(glob.glob(os.path.join(golden_dir, "resize_*.npz")))
    assert len(files) >= 6
    for f in files:
        g = np.load(f)
        oh, ow = (int(v) for v in g["out_hw"])
        img = torch.from_numpy(np.ascontiguousarray(g["image_hwc"].transpose(2, 0, 1))).cuda()
        out, flip = resize_bilinear_u8(img, (oh, ow), with_flip=True)
        want = np.ascontiguousarray(g["resized_hwc"].transpose(2, 0, 1))
        assert np.array_equal(out.cpu().numpy(), want), os.path.basename(f)
        assert np.array_equal(flip.cpu().numpy(), want[:, :, ::-1]), os.path.basename(f)
        assert np.array_equal(resize_bilinear_u8(img, (oh, ow)).cpu().numpy(), want)


@pytest.mark.parametrize("hw,out", [((375, 500), (480, 640)), ((375, 500), (1200, 1600)), ((500, 333), (1000, 666)),
                                    ((480, 640), (240, 320)), ((333, 500), (333, 700)), ((64, 64), (64, 64))])
def test_resize_recipe_sizes_equal_the_oracle(hw, out):
    """VOC-sized images to the recipe's MIN_SIZE_TRAIN scales (voc07_oicr_plus.yaml:30), a down-scale, one-axis and identity"""
    from sos_wsod_amd.resize import resize_bilinear_u8
    rng = np.random.RandomState(hw[0] + out[1])
    img = rng.randint(0, 256, (3, hw[0], hw[1])).astype(np.uint8)
    got, flip = resize_bilinear_u8(torch.from_numpy(img).cuda(), out, with_flip=True)
    want = oracle_resize(img, out[0], out[1])
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(flip.cpu().numpy(), want[:, :, ::-1])


def test_mappers_emit_pillow_pixels(golden_dir):
    """DeviceMultiInputMapper (dataset_mapper.py:303-352) and DeviceTTAMapper (test_time_augmentation_avg.py:199-310): every
    view's pixels == Pillow's resize of the source image (through the oracle), flipped views mirrored"""
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    from sos_wsod_amd.structures import Boxes, Instances
    from sos_wsod_amd.tta import DeviceTTAMapper
    rng = np.random.RandomState(3)
    h, w = 120, 160
    img = rng.randint(0, 256, (3, h, w)).astype(np.uint8)
    t = torch.from_numpy(img).cuda()
    boxes = np.array([[5.0, 6.0, 80.0, 90.0], [20.0, 10.0, 150.0, 110.0]], np.float32)
    d = {"image": t, "proposal_boxes": boxes, "proposal_objectness_logits": np.array([0.9, 0.4], np.float32)}
    out = DeviceMultiInputMapper(min_sizes=(96, 144, 200), max_size=400, seed=2)(d)
    for name in ("1", "2"):
        hh, ww = out["image" + name].shape[-2:]
        want = oracle_resize(img, hh, ww)
        assert np.array_equal(out["image" + name].cpu().numpy(), want)
        assert np.array_equal(out["image" + name + "_flip"].cpu().numpy(), want[:, :, ::-1])
    p = Instances((h, w)); p.proposal_boxes = Boxes(torch.from_numpy(boxes).cuda()); p.objectness_logits = torch.tensor([0.9, 0.4]).cuda()
    views = DeviceTTAMapper(min_sizes=(96, 150), max_size=170, flip=True)({"image": t, "proposals": p})
    assert len(views) == 4
    for view, tfm in views:
        hh, ww = tfm.new_hw
        want = oracle_resize(img, hh, ww)
        assert np.array_equal(view["image"].cpu().numpy(), want[:, :, ::-1] if tfm.flip else want)


def test_resize_of_a_crop_window_takes_strides():
    """a crop (CropTransform: image[y0:y0+h, x0:x0+w]) reaches the kernel as pointer + parent strides; same bits as the copy"""
    from sos_wsod_amd.resize import resize_bilinear_u8
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (3, 211, 307)).astype(np.uint8)
    t = torch.from_numpy(img).cuda()
    for (y0, x0, ch, cw), out in (((7, 13, 190, 280), (384, 566)), ((0, 0, 211, 300), (100, 142)), ((20, 30, 100, 200), (100, 300)),
                                  ((20, 30, 100, 200), (100, 200))):
        win = t[:, y0:y0 + ch, x0:x0 + cw]
        assert not win.is_contiguous() or (x0 == 0 and cw == 307)
        got, flip = resize_bilinear_u8(win, out, with_flip=True)
        want = oracle_resize(np.ascontiguousarray(img[:, y0:y0 + ch, x0:x0 + cw]), out[0], out[1])
        assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(flip.cpu().numpy(), want[:, :, ::-1])


def test_color_jitter_equals_the_blend_restatement():
    """sw_color_jitter_u8 against oracle.input_oracle.blend_u8 (itself equal to the reference run's pixels, tests/test_mapper_cpu.py):
    random pixels, grey pixels (B = G = R: the blend is mathematically the identity there, float rounding decides), saturated
    weights that clip at both ends, brightness-only / saturation-only"""
    from oracle import input_oracle as IO
    from sos_wsod_amd import ops
    rng = np.random.RandomState(11)
    h, w = 67, 131
    img = rng.randint(0, 256, (3, h, w)).astype(np.uint8)
    img[:, :20] = rng.randint(0, 256, (1, 20, w))                # grey rows
    img[:, 20:24] = 255; img[:, 24:28] = 0
    t = torch.from_numpy(img).cuda()
    hwc = np.ascontiguousarray(img.transpose(1, 2, 0))
    for wb, ws in ((0.9233439722, 0.8865590297), (1.5, 1.5), (1 / 1.5, 1 / 1.5), (1.0, 1.0), (1.4136172, 0.7), (0.7, 1.4999)):
        got, flip = ops.color_jitter_u8(t, wb, ws, with_flip=True)
        want = IO.blend_u8(hwc, wb, ws).transpose(2, 0, 1)
        assert np.array_equal(got.cpu().numpy(), want), (wb, ws)
        assert np.array_equal(flip.cpu().numpy(), want[:, :, ::-1])
    only_b = ops.color_jitter_u8(t, 1.3, None).cpu().numpy()
    assert np.array_equal(only_b, np.clip(np.float32(1.3) * img.astype(np.float32), 0, 255).astype(np.uint8))
    only_s = ops.color_jitter_u8(t, None, 0.8).cpu().numpy()
    g = (hwc[..., 0] * 0.299 + hwc[..., 1] * 0.587) + hwc[..., 2] * 0.114
    v = (1 - 0.8) * g[..., None] + (np.float32(0.8) * hwc.astype(np.float32)).astype(np.float64)
    assert np.array_equal(only_s, np.clip(v, 0, 255).astype(np.uint8).transpose(2, 0, 1))


@pytest.mark.parametrize("case", ["a", "v", "m"])
def test_multi_input_mapper_matches_the_reference_run(case):
    """the whole training mapper on the device — RandomCrop window, two PIL resizes, brightness + saturation blends, flips, the
    four index-aligned proposal sets, annotations — against fixtures written by RUNNING the reference's
    DatasetMapperMultiInput.__call__ (dataset_mapper.py:272-439; tests/golden/make_mapper_golden.py): same numpy seed -> same
    draws; boxes bit exact; pixels bit identical (case a: every pixel; v, m: CRC32 of each view)"""
    import zlib
    from test_mapper_cpu import VIEW_KEYS, check_mapper_boxes_against_reference_run
    g, out = check_mapper_boxes_against_reference_run(case, "cuda")
    for key in VIEW_KEYS:
        px = out["image" + key]
        assert px.is_cuda and px.dtype == torch.uint8 and px.is_contiguous()
        arr = px.cpu().numpy()
        if "image" + key in g:
            assert np.array_equal(arr, g["image" + key]), key
        assert zlib.crc32(arr.tobytes()) == int(g["crc_" + key]), key
    # and the model accepts what the mapper emits (4 views, aligned proposals)
    assert len({len(out["proposals" + k]) for k in VIEW_KEYS}) == 1
