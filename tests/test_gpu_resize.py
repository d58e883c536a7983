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
