"""CPU: the Pillow-bilinear restatement (oracle/resize_oracle.py) against the fixtures generated from Pillow
(tests/golden/resize_*.npz) and, where Pillow is installed, against Pillow itself over random sizes; the product's host-side
coefficient tables (sos_wsod_amd.resize._coeffs) against the oracle's."""
import glob
import os

import numpy as np
import pytest

from oracle.resize_oracle import bilinear_coeffs, resize_bilinear_u8


def test_oracle_equals_pillow_fixtures(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "resize_*.npz")))
    assert len(files) >= 6
    for f in files:
        g = np.load(f)
        oh, ow = (int(v) for v in g["out_hw"])
        got = resize_bilinear_u8(np.ascontiguousarray(g["image_hwc"].transpose(2, 0, 1)), oh, ow).transpose(1, 2, 0)
        assert np.array_equal(got, g["resized_hwc"]), os.path.basename(f)


def test_oracle_equals_pillow_on_random_sizes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(11)
    for t in range(25):
        h, w = rng.randint(4, 80, 2)
        oh, ow = rng.randint(2, 120, 2)
        oh = h if t % 6 == 0 else oh
        ow = w if t % 7 == 0 else ow
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((int(ow), int(oh)), Image.BILINEAR))
        got = resize_bilinear_u8(np.ascontiguousarray(img.transpose(2, 0, 1)), int(oh), int(ow)).transpose(1, 2, 0)
        assert np.array_equal(got, ref), (h, w, oh, ow)


def test_product_coefficient_tables_equal_the_oracle():
    from sos_wsod_amd.resize import _coeffs
    for n_in, n_out in [(375, 480), (500, 640), (1200, 700), (64, 65), (10, 3), (333, 1000), (7, 7 * 5)]:
        b0, k0 = _coeffs(n_in, n_out)
        b1, k1 = bilinear_coeffs(n_in, n_out)
        assert np.array_equal(b0, b1) and np.array_equal(k0, k1), (n_in, n_out)
        assert ((k0.sum(1) - (1 << 22)).__abs__() <= k0.shape[1]).all()         # weights sum to one up to rounding
