"""Generate tests/golden/resize_*.npz from Pillow itself (installed in the build container; the GPU box only reads the
fixtures): random uint8 images and `PIL.Image.fromarray(img).resize((w, h), Image.BILINEAR)` — the call behind the reference's
ResizeTransform.apply_image (dataset_mapper.py:303-352, test_time_augmentation_avg.py:199-310).  Data only.

    python tests/golden/make_resize_golden.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
# (name, in_h, in_w, out_h, out_w): up- and down-scaling, one-axis-only changes, non-integer ratios, a ResizeShortestEdge pair
CASES = [("up", 37, 53, 48, 69), ("down", 90, 120, 33, 44), ("wonly", 40, 64, 40, 97), ("honly", 50, 40, 83, 40),
         ("mixed", 61, 47, 29, 101), ("voc", 94, 125, 120, 160), ("big_down", 128, 96, 17, 13)]


def main():
    rng = np.random.RandomState(7)
    for name, h, w, oh, ow in CASES:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        if name == "voc":                                   # smooth content too (gradients + edges), not only noise
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 2 + xx) % 256, (xx * 3) % 256, ((yy // 8 + xx // 8) % 2) * 255], -1).astype(np.uint8)
        out = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        np.savez_compressed(os.path.join(HERE, f"resize_{name}.npz"), image_hwc=img, out_hw=np.asarray([oh, ow]), resized_hwc=out)
        print(name, img.shape, "->", out.shape)


if __name__ == "__main__":
    main()
