"""Multi-view training mapper, box side: the numpy restatement against the golden masks produced by the
reference's own `Boxes` class, and the package's tensor implementation against both.  Host logic — runs without a GPU
(the same checks run on the device in tests/test_gpu_e2e.py::test_multi_input_mapper_device)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "input_a.npz")
VIEWS = (("1", "hw1", False), ("2", "hw2", False), ("1_flip", "hw1", True), ("2_flip", "hw2", True))


def test_oracle_matches_reference_masks():
    from oracle import input_oracle as IO
    g = np.load(GOLD)
    ohw = tuple(g["orig_hw"])
    assert IO.shortest_edge_shape(*ohw, 480, 2000) == tuple(g["hw1"])
    joint = None
    for name, hwk, flip in VIEWS:
        b, _, keep = IO.transform_proposals_multi(g["boxes"], g["logits"], ohw, tuple(g[hwk]), flip, topk=10 ** 6)
        assert np.array_equal(b, g["boxes" + name])
        assert np.array_equal(keep, g["keep" + name])
        joint = keep if joint is None else joint & keep
    assert np.array_equal(joint, g["keep"])
    assert 0 < joint.sum() < len(joint)                         # the case does exercise the masks


def check_mapper_against_golden(device):
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    g = np.load(GOLD)
    h, w = (int(v) for v in g["orig_hw"])
    img = torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(device)
    d = {"image": img, "proposal_boxes": g["boxes"], "proposal_objectness_logits": g["logits"], "image_id": 7,
         "annotations": [{"bbox": [10.5, 20.0, 200.25, 300.0], "category_id": 3},
                         {"bbox": [0.0, 0.0, 499.0, 374.0], "category_id": 11, "iscrowd": 0},
                         {"bbox": [5.0, 5.0, 50.0, 50.0], "category_id": 1, "iscrowd": 1}]}
    m = DeviceMultiInputMapper(proposal_topk=2000, resize_pixels=device != "cpu")
    out = m(d, shapes=(tuple(int(v) for v in g["hw1"]), tuple(int(v) for v in g["hw2"])))
    keep = g["keep"]
    for name, hwk, flip in VIEWS:
        p = out["proposals" + name]
        assert tuple(p.image_size) == tuple(g[hwk])
        assert p.proposal_boxes.tensor.device.type == torch.device(device).type
        assert np.array_equal(p.proposal_boxes.tensor.cpu().numpy(), g["boxes" + name][keep])       # bit-exact float32
        assert np.array_equal(p.objectness_logits.cpu().numpy(), g["logits"][keep])
        im = out["image" + name]
        assert im.dtype == torch.uint8 and tuple(im.shape) == (3, *g[hwk])
        inst = out["instances" + name]
        assert inst.gt_classes.tolist() == [3, 11]                                                   # crowd dropped
    assert torch.equal(out["image1_flip"], out["image1"].flip(-1)) and torch.equal(out["image2_flip"], out["image2"].flip(-1))
    assert out["image_id"] == 7 and "proposal_boxes" not in out and "annotations" not in out
    # annotation boxes against the float64 restatement
    from oracle import input_oracle as IO
    gt = np.array([[10.5, 20.0, 200.25, 300.0], [0.0, 0.0, 499.0, 374.0]])
    for name, hwk, flip in VIEWS:
        want = IO.transform_annotation_boxes(gt, (h, w), tuple(g[hwk]), flip)
        assert np.array_equal(out["instances" + name].gt_boxes.tensor.cpu().numpy(), want)
    # flipped view: x-mirror of the plain view, row for row
    b1, b1f = out["proposals1"].proposal_boxes.tensor, out["proposals1_flip"].proposal_boxes.tensor
    W1 = float(g["hw1"][1])
    assert torch.equal(b1f[:, 0], W1 - b1[:, 2]) and torch.equal(b1f[:, 2], W1 - b1[:, 0]) and torch.equal(b1f[:, 1], b1[:, 1])


def test_mapper_matches_reference_masks_cpu():
    check_mapper_against_golden("cpu")


def test_unique_mask_against_oracle_random():
    from oracle import input_oracle as IO
    from sos_wsod_amd.mapper import unique_boxes_mask
    rng = np.random.RandomState(3)
    for n in (0, 1, 5, 1000):
        b = (rng.randint(0, 40, (n, 4)) + rng.choice([0.0, 0.5, 0.49, 0.51], (n, 4))).astype(np.float32)
        assert np.array_equal(unique_boxes_mask(torch.from_numpy(b)).numpy(), IO.unique_mask(b) if n else np.zeros(0, bool))


def test_scale_draw_rules():
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    m = DeviceMultiInputMapper(seed=0)
    seen = set()
    for _ in range(200):
        hw1, hw2 = m._draw_shapes(375, 500)
        assert hw1 != hw2 and min(hw1) in m.min_sizes and min(hw2) in m.min_sizes
        seen.add((min(hw1), min(hw2)))
    assert len(seen) > 20                                       # both draws cover the size list
    # first view honours max_size, the second is uncapped like the reference's rebuilt ResizeShortestEdge
    m2 = DeviceMultiInputMapper(min_sizes=(800, 1200), max_size=1000, seed=1)
    hw1, hw2 = m2._draw_shapes(300, 900)
    assert max(hw1) <= 1000 and max(hw2) > 1000
    with pytest.raises(AssertionError):
        DeviceMultiInputMapper(min_sizes=(480,))


def test_topk_slice_keeps_alignment():
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    g = np.load(GOLD)
    h, w = (int(v) for v in g["orig_hw"])
    d = {"image": torch.zeros(3, h, w, dtype=torch.uint8), "proposal_boxes": g["boxes"],
         "proposal_objectness_logits": g["logits"]}
    out = DeviceMultiInputMapper(proposal_topk=100, resize_pixels=False)(d, shapes=((480, 640), (1200, 1600)))
    n = int(g["keep"][:100].sum())
    for name, _, _ in VIEWS:
        assert len(out["proposals" + name].proposal_boxes) == n
        assert np.array_equal(out["proposals" + name].objectness_logits.numpy(), g["logits"][:100][g["keep"][:100]])


def test_proposal_file_reader(tmp_path):
    """`load_proposals_into_dataset` (data/build.py:100-161): both key spellings, ids compared as strings, descending score
    order, bbox_mode default / XYWH carried to the mapper"""
    import pickle
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    from sos_wsod_amd.proposals import load_proposals_into_dataset
    rng = np.random.RandomState(0)
    ids = [7, "000012", 99]
    boxes = [np.concatenate([rng.rand(n, 2) * 100, 20 + rng.rand(n, 2) * 100], 1).astype(np.float32) for n in (5, 8, 3)]
    scores = [rng.rand(n).astype(np.float32) for n in (5, 8, 3)]
    f1, f2 = str(tmp_path / "d2.pkl"), str(tmp_path / "d1.pkl")
    pickle.dump({"ids": ids, "boxes": boxes, "objectness_logits": scores}, open(f1, "wb"))
    pickle.dump({"indexes": ids, "boxes": boxes, "scores": scores, "bbox_mode": 1}, open(f2, "wb"))
    for path, mode in ((f1, 0), (f2, 1)):
        ds = load_proposals_into_dataset([{"image_id": "7"}, {"image_id": "000012"}], path)
        for rec, k in zip(ds, (0, 1)):
            order = scores[k].argsort()[::-1]
            assert np.array_equal(rec["proposal_boxes"], boxes[k][order])
            assert np.array_equal(rec["proposal_objectness_logits"], scores[k][order])
            assert (np.diff(rec["proposal_objectness_logits"]) <= 0).all() and rec["proposal_bbox_mode"] == mode
    with pytest.raises(KeyError):
        load_proposals_into_dataset([{"image_id": 1234}], f1)
    # XYWH proposals reach the views as XYXY
    rec = dict(ds[0], image=torch.zeros(3, 200, 300, dtype=torch.uint8))
    out = DeviceMultiInputMapper(proposal_topk=100, resize_pixels=False)(rec, shapes=((200, 300), (400, 600)))
    b = rec["proposal_boxes"]
    want = np.concatenate([b[:, :2], b[:, :2] + b[:, 2:]], 1)
    want[:, 0::2] = want[:, 0::2].clip(0, 300); want[:, 1::2] = want[:, 1::2].clip(0, 200)
    got = out["proposals1"].proposal_boxes.tensor.numpy()
    assert got.shape[0] <= len(b) and all(any(np.allclose(g, w_) for w_ in want) for g in got)


# ---- the whole mapper (RandomCrop, two scales, brightness / saturation blends, flips) against fixtures written by RUNNING the
# reference's DatasetMapperMultiInput.__call__ (tests/golden/make_mapper_golden.py)
MAPPER_CASES = ("a", "v", "m")
VIEW_KEYS = ("1", "2", "1_flip", "2_flip")


def mapper_fixture(case):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", f"mapper_{case}.npz"))


def mapper_for_fixture(g, resize_pixels):
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    return DeviceMultiInputMapper(min_sizes=tuple(int(s) for s in g["min_sizes"]), max_size=int(g["max_size"]),
                                  proposal_topk=int(g["topk"]), seed=int(g["seed"]), resize_pixels=resize_pixels,
                                  crop=("relative_range", [0.9, 0.9]), brightness=(1.0 / 1.5, 1.5), saturation=(1.0 / 1.5, 1.5))


def fixture_dataset_dict(g, device):
    annos = [{"bbox": [float(v) for v in b], "category_id": int(c), "iscrowd": int(k)}
             for b, c, k in zip(g["anno_boxes"], g["anno_classes"], g["anno_crowd"])]
    return {"image": torch.from_numpy(g["image"]).to(device), "proposal_boxes": g["boxes"], "proposal_objectness_logits": g["logits"],
            "annotations": annos, "image_id": 17}


def check_mapper_boxes_against_reference_run(case, device):
    """same numpy seed as the reference run -> the same crop window, view shapes and blend weights (to the bit), the same four
    proposal sets (float32, bit exact) and annotation sets; returns (fixture, mapper output) for the pixel checks"""
    g = mapper_fixture(case)
    m = mapper_for_fixture(g, resize_pixels=device != "cpu")
    out = m(fixture_dataset_dict(g, device))
    dr = m.last_draws
    assert tuple(dr["crop"]) == tuple(int(v) for v in g["crop"])
    assert tuple(dr["hw1"]) == tuple(g["hw1"]) and tuple(dr["hw2"]) == tuple(g["hw2"])
    assert tuple(dr["blend1"]) == tuple(g["blend1"]) and tuple(dr["blend2"]) == tuple(g["blend2"])      # float64, exact
    assert out["height"] == g["image"].shape[1] and out["width"] == g["image"].shape[2]                  # the uncropped size
    for key in VIEW_KEYS:
        p = out["proposals" + key]
        assert tuple(p.image_size) == tuple(g["psize_" + key]) == tuple(g["hw_" + key])
        assert np.array_equal(p.proposal_boxes.tensor.cpu().numpy(), g["pboxes_" + key])
        assert np.array_equal(p.objectness_logits.cpu().numpy(), g["plogits_" + key])
        inst = out["instances" + key]
        assert np.array_equal(inst.gt_boxes.tensor.cpu().numpy(), g["gboxes_" + key])
        assert np.array_equal(inst.gt_classes.cpu().numpy(), g["gclasses_" + key])
        assert tuple(out["image" + key].shape[1:]) == tuple(g["hw_" + key])
    assert 0 < len(g["plogits_1"]) < len(g["logits"])           # the crop + the masks did drop proposals
    return g, out


@pytest.mark.parametrize("case", MAPPER_CASES)
def test_mapper_matches_the_reference_run_cpu(case):
    check_mapper_boxes_against_reference_run(case, "cpu")


@pytest.mark.parametrize("case", MAPPER_CASES)
def test_input_oracle_matches_the_reference_run(case):
    """the numpy restatement (draw order, crop + resize + flip box maps, masks, blends + PIL resize -> pixels) against the same
    fixtures; this is the checker the GPU tests use beyond the fixtures' sizes"""
    import zlib
    from PIL import Image
    from oracle import input_oracle as IO
    g = mapper_fixture(case)
    h, w = g["image"].shape[1:]
    dr = IO.draw_views(int(g["seed"]), h, w, [int(s) for s in g["min_sizes"]], int(g["max_size"]))
    assert dr["crop"] == tuple(int(v) for v in g["crop"]) and dr["hw1"] == tuple(g["hw1"]) and dr["hw2"] == tuple(g["hw2"])
    assert dr["blend1"] == tuple(g["blend1"]) and dr["blend2"] == tuple(g["blend2"]) and dr["tries2"] == int(g["n_tries2"])
    y0, x0, ch, cw = dr["crop"]
    res, keep = IO.multi_input_proposals(g["boxes"], g["logits"], (ch, cw), dr["hw1"], dr["hw2"], int(g["topk"]), crop_xy=(x0, y0))
    live = g["anno_crowd"] == 0
    for key, hw, flip, blend in (("1", dr["hw1"], False, dr["blend1"]), ("2", dr["hw2"], False, dr["blend2"]),
                                 ("1_flip", dr["hw1"], True, dr["blend1"]), ("2_flip", dr["hw2"], True, dr["blend2"])):
        b, l = res["proposals" + key]
        assert np.array_equal(b, g["pboxes_" + key]) and np.array_equal(l, g["plogits_" + key])
        want = IO.transform_annotation_boxes(g["anno_boxes"][live], (ch, cw), hw, flip, crop_xy=(x0, y0))
        assert np.array_equal(want, g["gboxes_" + key])
        win = np.ascontiguousarray(g["image"].transpose(1, 2, 0)[y0:y0 + ch, x0:x0 + cw])
        px = IO.blend_u8(np.asarray(Image.fromarray(win).resize((hw[1], hw[0]), Image.BILINEAR)), *blend)
        px = np.ascontiguousarray((px[:, ::-1] if flip else px).transpose(2, 0, 1))
        assert zlib.crc32(px.tobytes()) == int(g["crc_" + key])
        if "image" + key in g:
            assert np.array_equal(px, g["image" + key])


def test_mapper_from_config_builds_the_reference_recipe():
    """voc07_oicr_plus.yaml:29-35: INPUT.CROP.ENABLED True with the defaults relative_range [0.9, 0.9] (config/defaults.py:63-74)"""
    from sos_wsod_amd.config import get_cfg, add_wsl_config
    from sos_wsod_amd.mapper import DeviceMultiInputMapper
    cfg = add_wsl_config(get_cfg())
    cfg.merge_from_list(["INPUT.CROP.ENABLED", "True", "INPUT.MIN_SIZE_TRAIN", "(480, 512, 544)", "INPUT.MAX_SIZE_TRAIN", "2000",
                         "MODEL.LOAD_PROPOSALS", "True", "DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN", "4000"])
    m = DeviceMultiInputMapper.from_config(cfg, seed=0, resize_pixels=False)
    assert m.crop == ("relative_range", (0.9, 0.9)) and m.min_sizes == (480, 512, 544) and m.max_size == 2000
    assert m.proposal_topk == 4000 and m.brightness == (1.0 / 1.5, 1.5) and m.saturation == (1.0 / 1.5, 1.5)
    cfg.INPUT.CROP.ENABLED = False
    assert DeviceMultiInputMapper.from_config(cfg, resize_pixels=False).crop is None
    # the other crop rules of RandomCrop.get_crop_size
    from sos_wsod_amd.mapper import crop_size_rule
    rng = np.random.RandomState(0)
    assert crop_size_rule("relative", (0.5, 0.25), 100, 200, rng) == (50, 50)
    assert crop_size_rule("absolute", (64, 300), 100, 200, rng) == (64, 200)
    ch, cw = crop_size_rule("absolute_range", (50, 150), 100, 200, rng)
    assert 50 <= ch <= 100 and 50 <= cw <= 150
    for _ in range(50):
        ch, cw = crop_size_rule("relative_range", (0.9, 0.9), 375, 500, rng)
        assert int(375 * 0.9) <= ch <= 375 and int(500 * 0.9) <= cw <= 500
