"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/soswsod_hip.h declares
(no compute calls without a GPU); the product path refuses to run without it."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "soswsod_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sw_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    import sos_wsod_amd._lib as L
    names = _declared_symbols()
    assert len(names) >= 25
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    assert set(L.SIGNATURES) == set(names)
    assert L.lib.sw_version().startswith(b"soswsod-hip")


def test_missing_extension_fails_loudly(tmp_path, monkeypatch):
    import sos_wsod_amd._lib as L
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        L.load()


def test_ops_refuse_cpu_tensors():
    import torch
    import sos_wsod_amd.ops as ops
    a = torch.zeros(8, 8)
    with pytest.raises(RuntimeError):
        ops.gemm(a, a, a, 8, 8, 8)


def test_config_reads_reference_style_yaml(tmp_path):
    from sos_wsod_amd.config import add_wsl_config, get_cfg
    base = tmp_path / "base.yaml"
    base.write_text("MODEL:\n  META_ARCHITECTURE: 'MultiInputRCNN'\n  ROI_BOX_HEAD:\n    NAME: 'DiscriminativeAdaptionNeck'\n    POOLER_RESOLUTION: 7\n")
    top = tmp_path / "top.yaml"
    top.write_text("_BASE_: 'base.yaml'\nMODEL:\n  BACKBONE:\n    NAME: 'build_vgg_backbone'\n    FREEZE_AT: 2\n  VGG:\n    CONV5_DILATION: 2\n"
                   "SOLVER:\n  STEPS: (35000, 50000)\nWSL:\n  REFINE_NUM: 4\n  REFINE_REG: [True, True, True, True]\n")
    cfg = add_wsl_config(get_cfg())
    cfg.merge_from_file(str(top))
    cfg.merge_from_list(["MODEL.AMD.COMPUTE_DTYPE", "fp32"])
    assert cfg.MODEL.META_ARCHITECTURE == "MultiInputRCNN" and cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION == 7
    assert cfg.SOLVER.STEPS == (35000, 50000) and cfg.WSL.REFINE_NUM == 4 and cfg.MODEL.VGG.CONV5_DILATION == 2
    assert cfg.MODEL.AMD.COMPUTE_DTYPE == "fp32"


def test_model_builds_from_cfg_with_reference_state_dict_names():
    """registry strings -> modules; parameter names/shapes = SURVEY A.3 (checkpoint contract)."""
    from sos_wsod_amd.config import add_wsl_config, get_cfg
    from sos_wsod_amd.rcnn_multi import build_model
    cfg = add_wsl_config(get_cfg())
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.META_ARCHITECTURE", "MultiInputRCNN", "MODEL.BACKBONE.NAME",
                         "build_vgg_backbone", "MODEL.VGG.CONV5_DILATION", 2, "MODEL.ROI_HEADS.NAME", "OICRPlusHeads",
                         "MODEL.ROI_HEADS.IN_FEATURES", ["plain5"], "MODEL.ROI_HEADS.NUM_CLASSES", 20,
                         "MODEL.ROI_HEADS.IOU_THRESHOLDS", [0.5, 0.6], "MODEL.ROI_HEADS.IOU_LABELS", [0, -1, 1],
                         "MODEL.ROI_BOX_HEAD.NAME", "DiscriminativeAdaptionNeck", "MODEL.ROI_BOX_HEAD.POOLER_TYPE", "ROIPool",
                         "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7, "MODEL.ROI_BOX_HEAD.DAN_DIM", [64, 64],
                         "MODEL.PROPOSAL_GENERATOR.NAME", "PrecomputedProposals", "WSL.REFINE_NUM", 4,
                         "WSL.REFINE_REG", [True] * 4, "WSL.REFINE_MIST", True])
    model = build_model(cfg)
    sd = model.state_dict()
    assert tuple(sd["backbone.plain1.0.conv1.weight"].shape) == (64, 3, 3, 3)
    assert tuple(sd["backbone.plain5.0.conv3.weight"].shape) == (512, 512, 3, 3)
    assert tuple(sd["roi_heads.box_head.fc1.weight"].shape) == (64, 25088)
    assert tuple(sd["roi_heads.box_predictor.det.weight"].shape) == (20, 64)
    assert tuple(sd["roi_heads.box_refinery_3.bbox_pred.weight"].shape) == (80, 64)
    assert tuple(sd["roi_heads.box_refinery_0.cls_score.bias"].shape) == (21,)
    assert tuple(sd["pixel_mean"].shape) == (3, 1, 1)
    frozen = [k for k, p in model.named_parameters() if not p.requires_grad]
    assert all(k.startswith(("backbone.plain1", "backbone.plain2")) for k in frozen) and len(frozen) == 8
    n = sum(p.numel() for p in model.parameters())
    assert n == 14714688 + 64 * 25088 + 64 + 64 * 64 + 64 + 2 * (20 * 64 + 20) + 4 * (21 * 64 + 21 + 80 * 64 + 80)
