"""Config boundary: keeps the reference's key names with a small PyYAML reader
instead of yacs.  get_cfg() + add_wsl_config() mirror uwsod/detectron2/config/defaults.py and
uwsod/projects/WSL/wsl/config/defaults.py:7-88 for the keys the OICR+ hot path reads; YAML files written
for the reference (e.g. configs/Detection/code_release/voc07_oicr_plus.yaml with its _BASE_ chain) load as is —
unknown keys are accepted and stored, never interpreted."""
import ast
import copy
import os

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], dict):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                if isinstance(v, str) and v.strip().startswith("(") and v.strip().endswith(")"):
                    try:                      # yacs-style "(a, b)" tuple literals in the reference's YAMLs
                        v = ast.literal_eval(v)
                    except (ValueError, SyntaxError):
                        pass
                self[k] = tuple(v) if isinstance(v, list) and isinstance(self.get(k), tuple) else v

    @staticmethod
    def _load_yaml_with_base(filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not base.startswith("/"):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode(CfgNode._load_yaml_with_base(base))
            merged._merge(cfg)
            return merged
        return cfg

    def merge_from_file(self, cfg_filename):
        self._merge(self._load_yaml_with_base(cfg_filename))

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0, "Override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            keys = full_key.split(".")
            for sub in keys[:-1]:
                d = d.setdefault(sub, CfgNode())
            if isinstance(v, str):
                try:
                    v = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            d[keys[-1]] = v

    def freeze(self):
        return self

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v) for k, v in n.items()}
        return yaml.safe_dump(plain(self))


CN = CfgNode


def get_cfg() -> CfgNode:
    """Defaults for the keys this path reads (values of uwsod/detectron2/config/defaults.py)."""
    C = CN()
    C.VERSION = 2
    C.SEED = -1
    C.VIS_PERIOD = 0
    C.MODEL = CN(dict(
        DEVICE="cuda", META_ARCHITECTURE="GeneralizedRCNN", WEIGHTS="", LOAD_PROPOSALS=False, MASK_ON=False,
        PIXEL_MEAN=[103.530, 116.280, 123.675], PIXEL_STD=[1.0, 1.0, 1.0],
        BACKBONE=dict(NAME="build_resnet_backbone", FREEZE_AT=2),
        PROPOSAL_GENERATOR=dict(NAME="RPN", MIN_SIZE=0),
        ROI_HEADS=dict(NAME="Res5ROIHeads", NUM_CLASSES=80, IN_FEATURES=["res4"], IOU_THRESHOLDS=[0.5], IOU_LABELS=[0, 1],
                       BATCH_SIZE_PER_IMAGE=512, POSITIVE_FRACTION=0.25, SCORE_THRESH_TEST=0.05, NMS_THRESH_TEST=0.5,
                       PROPOSAL_APPEND_GT=True),
        ROI_BOX_HEAD=dict(NAME="", BBOX_REG_LOSS_TYPE="smooth_l1", BBOX_REG_LOSS_WEIGHT=1.0,
                          BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0), SMOOTH_L1_BETA=0.0, POOLER_RESOLUTION=14,
                          POOLER_SAMPLING_RATIO=0, POOLER_TYPE="ROIAlignV2", NUM_FC=0, FC_DIM=1024, NUM_CONV=0,
                          CONV_DIM=256, NORM="", CLS_AGNOSTIC_BBOX_REG=False, TRAIN_ON_PRED_BOXES=False),
    ))
    C.INPUT = CN(dict(FORMAT="BGR", MIN_SIZE_TRAIN=(800,), MAX_SIZE_TRAIN=1333, MIN_SIZE_TRAIN_SAMPLING="choice", MIN_SIZE_TEST=800,
                      MAX_SIZE_TEST=1333, CROP=dict(ENABLED=False, TYPE="relative_range", SIZE=[0.9, 0.9])))
    C.DATASETS = CN(dict(TRAIN=(), TEST=(), PROPOSAL_FILES_TRAIN=(), PROPOSAL_FILES_TEST=(),
                         PRECOMPUTED_PROPOSAL_TOPK_TRAIN=2000, PRECOMPUTED_PROPOSAL_TOPK_TEST=1000))
    C.DATALOADER = CN(dict(NUM_WORKERS=4))
    C.SOLVER = CN(dict(LR_SCHEDULER_NAME="WarmupMultiStepLR", MAX_ITER=40000, BASE_LR=0.001, MOMENTUM=0.9, NESTEROV=False,
                       WEIGHT_DECAY=0.0001, WEIGHT_DECAY_NORM=0.0, GAMMA=0.1, STEPS=(30000,), WARMUP_FACTOR=1.0 / 1000,
                       WARMUP_ITERS=1000, WARMUP_METHOD="linear", CHECKPOINT_PERIOD=5000, IMS_PER_BATCH=16,
                       REFERENCE_WORLD_SIZE=0, BIAS_LR_FACTOR=1.0, WEIGHT_DECAY_BIAS=0.0001))
    C.TEST = CN(dict(EVAL_PERIOD=0, DETECTIONS_PER_IMAGE=100, AUG=dict(ENABLED=False)))
    C.OUTPUT_DIR = "./output"
    return C


def add_wsl_config(cfg: CfgNode):
    """uwsod/projects/WSL/wsl/config/defaults.py:7-88 — the WSL keys on this path — plus the
    MI355X-specific MODEL.AMD block (compute dtype of the HIP kernels)."""
    cfg.MODEL.VGG = CN(dict(DEPTH=16, OUT_FEATURES=["plain5"], CONV5_DILATION=1))
    cfg.MODEL.ROI_BOX_HEAD.DAN_DIM = [4096, 4096]
    cfg.WSL = CN(dict(VIS_TEST=False, ITER_SIZE=1, MEAN_LOSS=True, SIZE_EPOCH=5000, CMIL=False, USE_OBN=True,
                      CSC_MAX_ITER=35000, REFINE_NUM=3, REFINE_REG=[False, False, False], HAS_GAM=False,
                      REFINE_MIST=False, MIST_P=0.10, MIST_THRE=0.05, MIST_TYPE="nms", CLS_AGNOSTIC_BBOX_KNOWN=False))
    cfg.OICRPLUS = CN(dict(BBOX_UPDATE=False, PROPOSAL_NUM=100000))
    cfg.SOLVER.REFINE_LR_SCALE = 1.0
    cfg.SOLVER.REFINE_SCALE_ON = False
    cfg.SOLVER.AMP = False
    cfg.MODEL.AMD = CN(dict(COMPUTE_DTYPE="bf16"))     # "bf16" (MFMA bf16, fp32 accumulate) | "fp32" (exact f32 MFMA)
    return cfg
