"""Checkpoint I/O with the reference's key names and file formats — host logic, runs without a GPU."""
import pickle

import numpy as np
import torch


def _model():
    from sos_wsod_amd.config import add_wsl_config, get_cfg
    from sos_wsod_amd.rcnn_multi import build_model
    cfg = add_wsl_config(get_cfg())
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.META_ARCHITECTURE", "MultiInputRCNN", "MODEL.BACKBONE.NAME",
                         "build_vgg_backbone", "MODEL.VGG.CONV5_DILATION", 2, "MODEL.ROI_HEADS.NAME", "OICRPlusHeads",
                         "MODEL.ROI_HEADS.IN_FEATURES", ["plain5"], "MODEL.ROI_HEADS.NUM_CLASSES", 20,
                         "MODEL.ROI_HEADS.IOU_THRESHOLDS", [0.5, 0.6], "MODEL.ROI_HEADS.IOU_LABELS", [0, -1, 1],
                         "MODEL.ROI_BOX_HEAD.NAME", "DiscriminativeAdaptionNeck", "MODEL.ROI_BOX_HEAD.POOLER_TYPE", "ROIPool",
                         "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", 7, "MODEL.ROI_BOX_HEAD.DAN_DIM", [32, 32],
                         "MODEL.PROPOSAL_GENERATOR.NAME", "PrecomputedProposals", "WSL.REFINE_NUM", 4,
                         "WSL.REFINE_REG", [True] * 4, "WSL.REFINE_MIST", True])
    return build_model(cfg)


def test_pth_roundtrip_with_optimizer_scheduler_and_iteration(tmp_path):
    import sos_wsod_amd.ops as ops
    from sos_wsod_amd.checkpoint import DetectionCheckpointer
    from sos_wsod_amd.solver import WarmupMultiStepLR
    torch.manual_seed(0)
    m1 = _model()
    opt1 = torch.optim.SGD(m1.parameters(), lr=0.01, momentum=0.9)
    sch1 = WarmupMultiStepLR(opt1, [5, 8], warmup_iters=3)
    for _ in range(4):
        opt1.step(); sch1.step()
    ck = DetectionCheckpointer(m1, str(tmp_path), optimizer=opt1, scheduler=sch1)
    path = ck.save("model_0000003", iteration=3)
    assert path.endswith("model_0000003.pth") and ck.has_checkpoint() and ck.get_checkpoint_file() == path
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {"model", "optimizer", "scheduler", "iteration", "dropout_stream"}
    assert "backbone.plain3.0.conv2.weight" in raw["model"] and "roi_heads.box_refinery_2.bbox_pred.bias" in raw["model"]
    torch.manual_seed(1)
    m2 = _model()
    opt2 = torch.optim.SGD(m2.parameters(), lr=0.5, momentum=0.9)
    sch2 = WarmupMultiStepLR(opt2, [5, 8], warmup_iters=3)
    epoch = ops.PARAM_EPOCH
    extra = DetectionCheckpointer(m2, str(tmp_path), optimizer=opt2, scheduler=sch2).resume_or_load("", resume=True)
    assert extra == {"iteration": 3} and ops.PARAM_EPOCH > epoch          # cached weight copies are invalidated
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert sch2.last_epoch == sch1.last_epoch and opt2.param_groups[0]["lr"] == opt1.param_groups[0]["lr"]


def test_model_zoo_pkl_with_name_matching_fills_the_backbone_only(tmp_path):
    """the recipe's MODEL.WEIGHTS is a Detectron2-zoo pickle of numpy arrays holding ImageNet VGG16 convs only"""
    from sos_wsod_amd.checkpoint import DetectionCheckpointer
    torch.manual_seed(2)
    m = _model()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(0)
    zoo = {}
    for k, v in before.items():
        if k.startswith("backbone."):
            zoo[k[len("backbone."):]] = rng.standard_normal(tuple(v.shape)).astype(np.float32)      # keys without the prefix
    zoo["fc8.weight"] = np.zeros((1000, 4096), np.float32)                                          # a head the model lacks
    p = tmp_path / "VGG_ILSVRC_16_layers_v1_d2.pkl"
    with open(p, "wb") as f:
        pickle.dump({"model": zoo, "__author__": "someone", "matching_heuristics": True}, f)
    ck = DetectionCheckpointer(m)
    ck.resume_or_load(str(p), resume=False)
    after = m.state_dict()
    for k in before:
        if k.startswith("backbone."):
            assert np.array_equal(after[k].numpy(), zoo[k[len("backbone."):]]), k
        elif k not in ("pixel_mean", "pixel_std"):
            assert torch.equal(after[k], before[k]), k                                             # heads untouched
    assert any(k.startswith("roi_heads.") for k in ck.last_incompatible.missing_keys)
    assert not ck.last_incompatible.unexpected_keys                    # unmatched zoo entries are dropped by the heuristic


def test_shape_mismatch_is_reported_not_loaded(tmp_path):
    from sos_wsod_amd.checkpoint import DetectionCheckpointer
    m = _model()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["roi_heads.box_predictor.cls.weight"] = torch.zeros(81, 32)     # a COCO head into a VOC model
    torch.save({"model": sd}, tmp_path / "x.pth")
    keep = m.state_dict()["roi_heads.box_predictor.cls.weight"].clone()
    ck = DetectionCheckpointer(m)
    ck.load(str(tmp_path / "x.pth"))
    assert ck.last_incompatible.incorrect_shapes == [("roi_heads.box_predictor.cls.weight", (81, 32), (20, 32))]
    assert torch.equal(m.state_dict()["roi_heads.box_predictor.cls.weight"], keep)


def test_detection_wire_format_and_postprocess(tmp_path):
    """§8f row 2: `image_id score x1+1 y1+1 x2 y2` lines and the JSON records of pascal_voc_evaluation.py:57-118, after the
    rescale / clip / drop-empty of detector_postprocess"""
    import json
    from sos_wsod_amd.inference import VOCDetectionWriter, detector_postprocess
    from sos_wsod_amd.structures import Boxes, Instances
    r = Instances((100, 200))
    r.pred_boxes = Boxes(torch.tensor([[10.0, 20.0, 50.0, 60.0], [190.0, 90.0, 260.0, 140.0], [5.0, 5.0, 5.0, 9.0]]))
    r.scores = torch.tensor([0.98765, 0.5, 0.25])
    r.pred_classes = torch.tensor([3, 0, 3])
    out = detector_postprocess(r, 200, 400)                       # network saw 100x200, the dataset image is 200x400
    assert out.image_size == (200, 400) and len(out) == 2        # the zero-width box is dropped
    assert torch.equal(out.pred_boxes.tensor, torch.tensor([[20.0, 40.0, 100.0, 120.0], [380.0, 180.0, 400.0, 200.0]]))
    w = VOCDetectionWriter(20)
    w.process([{"image_id": "000012"}], [{"instances": out}])
    assert w.lines()[3] == ["000012 0.988 21.0 41.0 100.0 120.0"] and w.lines()[0] == ["000012 0.500 381.0 181.0 400.0 200.0"]
    w.dump(tmp_path / "det.json")
    assert json.load(open(tmp_path / "det.json")) == [
        {"image_id": 12, "category_id": 1, "score": 0.5, "bbox": [381.0, 181.0, 400.0, 200.0]},
        {"image_id": 12, "category_id": 4, "score": 0.988, "bbox": [21.0, 41.0, 100.0, 120.0]}]


def _resume_worker(rank, world, port, ckdir, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sos_wsod_amd.checkpoint import DetectionCheckpointer
    m = _model()
    m.roi_heads.seed = 1234
    first = m.roi_heads._dropout_stream_seed()
    m.roi_heads._drop_counter = 777 + rank                         # ranks at different stream positions
    if rank == 0:                                                  # the reference saves on rank 0 only
        DetectionCheckpointer(m, ckdir).save("model_0000009", iteration=9)
    dist.barrier()
    m2 = _model()
    DetectionCheckpointer(m2, ckdir).resume_or_load("", resume=True)
    torch.save({"first": first, "resumed": m2.roi_heads._dropout_stream_seed(), "counter": m2.roi_heads._drop_counter},
               f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_resume_keeps_one_dropout_stream_per_rank(tmp_path):
    """the checkpoint file is written by rank 0; after a resume every rank must again draw its OWN masks (seed derived from the
    rank-independent base seed in the file and the rank), not rank 0's"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    mp.spawn(_resume_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["first"] != r1["first"]
    assert r0["resumed"] == r0["first"] and r1["resumed"] == r1["first"]     # each rank continues with its own seed
    assert r0["counter"] == 777 and r1["counter"] == 777                      # the position saved by rank 0
