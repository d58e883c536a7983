"""Precomputed-proposal file reader (the reference's `load_proposals_into_dataset`,
uwsod/detectron2/data/build.py:100-161).

The file is a pickle of {"ids" | "indexes": image ids, "boxes": [N_i x 4 arrays], "objectness_logits" | "scores": [N_i arrays],
"bbox_mode": optional BoxMode value (default XYXY_ABS)}.  Every dataset record gets `proposal_boxes`,
`proposal_objectness_logits` (sorted by descending score with the reference's own `argsort()[::-1]`) and `proposal_bbox_mode`
— the fields `mapper.DeviceMultiInputMapper` (and the reference's `DatasetMapperMultiInput`) consume.
"""
import logging
import pickle
from typing import Dict, List

XYXY_ABS, XYWH_ABS = 0, 1          # structures/boxes.py BoxMode values the proposal files use

_RENAME = {"indexes": "ids", "scores": "objectness_logits"}       # Detectron1 proposal files (build.py:125-129)


def read_proposal_file(proposal_file: str) -> Dict:
    with open(proposal_file, "rb") as f:
        proposals = pickle.load(f, encoding="latin1")
    for key, new in _RENAME.items():
        if key in proposals:
            proposals[new] = proposals.pop(key)
    return proposals


def load_proposals_into_dataset(dataset_dicts: List[dict], proposal_file: str) -> List[dict]:
    logging.getLogger(__name__).info("Loading proposals from: %s", proposal_file)
    proposals = read_proposal_file(proposal_file)
    img_ids = {str(record["image_id"]) for record in dataset_dicts}             # ids may be int or str: compared as str
    id_to_index = {str(i): k for k, i in enumerate(proposals["ids"]) if str(i) in img_ids}
    bbox_mode = int(proposals["bbox_mode"]) if "bbox_mode" in proposals else XYXY_ABS
    if bbox_mode not in (XYXY_ABS, XYWH_ABS):
        raise ValueError(f"unsupported proposal bbox_mode {bbox_mode}")
    for record in dataset_dicts:
        i = id_to_index[str(record["image_id"])]                                # KeyError for an image without proposals, as there
        boxes, logits = proposals["boxes"][i], proposals["objectness_logits"][i]
        inds = logits.argsort()[::-1]
        record["proposal_boxes"] = boxes[inds]
        record["proposal_objectness_logits"] = logits[inds]
        record["proposal_bbox_mode"] = bbox_mode
    return dataset_dicts
