"""Checkpoint I/O with the reference's key names and file formats.

Mirrors `DetectionCheckpointer` (uwsod/detectron2/checkpoint/detection_checkpoint.py:11-75, on top of fvcore's
`Checkpointer`): a checkpoint is a dict {"model": state_dict, <checkpointable name>: state_dict ..., "iteration": int};
`.pth` files are `torch.save`d, `.pkl` files are pickles of numpy arrays in the Detectron2 model-zoo layout
({"model": {...}, "__author__": ..., "matching_heuristics": bool} — the ImageNet VGG16 the recipe starts from,
`MODEL.WEIGHTS: models/VGG/VGG_ILSVRC_16_layers_v1_d2.pkl`, is one) or bare Caffe2 blobs.

The model's parameter names and shapes are the reference's, so a reference checkpoint loads key for key.
Loading copies INTO the existing parameter storage: the predictor weights stay row slices of their flat master, and every
compute-dtype weight copy is invalidated through the parameter version / optimizer epoch (`ops.param_key`).
"""
import logging
import os
import pickle
from collections import namedtuple
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from . import ops

IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys", "incorrect_shapes"])


def _strip_prefix(sd: Dict[str, Any], prefix: str) -> Dict[str, Any]:
    if sd and all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return sd


def align_by_suffix(model_keys: List[str], ckpt: Dict[str, Any]) -> Dict[str, Any]:
    """The reference's name-matching heuristic (c2_model_loading.py `align_and_update_state_dicts`): every model key takes the
    checkpoint key that is its LONGEST suffix match (so a backbone-only file with keys `plain1.0.conv1.weight` fills
    `backbone.plain1.0.conv1.weight`); ambiguous ties are an error there and here."""
    out = {}
    ckeys = sorted(ckpt.keys())
    for mk in model_keys:
        best, best_len, tie = None, 0, False
        for ck in ckeys:
            if mk == ck or mk.endswith("." + ck):
                if len(ck) > best_len:
                    best, best_len, tie = ck, len(ck), False
                elif len(ck) == best_len:
                    tie = True
        if best is not None:
            if tie:
                raise ValueError(f"ambiguous checkpoint match for {mk}")
            out[mk] = ckpt[best]
    return out


class DetectionCheckpointer:
    def __init__(self, model: torch.nn.Module, save_dir: str = "", *, save_to_disk: Optional[bool] = None, **checkpointables):
        if isinstance(model, torch.nn.parallel.DistributedDataParallel):
            model = model.module
        self.model = model
        self.checkpointables = dict(checkpointables)          # e.g. optimizer=..., scheduler=...
        # the heads' dropout stream position (seed drawn from SEED + rank, element counter): a resumed run continues the
        # stream instead of replaying it.  Kept beside "model" so that the model's state dict has the reference's keys only.
        ds = getattr(getattr(model, "roi_heads", None), "dropout_stream", None)
        if ds is not None:
            self.checkpointables.setdefault("dropout_stream", ds)
        self.save_dir = save_dir
        if save_to_disk is None:
            import torch.distributed as dist
            save_to_disk = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        self.save_to_disk = save_to_disk
        self.logger = logging.getLogger(__name__)

    # ------------------------------------------------------------------ save (fvcore Checkpointer.save)
    def save(self, name: str, **kwargs: Any) -> Optional[str]:
        if not self.save_dir or not self.save_to_disk:
            return None
        data = {"model": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}}
        for key, obj in self.checkpointables.items():
            data[key] = obj.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, f"{name}.pth")
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:      # tag_last_checkpoint
            f.write(os.path.basename(path))
        return path

    # ------------------------------------------------------------------ load
    def has_checkpoint(self) -> bool:
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self) -> str:
        with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
            return os.path.join(self.save_dir, f.read().strip())

    def resume_or_load(self, path: str, *, resume: bool = True) -> Dict[str, Any]:
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])

    def load(self, path: str, checkpointables: Optional[List[str]] = None) -> Dict[str, Any]:
        if not path:
            return {}
        if not os.path.isfile(path):
            raise FileNotFoundError(f"Checkpoint {path} not found!")
        checkpoint = self._load_file(path)
        incompatible = self._load_model(checkpoint)
        self.last_incompatible = incompatible
        for key in (self.checkpointables if checkpointables is None else checkpointables):
            if key in checkpoint:
                self.checkpointables[key].load_state_dict(checkpoint.pop(key))
        checkpoint.pop("model", None)
        return checkpoint                                     # "iteration" and any user data

    def _load_file(self, filename: str) -> Dict[str, Any]:
        if filename.endswith(".pkl"):
            with open(filename, "rb") as f:
                data = pickle.load(f, encoding="latin1")
            if "model" in data and "__author__" in data:      # Detectron2 model-zoo format
                return data
            if "blobs" in data:                               # Caffe2 / Detectron1
                data = data["blobs"]
            data = {k: v for k, v in data.items() if not k.endswith("_momentum")}
            return {"model": data, "__author__": "Caffe2", "matching_heuristics": True}
        loaded = torch.load(filename, map_location="cpu", weights_only=False)
        if "model" not in loaded:
            loaded = {"model": loaded}
        return loaded

    def _load_model(self, checkpoint: Dict[str, Any]) -> IncompatibleKeys:
        sd = dict(checkpoint.pop("model"))
        for k, v in list(sd.items()):
            if isinstance(v, np.ndarray):
                sd[k] = torch.from_numpy(v)
            elif not isinstance(v, torch.Tensor):
                raise ValueError(f"Unsupported type found in checkpoint! {k}: {type(v)}")
        sd = _strip_prefix(sd, "module.")
        model_sd = self.model.state_dict()
        if checkpoint.get("matching_heuristics", False):
            sd = align_by_suffix(list(model_sd.keys()), sd)
        incorrect = []
        for k in list(sd.keys()):
            if k in model_sd and tuple(model_sd[k].shape) != tuple(sd[k].shape):
                incorrect.append((k, tuple(sd[k].shape), tuple(model_sd[k].shape)))
                sd.pop(k)
        res = self.model.load_state_dict(sd, strict=False)
        ops.invalidate_all_staged(); ops.BUFFER_EPOCH += 1      # every cached compute-dtype weight copy is stale now
        missing = [k for k in res.missing_keys if k not in ("pixel_mean", "pixel_std")]     # initialised from the config anyway
        if missing:
            self.logger.warning("missing keys: %s", missing)
        if res.unexpected_keys:
            self.logger.warning("unexpected keys: %s", list(res.unexpected_keys))
        return IncompatibleKeys(missing, list(res.unexpected_keys), incorrect)
