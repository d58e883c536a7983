"""Host-side data holders with the reference's interface (API shells only; SURVEY §2.1 "Structures"):
Boxes / pairwise-free box list (uwsod/detectron2/structures/boxes.py:96-327), Instances
(structures/instances.py:7-185), ImageList (structures/image_list.py:9-134), ShapeSpec
(layers/shape_spec.py).  Written for this path only: XYXY absolute boxes, no masks/keypoints/rotated."""
from collections import namedtuple
from typing import Any, Dict, List, Tuple

import torch


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, *, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


class Boxes:
    """(N,4) float tensor of XYXY absolute boxes."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]):
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w); self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w); self.tensor[:, 3].clamp_(min=0, max=h)

    def scale(self, scale_x: float, scale_y: float) -> None:
        """structures/boxes.py:296-302 (in place)"""
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Boxes with {} failed to return a matrix!".format(item)
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    @staticmethod
    def cat(boxes_list: List["Boxes"]):
        if len(boxes_list) == 0:
            return Boxes(torch.empty(0))
        return Boxes(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


class Instances:
    """Per-image bag of equally long fields (proposal_boxes, objectness_logits, gt_classes, ...)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name: str, val: Any):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name: str, value: Any):
        n = len(value)
        if len(self._fields):
            assert len(self) == n, "Adding a field of length {} to a Instances of length {}".format(n, len(self))
        self._fields[name] = value

    def has(self, name: str):
        return name in self._fields

    def remove(self, name: str):
        del self._fields[name]

    def get(self, name: str):
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args, **kwargs):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        if isinstance(item, int):
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists: List["Instances"]):
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        ret = Instances(instance_lists[0].image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = [x for v in values for x in v]
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret.set(k, values)
        return ret

    def __repr__(self):
        return "Instances(num_instances={}, image_size={}, fields=[{}])".format(
            len(self) if self._fields else 0, self._image_size, ", ".join(self._fields.keys()))


class ImageList:
    """Batch of images padded to a common size (no padding on this path: size_divisibility == 0)."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility: int = 0, pad_value: float = 0.0):
        assert len(tensors) > 0
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        mh, mw = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            mh = (mh + size_divisibility - 1) // size_divisibility * size_divisibility
            mw = (mw + size_divisibility - 1) // size_divisibility * size_divisibility
        if len(tensors) == 1 and sizes[0] == (mh, mw):
            return ImageList(tensors[0].unsqueeze(0), sizes)
        out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (mh, mw), pad_value)
        for t, o in zip(tensors, out):
            o[..., : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out.contiguous(), sizes)
