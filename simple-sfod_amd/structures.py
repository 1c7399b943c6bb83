"""Minimal mutable ``Boxes`` / ``Instances`` / ``ImageList`` containers.

The reference passes Detectron2 structures across the L4->L3 boundary and mutates them in
place (``add_label`` source_free_adaptive_teacher.py:319-322; proposal overwrite
source_free_adaptive_teacher_roi_heads.py:143).  These are from-scratch containers with the
same surface the hot path uses: ``inst[mask]``, ``len``, ``has/set/get/get_fields``,
``Instances.cat``, ``Boxes.clip/nonempty/area``, ``ImageList.from_tensors``.
They hold torch tensors (device memory plumbing only); no arithmetic of the hot path lives here
except trivial host-side clip/area helpers used by tests and the data layer.
"""
import itertools
from typing import Any, Dict, List, Tuple

import torch


class Boxes:
    """N x 4 (x1, y1, x2, y2) fp32 boxes."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor.to(torch.float32)

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

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

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0, 4))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def __iter__(self):
        yield from self.tensor


class Instances:
    """Per-image bag of equally long fields (``gt_boxes``, ``gt_classes``, ``scores`` ...)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name, value):
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, \
                "Adding a field of length {} to a Instances of length {}".format(data_len, len(self))
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
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
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret.set(k, values)
        return ret

    def __repr__(self):
        s = self.__class__.__name__ + "("
        s += "num_instances={}, image_height={}, image_width={}, fields=[{}])".format(
            len(self) if len(self._fields) else 0, self._image_size[0], self._image_size[1],
            ", ".join(f"{k}: {v}" for k, v in self._fields.items()))
        return s


class ImageList:
    """Batch of images padded bottom/right to a common size + the un-padded sizes (A.1)."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility: int = 0, pad_value: float = 0.0):
        sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
        hm = max(s[0] for s in sizes)
        wm = max(s[1] for s in sizes)
        if size_divisibility > 1:
            hm = (hm + size_divisibility - 1) // size_divisibility * size_divisibility
            wm = (wm + size_divisibility - 1) // size_divisibility * size_divisibility
        out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (hm, wm), pad_value)
        for i, t in enumerate(tensors):
            out[i, ..., : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out, sizes)


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride

    def __repr__(self):
        return f"ShapeSpec(channels={self.channels}, height={self.height}, width={self.width}, stride={self.stride})"
