"""Import surface for the reference files ``oracle/gen_golden.py`` loads by file path (``daod/loss/bpc_loss.py``,
``daod/modeling/roi_heads/source_free_fast_rcnn.py``, ``daod/engine/trainers/source_free_adaptive_teacher.py``): the
two per-image containers those functions touch, a few lines each, written from d2's documented behaviour
(TEST SCAFFOLDING, build container only -- no Detectron2 code)."""
import torch


class Boxes:
    """Nx4 xyxy container: ``tensor``, ``clip`` (x to [0, w], y to [0, h]), ``len``, indexing"""

    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clip(self, box_size):
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    @property
    def device(self):
        return self.tensor.device


class Instances:
    """per-image field container: attribute assignment = ``set`` (equal lengths asserted), ``len``, indexing"""

    def __init__(self, image_size, **kwargs):
        self.__dict__["_image_size"] = image_size
        self.__dict__["_fields"] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            self.__dict__[name] = val
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name, value):
        n = len(value)
        if len(self._fields):
            assert len(self) == n, "Adding a field of length {} to a Instances of length {}".format(n, len(self))
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __getitem__(self, item):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret


def pairwise_iou(a, b):
    raise NotImplementedError("imported by bpc_loss.py, not called")
