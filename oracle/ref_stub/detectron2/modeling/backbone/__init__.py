import torch.nn as nn


class Backbone(nn.Module):
    """Minimal stand-in for detectron2.modeling.backbone.Backbone (an nn.Module)."""


class _Registry:
    def __init__(self):
        self._m = {}

    def register(self, obj=None):
        def deco(fn):
            self._m[fn.__name__] = fn
            return fn
        return deco if obj is None else deco(obj)

    def get(self, name):
        return self._m[name]


BACKBONE_REGISTRY = _Registry()
