"""Registration by import side effect, like ``train_net_mt.py:25-27`` of the reference."""
from . import backbone_vgg, backbone_resnet, rpn, roi_heads, meta_arch, dann  # noqa: F401
from .meta_arch import build_model  # noqa: F401
