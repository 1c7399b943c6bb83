"""String -> class/function registries with the names the reference's yamls use.

Mirrors Detectron2's registry surface as used by the reference (registration by decorator at
import time, ``train_net_mt.py:25-27``):
  META_ARCH_REGISTRY           source_free_adaptive_teacher_rcnn.py:24
  BACKBONE_REGISTRY            vgg.py:116,121
  PROPOSAL_GENERATOR_REGISTRY  rpn.py:10
  ROI_HEADS_REGISTRY           source_free_adaptive_teacher_roi_heads.py:25
  ROI_BOX_HEAD_REGISTRY        box_head.py:13
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, \
            "An object named '{}' was already registered in '{}' registry!".format(name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")
