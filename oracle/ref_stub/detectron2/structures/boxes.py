from detectron2.structures import Boxes, pairwise_iou  # noqa: F401
