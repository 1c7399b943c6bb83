from detectron2.structures import Instances  # noqa: F401
