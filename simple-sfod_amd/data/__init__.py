from .synthetic import (SyntheticTargetDataset, TwoCropLoader, TrainingSampler, InferenceSampler,  # noqa: F401
                        TestLoader, CITYSCAPES_CLASSES)
from .augment import StrongAugmentation  # noqa: F401
from .coco import (CocoTargetDataset, load_coco_json, register_coco_instances, register_all_datasets,  # noqa: F401
                   register_datasets, build_dataset)
