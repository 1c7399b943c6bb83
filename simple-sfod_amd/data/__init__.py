from .synthetic import (SyntheticTargetDataset, TwoCropLoader, TrainingSampler, InferenceSampler,  # noqa: F401
                        TestLoader, CITYSCAPES_CLASSES)
from .augment import StrongAugmentation  # noqa: F401
