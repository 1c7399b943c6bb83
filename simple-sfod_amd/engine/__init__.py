from .trainer import (BaseTrainer, SourceFreeAdaptiveTeacherTrainer,  # noqa: F401
                      SourceFreeAdaptiveTeacherSingleTrainer, AdaptiveTeacherTrainer, adabn_refinement, test_refinement, get_trainer_class)
from .solver import FusedSGD, FlatModelState, WarmupMultiStepLR, build_optimizer  # noqa: F401
from . import planted  # noqa: F401
