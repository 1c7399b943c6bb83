from .synthetic import SyntheticTargetDataset, TwoCropLoader, TrainingSampler  # noqa: F401
