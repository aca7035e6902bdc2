from .configs import make_config
from .trainer import TrainStep, synthetic_batch

__all__ = ["make_config", "TrainStep", "synthetic_batch"]
