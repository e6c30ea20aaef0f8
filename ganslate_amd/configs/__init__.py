from . import base, config, inference, training, validation_testing  # noqa: F401
from .omegalite import II, MISSING, DictConfig, OmegaConf  # noqa: F401
