"""Root config (ganslate/configs/config.py:10-19)."""
from dataclasses import dataclass, field
from typing import Any, Optional

from .inference import InferenceConfig
from .training import TrainConfig
from .validation_testing import TestConfig, ValidationConfig


@dataclass
class Config:
    project: Optional[Any] = None
    mode: str = "train"
    train: TrainConfig = field(default_factory=TrainConfig)
    val: Optional[ValidationConfig] = None
    test: Optional[TestConfig] = None
    infer: Optional[InferenceConfig] = None
