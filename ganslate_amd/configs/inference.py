"""ganslate/configs/inference.py restated (parse-only)."""
from dataclasses import dataclass, field
from typing import Optional

from . import base, validation_testing


@dataclass
class InferenceConfig(base.BaseEngineConfig):
    is_deployment: bool = False
    dataset: Optional[base.BaseDatasetConfig] = None
    sliding_window: Optional[validation_testing.SlidingWindowConfig] = None
    checkpointing: base.CheckpointingConfig = field(default_factory=base.CheckpointingConfig)
