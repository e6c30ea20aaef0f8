"""ganslate/configs/training.py:7-47 restated (same names/defaults)."""
from dataclasses import dataclass, field
from typing import Optional

from . import base
from .omegalite import MISSING


@dataclass
class TrainMetricsConfig:
    discriminator_evolution: bool = False
    ssim: bool = False


@dataclass
class TrainCheckpointingConfig(base.CheckpointingConfig):
    freq: int = 2000
    start_after: int = 0
    load_optimizers: bool = True
    load_iter: Optional[int] = None


@dataclass
class TrainConfig(base.BaseEngineConfig):
    output_dir: str = MISSING
    batch_size: int = MISSING
    cuda: bool = True
    mixed_precision: bool = False
    opt_level: str = "O1"
    checkpointing: TrainCheckpointingConfig = field(default_factory=TrainCheckpointingConfig)
    logging: base.LoggingConfig = field(default_factory=base.LoggingConfig)
    n_iters: int = MISSING
    n_iters_decay: int = MISSING
    gan: base.BaseGANConfig = MISSING
    seed: Optional[int] = None
    metrics: TrainMetricsConfig = field(default_factory=TrainMetricsConfig)
