"""ganslate/configs/validation_testing.py restated — parsed so YAMLs with val/test sections load; the
validation/test engines themselves are out of the hot-path scope."""
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

from . import base
from .omegalite import MISSING


@dataclass
class SlidingWindowConfig:
    window_size: Tuple[int] = MISSING
    batch_size: int = 1
    overlap: float = 0.25
    mode: str = "gaussian"


@dataclass
class BaseValTestMetricsConfig:
    ssim: bool = True
    psnr: bool = True
    nmse: bool = True
    mse: bool = True
    mae: bool = True
    nmi: bool = False
    histogram_chi2: bool = False


@dataclass
class ValMetricsConfig(BaseValTestMetricsConfig):
    cycle_metrics: bool = True


@dataclass
class TestMetricsConfig(BaseValTestMetricsConfig):
    compute_over_input: bool = False
    save_to_csv: bool = True


@dataclass
class BaseValTestConfig(base.BaseEngineConfig):
    sliding_window: Optional[SlidingWindowConfig] = None
    dataset: Optional[base.BaseDatasetConfig] = None
    multi_dataset: Optional[Dict[str, base.BaseDatasetConfig]] = None


@dataclass
class ValidationConfig(BaseValTestConfig):
    freq: int = MISSING
    start_after: int = 0
    metrics: ValMetricsConfig = field(default_factory=ValMetricsConfig)


@dataclass
class TestConfig(BaseValTestConfig):
    checkpointing: base.CheckpointingConfig = field(default_factory=base.CheckpointingConfig)
    metrics: TestMetricsConfig = field(default_factory=TestMetricsConfig)
