"""Config dataclasses of the kept plugin surface — same field names and defaults as
ganslate/configs/base.py:8-129 so the reference's projects/*.yaml load unchanged."""
from dataclasses import dataclass, field
from typing import Optional, Tuple

from .omegalite import II, MISSING


@dataclass
class BaseDatasetConfig:
    _target_: str = MISSING
    root: str = MISSING
    num_workers: int = 4
    pin_memory: bool = True


@dataclass
class BaseOptimizerConfig:
    adversarial_loss_type: str = "lsgan"
    beta1: float = 0.5
    beta2: float = 0.999
    lr_D: float = 0.0001
    lr_G: float = 0.0002


@dataclass
class GeneratorInOutChannelsConfig:
    AB: Tuple[int, int] = MISSING
    BA: Optional[Tuple[int, int]] = II("train.gan.generator.in_out_channels.AB")


@dataclass
class BaseGeneratorConfig:
    _target_: str = MISSING
    in_out_channels: GeneratorInOutChannelsConfig = field(default_factory=GeneratorInOutChannelsConfig)


@dataclass
class DiscriminatorInChannelsConfig:
    B: int = MISSING
    A: Optional[int] = II("train.gan.discriminator.in_channels.B")


@dataclass
class BaseDiscriminatorConfig:
    _target_: str = MISSING
    in_channels: DiscriminatorInChannelsConfig = field(default_factory=DiscriminatorInChannelsConfig)


@dataclass
class BaseGANConfig:
    _target_: str = MISSING
    norm_type: str = "instance"
    weight_init_type: str = "normal"
    weight_init_gain: float = 0.02
    optimizer: BaseOptimizerConfig = MISSING
    generator: BaseGeneratorConfig = MISSING
    discriminator: Optional[BaseDiscriminatorConfig] = None


@dataclass
class WandbConfig:
    project: str = "ganslate-project"
    entity: Optional[str] = None
    run: Optional[str] = None
    id: Optional[str] = None


@dataclass
class CheckpointingConfig:
    load_iter: int = MISSING


@dataclass
class MultiModalitySplitConfig:
    A: Optional[Tuple[int]] = None
    B: Optional[Tuple[int]] = None


@dataclass
class LoggingConfig:
    freq: int = 50
    multi_modality_split: Optional[MultiModalitySplitConfig] = None
    tensorboard: bool = False
    wandb: Optional[WandbConfig] = None
    image_window: Optional[Tuple[float, float]] = None


@dataclass
class BaseEngineConfig:
    output_dir: str = II("train.output_dir")
    batch_size: int = II("train.batch_size")
    cuda: bool = II("train.cuda")
    mixed_precision: bool = II("train.mixed_precision")
    opt_level: str = II("train.opt_level")
    logging: LoggingConfig = II("train.logging")
    dataset: BaseDatasetConfig = MISSING
