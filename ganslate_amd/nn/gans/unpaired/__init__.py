from .cyclegan import CycleGAN, CycleGANConfig  # noqa: F401
