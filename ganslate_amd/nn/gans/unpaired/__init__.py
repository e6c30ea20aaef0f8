from .cut import CUT, CUTConfig  # noqa: F401
from .cyclegan import CycleGAN, CycleGANConfig  # noqa: F401
from .revgan import RevGAN, RevGANConfig  # noqa: F401
