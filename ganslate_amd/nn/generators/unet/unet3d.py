"""Unet3D generator on the HIP kernels — constructor, channel plan, block order, bias rule, dropout placement and
state_dict names of ganslate/nn/generators/unet/unet3d.py:17-156: the Conv3d / ConvTranspose3d / InstanceNorm3d twin of
Unet2D (same executor, volumes laid out NDHWC)."""
from dataclasses import dataclass

from .... import configs
from .unet2d import Unet2D


@dataclass
class Unet3DConfig(configs.base.BaseGeneratorConfig):
    num_downs: int = 7
    ngf: int = 64
    use_dropout: bool = False


class Unet3D(Unet2D):
    dims = 3
