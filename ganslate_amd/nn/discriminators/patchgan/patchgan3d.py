"""PatchGAN3D discriminator on the HIP executor — constructor, layer order and state_dict names of
ganslate/nn/discriminators/patchgan/patchgan3d.py:17-65 (the Conv3d / InstanceNorm3d twin of PatchGAN2D)."""
from dataclasses import dataclass
from typing import Tuple

from .... import configs
from ...native.net import NativeNet
from ...utils import is_bias_before_norm, require_instance_norm
from .patchgan2d import patchgan_nodes


@dataclass
class PatchGAN3DConfig(configs.base.BaseDiscriminatorConfig):
    ndf: int = 64
    n_layers: int = 3
    kernel_size: Tuple[int] = (4, 4, 4)


class PatchGAN3D(NativeNet):

    def __init__(self, in_channels, ndf, n_layers, kernel_size, norm_type):
        require_instance_norm(norm_type)
        nodes = patchgan_nodes(in_channels, ndf, n_layers, kernel_size, is_bias_before_norm(norm_type), dims=3)
        super().__init__(nodes, in_channels, 1, out_act="none")
