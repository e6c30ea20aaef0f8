"""Resnet3D generator on the HIP executor — same constructor, layer order, padding, bias rule and state_dict names
as ganslate/nn/generators/resnet/resnet3d.py:14-92: the 3-D twin of Resnet2D with nn.ReplicationPad3d instead of
reflection padding (resnet3d.py:15,24,78), Conv3d / ConvTranspose3d(3, s2, p1, op1) and InstanceNorm3d."""
from dataclasses import dataclass

from .... import configs
from ...native.net import NativeNet
from ...utils import is_bias_before_norm, require_instance_norm
from .resnet2d import resnet_nodes


@dataclass
class Resnet3DConfig(configs.base.BaseGeneratorConfig):
    n_residual_blocks: int = 9


class Resnet3D(NativeNet):

    def __init__(self, in_channels, out_channels, norm_type, n_residual_blocks=9):
        require_instance_norm(norm_type)
        nodes = resnet_nodes(in_channels, out_channels, is_bias_before_norm(norm_type), n_residual_blocks, dims=3)
        self.n_residual_blocks = n_residual_blocks
        super().__init__(nodes, in_channels, out_channels, out_act="tanh")
