"""SelfAttentionVnet3D on the HIP kernels — constructor, channel plan and state_dict names of
ganslate/nn/generators/vnet/selfattention_vnet3d.py:18-181: the partially-invertible V-Net of vnet3d.py with a
SelfAttentionBlock (nn/attention.py) on the output of the down blocks selected by `enable_attention_block`; the attended map
feeds both the next down block and the skip connection of its level (:150-160). Executor: `Vnet3D(attention=...)`."""
from dataclasses import dataclass
from typing import Tuple

from .... import configs
from .vnet3d import Vnet3D


@dataclass
class SelfAttentionVnet3DConfig(configs.base.BaseGeneratorConfig):
    """Partially-invertible V-Net generator with Self-Attention"""
    use_memory_saving: bool = True
    use_inverse: bool = True
    first_layer_channels: int = 16
    down_blocks: Tuple[int] = (1, 2, 3, 2)
    up_blocks: Tuple[int] = (2, 2, 1, 1)
    is_separable: bool = False
    # Need to correspond to the same length as the number of down blocks
    enable_attention_block: Tuple[bool] = (False, False, True, True)


class SelfAttentionVnet3D(Vnet3D):

    def __init__(self, in_channels, out_channels, norm_type, first_layer_channels=16, down_blocks=(1, 2, 3, 2),
                 up_blocks=(2, 2, 1, 1), use_memory_saving=True, use_inverse=True,
                 enable_attention_block=(True, True, True, True), is_separable=False):
        if len(enable_attention_block) != len(down_blocks):
            raise ValueError("`enable_attention_block` needs one entry per down block.")
        super().__init__(in_channels, out_channels, norm_type, first_layer_channels, down_blocks, up_blocks,
                         use_memory_saving, use_inverse, is_separable, attention=tuple(enable_attention_block))
