"""Resnet2D generator on the HIP executor — same constructor, layer order, padding, bias rule and state_dict
names as ganslate/nn/generators/resnet/resnet2d.py:14-93:
c7s1-64, d128, d256, n x R256, u128, u64, c7s1-out, tanh; ReflectionPad before the k7 and residual convs;
InstanceNorm2d(affine=False) + ReLU; ConvTranspose2d(3, s2, p1, op1) up-sampling (always biased)."""
from dataclasses import dataclass

from .... import configs
from ...native.net import NativeNet, Node
from ...native.spec import ConvSpec
from ...utils import is_bias_before_norm, require_instance_norm


@dataclass
class Resnet2DConfig(configs.base.BaseGeneratorConfig):
    n_residual_blocks: int = 9


class Resnet2D(NativeNet):

    def __init__(self, in_channels, out_channels, norm_type, n_residual_blocks=9):
        require_instance_norm(norm_type)
        use_bias = is_bias_before_norm(norm_type)
        n = n_residual_blocks
        enc = lambda i: (f"encoder.{i}",)   # `encoder` aliases the first 10+n modules of `model` (resnet2d.py:46)
        nodes = [Node(ConvSpec("conv", in_channels, 64, 7, 1, 3, pad_mode="reflect", bias=use_bias), True, "relu",
                      name="model.1", aliases=enc(1))]
        feats = 64
        for d in range(2):
            idx = 4 + 3 * d
            nodes.append(Node(ConvSpec("conv", feats, feats * 2, 3, 2, 1, bias=use_bias), True, "relu",
                              name=f"model.{idx}", aliases=enc(idx)))
            feats *= 2
        for b in range(n):
            idx = 10 + b
            src = len(nodes) - 1                      # node whose output enters the block (x + conv_block(x))
            nodes.append(Node(ConvSpec("conv", feats, feats, 3, 1, 1, pad_mode="reflect", bias=use_bias), True,
                              "relu", name=f"model.{idx}.conv_block.1", aliases=(f"encoder.{idx}.conv_block.1",)))
            nodes.append(Node(ConvSpec("conv", feats, feats, 3, 1, 1, pad_mode="reflect", bias=use_bias), True,
                              "none", res=src, name=f"model.{idx}.conv_block.5",
                              aliases=(f"encoder.{idx}.conv_block.5",)))
        for u in range(2):
            idx = 10 + n + 3 * u
            nodes.append(Node(ConvSpec("convT", feats, feats // 2, 3, 2, 1, 1), True, "relu", name=f"model.{idx}"))
            feats //= 2
        nodes.append(Node(ConvSpec("conv", feats, out_channels, 7, 1, 3, pad_mode="reflect", bias=use_bias), False,
                          "none", name=f"model.{17 + n}"))
        self.encoder_nodes = 3 + 2 * n              # nodes that make up the reference's `encoder`
        super().__init__(nodes, in_channels, out_channels, out_act="tanh")
