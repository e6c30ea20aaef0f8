"""PatchGAN2D discriminator on the HIP executor — constructor, layer order and state_dict names of
ganslate/nn/discriminators/patchgan/patchgan2d.py:17-66: C(ndf, s2, no norm) - [C(ndf*2^n, s2) - IN] x (n_layers-1)
- C(ndf*min(2^n_layers, 8), s1) - IN - C(1, s1); k4, pad 1, LeakyReLU(0.2); first and last conv always biased."""
from dataclasses import dataclass
from typing import Tuple

from .... import configs
from ...native.net import NativeNet, Node
from ...native.spec import ConvSpec
from ...utils import is_bias_before_norm, require_instance_norm


@dataclass
class PatchGAN2DConfig(configs.base.BaseDiscriminatorConfig):
    ndf: int = 64
    n_layers: int = 3
    kernel_size: Tuple[int] = (4, 4)


def patchgan_nodes(in_channels, ndf, n_layers, kernel_size, use_bias, dims=2):
    """layer list shared by PatchGAN2D (patchgan2d.py:24-63) and PatchGAN3D (patchgan3d.py:24-61)"""
    ks = list(kernel_size) if not isinstance(kernel_size, int) else [kernel_size] * dims
    assert len(set(ks)) == 1 and len(ks) == dims, "square / cubic kernels only"
    kw = int(ks[0])
    conv = lambda *a, **k: ConvSpec(*a, dims=dims, **k)
    nodes = [Node(conv("conv", in_channels, ndf, kw, 2, 1), False, "lrelu", name="model.0")]
    idx, mult = 2, 1
    for n in range(1, n_layers):
        prev, mult = mult, min(2 ** n, 8)
        nodes.append(Node(conv("conv", ndf * prev, ndf * mult, kw, 2, 1, bias=use_bias), True, "lrelu",
                          name=f"model.{idx}"))
        idx += 3
    prev, mult = mult, min(2 ** n_layers, 8)
    nodes.append(Node(conv("conv", ndf * prev, ndf * mult, kw, 1, 1, bias=use_bias), True, "lrelu",
                      name=f"model.{idx}"))
    idx += 3
    nodes.append(Node(conv("conv", ndf * mult, 1, kw, 1, 1), False, "none", name=f"model.{idx}"))
    return nodes


class PatchGAN2D(NativeNet):

    def __init__(self, in_channels, ndf, n_layers, kernel_size, norm_type):
        require_instance_norm(norm_type)
        nodes = patchgan_nodes(in_channels, ndf, n_layers, kernel_size, is_bias_before_norm(norm_type), dims=2)
        super().__init__(nodes, in_channels, 1, out_act="none")
