"""SelfAttentionPatchGAN3D on the HIP executor — constructor, layer order and state_dict names of
ganslate/nn/discriminators/patchgan/selfattention_patchgan3d.py:18-79: PatchGAN3D whose first conv has stride 3 ("to allow
memory to fit", :33) with a SelfAttentionBlock (nn/attention.py) behind the last stride-2 stage and behind the stride-1
stage:

  model.0  C(ndf, k4 s3, bias) - LReLU | [model.{2,5,..} C(ndf*2^n, s2) - IN - LReLU] x (n_layers-1) | attention(ndf*m)
  | C(ndf*m', s1) - IN - LReLU | attention(ndf*m') | C(1, s1, bias)"""
from dataclasses import dataclass
from typing import Tuple

from .... import configs
from ...native.net import NativeNet, Node, attention_extras
from ...native.spec import ConvSpec
from ...utils import is_bias_before_norm, require_instance_norm


@dataclass
class SelfAttentionPatchGAN3DConfig(configs.base.BaseDiscriminatorConfig):
    ndf: int = 64
    n_layers: int = 3
    kernel_size: Tuple[int] = (4, 4, 4)


class SelfAttentionPatchGAN3D(NativeNet):

    def __init__(self, in_channels, ndf, n_layers, kernel_size, norm_type):
        require_instance_norm(norm_type)
        use_bias = is_bias_before_norm(norm_type)
        ks = list(kernel_size) if not isinstance(kernel_size, int) else [kernel_size] * 3
        assert len(set(ks)) == 1 and len(ks) == 3, "cubic kernels only"
        assert n_layers >= 2, "the attention block sits behind a normalised stage (n_layers >= 2)"
        kw = int(ks[0])
        conv = lambda *a, **k: ConvSpec(*a, dims=3, **k)
        nodes = [Node(conv("conv", in_channels, ndf, kw, 3, 1), False, "lrelu", name="model.0")]
        order = ["model.0.weight", "model.0.bias"]
        extras = []

        def add(spec, norm, act, idx):
            nodes.append(Node(spec, norm, act, name=f"model.{idx}"))
            order.append(f"model.{idx}.weight")
            if spec.bias:
                order.append(f"model.{idx}.bias")

        def attention(idx, C):
            nodes[-1].attn = f"model.{idx}"
            ex = attention_extras(f"model.{idx}", C, dims=3)
            extras.extend(ex)
            order.extend(e.name for e in ex)

        idx, mult = 2, 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            add(conv("conv", ndf * prev, ndf * mult, kw, 2, 1, bias=use_bias), True, "lrelu", idx)
            idx += 3
        attention(idx, ndf * mult)
        idx += 1
        prev, mult = mult, min(2 ** n_layers, 8)
        add(conv("conv", ndf * prev, ndf * mult, kw, 1, 1, bias=use_bias), True, "lrelu", idx)
        idx += 3
        attention(idx, ndf * mult)
        idx += 1
        add(conv("conv", ndf * mult, 1, kw, 1, 1), False, "none", idx)
        super().__init__(nodes, in_channels, 1, out_act="none", extras=extras)
        self._param_order = order

    def reference_parameter_order(self):
        """torch's parameters() order of the reference nn.Sequential: layer by layer, an attention block as gamma, then its
        query / key / value convs"""
        return list(self._param_order)
