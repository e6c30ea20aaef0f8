"""MultiScalePatchGAN3D — ganslate/nn/discriminators/patchgan/multiscale_patchgan3d.py:14-60: `scales`
independent PatchGAN3Ds; discriminator s sees a random crop of the input with every spatial extent divided by s (s = 1: the
whole input) and the forward returns {"1": map, "2": map, ...}, which AdversarialLoss averages key by key
(adversarial_loss.py:92-94). Same constructor, config fields and state-dict names (`model.<s>.model.<i>.weight`).

Each scale is a PatchGAN3D on the HIP executor (its own flat parameter buffer: `parameters()` chains them in scale
order, which is nn.ModuleDict's order, so one optimiser spans them as in the reference). The crop window is the
reference's monai `RandSpatialCrop(roi, random_center=True, random_size=False)` applied to the (B, C, D, H, W) tensor:
one window per call, shared by the batch, start uniform over the valid starts per spatial axis. monai draws it from a
numpy RandomState nobody seeds (a fresh one per call, multiscale_patchgan3d.py:25-28), so the reference's crops are not
reproducible at all; here they come from Python's `random`, which the run's seed covers (one randint per axis whose
extent shrinks, D-H-W order, none for s = 1). Crops are drawn on the host per call, so a recipe using this discriminator
runs its step launch by launch (`graph_capturable = False`)."""
import random
from dataclasses import dataclass
from typing import Tuple

import torch

from .... import configs
from .patchgan3d import PatchGAN3D


def get_cropped_patch(input: torch.Tensor, scale: int = 1) -> torch.Tensor:
    """random sub-volume of a (B, C, D, H, W) tensor with spatial extents // scale (multiscale_patchgan3d.py:14-29)"""
    if scale == 1:
        return input
    sizes = [input.shape[a] // scale for a in (2, 3, 4)]
    starts = [random.randint(0, input.shape[a] - n) if input.shape[a] > n else 0 for a, n in zip((2, 3, 4), sizes)]
    z, y, x = starts
    return input[:, :, z:z + sizes[0], y:y + sizes[1], x:x + sizes[2]].contiguous()


@dataclass
class MultiScalePatchGAN3DConfig(configs.base.BaseDiscriminatorConfig):
    ndf: int = 64
    n_layers: int = 3
    kernel_size: Tuple[int] = (4, 4, 4)
    # discriminator s (1..scales) looks at a random patch of 1/s the size (multiscale_patchgan3d.py:38-41)
    scales: int = 2


class MultiScalePatchGAN3D:
    graph_capturable = False         # host-drawn crop windows per call

    def __init__(self, in_channels, ndf, n_layers, kernel_size, scales, norm_type):
        self.model = {str(s): PatchGAN3D(in_channels, ndf, n_layers, kernel_size, norm_type)
                      for s in range(1, scales + 1)}
        self.training = True

    # ---- what BaseGAN / the recipes use of a network ------------------------------------------------------------------
    def native_children(self):
        return list(self.model.values())

    @property
    def ops(self):
        return self.model["1"].ops

    @property
    def device(self):
        return self.model["1"].device

    def parameters(self):
        return [p for net in self.model.values() for p in net.parameters()]

    def train(self, mode=True):
        self.training = mode
        for net in self.model.values():
            net.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    def init_weights(self, *args, **kwargs):
        for net in self.model.values():       # scale order = the order nn.Module.apply visits the reference's ModuleDict
            net.init_weights(*args, **kwargs)

    def parallelize(self, *args, **kwargs):
        for net in self.model.values():
            net.parallelize(*args, **kwargs)

    def state_dict(self):
        return {f"model.{s}.{k}": v for s, net in self.model.items() for k, v in net.state_dict().items()}

    def load_state_dict(self, sd, strict=True):
        known = set()
        for s, net in self.model.items():
            prefix = f"model.{s}."
            sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
            known.update(prefix + k for k in sub)
            net.load_state_dict(sub, strict=strict)
        if strict and set(sd) - known:
            raise KeyError(f"unexpected keys {sorted(set(sd) - known)[:4]} ...")

    def grads_state_dict(self):
        return {f"model.{s}.{k}": v for s, net in self.model.items() for k, v in net.grads_state_dict().items()}

    def forward(self, input):
        return {s: net(get_cropped_patch(input, scale=int(s))) for s, net in self.model.items()}

    __call__ = forward
