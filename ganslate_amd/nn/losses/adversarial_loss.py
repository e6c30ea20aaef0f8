"""GAN objectives — interface of ganslate/nn/losses/adversarial_loss.py:7-98 (lsgan | vanilla | wgangp |
nonsaturating; dict-of-predictions averaged). Every mode is one fused reduction kernel forward and one elementwise kernel
backward (gs_mse_const / gs_adv_loss)."""
from typing import Dict, Union

import torch

from .functional import adversarial_loss, mse_const_loss


class AdversarialLoss:

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        if gan_mode not in ("lsgan", "vanilla", "wgangp", "nonsaturating"):
            raise NotImplementedError(f"GAN mode {gan_mode} not implemented.")
        self.gan_mode = gan_mode
        self.real_label, self.fake_label = float(target_real_label), float(target_fake_label)

    def to(self, device):
        return self

    def calculate_loss(self, prediction: torch.Tensor, target_is_real: bool):
        target = self.real_label if target_is_real else self.fake_label
        if self.gan_mode == "lsgan":
            return mse_const_loss(prediction, target)
        # nonsaturating: the reference raises NameError here (adversarial_loss.py:68-73 uses F without importing
        # it); the intended per-sample softplus form is implemented instead
        return adversarial_loss(prediction, self.gan_mode, target_is_real, target)

    def __call__(self, prediction: Union[Dict[str, torch.Tensor], torch.Tensor], target_is_real: bool):
        if isinstance(prediction, dict):
            return torch.stack([self.calculate_loss(p, target_is_real) for p in prediction.values()]).mean()
        return self.calculate_loss(prediction, target_is_real)

    forward = __call__
