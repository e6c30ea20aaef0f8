"""GAN objectives — interface of ganslate/nn/losses/adversarial_loss.py:7-98 (lsgan | vanilla | wgangp |
nonsaturating; dict-of-predictions averaged). `lsgan` runs the fused MSE-vs-constant kernel."""
from typing import Dict, Union

import torch
import torch.nn.functional as F

from .functional import mse_const_loss


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
        if self.gan_mode == "vanilla":
            return F.binary_cross_entropy_with_logits(prediction, torch.full_like(prediction, target))
        if self.gan_mode == "wgangp":
            return -prediction.mean() if target_is_real else prediction.mean()
        # nonsaturating: the reference raises NameError here (adversarial_loss.py:68-73 uses F without importing
        # it); the intended softplus form is implemented instead
        bs = prediction.size(0)
        sign = -1.0 if target_is_real else 1.0
        return F.softplus(sign * prediction).view(bs, -1).mean(dim=1)

    def __call__(self, prediction: Union[Dict[str, torch.Tensor], torch.Tensor], target_is_real: bool):
        if isinstance(prediction, dict):
            return torch.stack([self.calculate_loss(p, target_is_real) for p in prediction.values()]).mean()
        return self.calculate_loss(prediction, target_is_real)

    forward = __call__
