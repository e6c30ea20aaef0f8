"""Pix2PixConditionalGAN recipe — class surface, step order (G then D) and loss assembly of
ganslate/nn/gans/paired/pix2pix.py:11-152 on the HIP executors. The discriminator sees torch.cat([real_A, fake_B], 1)
(6 channels, :111,132,135); the concat of two 3-channel fp32 images is plumbing on the boundary tensors (its autograd
split hands fake_B's gradient slice back to the generator's autograd node)."""
from dataclasses import dataclass, field

import torch

from .... import configs
from ...losses.adversarial_loss import AdversarialLoss
from ...losses.functional import fanout, scalar_affine, scalar_sum
from ...losses.pix2pix_losses import Pix2PixLoss
from ...optim import NativeAdam
from ..base import BaseGAN


@dataclass
class OptimizerConfig(configs.base.BaseOptimizerConfig):
    lambda_pix2pix: float = 100.0


@dataclass
class Pix2PixConditionalGANConfig(configs.base.BaseGANConfig):
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)


class Pix2PixConditionalGAN(BaseGAN):
    graph_capturable = True      # fixed launch sequence, no image pool
    side_stream_names = ("D", "opt")

    def __init__(self, conf):
        super().__init__(conf)
        self.visuals = {name: None for name in ["real_A", "fake_B", "real_B"]}
        self.losses = {name: None for name in ["G", "D", "pix2pix"]}
        self.optimizers = {name: None for name in ["G", "D"]}
        self.networks = {name: None for name in (["G", "D"] if self.is_train else ["G"])}
        self.setup()

    def init_criterions(self):
        self.criterion_adv = AdversarialLoss(self.conf.train.gan.optimizer.adversarial_loss_type).to(self.device)
        self.criterion_pix2pix = Pix2PixLoss(self.conf)

    def init_optimizers(self):
        opt = self.conf.train.gan.optimizer
        self.optimizers["G"] = NativeAdam(self.networks["G"].parameters(), lr=opt.lr_G, betas=(opt.beta1, opt.beta2))
        self.optimizers["D"] = NativeAdam(self.networks["D"].parameters(), lr=opt.lr_D, betas=(opt.beta1, opt.beta2))

    def set_input(self, input):
        self.visuals["real_A"] = input["A"].to(self.device, non_blocking=True)
        self.visuals["real_B"] = input["B"].to(self.device, non_blocking=True)

    def optimize_parameters(self):
        self.forward()
        self.metrics.update(self.training_metrics.compute_metrics_G(self.visuals))
        # ------------------------ G ------------------------
        self.set_requires_grad(self.networks["D"], False)
        self.optimizers["G"].zero_grad(set_to_none=True)
        # (the generator takes one backward pass per step: its update runs layer group by layer group under that pass)
        self.arm_early_update("G")
        self.backward_G()
        self.optimizers["G"].step()
        # ------------------------ D ------------------------
        self.set_requires_grad(self.networks["D"], True)
        self.optimizers["D"].zero_grad(set_to_none=True)
        with self.side_work():       # launched beside the generator's backward pass (BaseGAN.fork_side_work)
            self.backward_D()
            self.metrics.update(self.training_metrics.compute_metrics_D("D", self.pred_real, self.pred_fake))
        self.join_side_work()
        self.optimizers["D"].step()

    def backward_G(self):
        real_A, real_B, fake_B = self.visuals["real_A"], self.visuals["real_B"], self.visuals["fake_B"]
        D = self.networks["D"]
        # the generated image feeds the discriminator and the L1 term: two aliases whose gradients the library adds
        # (losses/functional.py: fanout) instead of autograd's accumulation
        fake_B, fake_B2 = fanout(fake_B)
        if hasattr(D, "forward_parts_cat"):      # D(torch.cat([real_A, fake_B], dim=1)) without the concatenated tensor
            pred, = D.forward_parts_cat([(real_A, fake_B)])
        else:
            pred = D(torch.cat([real_A, fake_B], dim=1))
        self.fork_side_work()        # the discriminator's own update needs nothing that is launched after this point
        self.losses["G"] = self.criterion_adv(pred, target_is_real=True)
        # losses['pix2pix'] = lambda * L1 and G + pix2pix (pix2pix.py:74-77) as one launch
        lam = float(self.criterion_pix2pix.lambda_pix2pix)
        self.losses["pix2pix"], combined_loss_G = scalar_affine(
            [self.losses["G"], self.criterion_pix2pix.unweighted(fake_B2, real_B)], [[0.0, lam], [1.0, lam]])
        self.backward(loss=combined_loss_G, optimizer=self.optimizers["G"])

    def backward_D(self):
        real_A, real_B, fake_B = self.visuals["real_A"], self.visuals["real_B"], self.visuals["fake_B"]
        D = self.networks["D"]
        if hasattr(D, "forward_parts_cat"):  # D(real pair) and D(fake pair) as one pass over both batches (per-sample norm),
            # each pair converted side by side (no torch.cat)
            self.pred_real, self.pred_fake = D.forward_parts_cat([(real_A, real_B), (real_A, fake_B.detach())])
        else:
            pair_real, pair_fake = torch.cat([real_A, real_B], dim=1), torch.cat([real_A, fake_B.detach()], dim=1)
            if hasattr(D, "forward_parts"):
                self.pred_real, self.pred_fake = D.forward_parts((pair_real, pair_fake))
            else:
                self.pred_real, self.pred_fake = D(pair_real), D(pair_fake)
        loss_real = self.criterion_adv(self.pred_real, target_is_real=True)
        loss_fake = self.criterion_adv(self.pred_fake, target_is_real=False)
        self.losses["D"] = scalar_sum([loss_real, loss_fake])
        self.backward(loss=self.losses["D"], optimizer=self.optimizers["D"])

    def forward(self):
        self.visuals.update({"fake_B": self.networks["G"](self.visuals["real_A"])})

    def infer(self, input):
        with torch.no_grad():
            return self.networks["G"].forward(input)
