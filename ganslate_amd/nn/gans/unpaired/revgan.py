"""RevGAN recipe — class surface, step order and loss assembly of
ganslate/nn/gans/unpaired/revgan.py:19-220 ("Reversible GANs for Memory-efficient Image-to-Image Translation", van der
Ouderaa & Worrall, CVPR 2019): CycleGAN's losses with ONE partially-invertible generator used in both directions,
`G(x)` for A -> B and `G(x, inverse=True)` for B -> A (Piresnet3D / Vnet2D / Vnet3D with `use_inverse: true`), networks G, D_B, D_A and
one Adam over G.

Kept as the reference has it: the generator-side adversarial terms pair D_B with fake_A and D_A with fake_B
(revgan.py:187-193; CycleGAN pairs them the other way round). The generator's `use_memory_saving` is memcnn's
activation recompute, implemented by the generators' executors (vnet3d.py, piresnet3d.py): coupling inputs are rebuilt from
outputs during backward instead of being kept. Iterations replay as one hipGraph on one stream (four to six passes
through the same network share its gradient buffer)."""
from dataclasses import dataclass, field

import torch

from .... import configs
from ....data.utils.image_pool import ImagePool
from ...losses.adversarial_loss import AdversarialLoss
from ...losses.cyclegan_losses import CycleGANLosses
from ...optim import NativeAdam
from ..base import BaseGAN
from . import cyclegan


@dataclass
class OptimizerConfig(cyclegan.OptimizerConfig):
    pass


@dataclass
class RevGANConfig(configs.base.BaseGANConfig):
    pool_size: int = 50
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)


class RevGAN(BaseGAN):
    graph_capturable = True      # same fixed launch sequence as CycleGAN: pools' coin flips and Adam's scalars are device data

    def __init__(self, conf):
        super().__init__(conf)
        visual_names = ["real_A", "fake_B", "rec_A", "idt_A", "real_B", "fake_A", "rec_B", "idt_B"]
        self.visuals = {name: None for name in visual_names}
        loss_names = ["G_AB", "D_B", "cycle_A", "idt_A", "G_BA", "D_A", "cycle_B", "idt_B"]
        self.losses = {name: None for name in loss_names}
        self.optimizers = {name: None for name in ["G", "D"]}
        network_names = ["G", "D_B", "D_A"] if self.is_train else ["G"]
        self.networks = {name: None for name in network_names}
        if self.is_train:
            self.fake_A_pool = ImagePool(conf.train.gan.pool_size)
            self.fake_B_pool = ImagePool(conf.train.gan.pool_size)
        self.setup()
        if not getattr(self.networks["G"], "use_inverse", False):
            raise ValueError("RevGAN needs a generator with an inverse direction (Piresnet3D, or Vnet2D / Vnet3D with "
                             "use_inverse: true)")

    def _step_pools(self):
        return [self.fake_B_pool, self.fake_A_pool]      # backward_D("D_B") runs first

    def init_criterions(self):
        self.criterion_adv = AdversarialLoss(self.conf.train.gan.optimizer.adversarial_loss_type).to(self.device)
        self.criterion_G = CycleGANLosses(self.conf)

    def init_optimizers(self):
        import itertools
        opt = self.conf.train.gan.optimizer
        params_D = itertools.chain(self.networks["D_B"].parameters(), self.networks["D_A"].parameters())
        self.optimizers["G"] = NativeAdam(self.networks["G"].parameters(), lr=opt.lr_G, betas=(opt.beta1, opt.beta2))
        self.optimizers["D"] = NativeAdam(params_D, lr=opt.lr_D, betas=(opt.beta1, opt.beta2))

    def set_input(self, input):
        self.visuals["real_A"] = input["A"].to(self.device, non_blocking=True)
        self.visuals["real_B"] = input["B"].to(self.device, non_blocking=True)

    def optimize_parameters(self):
        discriminators = [self.networks["D_B"], self.networks["D_A"]]
        self.forward()
        self.metrics.update(self.training_metrics.compute_metrics_G(self.visuals))
        self.set_requires_grad(discriminators, False)
        self.optimizers["G"].zero_grad(set_to_none=True)
        self.backward_G()
        self.optimizers["G"].step()
        self.set_requires_grad(discriminators, True)
        self.optimizers["D"].zero_grad(set_to_none=True)
        self.backward_D("D_B")
        self.metrics.update(self.training_metrics.compute_metrics_D("D_B", self.pred_real, self.pred_fake))
        self.backward_D("D_A")
        self.metrics.update(self.training_metrics.compute_metrics_D("D_A", self.pred_real, self.pred_fake))
        self.optimizers["D"].step()

    def forward(self):
        G = self.networks["G"]
        real_A, real_B = self.visuals["real_A"], self.visuals["real_B"]
        fake_B = G(real_A)                          # G_AB
        rec_A = G(fake_B, inverse=True)             # G_BA
        fake_A = G(real_B, inverse=True)
        rec_B = G(fake_A)
        idt_B, idt_A = None, None
        if self.criterion_G.is_using_identity():
            idt_B = G(real_B)
            idt_A = G(real_A, inverse=True)
        self.visuals.update({"fake_B": fake_B, "rec_A": rec_A, "idt_A": idt_A,
                             "fake_A": fake_A, "rec_B": rec_B, "idt_B": idt_B})

    def backward_D(self, discriminator):
        if discriminator == "D_B":
            real, fake = self.visuals["real_B"], self.fake_B_pool.query(self.visuals["fake_B"])
            loss_id = 0
        elif discriminator == "D_A":
            real, fake = self.visuals["real_A"], self.fake_A_pool.query(self.visuals["fake_A"])
            loss_id = 1
        else:
            raise ValueError('The discriminator has to be either "D_A" or "D_B".')
        D = self.networks[discriminator]
        if hasattr(D, "forward_parts"):      # D(real) and D(fake) as one pass over both batches (InstanceNorm is per sample)
            self.pred_real, self.pred_fake = D.forward_parts((real, fake.detach()))
        else:
            self.pred_real, self.pred_fake = D(real), D(fake.detach())
        loss_real = self.criterion_adv(self.pred_real, target_is_real=True)
        loss_fake = self.criterion_adv(self.pred_fake, target_is_real=False)
        self.losses[discriminator] = loss_real + loss_fake
        self.backward(loss=self.losses[discriminator], optimizer=self.optimizers["D"], loss_id=loss_id)

    def backward_G(self):
        fake_B, fake_A = self.visuals["fake_B"], self.visuals["fake_A"]
        pred_B = self.networks["D_B"](fake_A)       # as written in the reference (revgan.py:187-188)
        pred_A = self.networks["D_A"](fake_B)
        self.losses["G_AB"] = self.criterion_adv(pred_B, target_is_real=True)
        self.losses["G_BA"] = self.criterion_adv(pred_A, target_is_real=True)
        losses_G = self.criterion_G(self.visuals)
        self.losses.update(losses_G)
        combined_loss_G = sum(losses_G.values()) + self.losses["G_AB"] + self.losses["G_BA"]
        self.backward(loss=combined_loss_G, optimizer=self.optimizers["G"], loss_id=2)

    def infer(self, input, direction="AB"):
        assert direction in ["AB", "BA"], "Specify which generator direction, AB or BA, to use."
        assert "G" in self.networks.keys()
        with torch.no_grad():
            return self.networks["G"](input, inverse=direction == "BA")
