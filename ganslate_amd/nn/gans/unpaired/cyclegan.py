"""CycleGAN recipe — same class surface, step order and loss assembly as
ganslate/nn/gans/unpaired/cyclegan.py:12-224, running on the HIP executors:
forward (4 or 6 generator passes) -> SSIM metrics -> G step (Ds frozen: no D weight-gradient kernels are
launched) -> D_B step, D_A step (image pools, `.detach()`), Adam on flat buffers."""
import itertools
import os
from dataclasses import dataclass, field

import torch

from .... import configs
from ....data.utils.image_pool import ImagePool
from ...losses.adversarial_loss import AdversarialLoss
from ...losses.cyclegan_losses import CycleGANLosses
from ...losses.functional import fanout, scalar_affine, scalar_sum
from ...native.twin import TwinNet
from ...optim import NativeAdam
from ..base import BaseGAN


@dataclass
class OptimizerConfig(configs.base.BaseOptimizerConfig):
    lambda_AB: float = 10.0
    lambda_BA: float = 10.0
    lambda_identity: float = 0
    proportion_ssim: float = 0.84


@dataclass
class CycleGANConfig(configs.base.BaseGANConfig):
    pool_size: int = 50
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)


class CycleGAN(BaseGAN):
    graph_capturable = True      # fixed launch sequence: the pools' coin flips and Adam's scalars live in device memory
    side_stream_names = ("cycle_B", "D")

    def __init__(self, conf):
        super().__init__(conf)
        visual_names = ["real_A", "fake_B", "rec_A", "idt_A", "real_B", "fake_A", "rec_B", "idt_B"]
        self.visuals = {name: None for name in visual_names}
        loss_names = ["G_AB", "D_B", "cycle_A", "idt_A", "G_BA", "D_A", "cycle_B", "idt_B"]
        self.losses = {name: None for name in loss_names}
        self.optimizers = {name: None for name in ["G", "D"]}
        network_names = ["G_AB", "G_BA", "D_B", "D_A"] if self.is_train else ["G_AB"]
        self.networks = {name: None for name in network_names}
        if self.is_train:
            self.fake_A_pool = ImagePool(conf.train.gan.pool_size)
            self.fake_B_pool = ImagePool(conf.train.gan.pool_size)
        self.setup()
        self._init_twins()

    def _init_twins(self):
        """The two generators (and the two discriminators) have the same layer list and see independent data: each pair
        runs lock-step as ONE batch of 2N images with per-image-range weights (nn/native/twin.py) — half the launches, and
        512 instead of 256 tiles per residual-conv launch. GS_TWIN=0 keeps the two passes apart (second cycle on its own
        stream); GS_TWIN=2d does so for volumes only. Volumes: twin passes are worth +2.9 % on the Resnet3D recipe; the V-Net
        executor runs them too since round 6 (Vnet3D._forward(..., tw)) but LOSES 2.2 % on the brats recipe (its launches fill
        the chip at batch 1 and the per-network PReLU-norm launches stay halves: profiles/r06_ab_vnet_twin.txt) — executors
        that set twin_default = False pair up only with GS_TWIN=all."""
        self.twin_G = self.twin_D = None
        for name in ("G_AB", "G_BA"):      # (GS_WGRAD_STREAM=1: their weight gradients beside the data-gradient chain, net.py)
            if name in self.networks and self.networks[name] is not None:
                self.networks[name].wgrad_side_stream = True
        mode = os.environ.get("GS_TWIN", "1")
        if not self.is_train or mode == "0":
            return
        ok = lambda a, b: TwinNet.compatible(a, b) and (a.dims == 2 or mode != "2d") and \
            (getattr(a, "twin_default", True) or mode == "all")
        if ok(self.networks["G_AB"], self.networks["G_BA"]):
            self.twin_G = TwinNet(self.networks["G_AB"], self.networks["G_BA"])
        if ok(self.networks["D_B"], self.networks["D_A"]):
            self.twin_D = TwinNet(self.networks["D_B"], self.networks["D_A"])

    def _step_pools(self):
        return [self.fake_B_pool, self.fake_A_pool]      # backward_D("D_B") runs first

    def init_criterions(self):
        self.criterion_adv = AdversarialLoss(self.conf.train.gan.optimizer.adversarial_loss_type).to(self.device)
        self.criterion_G = CycleGANLosses(self.conf)

    def init_optimizers(self):
        opt = self.conf.train.gan.optimizer
        params_G = itertools.chain(self.networks["G_AB"].parameters(), self.networks["G_BA"].parameters())
        params_D = itertools.chain(self.networks["D_B"].parameters(), self.networks["D_A"].parameters())
        self.optimizers["G"] = NativeAdam(params_G, lr=opt.lr_G, betas=(opt.beta1, opt.beta2))
        self.optimizers["D"] = NativeAdam(params_D, lr=opt.lr_D, betas=(opt.beta1, opt.beta2))

    def set_input(self, input):
        self.visuals["real_A"] = input["A"].to(self.device, non_blocking=True)
        self.visuals["real_B"] = input["B"].to(self.device, non_blocking=True)

    def optimize_parameters(self):
        discriminators = [self.networks["D_B"], self.networks["D_A"]]
        self.forward()
        self.metrics.update(self.training_metrics.compute_metrics_G(self.visuals))
        # ------------------------ G (A and B) ----------------------------------------------------
        self.set_requires_grad(discriminators, False)
        self.optimizers["G"].zero_grad(set_to_none=True)
        self.backward_G()
        self.optimizers["G"].step()
        # ------------------------ D_B and D_A ----------------------------------------------------
        self.set_requires_grad(discriminators, True)
        self.optimizers["D"].zero_grad(set_to_none=True)
        with self.side_work():       # launched beside the generators' backward pass (BaseGAN.fork_side_work)
            if self.twin_D is not None:
                self.backward_D_twin()
            else:
                self.backward_D("D_B")
                self.metrics.update(self.training_metrics.compute_metrics_D("D_B", self.pred_real, self.pred_fake))
                self.backward_D("D_A")
                self.metrics.update(self.training_metrics.compute_metrics_D("D_A", self.pred_real, self.pred_fake))
        self.join_side_work()
        self.optimizers["D"].step()

    def forward(self):
        """The two translation cycles A -> B -> A and B -> A -> B are independent until the losses: the second one is
        launched on its own stream (BaseGAN.side_work; autograd then runs its backward there too). Host order, and
        with it the autograd graph, is the reference's (cyclegan.py:109-124)."""
        real_A, real_B = self.visuals["real_A"], self.visuals["real_B"]
        if self.twin_G is not None:
            # both generators as one batch per phase: (G_AB(real_A), G_BA(real_B)), then (G_AB(fake_A), G_BA(fake_B))
            fake_B, fake_A = self.twin_G(real_A, real_B)
            # each generated image feeds the other generator now and its discriminator in the G step: two aliases whose
            # gradients the library adds (losses/functional.py:fanout) instead of autograd's accumulation
            (fake_B, fake_B2), (fake_A, fake_A2) = fanout(fake_B), fanout(fake_A)
            rec_B, rec_A = self.twin_G(fake_A2, fake_B2)
            idt_B, idt_A = self.twin_G(real_B, real_A) if self.criterion_G.is_using_identity() else (None, None)
            self.visuals.update({"fake_B": fake_B, "rec_A": rec_A, "idt_A": idt_A,
                                 "fake_A": fake_A, "rec_B": rec_B, "idt_B": idt_B})
            return
        for net in (self.networks["G_AB"], self.networks["G_BA"]):
            net.refresh_packs(real_A)          # weight packs are shared by both streams: refresh them before the fork
            net.multi_stream_passes = True     # ... and their backward passes are ordered per network (NativeNet)
        use_idt = self.criterion_G.is_using_identity()
        idt_B, idt_A = None, None
        self.fork_side_work("cycle_B")
        fake_B, fake_B2 = fanout(self.networks["G_AB"](real_A))
        rec_A = self.networks["G_BA"](fake_B2)
        with self.side_work("cycle_B"):
            fake_A, fake_A2 = fanout(self.networks["G_BA"](real_B))
            rec_B = self.networks["G_AB"](fake_A2)
            if use_idt:
                idt_B = self.networks["G_AB"](real_B)
        if use_idt:
            idt_A = self.networks["G_BA"](real_A)
        self.join_side_work("cycle_B", last=False)
        self.visuals.update({"fake_B": fake_B, "rec_A": rec_A, "idt_A": idt_A,
                             "fake_A": fake_A, "rec_B": rec_B, "idt_B": idt_B})

    def backward_D(self, discriminator):
        if discriminator == "D_B":
            real, fake = self.visuals["real_B"], self.fake_B_pool.query(self.visuals["fake_B"])
        elif discriminator == "D_A":
            real, fake = self.visuals["real_A"], self.fake_A_pool.query(self.visuals["fake_A"])
        else:
            raise ValueError('The discriminator has to be either "D_A" or "D_B".')
        D = self.networks[discriminator]
        if hasattr(D, "forward_parts"):      # D(real) and D(fake) as one pass over both batches (InstanceNorm is per sample)
            self.pred_real, self.pred_fake = D.forward_parts((real, fake.detach()))
        else:
            self.pred_real, self.pred_fake = D(real), D(fake.detach())
        loss_real = self.criterion_adv(self.pred_real, target_is_real=True)
        loss_fake = self.criterion_adv(self.pred_fake, target_is_real=False)
        self.losses[discriminator] = scalar_sum((loss_real, loss_fake))
        self.backward(loss=self.losses[discriminator], optimizer=self.optimizers["D"], loss_id=2)

    def backward_D_twin(self):
        """backward_D("D_B") and backward_D("D_A") (cyclegan.py:154-189) as one pass: D_B sees [real_B | pooled fake_B] and
        D_A sees [real_A | pooled fake_A] as a batch of 2N images each (InstanceNorm is per sample: every image's map is what
        the separate passes give), the two discriminators ride in one twin batch of 4N, and the two losses — which share
        no parameter — are differentiated together."""
        fake_B = self.fake_B_pool.query(self.visuals["fake_B"]).detach()      # pool order: D_B first (_step_pools)
        fake_A = self.fake_A_pool.query(self.visuals["fake_A"]).detach()
        pred_B, pred_A = self.twin_D((self.visuals["real_B"], fake_B), (self.visuals["real_A"], fake_A))
        terms = []
        for name, pred in (("D_B", pred_B), ("D_A", pred_A)):
            self.pred_real, self.pred_fake = pred
            terms += [self.criterion_adv(self.pred_real, target_is_real=True),
                      self.criterion_adv(self.pred_fake, target_is_real=False)]
            self.metrics.update(self.training_metrics.compute_metrics_D(name, self.pred_real, self.pred_fake))
        # D_B = real + fake, D_A = real + fake (cyclegan.py:182) and their sum, one launch
        self.losses["D_B"], self.losses["D_A"], total = scalar_affine(terms, [[1, 1, 0, 0], [0, 0, 1, 1], [1, 1, 1, 1]])
        self.backward(loss=total, optimizer=self.optimizers["D"], loss_id=2)

    def backward_G(self):
        fake_B, fake_A = self.visuals["fake_B"], self.visuals["fake_A"]
        if self.twin_D is not None:
            pred_B, pred_A = self.twin_D(fake_B, fake_A)
        else:
            pred_B = self.networks["D_B"](fake_B)
            pred_A = self.networks["D_A"](fake_A)
        # from here on the discriminators' own update may run: the images exist, the weight packs are refreshed
        self.fork_side_work()
        self.losses["G_AB"] = self.criterion_adv(pred_B, target_is_real=True)
        self.losses["G_BA"] = self.criterion_adv(pred_A, target_is_real=True)
        losses_G = self.criterion_G(self.visuals)
        self.losses.update(losses_G)
        combined_loss_G = scalar_sum(list(losses_G.values()) + [self.losses["G_AB"], self.losses["G_BA"]])
        self.backward(loss=combined_loss_G, optimizer=self.optimizers["G"], loss_id=0)
        self.join_side_work("cycle_B")      # the second cycle's backward ran on its own stream (see forward)

    def infer(self, input, direction="AB"):
        assert direction in ["AB", "BA"], "Specify which generator direction, AB or BA, to use."
        assert f"G_{direction}" in self.networks.keys()
        with torch.no_grad():
            return self.networks[f"G_{direction}"](input)
