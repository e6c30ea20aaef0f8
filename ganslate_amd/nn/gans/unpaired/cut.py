"""CUT / FastCUT recipe — class surface, step order (D first, then G + mlp) and loss assembly of
ganslate/nn/gans/unpaired/cut.py:16-226 on the HIP executors.

Differences that are deliberate: the reference reads `generator.in_channels`, a key that does not
exist (cut.py:83) — `in_out_channels.AB[0]` is used; under data parallelism it reaches for `.encoder` of the DDP
wrapper (cut.py:205-211) — here the encoder passes are partial passes of the same executor, so their weight gradients
land in the same flat buffer and are all-reduced with it; the `mlp` optimiser state is checkpointed too.

The encoder-only passes do not materialise NCHW feature maps: a "patch" is one pixel's channel vector, i.e. one
contiguous row of the NHWC activation, so the sampled patches are gathered straight out of the executor's
activations (Resnet2D.extract_patch_features) and their gradients scattered straight back."""
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np
import torch
from torch import nn

from .... import configs
from ...losses.adversarial_loss import AdversarialLoss
from ...losses.cut_losses import PatchNCELoss
from ...optim import NativeAdam
from ..base import BaseGAN


@dataclass
class OptimizerConfig(configs.base.BaseOptimizerConfig):
    lambda_adv: float = 1
    lambda_nce: float = 1
    lambda_nce_idt: float = 0.5
    nce_T: float = 0.07


@dataclass
class CUTConfig(configs.base.BaseGANConfig):
    nce_layers: Tuple[int] = (0, 4, 8, 12, 16)
    mlp_nc: int = 256
    num_patches: int = 256
    use_equivariance_flip: bool = False
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)


class CUT(BaseGAN):

    def __init__(self, conf):
        super().__init__(conf)
        opt = conf.train.gan.optimizer
        self.lambda_adv, self.lambda_nce, self.lambda_nce_idt = opt.lambda_adv, opt.lambda_nce, opt.lambda_nce_idt
        self.nce_layers = list(conf.train.gan.nce_layers)
        self.num_patches = conf.train.gan.num_patches
        self.use_equivariance_flip = conf.train.gan.use_equivariance_flip
        self.is_flipped = False
        self.visuals = {name: None for name in ["real_A", "fake_B", "real_B", "idt_B"]}
        self.losses = {name: None for name in ["D", "G", "NCE", "NCE_idt"]}
        self.networks = {name: None for name in (["G", "D", "mlp"] if self.is_train else ["G"])}
        self.setup()

    def init_networks(self):
        super().init_networks()           # builds G and D; 'mlp' matches neither prefix and is built here
        if self.is_train:
            G = self.networks["G"]
            channels = [G.encoder_tap(e)[1] for e in self.nce_layers]
            mlp = FeaturePatchMLP(channels, self.num_patches, self.conf.train.gan.mlp_nc).to(self.device)
            mlp.init_weights(self.conf.train.gan.weight_init_type, self.conf.train.gan.weight_init_gain)
            self.networks["mlp"] = mlp

    def init_optimizers(self):
        opt = self.conf.train.gan.optimizer
        betas = (opt.beta1, opt.beta2)
        self.optimizers["G"] = NativeAdam(self.networks["G"].parameters(), lr=opt.lr_G, betas=betas)
        self.optimizers["D"] = NativeAdam(self.networks["D"].parameters(), lr=opt.lr_D, betas=betas)
        self.optimizers["mlp"] = torch.optim.Adam(self.networks["mlp"].parameters(), lr=opt.lr_G, betas=betas)

    def init_criterions(self):
        self.criterion_adv = AdversarialLoss(self.conf.train.gan.optimizer.adversarial_loss_type).to(self.device)
        self.criterion_nce = [PatchNCELoss(self.conf).to(self.device) for _ in self.nce_layers]

    def parallelize_networks(self):
        for name, net in self.networks.items():
            net.parallelize()

    def optimize_parameters(self):
        self.forward()
        # ------------------------ Discriminator --------------------------------------------------
        self.set_requires_grad(self.networks["D"], True)
        self.optimizers["D"].zero_grad(set_to_none=True)
        self.backward_D()
        self.optimizers["D"].step()
        # ------------------------ Generator and MLP ----------------------------------------------
        self.set_requires_grad(self.networks["D"], False)
        self.optimizers["G"].zero_grad(set_to_none=True)
        self.optimizers["mlp"].zero_grad(set_to_none=True)
        self.backward_G_and_mlp()
        self.optimizers["G"].step()
        self.networks["mlp"].finish_grad_reduction()
        self.optimizers["mlp"].step()

    def set_input(self, input):
        self.visuals["real_A"] = input["A"].to(self.device, non_blocking=True)
        self.visuals["real_B"] = input["B"].to(self.device, non_blocking=True)

    def forward(self):
        using_idt = self.lambda_nce_idt > 0
        real_A = self.visuals["real_A"]
        real_B = self.visuals["real_B"] if using_idt else None
        if self.use_equivariance_flip and self.is_train:
            self.is_flipped = np.random.random() > 0.5
            if self.is_flipped:
                real_A = real_A.flip(-1)
                if using_idt:
                    real_B = real_B.flip(-1)
        self.visuals["fake_B"] = self.networks["G"](real_A)
        if using_idt:
            self.visuals["idt_B"] = self.networks["G"](real_B)

    def backward_D(self):
        real, fake = self.visuals["real_B"], self.visuals["fake_B"]
        pred_real = self.networks["D"](real)
        pred_fake = self.networks["D"](fake.detach())
        loss_real = self.criterion_adv(pred_real, True).mean()
        loss_fake = self.criterion_adv(pred_fake, False).mean()
        self.losses["D"] = loss_real + loss_fake
        self.backward(loss=self.losses["D"], optimizer=self.optimizers["D"], loss_id=0)

    def backward_G_and_mlp(self):
        real_A, real_B = self.visuals["real_A"], self.visuals["real_B"]
        fake_B, idt_B = self.visuals["fake_B"], self.visuals["idt_B"]
        adversarial_loss = 0
        if self.lambda_adv > 0:
            pred_fake = self.networks["D"](fake_B)
            adversarial_loss = self.criterion_adv(pred_fake, True).mean() * self.lambda_adv
            self.losses["G"] = adversarial_loss
        nce_loss = 0
        if self.lambda_nce > 0:
            nce_loss = self._calculate_nce_loss(real_A, fake_B)
            self.losses["NCE"] = nce_loss
            if self.lambda_nce_idt > 0:
                nce_idt_loss = self.lambda_nce_idt * self._calculate_nce_loss(real_B, idt_B)
                nce_loss = (1 - self.lambda_nce_idt) * nce_loss + nce_idt_loss
                self.losses["NCE_idt"] = nce_idt_loss
        combined_loss = adversarial_loss + nce_loss
        self.backward(loss=combined_loss, optimizer=(self.optimizers["G"], self.optimizers["mlp"]), loss_id=1)

    def sample_patch_ids(self, H, W):
        """one torch.randperm per feature level, in level order — the RNG call sequence of FeaturePatchMLP.forward on
        the source features (cut.py:262-268)"""
        G = self.networks["G"]
        ids = []
        for e in self.nce_layers:
            n = G.tap_extent(e, H, W)
            pid = torch.randperm(n, device=self.device)
            ids.append(pid[:int(min(self.num_patches, n))] if self.num_patches > 0 else torch.arange(n, device=self.device))
        return ids

    def _calculate_nce_loss(self, source, target, patch_ids=None):
        G, mlp = self.networks["G"], self.networks["mlp"]
        H, W = source.shape[-2:]
        ids = patch_ids if patch_ids is not None else self.sample_patch_ids(H, W)
        source_feats = G.extract_patch_features(source, self.nce_layers, ids)
        tgt_ids = ids
        if self.is_flipped:       # target features are flipped back along W before sampling (cut.py:214-215)
            tgt_ids = []
            for e, pid in zip(self.nce_layers, ids):
                w = (W + 6) if e == 0 else (W if e < 4 else (W // 2 if e < 7 else W // 4))
                tgt_ids.append((pid // w) * w + (w - 1 - pid % w))
        target_feats = G.extract_patch_features(target, self.nce_layers, tgt_ids)
        source_pool = mlp(source_feats)
        target_pool = mlp(target_feats)
        nce_loss = 0
        for target_feat, source_feat, criterion in zip(target_pool, source_pool, self.criterion_nce):
            nce_loss = nce_loss + (criterion(target_feat, source_feat) * self.lambda_nce).mean()
        return nce_loss / len(self.nce_layers)


class FeaturePatchMLP(nn.Module):
    """per-level Linear(C, nc) - ReLU - Linear(nc, nc) - L2 normalise on already-sampled patches [N, P, C]
    (cut.py:229-294; the patch sampling itself happens at the activation gather). Plain library GEMMs."""

    def __init__(self, channels_per_feature, num_patches=256, nc=256):
        super().__init__()
        self.num_patches = num_patches
        self.mlps = nn.ModuleList(
            nn.Sequential(nn.Linear(c, nc), nn.ReLU(), nn.Linear(nc, nc)) for c in channels_per_feature)
        self._dist = None

    def init_weights(self, init_type="normal", gain=0.02):
        assert init_type == "normal", "FeaturePatchMLP: only `normal` init is used by the reference configs"
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, gain)
                nn.init.constant_(m.bias, 0.0)

    def forward(self, feats):
        out = []
        for mlp, feat in zip(self.mlps, feats):
            f = mlp(feat.flatten(0, 1))
            norm = f.pow(2).sum(1, keepdim=True).pow(0.5)
            out.append(f.div(norm + 1e-7))
        return out

    def parallelize(self, process_group=None):
        import torch.distributed as dist
        self._dist = process_group if process_group is not None else dist.group.WORLD
        for p in self.parameters():
            dist.broadcast(p.data, 0, group=self._dist)
        return self

    def finish_grad_reduction(self):
        if self._dist is None:
            return
        import torch.distributed as dist
        world = dist.get_world_size(self._dist)
        for p in self.parameters():
            if p.grad is not None:
                dist.all_reduce(p.grad, group=self._dist)
                p.grad.div_(world)
