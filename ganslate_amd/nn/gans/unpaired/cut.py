"""CUT / FastCUT recipe — class surface, step order (D first, then G + mlp) and loss assembly of
ganslate/nn/gans/unpaired/cut.py:16-226 on the HIP executors.

Differences that are deliberate: the reference reads `generator.in_channels`, a key that does not
exist (cut.py:83) — `in_out_channels.AB[0]` is used; under data parallelism it reaches for `.encoder` of the DDP
wrapper (cut.py:205-211) — here the encoder passes are partial passes of the same executor, so their weight gradients
land in the same flat buffer and are all-reduced with it; the `mlp` optimiser state is checkpointed too.

The encoder-only passes do not materialise NCHW feature maps: a "patch" is one pixel's channel vector, i.e. one
contiguous row of the NHWC activation, so the sampled patches are gathered straight out of the executor's
activations (Resnet2D.extract_patch_features) and their gradients scattered straight back."""
import os
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np
import torch

from .... import configs
from ...losses.adversarial_loss import AdversarialLoss
from ...losses.functional import fanout, scalar_affine, scalar_sum
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


def _mean0(x):
    """`.mean()` of a criterion's result (cut.py:186-187,197): a 0-d loss is its own mean"""
    return x if torch.is_tensor(x) and x.dim() == 0 else x.mean()


class CUT(BaseGAN):
    # The launch sequence of an iteration is fixed once the patch ids are data: they are drawn on the host in the reference's
    # order before a replay and copied into static index tensors (see _prepare_host_state). With the round-2 kernels a CUT
    # iteration is ~17 ms of GPU work but ~22 ms of launch-by-launch enqueueing, so the captured step is what the GPU's
    # pace is. `use_equivariance_flip` (FastCUT) draws a host coin: the flip itself reads it from device memory
    # (gs_flip_w_if) and the flipped target ids are prepared with the other host state, so the step is captured either way.
    graph_capturable = True

    def __init__(self, conf):
        super().__init__(conf)
        self.external_draw_ids, self._pid_static, self._nce_call = False, None, 0
        self._flip_flag, self._tid_static = None, None
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
            mlp = FeaturePatchMLP(channels, self.num_patches, self.conf.train.gan.mlp_nc)
            mlp.init_weights(self.conf.train.gan.weight_init_type, self.conf.train.gan.weight_init_gain)
            self.networks["mlp"] = mlp

    def init_optimizers(self):
        opt = self.conf.train.gan.optimizer
        betas = (opt.beta1, opt.beta2)
        self.optimizers["G"] = NativeAdam(self.networks["G"].parameters(), lr=opt.lr_G, betas=betas)
        self.optimizers["D"] = NativeAdam(self.networks["D"].parameters(), lr=opt.lr_D, betas=betas)
        self.optimizers["mlp"] = NativeAdam(self.networks["mlp"].parameters(), lr=opt.lr_G, betas=betas)

    def init_criterions(self):
        self.criterion_adv = AdversarialLoss(self.conf.train.gan.optimizer.adversarial_loss_type).to(self.device)
        # PatchNCELoss (cut_losses.py) is fused with the patch MLP into FeaturePatchMLP.nce_loss (csrc/patchnce.hip)
        self.nce_T = self.conf.train.gan.optimizer.nce_T

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
        self.optimizers["mlp"].step()

    def set_input(self, input):
        self.visuals["real_A"] = input["A"].to(self.device, non_blocking=True)
        self.visuals["real_B"] = input["B"].to(self.device, non_blocking=True)

    def forward(self):
        using_idt = self.lambda_nce_idt > 0
        real_A = self.visuals["real_A"]
        real_B = self.visuals["real_B"] if using_idt else None
        if self.use_equivariance_flip and self.is_train:
            if not self.external_draw_ids:              # (a captured / replayed iteration drew it in _prepare_host_state)
                self._draw_flip()
            ops = self.networks["G"].ops
            real_A = ops.flip_w_if(real_A, self._flip_flag)
            if using_idt:
                real_B = ops.flip_w_if(real_B, self._flip_flag)
        G = self.networks["G"]
        if using_idt and self._batched(G):
            # G(real_A) and G(real_B) as one pass over both batches (per-sample InstanceNorm: same images out); the trunk
            # launches then carry 2N images — two tiles per workgroup (csrc/hconvw.hip)
            self.visuals["fake_B"], self.visuals["idt_B"] = G.forward_parts((real_A, real_B))
            return
        self.visuals["fake_B"] = G(real_A)
        if using_idt:
            self.visuals["idt_B"] = G(real_B)

    @staticmethod
    def _batched(G):
        """same-network passes over independent batches run as one pass (GS_CUT_BATCH=0: one pass each, as the reference)"""
        return os.environ.get("GS_CUT_BATCH", "1") != "0" and hasattr(G, "forward_parts") and \
            hasattr(G, "extract_patch_features_parts")

    def backward_D(self):
        real, fake = self.visuals["real_B"], self.visuals["fake_B"]
        D = self.networks["D"]
        if hasattr(D, "forward_parts") and os.environ.get("GS_CUT_BATCH", "1") != "0":
            pred_real, pred_fake = D.forward_parts((real, fake.detach()))      # one pass over both batches (per-sample norm)
        else:
            pred_real, pred_fake = D(real), D(fake.detach())
        loss_real = _mean0(self.criterion_adv(pred_real, True))
        loss_fake = _mean0(self.criterion_adv(pred_fake, False))
        self.losses["D"] = scalar_sum((loss_real, loss_fake))
        self.backward(loss=self.losses["D"], optimizer=self.optimizers["D"], loss_id=0)

    def backward_G_and_mlp(self):
        real_A, real_B = self.visuals["real_A"], self.visuals["real_B"]
        fake_B, idt_B = self.visuals["fake_B"], self.visuals["idt_B"]
        if self.lambda_adv > 0 and self.lambda_nce > 0:
            fake_B, fake_B_nce = fanout(fake_B)     # D and the encoder both send a gradient back: joined by gs_sum2_f32
        else:
            fake_B_nce = fake_B
        # every term with its weight in the combined loss (cut.py:193-227: adversarial * lambda_adv, and with the identity
        # term (1 - lambda_nce_idt) * NCE + lambda_nce_idt * NCE_idt); the logged losses and the combined one are rows of
        # ONE launch (losses/functional.py:scalar_affine)
        xs, combined, named = [], [], {}
        if self.lambda_adv > 0:
            pred_fake = self.networks["D"](fake_B)
            xs.append(_mean0(self.criterion_adv(pred_fake, True)))
            combined.append(self.lambda_adv)
            named["G"] = {len(xs) - 1: self.lambda_adv}
        if self.lambda_nce > 0:
            with_idt = self.lambda_nce_idt > 0
            if with_idt and self._batched(self.networks["G"]):
                # both PatchNCE terms out of ONE encoder pass over (fake_B, idt_B); ids drawn in the reference's order
                nce, nce_idt = self._calculate_nce_losses([(real_A, fake_B_nce), (real_B, idt_B)])
            else:
                nce = self._calculate_nce_loss(real_A, fake_B_nce)
                nce_idt = self._calculate_nce_loss(real_B, idt_B) if with_idt else None
            xs.append(nce)
            combined.append(1 - self.lambda_nce_idt if with_idt else 1.0)
            self.losses["NCE"] = nce
            if with_idt:
                xs.append(nce_idt)
                combined.append(self.lambda_nce_idt)
                named["NCE_idt"] = {len(xs) - 1: self.lambda_nce_idt}
        if not xs:
            combined_loss = 0
        else:
            rows = [[w.get(k, 0.0) for k in range(len(xs))] for w in named.values()] + [combined]
            out = scalar_affine(xs, rows)
            self.losses.update(zip(named, out))
            combined_loss = out[-1]
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

    # ---- captured step: the patch ids are host state ---------------------------------------------------------------
    def _nce_calls_per_step(self):
        return (1 if self.lambda_nce > 0 else 0) + (1 if self.lambda_nce > 0 and self.lambda_nce_idt > 0 else 0)

    def _set_external_host_state(self, on):
        super()._set_external_host_state(on)
        self.external_draw_ids = on

    def _draw_flip(self):
        """the iteration's coin (cut.py:147), kept on the host for the target ids and uploaded for the flip kernel"""
        self.is_flipped = bool(np.random.random() > 0.5)
        if self._flip_flag is None:
            self._flip_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        host = torch.tensor([int(self.is_flipped)], dtype=torch.int32)
        self._flip_flag.copy_(host.pin_memory() if self._flip_flag.is_cuda else host, non_blocking=True)

    def _flipped_ids(self, ids, H, W):
        """patch ids of the target features when the inputs were flipped along W (cut.py:213-215 flips the features back)"""
        G, out = self.networks["G"], []
        for e, pid in zip(self.nce_layers, ids):
            w = G.tap_dims(e, H, W)[1]          # the generator knows its own feature widths
            out.append((pid // w) * w + (w - 1 - pid % w))
        return out

    def _prepare_host_state(self):
        super()._prepare_host_state()
        if self.use_equivariance_flip and self.is_train:
            self._draw_flip()
        H, W = self.visuals["real_A"].shape[-2:]
        drawn = [self.sample_patch_ids(H, W) for _ in range(self._nce_calls_per_step())]      # the step's own draw order
        # (the targets' ids: the same, or mirrored along W when this iteration's inputs are flipped — data of static tensors
        # either way, so the captured launches do not change)
        tgt = [self._flipped_ids(c, H, W) if (self.use_equivariance_flip and self.is_flipped) else c for c in drawn]
        if self._pid_static is None or [[t.shape for t in c] for c in self._pid_static] != [[t.shape for t in c] for c in drawn]:
            self._pid_static = [[t.clone() for t in c] for c in drawn]
            self._tid_static = [[t.clone() for t in c] for c in tgt]
        else:
            for dst, src in zip(self._pid_static + self._tid_static, drawn + tgt):
                for d, t in zip(dst, src):
                    d.copy_(t)
        self._nce_call = 0

    def _rewind_host_state(self):
        super()._rewind_host_state()
        self._nce_call = 0          # (the static id tensors keep this iteration's draws)

    def _calculate_nce_losses(self, pairs):
        """_calculate_nce_loss for several (source, target) pairs with ONE encoder pass over all targets"""
        G, mlp = self.networks["G"], self.networks["mlp"]
        src_feats, tgt_ids = [], []
        for source, _ in pairs:
            H, W = source.shape[-2:]
            if self.external_draw_ids:                        # captured / replayed iteration: ids are static tensors
                ids, tids = self._pid_static[self._nce_call], self._tid_static[self._nce_call]
                self._nce_call += 1
            else:
                ids = self.sample_patch_ids(H, W)
                # target features are flipped back along W before sampling (cut.py:214-215)
                tids = self._flipped_ids(ids, H, W) if self.is_flipped else ids
            src_feats.append(G.extract_patch_features(source, self.nce_layers, ids, detached=True))
            tgt_ids.append(tids)
        tgt_feats = G.extract_patch_features_parts([t for _, t in pairs], self.nce_layers, tgt_ids)
        return [mlp.nce_loss(tf, sf, source.shape[0], self.nce_T, self.lambda_nce)
                for tf, sf, (source, _) in zip(tgt_feats, src_feats, pairs)]

    def _calculate_nce_loss(self, source, target, patch_ids=None):
        G, mlp = self.networks["G"], self.networks["mlp"]
        H, W = source.shape[-2:]
        tgt_ids = None
        if patch_ids is None and self.external_draw_ids:          # captured / replayed iteration: ids are static tensors
            patch_ids, tgt_ids = self._pid_static[self._nce_call], self._tid_static[self._nce_call]
            self._nce_call += 1
        ids = patch_ids if patch_ids is not None else self.sample_patch_ids(H, W)
        # (keys are detached in the loss: read out of this iteration's recorded pass over `source` where there is one)
        source_feats = G.extract_patch_features(source, self.nce_layers, ids, detached=True)
        if tgt_ids is None:       # target features are flipped back along W before sampling (cut.py:214-215)
            tgt_ids = self._flipped_ids(ids, H, W) if self.is_flipped else ids
        target_feats = G.extract_patch_features(target, self.nce_layers, tgt_ids)
        # both MLP passes, the logits, the cross-entropy and the mean over patches and levels: one autograd node
        return mlp.nce_loss(target_feats, source_feats, source.shape[0], self.nce_T, self.lambda_nce)


class _PatchNCEFn(torch.autograd.Function):
    """FeaturePatchMLP + PatchNCELoss over all feature levels as ONE autograd node on the HIP kernels of
    csrc/patchnce.hip: forward = 3 launches (MLP both sides, logits/loss/dL/dF, loss sum), backward = 2 (MLP data
    gradient, parameter gradients). Source (key) features are detached like in the reference (cut_losses.py:16)."""

    @staticmethod
    def forward(ctx, token, mlp, batch, nce_T, lambda_nce, n, *feats):
        target, source = list(feats[:n]), list(feats[n:])
        loss, saved = mlp.ops.patchnce_forward(target, source, mlp.master.detach(), batch=batch, nc=mlp.nc, nce_T=nce_T,
                                               lambda_nce=lambda_nce)
        ctx.mlp, ctx.saved, ctx.n = mlp, saved, n
        ctx.needs = [ctx.needs_input_grad[6 + i] for i in range(n)]
        return scalar_sum(loss.unbind(0)) if loss.numel() > 1 else loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        mlp = ctx.mlp
        if mlp.master.grad is None:
            mlp.master.grad = torch.zeros_like(mlp.master)
        dxq = mlp.ops.patchnce_backward(ctx.saved, mlp.master.detach(), mlp.master.grad if mlp.requires_grad else
                                        torch.zeros_like(mlp.master), grad_scale=g.contiguous().float())
        mlp.grad_dirty = True
        ctx.saved = None
        return (None,) * 6 + tuple(d if need else None for d, need in zip(dxq, ctx.needs)) + (None,) * ctx.n


class FeaturePatchMLP:
    """per-level Linear(C, nc) - ReLU - Linear(nc, nc) - L2 normalise on the sampled patches (cut.py:229-294; the patch
    sampling itself happens at the activation gather). Parameters live in ONE flat fp32 buffer — per level W1 [nc][C],
    b1 [nc], W2 [nc][nc], b2 [nc], the layout of gs_patchnce_* (include/ganslate_hip.h) — updated by NativeAdam like
    the conv networks' masters; state_dict() speaks the reference's key names (`mlps.{i}.{0,2}.{weight,bias}`)."""

    def __init__(self, channels_per_feature, num_patches=256, nc=256):
        from ...native.backend import get_ops
        self.ops = get_ops()
        self.device = self.ops.device
        self.channels, self.num_patches, self.nc = [int(c) for c in channels_per_feature], num_patches, nc
        self.numel = sum(nc * c + nc + nc * nc + nc for c in self.channels)
        self.master = torch.nn.Parameter(torch.zeros(self.numel, dtype=torch.float32, device=self.device))
        self.master.grad = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.master._owner_net = self
        self._token = torch.zeros(1, requires_grad=True, device=self.device)
        self.grad_dirty, self._dist, self.external_reduce, self.training = False, None, False, True

    # ---- nn.Module-like surface -------------------------------------------------------------------------------------
    def parameters(self):
        return [self.master]

    @property
    def requires_grad(self):
        return self.master.requires_grad

    def to(self, device):
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def mark_packs_dirty(self):
        pass

    def _views(self, flat):
        out, off, nc = [], 0, self.nc
        for c in self.channels:
            W1 = flat[off:off + nc * c].view(nc, c); off += nc * c
            b1 = flat[off:off + nc]; off += nc
            W2 = flat[off:off + nc * nc].view(nc, nc); off += nc * nc
            b2 = flat[off:off + nc]; off += nc
            out.append((W1, b1, W2, b2))
        return out

    def init_weights(self, init_type="normal", gain=0.02):
        """ganslate/nn/utils.py:13-36 applied to nn.Linear: N(0, gain) weights, zero biases, drawn on the CPU generator
        in module order (mlps.0.0, mlps.0.2, mlps.1.0, ...)"""
        assert init_type == "normal", "FeaturePatchMLP: only `normal` init is used by the reference configs"
        flat = torch.zeros(self.numel, dtype=torch.float32)
        for W1, b1, W2, b2 in self._views(flat):
            W1.copy_(torch.empty(W1.shape).normal_(0.0, gain))
            W2.copy_(torch.empty(W2.shape).normal_(0.0, gain))
        with torch.no_grad():
            self.master.copy_(flat.to(self.device))

    def reference_parameter_order(self):
        return [f"mlps.{i}.{m}.{k}" for i in range(len(self.channels)) for m in (0, 2) for k in ("weight", "bias")]

    def flat_to_tensors(self, flat):
        out = {}
        for i, (W1, b1, W2, b2) in enumerate(self._views(flat)):
            out.update({f"mlps.{i}.0.weight": W1.clone(), f"mlps.{i}.0.bias": b1.clone(),
                        f"mlps.{i}.2.weight": W2.clone(), f"mlps.{i}.2.bias": b2.clone()})
        return out

    def tensors_to_flat(self, tensors, flat):
        host = torch.zeros(self.numel, dtype=torch.float32)
        for i, views in enumerate(self._views(host)):
            for v, key in zip(views, ("0.weight", "0.bias", "2.weight", "2.bias")):
                v.copy_(tensors[f"mlps.{i}.{key}"].detach().float().cpu().reshape(v.shape))
        with torch.no_grad():
            flat.copy_(host.to(flat.device))

    def state_dict(self):
        return self.flat_to_tensors(self.master.detach())

    def load_state_dict(self, sd, strict=True):
        self.tensors_to_flat(sd, self.master)

    # ---- the loss -----------------------------------------------------------------------------------------------------
    def nce_loss(self, target_feats, source_feats, batch, nce_T, lambda_nce):
        """sum over the levels of mean_patches(PatchNCE(mlp(target), mlp(source))) * lambda_nce / len(levels)
        (cut.py:218-226)"""
        n = len(target_feats)
        return _PatchNCEFn.apply(self._token, self, batch, float(nce_T), float(lambda_nce), n, *target_feats,
                                 *[f.detach() for f in source_feats])

    # ---- data parallelism (the reference never gets this far: SURVEY.md §2.4) --------------------------------------------
    def parallelize(self, process_group=None):
        import torch.distributed as dist
        self._dist = process_group if process_group is not None else dist.group.WORLD
        with torch.no_grad():
            dist.broadcast(self.master.data, 0, group=self._dist)
        return self

    def flush_deferred_wgrads(self):
        pass

    def finish_grad_reduction(self):
        """called by NativeAdam before the update; returns the factor the summed gradient is scaled by"""
        if self._dist is None:
            return 1.0
        import torch.distributed as dist
        if not self.external_reduce and self.grad_dirty:
            dist.all_reduce(self.master.grad, op=dist.ReduceOp.SUM, group=self._dist)
        return 1.0 / dist.get_world_size(self._dist)
