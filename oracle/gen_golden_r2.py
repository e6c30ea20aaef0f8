"""Round-2 golden vectors from the REAL reference (/root/reference through oracle/ref_stubs), build container only:

    python -m oracle.gen_golden_r2 grads       # tests/golden/cyclegan_grads.json
    python -m oracle.gen_golden_r2 envelope    # tests/golden/envelope.json
    python -m oracle.gen_golden_r2 fullsize    # tests/golden/fullsize.json
    python -m oracle.gen_golden_r2 advmodes    # tests/golden/adv_modes.json
    python -m oracle.gen_golden_r2 vnet2d      # tests/golden/vnet2d.json
    python -m oracle.gen_golden_r2 revgan      # tests/golden/revgan.json
    python -m oracle.gen_golden_r2 volpatch    # tests/golden/volume_patches.json
    python -m oracle.gen_golden_r2 multiscale  # tests/golden/multiscale_patchgan3d.json
    python -m oracle.gen_golden_r2 recipegrads [case ...] # tests/golden/recipe_grads.json (all cases, or only the named SA_CASES)
    python -m oracle.gen_golden_r2 selfattention # tests/golden/selfattention.json

* cyclegan_grads.json — the parameter gradients `CycleGAN.optimize_parameters` (cyclegan.py:92-124) leaves in `.grad`
  after its first iteration (G gradients from backward_G :191-214, D gradients summed over backward_D("D_B") and
  ("D_A") :154-189): per tensor the L2 norm and 8 strided samples, plus the first two iterations' losses. Cases:
  `c64_default` (the 64x64 case of cyclegan_steps.json) and `cfg2_256_b8` (BASELINE configs[1] shape: 256x256, batch 8).
* recipe_grads.json — the same record (per-tensor `.grad` norm + 8 strided samples after the FIRST iteration, and that
  iteration's losses) for the other recipes: Pix2Pix (`p2p_64x128` and BASELINE configs[2] at full width, dropout off),
  CUT (`cut_64`: G, D and the patch MLP), 3-D CycleGAN (`v32_default`, `vnet_16x32x32`), RevGAN (`rev3d_16x32x32`,
  `rev3d_piresnet`) — the oracle's step classes are pinned to it, and the GPU tests compare the HIP gradients with the
  oracle's full tensors (tests/test_recipe_gradients_*.py).
* envelope.json — the reference against ITSELF: the same 100 iterations (64x64, batch 2, horse2zebra hyper-parameters)
  run with 1 and with 8 intra-op threads. The arithmetic is identical, only the summation order of MKL-DNN's
  reductions differs; the gap between the two curves is the floor no other implementation can be asked to beat and
  the tolerances of the step tests are derived from it (tests/envelope.py).
* adv_modes.json — the objectives of AdversarialLoss other than lsgan (adversarial_loss.py:26-34,60-67): its value and
  input gradient on a seeded discriminator map, and four CycleGAN iterations with `adversarial_loss_type` vanilla /
  wgangp. (`nonsaturating` cannot be recorded: the reference's branch raises NameError, :68-73.)
* volume_patches.json — the 3-D training-patch path of the volume datasets (projects/brats_mri_sequence_translation/
  datasets/train_dataset.py:76-85): `StochasticFocalPatchSampler.get_patch_pair` (data/utils/stochastic_focal_patching.py)
  under `random.seed(s)` — the patch start coordinates it draws for volumes of given shapes, several draws per case — and
  `z_score_normalize(patch, scale_to_range=(-1, 1))` (data/utils/normalization.py:18-30) of the first pair's patches on
  seeded volumes (samples + moments), plus `min_max_normalize` and `z_score_normalize_with_precomputed_stats`.
"""
import json
import random
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"

GRAD_CASES = {
    "c64_default": dict(size=64, batch=2, steps=2, n_iters=20, n_iters_decay=10, pool_size=50,
                        lambda_identity=0.0, proportion_ssim=0.0, seed=11),
    # BASELINE configs[1]: horse2zebra CycleGAN ResNet-9 256x256 batch 8 (yaml hyper-parameters)
    "cfg2_256_b8": dict(size=256, batch=8, steps=2, n_iters=100, n_iters_decay=100, pool_size=50,
                        lambda_identity=0.0, proportion_ssim=0.0, seed=14),
}
ENVELOPE_CASE = dict(size=64, batch=2, steps=100, n_iters=50, n_iters_decay=50, pool_size=50,
                     lambda_identity=0.0, proportion_ssim=0.0, seed=11)


def _model(c):
    from oracle import gen_golden as G          # imports the reference
    from oracle.torch_ref import seeded_state_dict
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    model = G.CycleGAN(G.make_conf(c))
    for k, (n, net) in enumerate(model.networks.items()):
        net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
    random.seed(c["seed"])
    return model, G.inputs


def _record(model):
    lrs, losses, visuals, metrics = model.get_loggable_data()
    return {"lrs": {k: float(v) for k, v in lrs.items()},
            "losses": {k: float(v) for k, v in losses.items() if v is not None},
            "metrics": {k: float(v) for k, v in metrics.items() if v is not None}}


def grad_case(name, c):
    model, inputs = _model(c)
    rec, grads = [], None
    for s in range(c["steps"]):
        A, B = inputs(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        rec.append(_record(model))
        if s == 0:
            grads = {}
            for net_name, net in model.networks.items():
                per = {}
                for n, p in net.named_parameters(remove_duplicate=False):
                    if n.startswith("encoder."):      # Resnet2D registers the same tensors twice (resnet2d.py:46)
                        continue
                    flat = p.grad.detach().flatten()
                    idx = torch.linspace(0, flat.numel() - 1, min(8, flat.numel())).long()
                    per[n] = {"norm": float(flat.double().norm()), "idx": [int(i) for i in idx],
                              "samples": [float(v) for v in flat[idx]]}
                grads[net_name] = per
        model.update_learning_rate()
        print(name, s, rec[-1]["losses"], flush=True)
    return {"config": c, "steps": rec, "step0_grads": grads}


def envelope_run(threads):
    torch.set_num_threads(threads)
    model, inputs = _model(ENVELOPE_CASE)
    rec = []
    for s in range(ENVELOPE_CASE["steps"]):
        A, B = inputs(ENVELOPE_CASE, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        rec.append(_record(model))
        model.update_learning_rate()
    return rec


def fullsize():
    """BASELINE configs[3] / [4] shapes the round-1 goldens only covered at 64x64 / 16x32x32: CUT at 256x256 batch 2 and
    the brats-yaml 3-D CycleGAN (Vnet3D + PatchGAN3D n_layers 2) at 64^3 -> tests/golden/fullsize.json"""
    from oracle import gen_golden as G
    torch.set_num_threads(8)
    out = {}
    c = dict(size=256, batch=2, steps=2, n_iters=100, n_iters_decay=100, num_patches=256, seed=42)
    out["cut_256_b2"] = G.run_cut_case("cut_256_b2", c)
    c = dict(size=[64, 64, 64], batch=1, steps=2, n_iters=100, n_iters_decay=100, pool_size=50, lambda_identity=0.0,
             proportion_ssim=0.0, d_layers=2, seed=54,
             vnet=dict(first_layer_channels=16, down_blocks=[2, 2, 3], up_blocks=[3, 3, 3]))
    out["vnet_64"] = G.run_case_3d("vnet_64", c)
    (OUT / "fullsize.json").write_text(json.dumps(out, indent=1))


ADV_CASES = {
    "c64_vanilla": dict(size=64, batch=2, steps=4, n_iters=20, n_iters_decay=10, pool_size=50, lambda_identity=0.0,
                        proportion_ssim=0.0, seed=15, adv="vanilla"),
    "c64_wgangp": dict(size=64, batch=2, steps=4, n_iters=20, n_iters_decay=10, pool_size=50, lambda_identity=0.0,
                       proportion_ssim=0.0, seed=16, adv="wgangp"),
}


def adv_pred(seed=21):
    """the seeded discriminator map the op-level vectors are taken on: logits of a few units, both signs"""
    return torch.randn(8, 1, 30, 30, generator=torch.Generator().manual_seed(seed)) * 3.0


def advmodes():
    from oracle import gen_golden as G          # imports the reference
    from ganslate.nn.losses.adversarial_loss import AdversarialLoss
    torch.set_num_threads(8)
    ops = {}
    for mode in ("lsgan", "vanilla", "wgangp"):
        crit = AdversarialLoss(mode)
        for real in (True, False):
            x = adv_pred().requires_grad_()
            val = crit(x, real)
            (g,) = torch.autograd.grad(val, x)
            flat = g.flatten()
            idx = torch.linspace(0, flat.numel() - 1, 8).long()
            ops[f"{mode}_{'real' if real else 'fake'}"] = {
                "loss": float(val), "grad_norm": float(flat.double().norm()), "idx": [int(i) for i in idx],
                "grad_samples": [float(v) for v in flat[idx]]}
        # dict of predictions: mean over the keys (adversarial_loss.py:91-94)
        d = {"a": adv_pred(22), "b": adv_pred(23)[:, :, :7, :7]}
        ops[f"{mode}_dict_real"] = {"loss": float(crit(d, True))}
    steps = {n: G.run_case(n, c) for n, c in ADV_CASES.items()}
    (OUT / "adv_modes.json").write_text(json.dumps({"ops": ops, "steps": steps}, indent=1))


def vnet2d():
    """the reference's Vnet2D (nn/generators/vnet/vnet2d.py) with use_memory_saving=False, use_inverse=False over the memcnn
    stand-in: forward output, input gradient and every parameter-gradient norm (gen_golden.net_case)"""
    from oracle import gen_golden as G          # imports the reference
    from ganslate.nn.generators.vnet.vnet2d import Vnet2D
    torch.set_num_threads(8)
    out = {
        "vnet2d_default_blocks": G.net_case("v2d", Vnet2D(2, 3, "instance", 16, (1, 2, 3, 2), (2, 2, 1, 1), False, False),
                                            (2, 2, 32, 48), 67),
        "vnet2d_1ch_small": G.net_case("v2s", Vnet2D(1, 1, "instance", 8, (1, 2), (2, 1), False, False),
                                       (1, 1, 16, 24), 68),
    }
    (OUT / "vnet2d.json").write_text(json.dumps(out, indent=1))


REVGAN_CASES = {
    # RevGAN (nn/gans/unpaired/revgan.py) with ONE partially-invertible V-Net used in both directions and the CycleGAN losses
    "rev3d_16x32x32": dict(size=[16, 32, 32], batch=1, steps=3, n_iters=100, n_iters_decay=100, pool_size=50,
                           lambda_identity=0.0, proportion_ssim=0.0, d_layers=2, seed=55, dims=3,
                           vnet=dict(first_layer_channels=8, down_blocks=[1, 2], up_blocks=[2, 1])),
    "rev3d_piresnet": dict(size=[16, 32, 32], batch=1, steps=3, n_iters=100, n_iters_decay=100, pool_size=50,
                           lambda_identity=0.0, proportion_ssim=0.0, d_layers=2, seed=57, dims=3,
                           piresnet=dict(first_layer_channels=16, depth=2)),
    "rev2d_64x64_idt": dict(size=[64, 64], batch=2, steps=3, n_iters=100, n_iters_decay=100, pool_size=50,
                            lambda_identity=0.5, proportion_ssim=0.0, d_layers=2, seed=56, dims=2,
                            vnet=dict(first_layer_channels=8)),
}


def _both_directions_case(net, x_shape, seed):
    """y = net(x); r = net(y, inverse=True); gradients of sum(y*gy) + sum(r*gr)"""
    from oracle.torch_ref import seeded_state_dict
    net.load_state_dict(seeded_state_dict(net, seed))
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(x_shape, generator=g) * 2 - 1).requires_grad_()
    y = net(x)
    r = net(y, inverse=True)
    gy, gr = torch.randn(y.shape, generator=g), torch.randn(r.shape, generator=g)
    ((y * gy).sum() + (r * gr).sum()).backward()
    idx = torch.linspace(0, y.numel() - 1, 32).long()
    return {"seed": seed, "x_shape": list(x_shape), "state_dict_keys": list(net.state_dict().keys()),
            "n_params": sum(p.numel() for p in net.parameters()), "sample_idx": [int(i) for i in idx],
            "y_samples": [float(v) for v in y.detach().flatten()[idx]],
            "r_samples": [float(v) for v in r.detach().flatten()[idx]],
            "y_abs_sum": float(y.detach().double().abs().sum()), "r_abs_sum": float(r.detach().double().abs().sum()),
            "x_grad_abs_sum": float(x.grad.double().abs().sum()),
            "param_grad_norms": {n: float(p.grad.norm()) for n, p in net.named_parameters()}}


def _revgan_model(c):
    """the reference's RevGAN for a REVGAN_CASES entry with the seeded weights loaded -> (model, input shape)"""
    from oracle import gen_golden as G          # imports the reference
    from omegaconf import DictConfig
    from ganslate.nn.gans.unpaired.revgan import RevGAN
    from oracle.torch_ref import seeded_state_dict
    conf = G.make_conf(c)
    conf.train.metrics["ssim"] = False
    gan = conf.train.gan
    gan["_target_"] = "ganslate.nn.gans.unpaired.RevGAN"
    if "piresnet" in c:
        gan["generator"] = DictConfig({"_target_": "ganslate.nn.generators.Piresnet3D", "use_memory_saving": True,
                                       "use_inverse": True, "depth": c["piresnet"]["depth"],
                                       "first_layer_channels": c["piresnet"]["first_layer_channels"],
                                       "in_out_channels": {"AB": [1, 1]}})
        gan["discriminator"] = DictConfig({"_target_": "ganslate.nn.discriminators.PatchGAN3D", "ndf": 64,
                                           "n_layers": c["d_layers"], "kernel_size": [4, 4, 4],
                                           "in_channels": {"B": 1, "A": 1}})
        shape = (c["batch"], 1, *c["size"])
    elif c["dims"] == 3:
        gan["generator"] = DictConfig({"_target_": "ganslate.nn.generators.Vnet3D", "use_memory_saving": True,
                                       "use_inverse": True, "is_separable": False,
                                       "first_layer_channels": c["vnet"]["first_layer_channels"],
                                       "down_blocks": c["vnet"]["down_blocks"], "up_blocks": c["vnet"]["up_blocks"],
                                       "in_out_channels": {"AB": [1, 1]}})
        gan["discriminator"] = DictConfig({"_target_": "ganslate.nn.discriminators.PatchGAN3D", "ndf": 64,
                                           "n_layers": c["d_layers"], "kernel_size": [4, 4, 4],
                                           "in_channels": {"B": 1, "A": 1}})
        shape = (c["batch"], 1, *c["size"])
    else:
        gan["generator"] = DictConfig({"_target_": "ganslate.nn.generators.Vnet2D", "use_memory_saving": True,
                                       "use_inverse": True, "first_layer_channels": c["vnet"]["first_layer_channels"],
                                       "in_out_channels": {"AB": [2, 2]}})
        gan["discriminator"] = DictConfig({"_target_": "ganslate.nn.discriminators.PatchGAN2D", "ndf": 64,
                                           "n_layers": c["d_layers"], "kernel_size": [4, 4],
                                           "in_channels": {"B": 2, "A": 2}})
        shape = (c["batch"], 2, *c["size"])
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    model = RevGAN(conf)
    for k, (n, net) in enumerate(model.networks.items()):
        net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
    random.seed(c["seed"])
    return model, shape


def revgan():
    """the reference's RevGAN and its generators' inverse direction over the memcnn stand-in (oracle/ref_stubs/memcnn:
    additive coupling forward / inverse with plain autograd — what memcnn's memory saving recomputes, it does not change)"""
    from oracle import gen_golden as G          # imports the reference
    from omegaconf import DictConfig
    from ganslate.nn.gans.unpaired.revgan import RevGAN
    from ganslate.nn.generators.vnet.vnet2d import Vnet2D
    from ganslate.nn.generators.vnet.vnet3d import Vnet3D
    from oracle.torch_ref import seeded_state_dict
    torch.set_num_threads(8)
    from ganslate.nn.generators.resnet.piresnet3d import Piresnet3D
    nets = {
        "piresnet3d": _both_directions_case(Piresnet3D(1, 1, "instance", 3, 16, True, True), (1, 1, 8, 12, 16), 71),
        "vnet3d_inverse": _both_directions_case(Vnet3D(1, 1, "instance", 8, (1, 2), (2, 1), True, True, False),
                                                (1, 1, 8, 12, 16), 69),
        "vnet2d_inverse_default_blocks": _both_directions_case(Vnet2D(2, 2, "instance", 8), (1, 2, 64, 96), 70),
    }
    steps = {}
    for name, c in REVGAN_CASES.items():
        model, shape = _revgan_model(c)
        rec = []
        for s_ in range(c["steps"]):
            g = torch.Generator().manual_seed(c["seed"] * 100 + s_)
            A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
            model.set_input({"A": A, "B": B})
            model.optimize_parameters()
            rec.append(_record(model))
            model.update_learning_rate()
            print(name, s_, rec[-1]["losses"], flush=True)
        norms = {n: float(torch.sqrt(sum((p.detach() ** 2).sum() for p in net.parameters())))
                 for n, net in model.networks.items()}
        steps[name] = {"config": c, "steps": rec, "final_param_norms": norms,
                       "network_names": list(model.networks.keys())}
    (OUT / "revgan.json").write_text(json.dumps({"nets": nets, "steps": steps}, indent=1))


VOLPATCH_CASES = {
    # name: (shape_A, shape_B, patch_size, focal_region_proportion, seed, draws)
    "brats_like": ((20, 36, 30), (24, 33, 31), (8, 16, 16), 0.0, 5, 4),
    "focal_0.2": ((20, 36, 30), (24, 33, 31), (8, 16, 16), 0.2, 6, 4),
    "focal_0.5_tight": ((9, 18, 17), (8, 16, 20), (8, 16, 16), 0.5, 7, 4),
    "patch_2d": ((5, 40, 44), (6, 41, 39), (32, 32), 0.3, 8, 3),
    "exact_fit": ((8, 16, 16), (8, 16, 16), (8, 16, 16), 0.4, 9, 2),
}


def _seeded_volume(shape, seed):
    """MRI-like intensities: non-negative, skewed, with a zero background slab (numpy Generator: stable across versions)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    v = rng.gamma(2.0, 180.0, size=shape).astype(np.float32)
    v[: max(1, shape[0] // 5)] = 0.0
    return torch.from_numpy(np.round(v))


def volpatch():
    from oracle import gen_golden as G          # noqa: F401  (puts the reference on sys.path)
    import numpy as np
    from ganslate.data.utils.normalization import (min_max_normalize, z_score_normalize,
                                                   z_score_normalize_with_precomputed_stats)
    from ganslate.data.utils.stochastic_focal_patching import StochasticFocalPatchSampler
    out = {}
    for name, (sa, sb, ps, frp, seed, draws) in VOLPATCH_CASES.items():
        A, B = _seeded_volume(sa, seed), _seeded_volume(sb, seed + 100)
        sampler = StochasticFocalPatchSampler(np.array(ps), frp)
        random.seed(seed)
        recs = []
        for d in range(draws):
            st = random.getstate()
            pa, pb = sampler.get_patch_pair(A, B)
            # recover the starts the sampler drew: replay its two public steps under the same RNG state
            random.setstate(st)
            za, xa, ya = sampler.pick_random_start(A)
            rel = sampler.calculate_relative_focal_point(za, xa, ya, A)
            zb, xb, yb = sampler.pick_stochastic_focal_start(B, rel)
            full = [1, *ps] if len(ps) == 2 else list(ps)
            assert torch.equal(pa.reshape(full), A[za:za + full[0], xa:xa + full[1], ya:ya + full[2]])
            assert torch.equal(pb.reshape(full), B[zb:zb + full[0], xb:xb + full[1], yb:yb + full[2]])
            rec = {"start_A": [int(za), int(xa), int(ya)], "start_B": [int(zb), int(xb), int(yb)],
                   "shape": list(pa.shape)}
            if d == 0:
                for key, patch in (("A", pa), ("B", pb)):
                    z = z_score_normalize(patch.clone(), scale_to_range=(-1, 1))
                    flat = z.flatten()
                    idx = torch.linspace(0, flat.numel() - 1, 16).long()
                    rec["z_" + key] = {"samples_at": idx.tolist(), "samples": flat[idx].tolist(),
                                       "mean": float(flat.double().mean()), "sq": float((flat.double() ** 2).mean()),
                                       "min": float(flat.min()), "max": float(flat.max()),
                                       "patch_mean": float(patch.mean()), "patch_std": float(patch.std())}
                    plain = z_score_normalize(patch.clone())
                    rec["z_plain_" + key] = plain.flatten()[idx].tolist()
                mm = min_max_normalize(pa.clone(), 0.0, 1500.0).flatten()
                rec["minmax_A"] = mm[idx].tolist()
                pre = z_score_normalize_with_precomputed_stats(pa.clone(), (210.0, 95.0), original_scale=(0.0, 1800.0),
                                                               scale_to_range=(-1, 1)).flatten()
                rec["precomputed_A"] = pre[idx].tolist()
            recs.append(rec)
        out[name] = {"shape_A": list(sa), "shape_B": list(sb), "patch_size": list(ps), "focal_region_proportion": frp,
                     "seed": seed, "draws": recs}
    (OUT / "volume_patches.json").write_text(json.dumps({"torch": torch.__version__, "cases": out}, indent=1))


MULTISCALE_CASES = {
    # name: (in_channels, ndf, n_layers, scales, batch, (D, H, W), seed)
    "s2_default_layers": (1, 4, 3, 2, 1, (64, 64, 72), 3),
    "s2_two_layers": (1, 8, 2, 2, 2, (32, 40, 48), 5),
    "s3_one_layer": (2, 8, 1, 3, 1, (24, 36, 30), 4),
}


def multiscale():
    """the REAL MultiScalePatchGAN3D (multiscale_patchgan3d.py:44-60, over the monai stand-in) with seeded weights: the
    per-scale maps, input gradient and parameter-gradient norms of sum_s mean(map_s^2) under random.seed(seed)"""
    from oracle import gen_golden as G          # noqa: F401  (puts the reference on sys.path)
    from ganslate.nn.discriminators.patchgan.multiscale_patchgan3d import MultiScalePatchGAN3D
    from oracle.torch_ref import seeded_state_dict
    out = {}
    for name, (cin, ndf, nl, scales, B, dims, seed) in MULTISCALE_CASES.items():
        net = MultiScalePatchGAN3D(cin, ndf, nl, (4, 4, 4), scales, "instance")
        net.load_state_dict(seeded_state_dict(net, seed))
        g = torch.Generator().manual_seed(seed)
        x = (torch.rand((B, cin, *dims), generator=g) * 2 - 1).requires_grad_()
        random.seed(seed)
        maps = net(x)
        loss = sum((m ** 2).mean() for m in maps.values())
        loss.backward()
        rec = {"config": dict(in_channels=cin, ndf=ndf, n_layers=nl, scales=scales, batch=B, dims=list(dims), seed=seed),
               "keys": list(net.state_dict().keys()), "maps": {}, "loss": float(loss),
               "input_grad": {"norm": float(x.grad.norm()), "nonzero": int((x.grad != 0).sum())},
               "param_grad_norms": {k: float(p.grad.norm()) for k, p in net.named_parameters()}}
        for s, m in maps.items():
            flat = m.detach().flatten()
            idx = torch.linspace(0, flat.numel() - 1, 12).long()
            rec["maps"][s] = {"shape": list(m.shape), "samples_at": idx.tolist(), "samples": flat[idx].tolist(),
                              "norm": float(flat.norm())}
        out[name] = rec
    (OUT / "multiscale_patchgan3d.json").write_text(json.dumps(out, indent=1))


def _grad_record(model):
    grads = {}
    for net_name, net in model.networks.items():
        per = {}
        for n, p in net.named_parameters(remove_duplicate=False):
            if n.startswith("encoder.") or p.grad is None:      # Resnet2D registers the same tensors twice (resnet2d.py:46)
                continue
            flat = p.grad.detach().flatten()
            k = min(8, flat.numel())          # (integer arithmetic: float32 linspace overshoots on the 33 M-element layers)
            idx = torch.tensor([(flat.numel() - 1) * i // max(k - 1, 1) for i in range(k)])
            per[n] = {"norm": float(flat.double().norm()), "numel": flat.numel(), "idx": [int(i) for i in idx],
                      "samples": [float(v) for v in flat[idx]]}
        grads[net_name] = per
    return grads


P2P_FULL = dict(size=[256, 512], batch=1, steps=1, n_iters=100, n_iters_decay=100, num_downs=7, ngf=128,
                use_dropout=False, n_layers=4, lambda_pix2pix=30.0, seed=35)


# CycleGAN over the self-attention networks (round 4): SelfAttentionVnet3D(8; down 1,2; up 2,1; a block on both down
# levels) + SelfAttentionPatchGAN3D(ndf 16, 2 layers) — tests/configs/cyclegan_selfattention_synthetic.yaml
SA_CASES = {
    "sa_32x48x48": dict(size=[32, 48, 48], batch=1, steps=1, n_iters=100, n_iters_decay=100, pool_size=50,
                        lambda_identity=0.0, proportion_ssim=0.0, d_layers=2, seed=81,
                        sa=dict(first_layer_channels=8, down_blocks=[1, 2], up_blocks=[2, 1],
                                enable_attention_block=[True, True], ndf=16)),
}


def recipegrads(only=()):
    """one iteration of the reference's recipes; what it leaves in every parameter's .grad. `only`: case names to (re)generate
    into the existing file (the others keep their recorded bytes)"""
    from oracle import gen_golden as G
    from oracle.torch_ref import seeded_state_dict
    torch.set_num_threads(8)
    out = {}
    if only:
        out = json.loads((OUT / "recipe_grads.json").read_text())
        for name in only:
            c = SA_CASES[name]
            torch.manual_seed(c["seed"])
            random.seed(c["seed"])
            model = G.CycleGAN(G.make_conf_3d(c))
            for k, (n, net) in enumerate(model.networks.items()):
                net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
            random.seed(c["seed"])
            model.set_input(dict(zip("AB", G.inputs_3d(c, 0))))
            model.optimize_parameters()
            out[name] = {"kind": "cyclegan3d", "config": c, "losses": _record(model)["losses"],
                         "step0_grads": _grad_record(model)}
            print(name, out[name]["losses"], flush=True)
        (OUT / "recipe_grads.json").write_text(json.dumps(out, indent=1))
        return

    def one_step(name, kind, c, model, A, B):
        if kind == "cut":
            torch.manual_seed(1000)          # pins the torch.randperm patch ids of the step (as in cut_steps.json)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        out[name] = {"kind": kind, "config": c, "losses": _record(model)["losses"], "step0_grads": _grad_record(model)}
        print(name, out[name]["losses"], flush=True)

    for name, c in list(G.PIX2PIX_CASES.items())[:1] + [("p2p_cfg3_full", P2P_FULL)]:
        torch.manual_seed(c["seed"])
        model = G.Pix2PixConditionalGAN(G.make_pix2pix_conf(c))
        for k, (n, net) in enumerate(model.networks.items()):
            net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
        one_step(name, "pix2pix", c, model, *G.p2p_inputs(c, 0))
    for name, c in G.CUT_CASES.items():
        torch.manual_seed(c["seed"])
        model = G.CUT(G.make_cut_conf(c))
        for k, (n, net) in enumerate(model.networks.items()):
            net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
        one_step(name, "cut", c, model, *G.inputs(c, 0))
    for name in ("v32_default", "vnet_16x32x32"):
        c = G.CASES_3D[name]
        torch.manual_seed(c["seed"])
        random.seed(c["seed"])
        model = G.CycleGAN(G.make_conf_3d(c))
        for k, (n, net) in enumerate(model.networks.items()):
            net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
        random.seed(c["seed"])
        one_step(name, "cyclegan3d", c, model, *G.inputs_3d(c, 0))
    for name in ("rev3d_16x32x32", "rev3d_piresnet"):
        c = REVGAN_CASES[name]
        model, shape = _revgan_model(c)
        g = torch.Generator().manual_seed(c["seed"] * 100)
        A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
        one_step(name, "revgan", c, model, A, B)
    (OUT / "recipe_grads.json").write_text(json.dumps(out, indent=1))


def selfattention():
    """the reference's self-attention networks (nn/attention.py through selfattention_patchgan3d.py:18-79 and
    selfattention_vnet3d.py:44-181, the latter over the memcnn stand-in) on seeded weights and inputs: output samples, input
    gradient and every parameter's gradient norm (the record of oracle/gen_golden.net_case)"""
    from oracle import gen_golden as G          # imports the reference
    from ganslate.nn.discriminators.patchgan.selfattention_patchgan3d import SelfAttentionPatchGAN3D
    from ganslate.nn.generators.vnet.selfattention_vnet3d import SelfAttentionVnet3D
    torch.set_num_threads(8)
    nets = {
        "sa_patchgan3d_64": G.net_case("sapg", SelfAttentionPatchGAN3D(1, 32, 3, 4, "instance"), (1, 1, 64, 64, 64), 75),
        "sa_patchgan3d_2ch_2layers": G.net_case("sapg2", SelfAttentionPatchGAN3D(2, 16, 2, 4, "instance"),
                                                (2, 2, 34, 40, 46), 76),
        "sa_vnet3d_small": G.net_case("savn", SelfAttentionVnet3D(1, 1, "instance", 8, (1, 2), (2, 1), False, False,
                                                                  (True, True), False), (1, 1, 8, 16, 16), 77),
        "sa_vnet3d_default_flags": G.net_case("savn2", SelfAttentionVnet3D(1, 1, "instance", 8, (1, 1, 2, 1), (1, 2, 1, 1),
                                                                           False, False, (False, False, True, True), False),
                                              (1, 1, 16, 32, 32), 78),
    }
    (OUT / "selfattention.json").write_text(json.dumps(nets, indent=1))


def patchnce_class():
    """the reference's stand-alone criterion (nn/losses/cut_losses.py:5-43) on seeded, L2-normalised features"""
    from oracle import gen_golden as G          # imports the reference  # noqa: F401
    from ganslate.nn.losses.cut_losses import PatchNCELoss
    from omegaconf import DictConfig
    out = {}
    for name, (batch, patches, dim, T, seed) in {"b2_p16_d32": (2, 16, 32, 0.07, 91), "b3_p8_d256": (3, 8, 256, 0.1, 92)}.items():
        conf = DictConfig({"train": {"batch_size": batch, "gan": {"optimizer": {"nce_T": T}}}})
        g = torch.Generator().manual_seed(seed)
        q = torch.nn.functional.normalize(torch.randn(batch * patches, dim, generator=g), dim=1).requires_grad_()
        k = torch.nn.functional.normalize(torch.randn(batch * patches, dim, generator=g), dim=1)
        loss = PatchNCELoss(conf)(q, k)
        loss.sum().backward()
        out[name] = {"batch": batch, "patches": patches, "dim": dim, "nce_T": T, "seed": seed,
                     "loss": [float(v) for v in loss.detach()], "grad_q_norm": float(q.grad.norm()),
                     "grad_q_samples": [float(v) for v in q.grad.flatten()[::max(1, q.grad.numel() // 8)][:8]]}
    (OUT / "patchnce_class.json").write_text(json.dumps(out, indent=1))


def main():
    what = sys.argv[1]
    if what == "patchnce_class":
        return patchnce_class()
    if what == "multiscale":
        multiscale()
    elif what == "recipegrads":
        recipegrads(tuple(sys.argv[2:]))      # e.g. `recipegrads sa_32x48x48`: add / refresh single cases
    elif what == "selfattention":
        selfattention()
    elif what == "volpatch":
        volpatch()
    elif what == "fullsize":
        fullsize()
    elif what == "revgan":
        revgan()
    elif what == "vnet2d":
        vnet2d()
    elif what == "advmodes":
        advmodes()
    elif what == "grads":
        torch.set_num_threads(8)
        out = {n: grad_case(n, c) for n, c in GRAD_CASES.items()}
        (OUT / "cyclegan_grads.json").write_text(json.dumps(out, indent=1))
    elif what == "envelope-run":            # one curve, printed as JSON (child process: thread pools are per process)
        print("CURVE" + json.dumps(envelope_run(int(sys.argv[2]))))
    elif what == "envelope":
        curves = {}
        for t in (1, 8):
            r = subprocess.run([sys.executable, "-m", "oracle.gen_golden_r2", "envelope-run", str(t)], cwd=ROOT,
                               capture_output=True, text=True, check=True)
            curves[f"threads_{t}"] = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("CURVE"))[5:])
        (OUT / "envelope.json").write_text(json.dumps({"config": ENVELOPE_CASE, "torch": torch.__version__,
                                                       **curves}, indent=1))
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
