"""Generates tests/golden/*.json by running the REAL reference (/root/reference, imported through the stand-in
modules under oracle/ref_stubs) on the CPU. Only runs in the build container; the fixtures it writes are
data (seeds, scalars, a few samples) — no reference source travels.

    python -m oracle.gen_golden            # rewrites tests/golden/cyclegan_steps.json, nets.json

Inputs are U(-1,1) from torch.Generator(seed); weights come from oracle.torch_ref.seeded_state_dict so the
restatement and the HIP nets can be given bit-identical parameters without storing them.
"""
import json
import random
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle" / "ref_stubs"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, str(ROOT))

from omegaconf import DictConfig  # noqa: E402  (the stand-in)
import ganslate.configs.base  # noqa: E402,F401
from ganslate.nn.gans.unpaired.cyclegan import CycleGAN  # noqa: E402
from ganslate.nn.generators import Resnet2D, Unet2D  # noqa: E402
from ganslate.nn.gans.paired.pix2pix import Pix2PixConditionalGAN  # noqa: E402
from ganslate.nn.gans.unpaired.cut import CUT  # noqa: E402
from ganslate.nn.discriminators import PatchGAN2D  # noqa: E402

from oracle.torch_ref import seeded_state_dict  # noqa: E402

CASES = {
    # horse2zebra yaml hyper-parameters (projects/horse2zebra/experiments/default.yaml:28-53) at 64x64
    "c64_default": dict(size=64, batch=2, steps=30, n_iters=20, n_iters_decay=10, pool_size=50,
                        lambda_identity=0.0, proportion_ssim=0.0, seed=11),
    # first-run template values: SSIM-weighted cycle loss + identity loss
    "c64_idt_ssim": dict(size=64, batch=1, steps=5, n_iters=100, n_iters_decay=100, pool_size=50,
                         lambda_identity=0.5, proportion_ssim=0.84, seed=12),
    # BASELINE config 1 shape: 256x256, batch 1
    "cfg1_256": dict(size=256, batch=1, steps=2, n_iters=100, n_iters_decay=100, pool_size=50,
                     lambda_identity=0.0, proportion_ssim=0.0, seed=13),
}


PIX2PIX_CASES = {
    # cityscapes pix2pix yaml hyper-parameters (projects/cityscapes_label2photo/experiments/pix2pix.yaml:25-48:
    # PatchGAN n_layers 4 on 6 channels, lambda 30, lr_D 1e-4) with a narrow U-Net and no dropout -> deterministic
    "p2p_64x128": dict(size=[64, 128], batch=2, steps=6, n_iters=4, n_iters_decay=4, num_downs=5, ngf=16,
                       use_dropout=False, n_layers=4, lambda_pix2pix=30.0, seed=31),
    # cfg3 network shape (num_downs 7, 256x512) at reduced width, batch 1
    "p2p_cfg3_shape": dict(size=[256, 512], batch=1, steps=2, n_iters=100, n_iters_decay=100, num_downs=7, ngf=16,
                           use_dropout=False, n_layers=4, lambda_pix2pix=30.0, seed=32),
}


def make_pix2pix_conf(c):
    return DictConfig({
        "mode": "train",
        "train": {
            "output_dir": "/tmp/ganslate_ref_out", "cuda": False, "mixed_precision": False, "opt_level": "O1",
            "batch_size": c["batch"], "n_iters": c["n_iters"], "n_iters_decay": c["n_iters_decay"],
            "checkpointing": {"load_iter": None, "freq": 10 ** 9, "start_after": 0, "load_optimizers": True},
            "metrics": {"discriminator_evolution": True, "ssim": False},
            "gan": {
                "_target_": "ganslate.nn.gans.paired.Pix2PixConditionalGAN", "norm_type": "instance",
                "weight_init_type": "normal", "weight_init_gain": 0.02,
                "generator": {"_target_": "ganslate.nn.generators.Unet2D", "num_downs": c["num_downs"],
                              "ngf": c["ngf"], "use_dropout": c["use_dropout"],
                              "in_out_channels": {"AB": [3, 3], "BA": [3, 3]}},
                "discriminator": {"_target_": "ganslate.nn.discriminators.PatchGAN2D", "ndf": 64,
                                  "n_layers": c["n_layers"], "kernel_size": [4, 4], "in_channels": {"B": 6, "A": 6}},
                "optimizer": {"adversarial_loss_type": "lsgan", "beta1": 0.5, "beta2": 0.999, "lr_D": 0.0001,
                              "lr_G": 0.0002, "lambda_pix2pix": c["lambda_pix2pix"]},
            },
        },
    })


def p2p_inputs(c, step):
    g = torch.Generator().manual_seed(c["seed"] * 100 + step)
    shape = (c["batch"], 3, *c["size"])
    return torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1


def run_pix2pix_case(name, c):
    torch.manual_seed(c["seed"])
    model = Pix2PixConditionalGAN(make_pix2pix_conf(c))
    for k, (n, net) in enumerate(model.networks.items()):
        net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
    rec = []
    for s in range(c["steps"]):
        A, B = p2p_inputs(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        rec.append({"lrs": {k: float(v) for k, v in lrs.items()},
                    "losses": {k: float(v) for k, v in losses.items() if v is not None},
                    "metrics": {k: float(v) for k, v in metrics.items() if v is not None}})
        model.update_learning_rate()
        print(name, s, rec[-1]["losses"], flush=True)
    return {"config": c, "steps": rec}


CUT_CASES = {
    # CUT defaults (cut.py:16-40): nce_layers (0,4,8,12,16), 256 patches, T 0.07, lambda_nce_idt 0.5; horse2zebra lrs
    "cut_64": dict(size=64, batch=2, steps=4, n_iters=100, n_iters_decay=100, num_patches=256, seed=41),
}


def make_cut_conf(c):
    conf = make_conf(dict(c, pool_size=0, lambda_identity=0.0, proportion_ssim=0.0))
    gan = conf.train.gan
    gan["_target_"] = "ganslate.nn.gans.unpaired.CUT"
    gan["nce_layers"] = [0, 4, 8, 12, 16]
    gan["mlp_nc"] = 256
    gan["num_patches"] = c["num_patches"]
    gan["use_equivariance_flip"] = False
    gan.generator["in_channels"] = 3      # the key cut.py:83 reads; absent from the v1 schema
    gan["optimizer"] = DictConfig({"adversarial_loss_type": "lsgan", "beta1": 0.5, "beta2": 0.999, "lr_D": 0.0002,
                                   "lr_G": 0.0002, "lambda_adv": 1, "lambda_nce": 1, "lambda_nce_idt": 0.5,
                                   "nce_T": 0.07})
    return conf


def run_cut_case(name, c):
    torch.manual_seed(c["seed"])
    model = CUT(make_cut_conf(c))
    for k, (n, net) in enumerate(model.networks.items()):
        net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
    rec = []
    for s in range(c["steps"]):
        A, B = inputs(c, s)
        torch.manual_seed(1000 + s)          # pins the torch.randperm patch ids of this step
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        rec.append({"lrs": {k: float(v) for k, v in lrs.items()},
                    "losses": {k: float(v) for k, v in losses.items() if v is not None}})
        model.update_learning_rate()
        print(name, s, rec[-1]["losses"], flush=True)
    return {"config": c, "steps": rec}


def make_conf(c):
    return DictConfig({
        "mode": "train",
        "train": {
            "output_dir": "/tmp/ganslate_ref_out", "cuda": False, "mixed_precision": False, "opt_level": "O1",
            "batch_size": c["batch"], "n_iters": c["n_iters"], "n_iters_decay": c["n_iters_decay"],
            "checkpointing": {"load_iter": None, "freq": 10 ** 9, "start_after": 0, "load_optimizers": True},
            "metrics": {"discriminator_evolution": True, "ssim": True},
            "gan": {
                "_target_": "ganslate.nn.gans.unpaired.CycleGAN", "norm_type": "instance",
                "weight_init_type": "normal", "weight_init_gain": 0.02, "pool_size": c["pool_size"],
                "generator": {"_target_": "ganslate.nn.generators.Resnet2D", "n_residual_blocks": 9,
                              "in_out_channels": {"AB": [3, 3], "BA": [3, 3]}},
                "discriminator": {"_target_": "ganslate.nn.discriminators.PatchGAN2D", "ndf": 64, "n_layers": 3,
                                  "kernel_size": [4, 4], "in_channels": {"B": 3, "A": 3}},
                "optimizer": {"adversarial_loss_type": c.get("adv", "lsgan"), "beta1": 0.5, "beta2": 0.999,
                              "lr_D": 0.0002, "lr_G": 0.0002, "lambda_AB": 10.0, "lambda_BA": 10.0,
                              "lambda_identity": c["lambda_identity"], "proportion_ssim": c["proportion_ssim"]},
            },
        },
    })


def inputs(c, step):
    g = torch.Generator().manual_seed(c["seed"] * 100 + step)
    shape = (c["batch"], 3, c["size"], c["size"])
    return torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1


def run_case(name, c):
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    model = CycleGAN(make_conf(c))
    for k, (n, net) in enumerate(model.networks.items()):
        net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
    random.seed(c["seed"])
    rec = []
    for s in range(c["steps"]):
        A, B = inputs(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        rec.append({
            "lrs": {k: float(v) for k, v in lrs.items()},
            "losses": {k: float(v) for k, v in losses.items() if v is not None},
            "metrics": {k: float(v) for k, v in metrics.items() if v is not None},
        })
        model.update_learning_rate()
        print(name, s, rec[-1]["losses"], flush=True)
    norms = {n: float(torch.sqrt(sum((p.detach() ** 2).sum() for p in net.parameters())))
             for n, net in model.networks.items()}
    return {"config": c, "steps": rec, "final_param_norms": norms}


def net_case(name, net, x_shape, seed):
    net.load_state_dict(seeded_state_dict(net, seed))
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(x_shape, generator=g) * 2 - 1).requires_grad_()
    y = net(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    sd_keys = list(net.state_dict().keys())
    flat = y.detach().flatten()
    idx = torch.linspace(0, flat.numel() - 1, 32).long()
    return {
        "seed": seed, "x_shape": list(x_shape), "y_shape": list(y.shape), "state_dict_keys": sd_keys,
        "n_params": sum(p.numel() for p in net.parameters()),
        "y_sum": float(flat.double().sum()), "y_abs_sum": float(flat.double().abs().sum()),
        "y_samples": [float(v) for v in flat[idx]], "sample_idx": [int(i) for i in idx],
        "x_grad_abs_sum": float(x.grad.double().abs().sum()),
        "x_grad_samples": [float(v) for v in x.grad.flatten()[torch.linspace(0, x.numel() - 1, 32).long()]],
        "param_grad_norms": {n: float(p.grad.norm()) for n, p in net.named_parameters()},
    }


# ---- 3-D (BASELINE configs[4] shape family: Resnet3D-class generator + PatchGAN3D on single-channel volumes) --------
CASES_3D = {
    # CycleGAN hyper-parameters of the horse2zebra yaml on 1-channel volumes; Resnet3D(9 blocks) + PatchGAN3D(2 layers)
    "v32_default": dict(size=[32, 32, 32], batch=1, steps=4, n_iters=100, n_iters_decay=100, pool_size=50,
                        lambda_identity=0.0, proportion_ssim=0.0, n_residual_blocks=9, d_layers=2, seed=51),
    # anisotropic patch, identity loss on, batch 2
    "v16x24x32_idt": dict(size=[16, 24, 32], batch=2, steps=3, n_iters=100, n_iters_decay=100, pool_size=50,
                          lambda_identity=0.5, proportion_ssim=0.0, n_residual_blocks=3, d_layers=2, seed=52),
    # brats yaml networks: Vnet3D(16; down 2,2,3; up 3,3,3) + PatchGAN3D(2 layers), lr_G 4e-4 is the CUT yaml's; the
    # cyclegan yaml keeps the defaults
    "vnet_16x32x32": dict(size=[16, 32, 32], batch=1, steps=3, n_iters=100, n_iters_decay=100, pool_size=50,
                          lambda_identity=0.0, proportion_ssim=0.0, d_layers=2, seed=53,
                          vnet=dict(first_layer_channels=16, down_blocks=[2, 2, 3], up_blocks=[3, 3, 3])),
}


def make_conf_3d(c):
    conf = make_conf(c)
    conf.train.metrics["ssim"] = False
    gan = conf.train.gan
    if "vnet" in c:
        # brats yaml generator (projects/brats_mri_sequence_translation/experiments/*.yaml): Vnet3D, no memory saving
        gan["generator"] = DictConfig({"_target_": "ganslate.nn.generators.Vnet3D", "use_memory_saving": False,
                                       "use_inverse": False, "first_layer_channels": c["vnet"]["first_layer_channels"],
                                       "down_blocks": c["vnet"]["down_blocks"], "up_blocks": c["vnet"]["up_blocks"],
                                       "is_separable": False, "in_out_channels": {"AB": [1, 1], "BA": [1, 1]}})
    elif "sa" in c:
        # the self-attention networks (selfattention_vnet3d.py:44-181, selfattention_patchgan3d.py:18-79)
        gan["generator"] = DictConfig({"_target_": "ganslate.nn.generators.SelfAttentionVnet3D", "use_memory_saving": False,
                                       "use_inverse": False, "first_layer_channels": c["sa"]["first_layer_channels"],
                                       "down_blocks": c["sa"]["down_blocks"], "up_blocks": c["sa"]["up_blocks"],
                                       "is_separable": False, "enable_attention_block": c["sa"]["enable_attention_block"],
                                       "in_out_channels": {"AB": [1, 1], "BA": [1, 1]}})
    else:
        gan["generator"] = DictConfig({"_target_": "ganslate.nn.generators.Resnet3D",
                                       "n_residual_blocks": c["n_residual_blocks"],
                                       "in_out_channels": {"AB": [1, 1], "BA": [1, 1]}})
    if "sa" in c:
        gan["discriminator"] = DictConfig({"_target_": "ganslate.nn.discriminators.SelfAttentionPatchGAN3D",
                                           "ndf": c["sa"]["ndf"], "n_layers": c["d_layers"], "kernel_size": [4, 4, 4],
                                           "in_channels": {"B": 1, "A": 1}})
    else:
        gan["discriminator"] = DictConfig({"_target_": "ganslate.nn.discriminators.PatchGAN3D", "ndf": 64,
                                           "n_layers": c["d_layers"], "kernel_size": [4, 4, 4],
                                           "in_channels": {"B": 1, "A": 1}})
    return conf


def inputs_3d(c, step):
    g = torch.Generator().manual_seed(c["seed"] * 100 + step)
    shape = (c["batch"], 1, *c["size"])
    return torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1


def run_case_3d(name, c):
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    model = CycleGAN(make_conf_3d(c))
    for k, (n, net) in enumerate(model.networks.items()):
        net.load_state_dict(seeded_state_dict(net, c["seed"] + k))
    random.seed(c["seed"])
    rec = []
    for s in range(c["steps"]):
        A, B = inputs_3d(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        rec.append({
            "lrs": {k: float(v) for k, v in lrs.items()},
            "losses": {k: float(v) for k, v in losses.items() if v is not None},
            "metrics": {k: float(v) for k, v in metrics.items() if v is not None},
        })
        model.update_learning_rate()
        print(name, s, rec[-1]["losses"], flush=True)
    norms = {n: float(torch.sqrt(sum((p.detach() ** 2).sum() for p in net.parameters())))
             for n, net in model.networks.items()}
    return {"config": c, "steps": rec, "final_param_norms": norms}


def main_3d(out):
    from ganslate.nn.generators import Resnet3D, Unet3D, Vnet3D
    from ganslate.nn.discriminators import PatchGAN3D
    vol = {
        "nets": {
            "resnet3d_16x24x32_3blocks": net_case("r3", Resnet3D(1, 1, "instance", 3), (1, 1, 16, 24, 32), 61),
            "patchgan3d_32_3layers": net_case("p3", PatchGAN3D(1, 64, 3, 4, "instance"), (2, 1, 32, 32, 32), 62),
            "patchgan3d_2ch_2layers": net_case("p2", PatchGAN3D(2, 64, 2, 4, "instance"), (1, 2, 16, 24, 20), 63),
            "vnet3d_brats_blocks": net_case("v3", Vnet3D(1, 1, "instance", 16, (2, 2, 3), (3, 3, 3), False, False),
                                            (1, 1, 16, 24, 32), 65),
            "vnet3d_2ch_small": net_case("v2", Vnet3D(2, 1, "instance", 8, (1, 2), (2, 1), False, False),
                                         (2, 2, 8, 12, 16), 66),
            "unet3d_5downs": net_case("u3", Unet3D(1, 1, 5, "instance", ngf=8), (1, 1, 32, 32, 64), 64),
        },
        "steps": {name: run_case_3d(name, c) for name, c in CASES_3D.items()},
    }
    (out / "volumes.json").write_text(json.dumps(vol, indent=1))


def main():
    out = ROOT / "tests" / "golden"
    out.mkdir(parents=True, exist_ok=True)
    if "--only-3d" in sys.argv:
        return main_3d(out)
    nets = {
        "resnet2d_64": net_case("resnet2d_64", Resnet2D(3, 3, "instance", 9), (2, 3, 64, 64), 21),
        "resnet2d_40x56_3blocks": net_case("r", Resnet2D(3, 3, "instance", 3), (1, 3, 40, 56), 22),
        "patchgan2d_64": net_case("patchgan2d_64", PatchGAN2D(3, 64, 3, 4, "instance"), (2, 3, 64, 64), 23),
        "patchgan2d_6ch_4layers": net_case("p", PatchGAN2D(6, 64, 4, 4, "instance"), (1, 6, 96, 128), 24),
    }
    nets["unet2d_5downs"] = net_case("u5", Unet2D(3, 3, 5, "instance", ngf=16), (2, 3, 32, 64), 25)
    nets["unet2d_7downs"] = net_case("u7", Unet2D(3, 3, 7, "instance", ngf=8), (1, 3, 128, 256), 26)
    (out / "nets.json").write_text(json.dumps(nets, indent=1))
    p2p = {name: run_pix2pix_case(name, c) for name, c in PIX2PIX_CASES.items()}
    (out / "pix2pix_steps.json").write_text(json.dumps(p2p, indent=1))
    cut = {name: run_cut_case(name, c) for name, c in CUT_CASES.items()}
    (out / "cut_steps.json").write_text(json.dumps(cut, indent=1))
    main_3d(out)
    if "--only-new" in sys.argv:
        return
    steps = {name: run_case(name, c) for name, c in CASES.items()}
    (out / "cyclegan_steps.json").write_text(json.dumps(steps, indent=1))


if __name__ == "__main__":
    main()
