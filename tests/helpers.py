"""Shared helpers for step-level parity tests (CPU oracle backend and GPU)."""
import json
import random
from pathlib import Path

import torch

GOLD = Path(__file__).parent / "golden"
CONF = Path(__file__).parent / "configs" / "cyclegan_synthetic.yaml"


def golden_inputs(c, step):
    g = torch.Generator().manual_seed(c["seed"] * 100 + step)
    shape = (c["batch"], 3, c["size"], c["size"])
    return torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1


def load_golden_steps():
    return json.loads((GOLD / "cyclegan_steps.json").read_text())


def build_product_cyclegan(c, extra=()):
    """product CycleGAN configured like golden case `c`, with the oracle's seeded weights loaded"""
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle import torch_ref
    conf = build_conf([f"config={CONF}", f"train.batch_size={c['batch']}", f"train.n_iters={c['n_iters']}",
                       f"train.n_iters_decay={c['n_iters_decay']}", f"train.gan.pool_size={c['pool_size']}",
                       f"train.gan.optimizer.lambda_identity={c['lambda_identity']}",
                       f"train.gan.optimizer.proportion_ssim={c['proportion_ssim']}", *extra])
    torch.manual_seed(c["seed"])
    model = build_gan(conf)
    shadow = {"G_AB": torch_ref.Resnet2D(3, 3, 9), "G_BA": torch_ref.Resnet2D(3, 3, 9),
              "D_B": torch_ref.PatchGAN2D(3), "D_A": torch_ref.PatchGAN2D(3)}
    for k, name in enumerate(["G_AB", "G_BA", "D_B", "D_A"]):
        model.networks[name].load_state_dict(torch_ref.seeded_state_dict(shadow[name], c["seed"] + k))
    random.seed(c["seed"])
    return model


def run_product_steps(model, c, n_steps):
    out = []
    for s in range(n_steps):
        A, B = golden_inputs(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        out.append({"lrs": dict(lrs),
                    "losses": {k: float(v.detach()) for k, v in losses.items() if v is not None},
                    "metrics": {k: float(v) for k, v in metrics.items() if v is not None}})
        model.update_learning_rate()
    return out


# ---- volumes (Resnet3D + PatchGAN3D) ------------------------------------------------------------------------------------
VOL_CONF = Path(__file__).parent / "configs" / "cyclegan3d_synthetic.yaml"
VNET_CONF = Path(__file__).parent / "configs" / "cyclegan_vnet_synthetic.yaml"


def load_golden_volumes():
    return json.loads((GOLD / "volumes.json").read_text())


def volume_inputs(c, step):
    g = torch.Generator().manual_seed(c["seed"] * 100 + step)
    shape = (c["batch"], 1, *c["size"])
    return torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1


def build_product_cyclegan3d(c, extra=()):
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle import torch_ref
    common = [f"train.batch_size={c['batch']}", f"train.n_iters={c['n_iters']}",
              f"train.n_iters_decay={c['n_iters_decay']}", f"train.gan.pool_size={c['pool_size']}",
              f"train.gan.discriminator.n_layers={c['d_layers']}",
              f"train.gan.optimizer.lambda_identity={c['lambda_identity']}", *extra]
    if "vnet" in c:
        v = c["vnet"]
        conf = build_conf([f"config={VNET_CONF}", f"train.gan.generator.first_layer_channels={v['first_layer_channels']}",
                           *common])
        assert list(conf.train.gan.generator.down_blocks) == v["down_blocks"]
        assert list(conf.train.gan.generator.up_blocks) == v["up_blocks"]
        G = lambda: torch_ref.Vnet3D(1, 1, v["first_layer_channels"], tuple(v["down_blocks"]), tuple(v["up_blocks"]))
    else:
        conf = build_conf([f"config={VOL_CONF}", f"train.gan.generator.n_residual_blocks={c['n_residual_blocks']}",
                           *common])
        G = lambda: torch_ref.Resnet3D(1, 1, c["n_residual_blocks"])
    torch.manual_seed(c["seed"])
    model = build_gan(conf)
    D = lambda: torch_ref.PatchGAN3D(1, 64, c["d_layers"])
    shadow = {"G_AB": G(), "G_BA": G(), "D_B": D(), "D_A": D()}
    for k, name in enumerate(["G_AB", "G_BA", "D_B", "D_A"]):
        model.networks[name].load_state_dict(torch_ref.seeded_state_dict(shadow[name], c["seed"] + k))
    random.seed(c["seed"])
    return model


def run_product_volume_steps(model, c, n_steps):
    out = []
    for s in range(n_steps):
        A, B = volume_inputs(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        out.append({"lrs": dict(lrs),
                    "losses": {k: float(v.detach()) for k, v in losses.items() if v is not None},
                    "metrics": {k: float(v) for k, v in metrics.items() if v is not None}})
        model.update_learning_rate()
    return out


# ---- pix2pix ---------------------------------------------------------------------------------------------------------
P2P_CONF = Path(__file__).parent / "configs" / "pix2pix_synthetic.yaml"


def load_golden_pix2pix():
    return json.loads((GOLD / "pix2pix_steps.json").read_text())


def p2p_inputs(c, step):
    g = torch.Generator().manual_seed(c["seed"] * 100 + step)
    shape = (c["batch"], 3, *c["size"])
    return torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1


def build_product_pix2pix(c, extra=()):
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle import torch_ref
    conf = build_conf([f"config={P2P_CONF}", f"train.batch_size={c['batch']}", f"train.n_iters={c['n_iters']}",
                       f"train.n_iters_decay={c['n_iters_decay']}",
                       f"train.gan.generator.num_downs={c['num_downs']}", f"train.gan.generator.ngf={c['ngf']}",
                       f"train.gan.generator.use_dropout={c['use_dropout']}",
                       f"train.gan.discriminator.n_layers={c['n_layers']}",
                       f"train.gan.optimizer.lambda_pix2pix={c['lambda_pix2pix']}", *extra])
    torch.manual_seed(c["seed"])
    model = build_gan(conf)
    shadow = {"G": torch_ref.Unet2D(3, 3, c["num_downs"], c["ngf"], c["use_dropout"]),
              "D": torch_ref.PatchGAN2D(6, 64, c["n_layers"])}
    for k, name in enumerate(["G", "D"]):
        model.networks[name].load_state_dict(torch_ref.seeded_state_dict(shadow[name], c["seed"] + k))
    return model


def run_product_pix2pix_steps(model, c, n_steps):
    out = []
    for s in range(n_steps):
        A, B = p2p_inputs(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        out.append({"lrs": dict(lrs),
                    "losses": {k: float(v.detach()) for k, v in losses.items() if v is not None},
                    "metrics": {k: float(v) for k, v in metrics.items() if v is not None}})
        model.update_learning_rate()
    return out


# ---- CUT -------------------------------------------------------------------------------------------------------------
CUT_CONF = Path(__file__).parent / "configs" / "cut_synthetic.yaml"


def load_golden_cut():
    return json.loads((GOLD / "cut_steps.json").read_text())


def build_product_cut(c, extra=()):
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle import torch_ref
    conf = build_conf([f"config={CUT_CONF}", f"train.batch_size={c['batch']}", f"train.n_iters={c['n_iters']}",
                       f"train.n_iters_decay={c['n_iters_decay']}", f"train.gan.num_patches={c['num_patches']}",
                       *extra])
    torch.manual_seed(c["seed"])
    model = build_gan(conf)
    ref = torch_ref.CUTStep(c["batch"], num_patches=c["num_patches"], seed=c["seed"])
    for k, name in enumerate(["G", "D", "mlp"]):
        model.networks[name].load_state_dict(torch_ref.seeded_state_dict(ref.nets[name], c["seed"] + k))
    # patch ids drawn on the CPU generator (as the reference does on CPU), whatever the device
    G = model.networks["G"]

    def sample(H, W):
        ids = []
        for e in model.nce_layers:
            pid = torch.randperm(G.tap_extent(e, H, W))
            ids.append(pid[:int(min(model.num_patches, len(pid)))].to(model.device))
        return ids

    model.sample_patch_ids = sample
    return model


def run_product_cut_steps(model, c, n_steps):
    out = []
    for s in range(n_steps):
        A, B = golden_inputs(c, s)
        torch.manual_seed(1000 + s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        lrs, losses, visuals, metrics = model.get_loggable_data()
        out.append({"lrs": dict(lrs), "losses": {k: float(v.detach()) for k, v in losses.items() if v is not None}})
        model.update_learning_rate()
    return out


# ---- step-0 parameter gradients -----------------------------------------------------------------------------------------
def load_golden_grads():
    return json.loads((GOLD / "cyclegan_grads.json").read_text())


FROZEN = ("train.gan.optimizer.lr_G=0.0", "train.gan.optimizer.lr_D=0.0")


def adam_first_moments(model):
    """{network: {tensor name: exp_avg in torch layout}} of a product model. With the learning rates at 0 the weights
    never move, and after ONE iteration exp_avg = (1 - beta1) * g: the gradients exactly as the optimiser consumed them
    (after gradient accumulation over the backward passes, the merged weight-gradient launches, the data-parallel
    average), which `.grad` no longer shows because the update kernel clears it."""
    out = {}
    for optim in model.optimizers.values():
        if not hasattr(optim, "param_groups"):
            continue
        for group in optim.param_groups:
            for p in group["params"]:
                net = getattr(p, "_owner_net", None)
                if net is None or "exp_avg" not in optim.state[p]:
                    continue
                name = next(n for n, v in model.networks.items() if v is net)
                if not hasattr(net, "grads_state_dict"):      # CUT's patch MLP: one flat buffer, reference key names
                    out[name] = {k: v.detach().float().cpu() for k, v in net.flat_to_tensors(optim.state[p]["exp_avg"]).items()}
                    continue
                keep, net.master.grad = net.master.grad, optim.state[p]["exp_avg"]
                try:
                    out[name] = {k: v.detach().float().cpu() for k, v in net.grads_state_dict().items()}
                finally:
                    net.master.grad = keep
    return out
