"""Step-0 parameter gradients of the recipes other than 2-D CycleGAN (VERDICT r2 Weak #1): the cases of
tests/golden/recipe_grads.json (written by `python -m oracle.gen_golden_r2 recipegrads` from the REAL reference), the
oracle's restatement of each recipe's first iteration, and the product's — read back from Adam's first moment with the
learning rates at 0 (tests/helpers.adam_first_moments: exp_avg = (1 - beta1) g after one iteration)."""
import json
import random

import torch

from oracle import torch_ref

from .helpers import (FROZEN, GOLD, adam_first_moments, build_product_cut, build_product_cyclegan3d, build_product_pix2pix,
                      golden_inputs, p2p_inputs, volume_inputs)

REVGAN_CONF = {"rev3d_16x32x32": "revgan3d_synthetic.yaml", "rev3d_piresnet": "revgan3d_piresnet_synthetic.yaml"}


def load_recipe_grads():
    return json.loads((GOLD / "recipe_grads.json").read_text())


def _inputs(kind, c):
    if kind == "pix2pix":
        return p2p_inputs(c, 0)
    if kind == "cut":
        return golden_inputs(c, 0)
    if kind == "cyclegan3d":
        return volume_inputs(c, 0)
    g = torch.Generator().manual_seed(c["seed"] * 100)
    shape = (c["batch"], 1, *c["size"])
    return torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1


def oracle_step0(kind, c, frozen=False):
    """(losses, {network: {parameter name: .grad}}) of the oracle's first iteration (fp32 torch autograd).
    frozen: learning rates 0 like the product's run — only CUT needs it (it updates D BEFORE the generator's loss is
    evaluated, cut.py:131-148, so its G-side gradients depend on whether that update happened)"""
    torch.set_num_threads(8)
    if kind == "pix2pix":
        ref = torch_ref.Pix2PixStep(num_downs=c["num_downs"], ngf=c["ngf"], use_dropout=c["use_dropout"],
                                    n_layers=c["n_layers"], lambda_pix2pix=c["lambda_pix2pix"], n_iters=c["n_iters"],
                                    n_iters_decay=c["n_iters_decay"], seed=c["seed"])
    elif kind == "cut":
        ref = torch_ref.CUTStep(c["batch"], num_patches=c["num_patches"], n_iters=c["n_iters"],
                                n_iters_decay=c["n_iters_decay"], seed=c["seed"], **({"lr": 0.0} if frozen else {}))
        torch.manual_seed(1000)
    elif kind == "cyclegan3d":
        mk = {}
        if "sa" in c:       # the self-attention networks (SelfAttentionVnet3D / SelfAttentionPatchGAN3D)
            sa = c["sa"]
            mk = dict(make_G=lambda i, o: torch_ref.SelfAttentionVnet3D(i, o, sa["first_layer_channels"],
                                                                       tuple(sa["down_blocks"]), tuple(sa["up_blocks"]), False,
                                                                       tuple(sa["enable_attention_block"])),
                      make_D=lambda i: torch_ref.SelfAttentionPatchGAN3D(i, sa["ndf"], c["d_layers"], 4))
        ref = torch_ref.CycleGANStep(in_ch=1, out_ch=1, n_blocks=c.get("n_residual_blocks", 0), n_layers=c["d_layers"],
                                     vnet=c.get("vnet"), n_iters=c["n_iters"], n_iters_decay=c["n_iters_decay"],
                                     pool_size=c["pool_size"], lambda_identity=c["lambda_identity"], proportion_ssim=0.0,
                                     metrics_ssim=False, seed=c["seed"], dims=3, **mk)
        random.seed(c["seed"])
    else:
        ref = torch_ref.RevGANStep(ch=1, n_layers=c["d_layers"], n_iters=c["n_iters"], n_iters_decay=c["n_iters_decay"],
                                   pool_size=c["pool_size"], lambda_identity=c["lambda_identity"], seed=c["seed"], dims=3,
                                   vnet=c.get("vnet"), piresnet=c.get("piresnet"))
        random.seed(c["seed"])
    out = ref.step(*_inputs(kind, c))
    losses = out[0] if isinstance(out, tuple) else out
    grads = {}
    for name, net in ref.nets.items():
        grads[name] = {n: p.grad.detach().clone() for n, p in net.named_parameters(remove_duplicate=False)
                       if not n.startswith("encoder.") and p.grad is not None}
    return losses, grads


def build_product_sa_cyclegan(c, extra=()):
    """CycleGAN over the self-attention networks from tests/configs/cyclegan_selfattention_synthetic.yaml with the case's
    seeded weights (the oracle twins' state dicts: the reference's key names)"""
    from pathlib import Path
    from ganslate_amd.utils.builders import build_conf, build_gan
    sa = c["sa"]
    conf = build_conf([f"config={Path(__file__).parent / 'configs' / 'cyclegan_selfattention_synthetic.yaml'}",
                       f"train.batch_size={c['batch']}", f"train.n_iters={c['n_iters']}",
                       f"train.n_iters_decay={c['n_iters_decay']}", f"train.gan.pool_size={c['pool_size']}", *extra])
    g, d = conf.train.gan.generator, conf.train.gan.discriminator
    assert (g.first_layer_channels, list(g.down_blocks), list(g.up_blocks), list(g.enable_attention_block), d.ndf, d.n_layers) == \
        (sa["first_layer_channels"], sa["down_blocks"], sa["up_blocks"], sa["enable_attention_block"], sa["ndf"], c["d_layers"])
    torch.manual_seed(c["seed"])
    model = build_gan(conf)
    mkG = lambda: torch_ref.SelfAttentionVnet3D(1, 1, sa["first_layer_channels"], tuple(sa["down_blocks"]),
                                                tuple(sa["up_blocks"]), False, tuple(sa["enable_attention_block"]))
    mkD = lambda: torch_ref.SelfAttentionPatchGAN3D(1, sa["ndf"], c["d_layers"], 4)
    for k, (name, mk) in enumerate((("G_AB", mkG), ("G_BA", mkG), ("D_B", mkD), ("D_A", mkD))):
        model.networks[name].load_state_dict(torch_ref.seeded_state_dict(mk(), c["seed"] + k))
    random.seed(c["seed"])
    return model


def product_step0(kind, name, c, extra=()):
    """the product's first iteration with frozen weights -> (losses, {network: {parameter name: gradient}})"""
    extra = tuple(extra) + FROZEN
    if kind == "pix2pix":
        model = build_product_pix2pix(c, extra)
    elif kind == "cut":
        model = build_product_cut(c, extra)
        torch.manual_seed(1000)
    elif kind == "cyclegan3d" and "sa" in c:
        model = build_product_sa_cyclegan(c, extra)
    elif kind == "cyclegan3d":
        model = build_product_cyclegan3d(c, extra)
    else:
        from .test_revgan_cpu import _product_revgan
        model = _product_revgan(c, REVGAN_CONF[name], extra=extra + (("train.cuda=True",) if torch.cuda.is_available() else ()))
    A, B = _inputs(kind, c)
    model.set_input({"A": A, "B": B})
    model.optimize_parameters()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    losses = {k: float(v.detach()) for k, v in model.losses.items() if v is not None}
    return losses, {net: {k: 2.0 * v for k, v in per.items()} for net, per in adam_first_moments(model).items()}
