"""RevGAN (ganslate/nn/gans/unpaired/revgan.py) and the inverse direction of the partially-invertible
V-Nets: the oracle (oracle/torch_ref.py: Vnet3D / Vnet2D(use_inverse=True), RevGANStep) pinned against vectors of the REAL
reference run over the memcnn stand-in (tests/golden/revgan.json, oracle/gen_golden_r2.py revgan). memcnn itself is absent
and unpinned in the reference: the coupling's forward / inverse equations are restated from release 1.5.1
(oracle/memcnn_ref.py, tests/test_memcnn_semantics_cpu.py), so this parity is "unpinned" for the memcnn-specific part —
memory saving changes what is kept alive, not what is computed. CPU only."""
import json
import random
from pathlib import Path

import pytest
import torch

from oracle import torch_ref

GOLD = json.loads((Path(__file__).parent / "golden" / "revgan.json").read_text())


@pytest.fixture(autouse=True)
def _eight_threads():
    before = torch.get_num_threads()
    torch.set_num_threads(8)
    yield
    torch.set_num_threads(before)


@pytest.mark.parametrize("name", list(GOLD["nets"]))
def test_inverse_direction_of_the_oracle_networks(name):
    gold = GOLD["nets"][name]
    net = {"vnet3d_inverse": lambda: torch_ref.Vnet3D(1, 1, 8, (1, 2), (2, 1), use_inverse=True),
           "vnet2d_inverse_default_blocks": lambda: torch_ref.Vnet2D(2, 2, 8, use_inverse=True),
           "piresnet3d": lambda: torch_ref.Piresnet3D(1, 1, 3, 16, use_inverse=True)}[name]()
    assert list(net.state_dict().keys()) == gold["state_dict_keys"]
    assert sum(p.numel() for p in net.parameters()) == gold["n_params"]
    net.load_state_dict(torch_ref.seeded_state_dict(net, gold["seed"]))
    g = torch.Generator().manual_seed(gold["seed"])
    x = (torch.rand(gold["x_shape"], generator=g) * 2 - 1).requires_grad_()
    y = net(x)
    r = net(y, inverse=True)
    gy, gr = torch.randn(y.shape, generator=g), torch.randn(r.shape, generator=g)
    ((y * gy).sum() + (r * gr).sum()).backward()
    idx = gold["sample_idx"]
    assert torch.allclose(y.detach().flatten()[idx], torch.tensor(gold["y_samples"]), atol=1e-6, rtol=1e-5)
    assert torch.allclose(r.detach().flatten()[idx], torch.tensor(gold["r_samples"]), atol=1e-5, rtol=1e-4)
    assert abs(float(x.grad.double().abs().sum()) - gold["x_grad_abs_sum"]) <= 1e-3 * gold["x_grad_abs_sum"]
    for n, p in net.named_parameters():
        want = gold["param_grad_norms"][n]
        assert abs(float(p.grad.norm()) - want) <= 1e-3 * want + 1e-7, n


@pytest.mark.parametrize("name", list(GOLD["steps"]))
def test_revgan_step_restatement_matches_reference(name):
    gold = GOLD["steps"][name]
    c = gold["config"]
    model = torch_ref.RevGANStep(ch=1 if c["dims"] == 3 else 2, n_layers=c["d_layers"], n_iters=c["n_iters"],
                                 n_iters_decay=c["n_iters_decay"], pool_size=c["pool_size"],
                                 lambda_identity=c["lambda_identity"], proportion_ssim=c["proportion_ssim"],
                                 seed=c["seed"], dims=c["dims"], vnet=c.get("vnet"), piresnet=c.get("piresnet"))
    assert list(model.nets) == gold["network_names"]
    random.seed(c["seed"])
    ch = 1 if c["dims"] == 3 else 2
    for s in range(c["steps"]):
        g = torch.Generator().manual_seed(c["seed"] * 100 + s)
        shape = (c["batch"], ch, *c["size"])
        A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
        losses, metrics = model.step(A, B)
        want = gold["steps"][s]
        assert set(losses) == set(want["losses"])
        for k, v in want["losses"].items():
            assert abs(losses[k] - v) <= 5e-4 * abs(v) + 1e-6, (s, k, losses[k], v)
        model.update_learning_rate()


# ---- the product's recipe on the fp32 oracle backend (host logic: pass order, shared generator, loss assembly, Adam) ------
def _product_revgan(c, conf_name, extra=()):
    from ganslate_amd.utils.builders import build_conf, build_gan
    conf = build_conf([f"config=tests/configs/{conf_name}", f"train.batch_size={c['batch']}",
                       f"train.n_iters={c['n_iters']}", f"train.n_iters_decay={c['n_iters_decay']}",
                       f"train.gan.pool_size={c['pool_size']}",
                       f"train.gan.optimizer.lambda_identity={c['lambda_identity']}", "train.cuda=False", *extra])
    torch.manual_seed(c["seed"])
    model = build_gan(conf)
    ch = 1 if c["dims"] == 3 else 2
    V, D = (torch_ref.Vnet3D, torch_ref.PatchGAN3D) if c["dims"] == 3 else (torch_ref.Vnet2D, torch_ref.PatchGAN2D)
    if "piresnet" in c:
        G = torch_ref.Piresnet3D(ch, ch, c["piresnet"]["depth"], c["piresnet"]["first_layer_channels"], use_inverse=True)
    else:
        kw = dict(first_layer_channels=c["vnet"]["first_layer_channels"])
        if "down_blocks" in c["vnet"]:
            kw.update(down_blocks=tuple(c["vnet"]["down_blocks"]), up_blocks=tuple(c["vnet"]["up_blocks"]))
        G = V(ch, ch, use_inverse=True, **kw)
    shadow = {"G": G, "D_B": D(ch, 64, c["d_layers"]), "D_A": D(ch, 64, c["d_layers"])}
    assert list(model.networks) == ["G", "D_B", "D_A"]
    for k, name in enumerate(["G", "D_B", "D_A"]):
        model.networks[name].load_state_dict(torch_ref.seeded_state_dict(shadow[name], c["seed"] + k))
    random.seed(c["seed"])
    return model


@pytest.mark.parametrize("name,conf_name", [("rev3d_16x32x32", "revgan3d_synthetic.yaml"),
                                            ("rev3d_piresnet", "revgan3d_piresnet_synthetic.yaml"),
                                            ("rev2d_64x64_idt", "revgan2d_synthetic.yaml")])
def test_product_recipe_on_the_oracle_backend_matches_reference(name, conf_name):
    from ganslate_amd.nn.native import backend
    from oracle.ops_ref import RefOps
    gold = GOLD["steps"][name]
    c = gold["config"]
    backend.set_ops(RefOps(act_dtype=torch.float32))
    try:
        model = _product_revgan(c, conf_name)
        ch = 1 if c["dims"] == 3 else 2
        for s in range(2 if c["dims"] == 2 else 1):          # (3-D: one iteration keeps the CPU suite short)
            g = torch.Generator().manual_seed(c["seed"] * 100 + s)
            shape = (c["batch"], ch, *c["size"])
            A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
            model.set_input({"A": A, "B": B})
            model.optimize_parameters()
            losses = {k: float(v.detach()) for k, v in model.losses.items() if v is not None}
            want = gold["steps"][s]["losses"]
            assert set(losses) == set(want)
            for k, v in want.items():
                # iteration 0: arithmetic parity; iteration 1 carries one Adam update (+-lr per weight, see tests/envelope.py)
                assert abs(losses[k] - v) <= (1e-3 if s == 0 else 3e-2) * abs(v) + 1e-5, (s, k, losses[k], v)
            model.update_learning_rate()
        assert model.infer(A, "BA").shape == A.shape
    finally:
        backend.set_ops(None)


def test_parameter_order_with_the_inverse_layers_is_the_reference_registration_order():
    """optimizer_G in checkpoints is stored per parameter in the reference's order (base.py:244-287): in_ab, in_ba, out_ab,
    out_ba, downs (down_conv_ab, down_conv_ba, core, relu), ups (vnet3d.py:60-104)"""
    from ganslate_amd.nn.generators import Vnet3D
    from ganslate_amd.nn.native import backend
    from oracle.ops_ref import RefOps
    backend.set_ops(RefOps(act_dtype=torch.float32))
    try:
        native = Vnet3D(1, 1, "instance", 8, (1, 2), (2, 1))
        shadow = torch_ref.Vnet3D(1, 1, 8, (1, 2), (2, 1), use_inverse=True)
        want = [n for n, _ in shadow.named_parameters()]          # named_parameters() drops the `encoder.*` aliases
        assert native.reference_parameter_order() == want
    finally:
        backend.set_ops(None)


@pytest.mark.parametrize("saving", [False, True])
def test_piresnet3d_product_vs_oracle_both_directions(saving):
    """Piresnet3D (piresnet3d.py:29-108): the product's executor on the fp32 oracle backend against the oracle twin (pinned to
    the real reference above) — A -> B, then B -> A on its output, gradients of a loss on both; with and without the
    activation recompute"""
    from ganslate_amd.nn.generators import Piresnet3D
    from ganslate_amd.nn.native import backend
    from oracle.ops_ref import RefOps
    from .test_networks_cpu import _compare_both_directions
    backend.set_ops(RefOps(act_dtype=torch.float32))
    try:
        native = Piresnet3D(1, 1, "instance", 3, 16, use_memory_saving=saving, use_inverse=True)
        shadow = torch_ref.Piresnet3D(1, 1, 3, 16, use_inverse=True)
        assert native.reference_parameter_order() == [n for n, _ in shadow.named_parameters()]
        _compare_both_directions(native, shadow, (1, 1, 8, 12, 16), 72)
    finally:
        backend.set_ops(None)
