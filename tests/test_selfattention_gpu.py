"""The self-attention networks (ganslate/nn/attention.py, selfattention_patchgan3d.py,
selfattention_vnet3d.py) on the HIP path: network forward / backward against the fp32 oracle twins (pinned to the real classes
in tests/test_oracle_pinned.py, tests/golden/selfattention.json) and the bf16 emulation, the reference's golden outputs, and
a CycleGAN training run with both networks built from a YAML config through the plugin surface."""
import json
import random
from pathlib import Path

import pytest
import torch

from oracle import torch_ref

from .test_cyclegan_gpu import _net_case, rel_l2

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / "golden" / "selfattention.json").read_text())


def test_selfattention_patchgan3d_hip_vs_oracle(hip_ops):
    from ganslate_amd.nn.discriminators import SelfAttentionPatchGAN3D
    _net_case(hip_ops, lambda: SelfAttentionPatchGAN3D(1, 32, 3, (4, 4, 4), "instance"),
              torch_ref.SelfAttentionPatchGAN3D(1, 32, 3, 4), (1, 1, 64, 64, 64), 75, grad_tol=0.30, grad_cos=0.95)


def test_selfattention_vnet3d_hip_vs_oracle(hip_ops):
    from ganslate_amd.nn.generators import SelfAttentionVnet3D
    _net_case(hip_ops, lambda: SelfAttentionVnet3D(1, 1, "instance", 8, (1, 2), (2, 1), False, False, (True, True)),
              torch_ref.SelfAttentionVnet3D(1, 1, 8, (1, 2), (2, 1), False, (True, True)), (1, 1, 16, 32, 32), 77,
              grad_tol=0.45, grad_cos=0.90)


@pytest.mark.parametrize("name", ["sa_patchgan3d_64", "sa_vnet3d_default_flags"])
def test_selfattention_networks_match_reference_golden(hip_ops, name):
    """same network, weights and input as the golden case recorded from the REAL reference classes"""
    from ganslate_amd.nn.discriminators import SelfAttentionPatchGAN3D
    from ganslate_amd.nn.generators import SelfAttentionVnet3D
    gold = GOLD[name]
    if name == "sa_patchgan3d_64":
        net, shadow = SelfAttentionPatchGAN3D(1, 32, 3, (4, 4, 4), "instance"), torch_ref.SelfAttentionPatchGAN3D(1, 32, 3, 4)
    else:
        net = SelfAttentionVnet3D(1, 1, "instance", 8, (1, 1, 2, 1), (1, 2, 1, 1), False, False, (False, False, True, True))
        shadow = torch_ref.SelfAttentionVnet3D(1, 1, 8, (1, 1, 2, 1), (1, 2, 1, 1), False, (False, False, True, True))
    net.load_state_dict(torch_ref.seeded_state_dict(shadow, gold["seed"]))
    g = torch.Generator().manual_seed(gold["seed"])
    x = torch.rand(gold["x_shape"], generator=g) * 2 - 1
    y = net(x.to(hip_ops.device)).float().cpu().flatten()
    ref = torch.tensor(gold["y_samples"])
    assert list(net(x.to(hip_ops.device)).shape) == gold["y_shape"]
    assert (y[gold["sample_idx"]] - ref).abs().max().item() <= 0.12 * ref.abs().max().item()
    assert abs(y.double().abs().sum().item() - gold["y_abs_sum"]) <= 3e-2 * gold["y_abs_sum"]


def test_cyclegan_with_selfattention_networks_trains_like_the_oracle(hip_ops):
    """CycleGAN over SelfAttentionVnet3D + SelfAttentionPatchGAN3D built from tests/configs/cyclegan_selfattention_synthetic.yaml
    (the reference's `_target_` strings): iteration 0 against the fp32 oracle step on the same weights and batch (2e-2), two
    more iterations (the second one captured as a hipGraph) finite and moving, checkpoint keys = the reference's names"""
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle.torch_ref import CycleGANStep
    conf = build_conf([f"config={Path(__file__).parent / 'configs' / 'cyclegan_selfattention_synthetic.yaml'}"])
    seed = 81
    torch.manual_seed(seed)
    model = build_gan(conf)
    mkG = lambda i, o: torch_ref.SelfAttentionVnet3D(i, o, 8, (1, 2), (2, 1), False, (True, True))
    mkD = lambda i: torch_ref.SelfAttentionPatchGAN3D(i, 16, 2, 4)
    ref = CycleGANStep(in_ch=1, out_ch=1, n_iters=100, n_iters_decay=100, pool_size=50, metrics_ssim=False, seed=seed, dims=3,
                       make_G=mkG, make_D=mkD)
    for name, net in ref.nets.items():
        model.networks[name].load_state_dict(net.state_dict())
        got, want = set(model.networks[name].state_dict()), set(net.state_dict())
        assert got == want, sorted(got ^ want)[:6]
    random.seed(seed)
    g = torch.Generator().manual_seed(seed * 100)
    shape = (1, 1, 32, 48, 48)
    A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
    want, _ = ref.step(A, B)
    random.seed(seed)
    model.set_input({"A": A, "B": B})
    model.optimize_parameters()
    torch.cuda.synchronize()
    for k, v in want.items():
        assert float(model.losses[k].detach()) == pytest.approx(v, rel=2e-2), (k, float(model.losses[k].detach()), v)
    first = {k: float(v.detach()) for k, v in model.losses.items() if v is not None}
    for s in range(1, 4):
        model.update_learning_rate()
        g = torch.Generator().manual_seed(seed * 100 + s)
        A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
    torch.cuda.synchronize()
    last = {k: float(v.detach()) for k, v in model.losses.items() if v is not None}
    assert all(v == v and abs(v) < 1e3 for v in last.values()), last
    assert any(abs(last[k] - first[k]) > 1e-4 for k in last), "the losses must move"
