"""3-D parity of the HIP path (Resnet3D + PatchGAN3D, the conv3d family of BASELINE configs[4]) against the fp32
oracle, the bf16 CPU emulation and the golden vectors of the real reference (tests/golden/volumes.json).
Tolerances as in tests/test_cyclegan_gpu.py (bf16 storage, fp32 accumulation)."""
import json
from pathlib import Path

import pytest
import torch

from oracle import torch_ref

from .helpers import build_product_cyclegan3d, load_golden_volumes, run_product_volume_steps
from .test_cyclegan_gpu import _net_case, rel_l2

pytestmark = pytest.mark.gpu


def test_resnet3d_hip_vs_oracle(hip_ops):
    from ganslate_amd.nn.generators import Resnet3D
    _net_case(hip_ops, lambda: Resnet3D(1, 1, "instance", 3), torch_ref.Resnet3D(1, 1, 3), (1, 1, 16, 24, 32), 71,
              grad_tol=0.30, grad_cos=0.95)


@pytest.mark.parametrize("in_ch,n_layers,dhw", [(1, 3, (32, 32, 32)), (2, 2, (16, 24, 20))])
def test_patchgan3d_hip_vs_oracle(hip_ops, in_ch, n_layers, dhw):
    from ganslate_amd.nn.discriminators import PatchGAN3D
    _net_case(hip_ops, lambda: PatchGAN3D(in_ch, 64, n_layers, (4, 4, 4), "instance"),
              torch_ref.PatchGAN3D(in_ch, 64, n_layers, 4), (1, in_ch, *dhw), 72)


def test_unet3d_hip_vs_oracle(hip_ops):
    from ganslate_amd.nn.generators import Unet3D
    _net_case(hip_ops, lambda: Unet3D(1, 1, 5, "instance", ngf=16), torch_ref.Unet3D(1, 1, 5, 16),
              (1, 1, 32, 32, 64), 73, grad_tol=0.30, grad_cos=0.95)


@pytest.mark.parametrize("cin,c,downs,ups,dhw", [(1, 16, (2, 2, 3), (3, 3, 3), (16, 24, 32)),
                                                 (2, 8, (1, 2), (2, 1), (16, 24, 32))])
def test_vnet3d_hip_vs_oracle(hip_ops, cin, c, downs, ups, dhw):
    from ganslate_amd.nn.generators import Vnet3D
    _net_case(hip_ops, lambda: Vnet3D(cin, 1, "instance", c, downs, ups, use_memory_saving=False, use_inverse=False),
              torch_ref.Vnet3D(cin, 1, c, downs, ups), (1, cin, *dhw), 74, grad_tol=0.30, grad_cos=0.95)


def test_resnet3d_matches_reference_golden(hip_ops):
    from ganslate_amd.nn.generators import Resnet3D
    gold = load_golden_volumes()["nets"]["resnet3d_16x24x32_3blocks"]
    net, shadow = Resnet3D(1, 1, "instance", 3), torch_ref.Resnet3D(1, 1, 3)
    net.load_state_dict(torch_ref.seeded_state_dict(shadow, gold["seed"]))
    g = torch.Generator().manual_seed(gold["seed"])
    x = torch.rand(gold["x_shape"], generator=g) * 2 - 1
    y = net(x.to(hip_ops.device)).cpu().flatten()
    ref = torch.tensor(gold["y_samples"])
    assert (y[gold["sample_idx"]] - ref).abs().max().item() <= 0.12 * ref.abs().max().item()
    assert rel_l2(y[gold["sample_idx"]], ref) <= 5e-2
    assert abs(y.double().abs().sum().item() - gold["y_abs_sum"]) <= 2e-2 * gold["y_abs_sum"]


@pytest.mark.parametrize("name", ["v32_default", "v16x24x32_idt", "vnet_16x32x32"])
def test_volume_training_step_matches_reference_golden(hip_ops, name):
    gold = load_golden_volumes()["steps"][name]
    c = gold["config"]
    model = build_product_cyclegan3d(c)
    got = run_product_volume_steps(model, c, c["steps"])
    for s in range(c["steps"]):
        g = gold["steps"][s]
        assert set(got[s]["losses"]) == set(g["losses"])
        # after the first updates the trajectory is only statistically pinned: on these tiny volumes the
        # third iteration's cycle terms sit 2.9-3.5 % from the reference's whichever summation order the kernels use
        tol_adv, tol_cyc = (2e-2, 2e-2) if s == 0 else (0.25, 0.05)
        for k, v in g["losses"].items():
            tol = tol_cyc if k.startswith(("cycle", "idt")) else tol_adv
            assert got[s]["losses"][k] == pytest.approx(v, rel=tol), (s, k, got[s]["losses"][k], v)
        if s == 0:
            for k, v in g["metrics"].items():
                assert got[s]["metrics"][k] == pytest.approx(v, rel=2e-2, abs=1e-2), (s, k)
