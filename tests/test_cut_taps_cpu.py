"""Feature-tap geometry of CUT (ganslate/nn/gans/unpaired/cut.py:297-312 walks the generator's `encoder` module by module;
:214-215 flips the target features back along W when `use_equivariance_flip` drew a flip): the generator's own
`tap_dims` against the shapes the oracle's encoder really produces, and the patch-id remap CUT derives from it."""
import pytest
import torch

from ganslate_amd.nn.native import backend
from oracle import torch_ref
from oracle.ops_ref import RefOps


@pytest.fixture()
def fp32_oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


@pytest.mark.parametrize("H,W", [(64, 64), (40, 56), (37, 53)])
def test_tap_dims_are_the_encoder_feature_shapes(fp32_oracle_backend, H, W):
    from ganslate_amd.nn.generators import Resnet2D
    net = Resnet2D(3, 3, "instance", 3)
    ref = torch_ref.Resnet2D(3, 3, 3)
    x = torch.rand(1, 3, H, W)
    feat = x
    for e, layer in enumerate(ref.encoder):
        feat = layer(feat)
        assert net.tap_dims(e, H, W) == tuple(feat.shape[-2:]), e
        assert net.tap_extent(e, H, W) == feat.shape[-2] * feat.shape[-1]


@pytest.mark.parametrize("H,W", [(64, 64), (37, 53)])
def test_flip_remap_addresses_the_mirrored_pixel(fp32_oracle_backend, H, W):
    from ganslate_amd.nn.generators import Resnet2D
    net = Resnet2D(3, 3, "instance", 9)
    g = torch.Generator().manual_seed(5)
    for e in (0, 4, 8, 12, 16):
        h, w = net.tap_dims(e, H, W)
        F = torch.rand(h, w, generator=g)
        pid = torch.randperm(h * w, generator=g)[:64]
        remapped = (pid // w) * w + (w - 1 - pid % w)          # CUT._calculate_nce_loss
        assert torch.equal(F.flip(-1).flatten()[remapped], F.flatten()[pid]), e


def test_stand_alone_patchnce_criterion_matches_the_reference():
    """ganslate_amd.nn.losses.cut_losses.PatchNCELoss (a user recipe may instantiate the reference's criterion by name): loss
    per row and the gradient w.r.t. the target features against values recorded from the reference's class
    (oracle/gen_golden_r2.py patchnce_class -> tests/golden/patchnce_class.json)"""
    import json
    from pathlib import Path
    from types import SimpleNamespace as NS
    from ganslate_amd.nn.losses.cut_losses import PatchNCELoss
    gold = json.loads((Path(__file__).parent / "golden" / "patchnce_class.json").read_text())
    for name, c in gold.items():
        conf = NS(train=NS(batch_size=c["batch"], gan=NS(optimizer=NS(nce_T=c["nce_T"]))))
        g = torch.Generator().manual_seed(c["seed"])
        q = torch.nn.functional.normalize(torch.randn(c["batch"] * c["patches"], c["dim"], generator=g), dim=1).requires_grad_()
        k = torch.nn.functional.normalize(torch.randn(c["batch"] * c["patches"], c["dim"], generator=g), dim=1)
        loss = PatchNCELoss(conf)(q, k)
        assert torch.allclose(loss.detach(), torch.tensor(c["loss"]), rtol=1e-5, atol=1e-6), name
        loss.sum().backward()
        assert abs(float(q.grad.norm()) - c["grad_q_norm"]) <= 1e-5 * c["grad_q_norm"]
        got = q.grad.flatten()[::max(1, q.grad.numel() // 8)][:8]
        assert torch.allclose(got, torch.tensor(c["grad_q_samples"]), rtol=1e-4, atol=1e-7), name
