"""MultiScalePatchGAN3D: the oracle twin against vectors recorded from the REAL reference class
(oracle/gen_golden_r2.py multiscale, over the monai stand-in: the window RNG is unpinned, see
oracle/ref_stubs/monai/transforms), the product on the fp32 oracle backend against the twin — per-scale maps, input
gradient, every parameter gradient, state-dict names — and a CycleGAN run whose discriminators are multi-scale."""
import json
import random
from pathlib import Path

import pytest
import torch

from ganslate_amd.nn.native import backend
from oracle import torch_ref
from oracle.ops_ref import RefOps

GOLD = json.loads((Path(__file__).parent / "golden" / "multiscale_patchgan3d.json").read_text())


@pytest.fixture()
def fp32_oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


def _input(c):
    g = torch.Generator().manual_seed(c["seed"])
    return torch.rand((c["batch"], c["in_channels"], *c["dims"]), generator=g) * 2 - 1


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_twin_against_the_reference(name):
    rec = GOLD[name]
    c = rec["config"]
    twin = torch_ref.MultiScalePatchGAN3D(c["in_channels"], c["ndf"], c["n_layers"], 4, c["scales"])
    assert list(twin.state_dict().keys()) == rec["keys"]
    twin.load_state_dict(torch_ref.seeded_state_dict(twin, c["seed"]))
    x = _input(c).requires_grad_()
    random.seed(c["seed"])
    maps = twin(x)
    loss = sum((m ** 2).mean() for m in maps.values())
    loss.backward()
    assert float(loss) == pytest.approx(rec["loss"], rel=1e-5)
    for s, g in rec["maps"].items():
        assert list(maps[s].shape) == g["shape"]
        got = maps[s].detach().flatten()[torch.tensor(g["samples_at"])]
        assert torch.allclose(got, torch.tensor(g["samples"]), atol=1e-5, rtol=1e-4)
    assert float(x.grad.norm()) == pytest.approx(rec["input_grad"]["norm"], rel=1e-4)
    assert int((x.grad != 0).sum()) == rec["input_grad"]["nonzero"]
    for k, p in twin.named_parameters():
        assert float(p.grad.norm()) == pytest.approx(rec["param_grad_norms"][k], rel=2e-4, abs=1e-9), k


@pytest.mark.parametrize("name", ["s2_two_layers", "s3_one_layer"])
def test_product_against_the_twin(fp32_oracle_backend, name):
    from ganslate_amd.nn.discriminators import MultiScalePatchGAN3D
    c = GOLD[name]["config"]
    twin = torch_ref.MultiScalePatchGAN3D(c["in_channels"], c["ndf"], c["n_layers"], 4, c["scales"])
    sd = torch_ref.seeded_state_dict(twin, c["seed"])
    twin.load_state_dict(sd)
    net = MultiScalePatchGAN3D(c["in_channels"], c["ndf"], c["n_layers"], (4, 4, 4), c["scales"], "instance")
    assert net.graph_capturable is False and len(net.native_children()) == c["scales"]
    net.load_state_dict(sd)
    assert list(net.state_dict().keys()) == GOLD[name]["keys"]
    for k, v in net.state_dict().items():
        assert torch.allclose(v.cpu(), sd[k], atol=0, rtol=0), k
    with pytest.raises(KeyError):
        net.load_state_dict({**sd, "model.9.model.0.weight": torch.zeros(1)})
    xa, xb = _input(c).requires_grad_(), _input(c).requires_grad_()
    random.seed(c["seed"])
    ma = twin(xa)
    state = random.getstate()
    random.seed(c["seed"])
    mb = net(xb)
    assert random.getstate() == state                  # same window draws
    assert list(mb) == list(ma)
    for s in ma:
        assert torch.allclose(mb[s], ma[s], atol=2e-5, rtol=1e-4), s
    sum((m ** 2).mean() for m in ma.values()).backward()
    sum((m ** 2).mean() for m in mb.values()).backward()
    scale = xa.grad.abs().max().item()
    assert (xa.grad - xb.grad).abs().max().item() <= 1e-3 * scale
    grads = net.grads_state_dict()
    normed = {f"model.{s}.{nd.name}" for s, sub in net.model.items() for nd in sub.nodes if nd.norm}
    for k, p in twin.named_parameters():
        if k.endswith(".bias") and k[:-5] in normed:
            continue                                    # exactly-zero true gradient: rounding noise on both sides
        ref = p.grad
        assert (ref - grads[k]).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-7, k
    assert len(net.parameters()) == c["scales"] and all(p._owner_net is sub for p, sub in
                                                         zip(net.parameters(), net.model.values()))


def test_cyclegan_with_multiscale_discriminators(fp32_oracle_backend, tmp_path):
    """`_target_: ganslate.nn.discriminators.MultiScalePatchGAN3D` through the builders: the adversarial losses are the
    mean over scales (adversarial_loss.py:92-94), the D-output metrics are None (the reference's dict branch returns
    nothing, train_metrics.py:22-25), Adam spans the sub-networks, checkpoints carry `model.<s>.` names"""
    from ganslate_amd.engines import init_engine
    args = ["config=tests/configs/cyclegan3d_synthetic.yaml", f"train.output_dir={tmp_path}", "train.cuda=False",
            "train.n_iters=2", "train.n_iters_decay=0", "train.dataset.final_size=[16,24,24]",
            "train.gan.generator.n_residual_blocks=1",
            "train.gan.discriminator._target_=ganslate.nn.discriminators.MultiScalePatchGAN3D",
            "train.gan.discriminator.n_layers=1", "train.gan.discriminator.ndf=8", "train.gan.discriminator.scales=2",
            "train.checkpointing.freq=2", "train.seed=3"]
    trainer = init_engine("train", args)
    model = trainer.model
    before = {k: v.clone() for k, v in model.networks["D_A"].state_dict().items()}
    trainer.run()
    assert model.step_graph_enabled is False
    assert all(float(v) == float(v) for v in model.losses.values() if v is not None)
    assert isinstance(model.pred_real, dict) and sorted(model.pred_real) == ["1", "2"]
    assert model.pred_real["2"].shape[2:] != model.pred_real["1"].shape[2:]
    want = torch.stack([((p - 1) ** 2).mean() for p in model.pred_real.values()]).mean() + \
        torch.stack([(p ** 2).mean() for p in model.pred_fake.values()]).mean()
    assert float(model.losses["D_A"]) == pytest.approx(float(want), rel=1e-5)
    assert model.metrics.get("D_A_real") is None and model.metrics.get("D_A_fake") is None
    after = model.networks["D_A"].state_dict()
    assert all(k.startswith(("model.1.", "model.2.")) for k in after)
    for s in ("1", "2"):                                  # both scales were updated by the one D optimiser
        assert not torch.equal(after[f"model.{s}.model.0.weight"], before[f"model.{s}.model.0.weight"])
    ck = torch.load(tmp_path / "checkpoints" / "2.pth", map_location="cpu")
    assert sorted(ck["D_A"]) == sorted(after)
    n_ref_params = len(after)                              # reference optimizer state: one entry per tensor of D_B then D_A
    assert len(ck["optimizer_D"]["state"]) == 2 * n_ref_params
