"""The GAN objectives other than lsgan (ganslate/nn/losses/adversarial_loss.py:26-34,60-67,91-94): the oracle's restatement
(oracle/torch_ref.adversarial_loss, oracle/ops_ref.RefOps.adv_loss, CycleGANStep(adv=...)) pinned against vectors of the
REAL reference (tests/golden/adv_modes.json, oracle/gen_golden_r2.py advmodes). CPU only."""
import json
import random
from pathlib import Path

import pytest
import torch

from oracle import torch_ref
from oracle.ops_ref import RefOps

GOLD = json.loads((Path(__file__).parent / "golden" / "adv_modes.json").read_text())


def adv_pred(seed=21):
    return torch.randn(8, 1, 30, 30, generator=torch.Generator().manual_seed(seed)) * 3.0


@pytest.mark.parametrize("mode", ["lsgan", "vanilla", "wgangp"])
@pytest.mark.parametrize("real", [True, False])
def test_objective_value_and_gradient(mode, real):
    g = GOLD["ops"][f"{mode}_{'real' if real else 'fake'}"]
    x = adv_pred().requires_grad_()
    val = torch_ref.adversarial_loss(x, real, mode)
    (gx,) = torch.autograd.grad(val, x)
    assert float(val) == pytest.approx(g["loss"], rel=1e-6, abs=1e-7)
    assert float(gx.double().norm()) == pytest.approx(g["grad_norm"], rel=1e-6)
    assert torch.allclose(gx.flatten()[g["idx"]], torch.tensor(g["grad_samples"]), rtol=1e-5, atol=1e-9)
    # the op-level oracle the HIP kernel is compared with
    loss, grad = torch.zeros(()), torch.empty(8, 1, 30, 30)
    RefOps().adv_loss(adv_pred(), mode, real, 1.0 if real else 0.0, loss=loss, grad=grad, grad_scale=torch.tensor(1.0))
    assert float(loss) == pytest.approx(g["loss"], rel=1e-6, abs=1e-7)
    assert torch.allclose(grad.flatten()[g["idx"]], torch.tensor(g["grad_samples"]), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("mode", ["lsgan", "vanilla", "wgangp"])
def test_dict_of_predictions_is_the_mean_over_keys(mode):
    d = {"a": adv_pred(22), "b": adv_pred(23)[:, :, :7, :7]}
    val = torch.stack([torch_ref.adversarial_loss(p, True, mode) for p in d.values()]).mean()
    assert float(val) == pytest.approx(GOLD["ops"][f"{mode}_dict_real"]["loss"], rel=1e-6, abs=1e-7)


@pytest.mark.parametrize("name", list(GOLD["steps"]))
def test_cyclegan_step_with_other_objectives(name):
    gold = GOLD["steps"][name]
    c = gold["config"]
    torch.set_num_threads(8)
    model = torch_ref.CycleGANStep(n_iters=c["n_iters"], n_iters_decay=c["n_iters_decay"], pool_size=c["pool_size"],
                                   lambda_identity=c["lambda_identity"], proportion_ssim=c["proportion_ssim"],
                                   adv=c["adv"], seed=c["seed"])
    random.seed(c["seed"])
    for s in range(c["steps"]):
        g = torch.Generator().manual_seed(c["seed"] * 100 + s)
        shape = (c["batch"], 3, c["size"], c["size"])
        A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
        losses, metrics = model.step(A, B)
        for k, v in gold["steps"][s]["losses"].items():
            # wgangp's D loss is a difference of two means of order 0.5: absolute floor instead of a relative one
            assert abs(losses[k] - v) <= 2e-4 * abs(v) + 2e-5, (s, k, losses[k], v)
        model.update_learning_rate()
