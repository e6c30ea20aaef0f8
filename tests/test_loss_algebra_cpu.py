"""The scalar algebra of the loss assembly (`losses/functional.py:scalar_affine`, `fanout`) on the oracle backend: the values
and gradients of the reference's written-out expressions (cyclegan_losses.py:21-32,70-101, cyclegan.py:150,182)."""
import pytest
import torch

from ganslate_amd.nn.losses.functional import fanout, scalar_affine, scalar_sum
from ganslate_amd.nn.native import backend
from oracle.ops_ref import RefOps


@pytest.fixture
def oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


def test_weighted_losses_and_their_gradients(oracle_backend):
    lam, alpha = 10.0, 0.84
    ssim, l1, adv = (torch.tensor(v, requires_grad=True) for v in (0.31, 0.27, 0.9))
    ref = lam * (alpha * ssim + (1 - alpha) * l1)          # cyclegan_losses.py:23, 86-90
    total_ref = ref + adv
    total_ref.backward()
    want = [t.grad.clone() for t in (ssim, l1, adv)]
    for t in (ssim, l1, adv):
        t.grad = None
    cyc, = scalar_affine([ssim, l1], [[lam * alpha, lam * (1 - alpha)]])
    total = scalar_sum([cyc, adv])
    assert cyc.item() == pytest.approx(ref.item(), rel=1e-6) and total.item() == pytest.approx(total_ref.item(), rel=1e-6)
    total.backward()
    for t, w in zip((ssim, l1, adv), want):
        assert t.grad.item() == pytest.approx(w.item(), rel=1e-6)


def test_terms_that_are_not_device_scalars_fall_back_to_torch_operators(oracle_backend):
    x = torch.tensor([1.0, 2.0], requires_grad=True)        # a user criterion that returns a vector
    out, = scalar_affine([x, 3.0], [[2.0, 1.0]], [0.5])
    assert torch.equal(out.detach(), torch.tensor([5.5, 7.5]))
    out.sum().backward()
    assert torch.equal(x.grad, torch.tensor([2.0, 2.0]))


def test_fanout_joins_the_two_gradients(oracle_backend):
    x = torch.randn(2, 3, 8, 8, requires_grad=True)
    y = x * 1.0
    a, b = fanout(y)
    (a.sum() * 2 + (b * b).sum()).backward()
    assert torch.allclose(x.grad, 2 + 2 * x.detach())
    y = x.detach()
    assert fanout(y)[0] is y
