"""Twin execution (ganslate_amd/nn/native/twin.py): two networks of identical architecture as ONE batch must compute what
the two separate passes compute — outputs, input gradients, every parameter gradient of BOTH networks — including the
merged weight-gradient launches of two passes per network (cyclegan.py:139-150), multi-part batches (D(real) and D(fake)
as one pass, cyclegan.py:154-189) and a whole CycleGAN iteration. fp32 oracle backend, CPU: this pins the executor's host
logic; the native twin kernels are pinned in tests/test_twin_gpu.py."""
import os
import random

import pytest
import torch

from ganslate_amd.nn.native import backend
from ganslate_amd.nn.native.twin import Twin, TwinNet
from oracle.ops_ref import RefOps

from .helpers import build_product_cyclegan, golden_inputs, load_golden_steps


@pytest.fixture()
def fp32_oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


def _pair(make, seed):
    torch.manual_seed(seed)
    a, b = make(), make()
    a.init_weights("normal", 0.05)
    b.init_weights("normal", 0.05)
    return a, b


def _clone(make, src):
    net = make()
    net.load_state_dict(src.state_dict())
    return net


def _close(x, y, tol=2e-5):
    scale = max(x.abs().max().item(), 1e-12)
    assert (x - y).abs().max().item() <= tol * scale, ((x - y).abs().max().item(), scale)


def test_twin_container_slices_like_a_tensor():
    t = Twin(torch.arange(10.), torch.arange(10.) + 100)
    s = t[2:5]
    assert torch.equal(s.a, torch.tensor([2., 3., 4.])) and torch.equal(s.b, torch.tensor([102., 103., 104.]))
    assert t.half(0) is t.a and t.half(1) is t.b


def test_twin_generators_equal_the_two_separate_passes(fp32_oracle_backend):
    from ganslate_amd.nn.generators import Resnet2D
    make = lambda: Resnet2D(3, 3, "instance", 2)
    a, b = _pair(make, 5)
    a1, b1 = _clone(make, a), _clone(make, b)
    g = torch.Generator().manual_seed(7)
    xa, xb = (torch.rand(2, 3, 32, 32, generator=g) * 2 - 1 for _ in range(2))
    ga, gb = (torch.randn(2, 3, 32, 32, generator=g) for _ in range(2))
    # two passes per network, like the two phases of a CycleGAN step: the weight gradients of the passes are merged
    xs = [t.clone().requires_grad_() for t in (xa, xb, xa, xb)]
    ya, yb = TwinNet(a, b)(xs[0], xs[1])
    za, zb = TwinNet(a, b)(yb, ya)
    ((za * ga).sum() + (zb * gb).sum() + (ya * gb).sum()).backward()
    a.flush_deferred_wgrads(); b.flush_deferred_wgrads()
    ya1, yb1 = a1(xs[2]), b1(xs[3])
    za1, zb1 = a1(yb1), b1(ya1)
    ((za1 * ga).sum() + (zb1 * gb).sum() + (ya1 * gb).sum()).backward()
    a1.flush_deferred_wgrads(); b1.flush_deferred_wgrads()
    for got, ref in ((ya, ya1), (yb, yb1), (za, za1), (zb, zb1), (xs[0].grad, xs[2].grad), (xs[1].grad, xs[3].grad)):
        _close(got, ref)
    for net, ref in ((a, a1), (b, b1)):
        for (k, v), (k1, v1) in zip(net.grads_state_dict().items(), ref.grads_state_dict().items()):
            assert k == k1
            if v1.abs().max() > 0:
                _close(v, v1, 5e-5)


def test_twin_discriminators_take_real_and_fake_as_one_batch(fp32_oracle_backend):
    from ganslate_amd.nn.discriminators import PatchGAN2D
    make = lambda: PatchGAN2D(3, 64, 3, (4, 4), "instance")
    a, b = _pair(make, 11)
    a1, b1 = _clone(make, a), _clone(make, b)
    g = torch.Generator().manual_seed(3)
    rb, fb, ra, fa = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1 for _ in range(4))
    (pr_b, pf_b), (pr_a, pf_a) = TwinNet(a, b)((rb, fb), (ra, fa))
    loss = ((pr_b - 1) ** 2).mean() + (pf_b ** 2).mean() + ((pr_a - 1) ** 2).mean() + (pf_a ** 2).mean()
    loss.backward()
    refs = []
    for net, real, fake in ((a1, rb, fb), (b1, ra, fa)):
        pr, pf = net(real), net(fake)
        (((pr - 1) ** 2).mean() + (pf ** 2).mean()).backward()
        net.flush_deferred_wgrads()
        refs += [pr, pf]
    for got, ref in zip((pr_b, pf_b, pr_a, pf_a), refs):
        _close(got, ref)
    for net, ref in ((a, a1), (b, b1)):
        for (k, v), (_, v1) in zip(net.grads_state_dict().items(), ref.grads_state_dict().items()):
            if v1.abs().max() > 0:
                _close(v, v1, 5e-5)


def test_frozen_twin_discriminators_pass_input_gradients_only(fp32_oracle_backend):
    from ganslate_amd.nn.discriminators import PatchGAN2D
    make = lambda: PatchGAN2D(3, 64, 3, (4, 4), "instance")
    a, b = _pair(make, 13)
    for net in (a, b):
        net.master.requires_grad = False
    g = torch.Generator().manual_seed(4)
    xa, xb = (torch.rand(1, 3, 64, 64, generator=g).requires_grad_() for _ in range(2))
    pa, pb = TwinNet(a, b)(xa, xb)
    (pa.sum() + 2 * pb.sum()).backward()
    xa1, xb1 = xa.detach().clone().requires_grad_(), xb.detach().clone().requires_grad_()
    (a(xa1).sum() + 2 * b(xb1).sum()).backward()
    _close(xa.grad, xa1.grad); _close(xb.grad, xb1.grad)
    assert a.master.grad.abs().max() == 0 and b.master.grad.abs().max() == 0


def test_incompatible_networks_are_refused(fp32_oracle_backend):
    from ganslate_amd.nn.generators import Resnet2D
    a, b = Resnet2D(3, 3, "instance", 2), Resnet2D(3, 3, "instance", 3)
    assert not TwinNet.compatible(a, b) and not TwinNet.compatible(a, a)
    with pytest.raises(ValueError):
        TwinNet(a, b)


@pytest.mark.parametrize("name", ["c64_default", "c64_idt_ssim"])
def test_cyclegan_iterations_with_and_without_twin_passes_agree(fp32_oracle_backend, name, monkeypatch):
    c = load_golden_steps()[name]["config"]
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GS_TWIN", mode)
        model = build_product_cyclegan(c)
        assert (model.twin_G is not None) == (mode == "1") and (model.twin_D is not None) == (mode == "1")
        random.seed(c["seed"])
        out = []
        for step in range(2):
            a, b = golden_inputs(c, step)
            model.set_input({"A": a, "B": b})
            model.optimize_parameters()
            out.append({k: float(v.detach()) for k, v in model.losses.items() if v is not None})
        runs[mode] = (out, {n: net.master.detach().clone() for n, net in model.networks.items()})
    for s in range(2):
        for k, v in runs["0"][0][s].items():
            assert runs["1"][0][s][k] == pytest.approx(v, rel=1e-4 if s == 0 else 2e-2, abs=1e-6), (s, k)
    for n, w in runs["0"][1].items():       # two Adam steps: +-lr per step on every weight (sign noise flips a few)
        assert (runs["1"][1][n] - w).abs().mean().item() <= 2e-5, n


@pytest.mark.parametrize("memory_saving", [False, True], ids=["plain", "recompute"])
def test_twin_vnets_equal_the_two_separate_passes(fp32_oracle_backend, memory_saving):
    """Vnet3D._forward / _backward with a twin partner (round 6): the PReLU slopes, their gradients and the bias gradients out of
    the norm reductions are per network (pnorm launches as halves), everything else one batch. Outputs, input gradients and the
    whole flat gradient of BOTH networks against the two separate passes."""
    from ganslate_amd.nn.generators import Vnet3D
    make = lambda: Vnet3D(1, 1, "instance", first_layer_channels=8, down_blocks=(1, 1), up_blocks=(1, 1),
                          use_memory_saving=memory_saving, use_inverse=False)
    a, b = _pair(make, 11)
    assert TwinNet.compatible(a, b)
    a1, b1 = _clone(make, a), _clone(make, b)
    g = torch.Generator().manual_seed(12)
    xa, xb = (torch.rand(1, 1, 8, 8, 8, generator=g) * 2 - 1 for _ in range(2))
    ga, gb = (torch.randn(1, 1, 8, 8, 8, generator=g) for _ in range(2))
    xs = [t.clone().requires_grad_() for t in (xa, xb, xa, xb)]
    ya, yb = TwinNet(a, b)(xs[0], xs[1])
    ((ya * ga).sum() + (yb * gb).sum()).backward()
    ya1, yb1 = a1(xs[2]), b1(xs[3])
    ((ya1 * ga).sum() + (yb1 * gb).sum()).backward()
    for got, ref in ((ya, ya1), (yb, yb1), (xs[0].grad, xs[2].grad), (xs[1].grad, xs[3].grad),
                     (a.master.grad, a1.master.grad), (b.master.grad, b1.master.grad)):
        _close(got.detach(), ref.detach())
    # a network with the inverse path (RevGAN's shared V-Net) does not pair up
    inv = lambda: Vnet3D(1, 1, "instance", first_layer_channels=8, down_blocks=(1, 1), up_blocks=(1, 1), use_inverse=True)
    assert not TwinNet.compatible(*_pair(inv, 13))
