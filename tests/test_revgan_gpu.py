"""RevGAN on the HIP path: the inverse direction of the partially-invertible V-Nets (gs_pnorm_forward
res_mode 3, *_ba layers) against the fp32 oracle — which tests/test_revgan_cpu.py pins to the real reference — and the
RevGAN training step against the reference's golden iterations (tests/golden/revgan.json)."""
import json
import random
from pathlib import Path

import pytest
import torch

from oracle import torch_ref

from .envelope import step_tolerance
from .test_cyclegan_gpu import cosine, rel_l2
from .test_revgan_cpu import _product_revgan

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / "golden" / "revgan.json").read_text())


@pytest.mark.parametrize("dims,ch,c,blocks,shape", [(3, 1, 8, ((1, 2), (2, 1)), (1, 1, 16, 24, 32)),
                                                     (2, 2, 8, ((1, 2), (2, 1)), (2, 2, 64, 96)),
                                                     (2, 2, 8, None, (1, 2, 128, 192))])
def test_both_directions_hip_vs_oracle(hip_ops, dims, ch, c, blocks, shape):
    """y = G(x), r = G(y, inverse=True) and the gradients of a loss on both (RevGAN's use of the network, revgan.py:120-130)
    on the HIP kernels, against the fp32 oracle and against the bf16 CPU emulation of the same executor.

    Outputs: 3e-2 / 4e-2 relative L2 (measured 0.9 % and 1.6-2.4 %). Conv-weight gradients of the shallow networks:
    measured 0.23-0.28 / cos 0.96-0.97 against the emulation and 0.35 / 0.94 against fp32 — the emulation itself is 0.34-0.37
    from fp32: two chained passes double the PReLU-kink flips of tests/test_cyclegan_gpu.py::_net_case. The PReLU slope
    gradients (sums over the negative pre-activations only, with the seeded slopes near zero) scatter by 0.5-0.8 between
    the emulation and fp32 alike and are not compared; the reference's default 14-coupling Vnet2D (112 stages chained)
    loses the gradient direction to bf16 storage in the emulation too (0.62 / cos 0.82), so only its outputs and input
    gradient are checked. The arithmetic of res_mode 3 is pinned at op level (tests/test_ops_gpu.py::test_pnorm_*) and the
    executor's inverse path in fp32 (tests/test_networks_cpu.py::test_vnet3d_inverse_direction)."""
    from ganslate_amd.nn.generators import Vnet2D, Vnet3D
    from ganslate_amd.nn.native import backend
    from oracle.ops_ref import RefOps
    V, R = (Vnet3D, torch_ref.Vnet3D) if dims == 3 else (Vnet2D, torch_ref.Vnet2D)
    kw = {} if blocks is None else dict(down_blocks=blocks[0], up_blocks=blocks[1])
    deep = blocks is None
    shadow = R(ch, ch, c, use_inverse=True, **kw)
    sd = torch_ref.seeded_state_dict(shadow, 81)
    shadow.load_state_dict(sd)
    g = torch.Generator().manual_seed(82)
    x = torch.rand(shape, generator=g) * 2 - 1
    gy, gr = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    xa = x.clone().requires_grad_()
    ya = shadow(xa)
    ra = shadow(ya, inverse=True)
    ((ya * gy).sum() + (ra * gr).sum()).backward()
    res = {}
    for name, ops in (("hip", hip_ops),) + (() if deep else (("emu", RefOps(act_dtype=torch.bfloat16)),)):
        backend.set_ops(ops)
        try:
            net = V(ch, ch, "instance", c, **kw)
            net.load_state_dict(sd)
            xb = x.clone().to(ops.device).requires_grad_()
            yb = net(xb)
            rb = net(yb, inverse=True)
            ((yb * gy.to(ops.device)).sum() + (rb * gr.to(ops.device)).sum()).backward()
            if ops.device.type == "cuda":
                torch.cuda.synchronize()
            res[name] = (yb.detach().cpu(), rb.detach().cpu(), xb.grad.cpu(),
                         {k: v.float().cpu() for k, v in net.grads_state_dict().items()})
        finally:
            backend.set_ops(hip_ops)
    yh, rh, gxh, gh = res["hip"]
    assert rel_l2(yh, ya.detach()) <= 3e-2 and rel_l2(rh, ra.detach()) <= 4e-2
    if deep:
        assert cosine(gxh, xa.grad) >= 0.85, cosine(gxh, xa.grad)       # measured 0.894
        return
    assert rel_l2(gxh, xa.grad) <= 0.35 and cosine(gxh, xa.grad) >= 0.94, (rel_l2(gxh, xa.grad), cosine(gxh, xa.grad))
    bad = []
    for n, p in shadow.named_parameters():
        if n.startswith("encoder.") or p.dim() == 1:
            continue
        a, e = gh[n], res["emu"][3][n]
        if rel_l2(a, e) > 0.33 or cosine(a, e) < 0.94 or rel_l2(a, p.grad) > 0.42 or cosine(a, p.grad) < 0.91:
            bad.append((n, round(rel_l2(a, e), 3), round(cosine(a, e), 3), round(rel_l2(a, p.grad), 3),
                        round(cosine(a, p.grad), 3)))
    assert not bad, bad


@pytest.mark.parametrize("name,conf_name", [("rev3d_16x32x32", "revgan3d_synthetic.yaml"),
                                            ("rev3d_piresnet", "revgan3d_piresnet_synthetic.yaml"),
                                            ("rev2d_64x64_idt", "revgan2d_synthetic.yaml")])
def test_revgan_training_step_matches_reference_golden(hip_ops, name, conf_name):
    gold = GOLD["steps"][name]
    c = gold["config"]
    model = _product_revgan(c, conf_name, extra=("train.cuda=True",))
    ch = 1 if c["dims"] == 3 else 2
    for s in range(c["steps"]):
        g = torch.Generator().manual_seed(c["seed"] * 100 + s)
        shape = (c["batch"], ch, *c["size"])
        A, B = torch.rand(shape, generator=g) * 2 - 1, torch.rand(shape, generator=g) * 2 - 1
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        torch.cuda.synchronize()
        losses = {k: float(v.detach()) for k, v in model.losses.items() if v is not None}
        want = gold["steps"][s]["losses"]
        assert set(losses) == set(want)
        for k, v in want.items():
            assert losses[k] == pytest.approx(v, rel=step_tolerance(k, s)), (s, k, losses[k], v)
        model.update_learning_rate()
    out = model.infer(A.to(hip_ops.device), "BA")
    assert out.shape == A.shape and bool(torch.isfinite(out).all())


def test_memory_saving_frees_the_couplings_activations(hip_ops):
    """use_memory_saving on the GPU: the live activation memory between forward and backward no longer grows with the number
    of couplings (measured through the caching allocator's peak), and the gradients stay those of the kept-activation run
    up to the bf16 rounding of the rebuilt inputs"""
    from ganslate_amd.nn.generators import Vnet3D
    shadow = torch_ref.Vnet3D(1, 1, 16, (4, 4), (4, 4), use_inverse=True)
    sd = torch_ref.seeded_state_dict(shadow, 83)
    g = torch.Generator().manual_seed(83)
    x = torch.rand(1, 1, 32, 64, 64, generator=g) * 2 - 1
    gy = torch.randn(x.shape, generator=g)
    res = {}
    for saving in (False, True):
        net = Vnet3D(1, 1, "instance", 16, (4, 4), (4, 4), use_memory_saving=saving, use_inverse=True)
        net.load_state_dict(sd)
        xi = x.clone().to(hip_ops.device).requires_grad_()
        net(xi).sum().backward()                       # warm-up: packs, gradient buffer, workspaces
        xi.grad = None
        net.master.grad.zero_()
        import gc
        gc.collect()                                   # (cyclic garbage of earlier tests freed DURING the pass below would be
        torch.cuda.synchronize()                       #  subtracted from what the pass holds: seen as a negative reading)
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = net(xi)
        held = torch.cuda.memory_allocated() - base    # what the recorded pass keeps alive for backward
        y.backward(gy.to(hip_ops.device))
        torch.cuda.synchronize()
        res[saving] = (held, y.detach().float().cpu(), xi.grad.cpu(),
                       {k: v.float().cpu() for k, v in net.grads_state_dict().items()})
        del net, xi, y
    print(f"\nactivations held between forward and backward: {res[False][0] / 2**20:.0f} MiB kept, "
          f"{res[True][0] / 2**20:.0f} MiB with memory saving")
    assert res[True][0] < 0.6 * res[False][0]
    assert torch.equal(res[True][1], res[False][1])
    assert rel_l2(res[True][2], res[False][2]) <= 0.15 and cosine(res[True][2], res[False][2]) >= 0.98
    for k, a in res[False][3].items():
        if a.dim() > 1 and float(a.abs().max()) > 0:         # (the *_ba layers saw no pass here)
            assert cosine(res[True][3][k], a) >= 0.97, (k, cosine(res[True][3][k], a))


def test_piresnet3d_both_directions_hip_vs_oracle(hip_ops):
    """Piresnet3D on the HIP kernels (replicate-padded k5 / k3 convs with their padding fold in the norm backward,
    gs_slice_stats pre-norms, res_mode 3 inverse) against the fp32 oracle twin and the bf16 CPU emulation"""
    from ganslate_amd.nn.generators import Piresnet3D
    from ganslate_amd.nn.native import backend
    from oracle.ops_ref import RefOps
    shadow = torch_ref.Piresnet3D(1, 1, 3, 16, use_inverse=True)
    sd = torch_ref.seeded_state_dict(shadow, 84)
    shadow.load_state_dict(sd)
    g = torch.Generator().manual_seed(85)
    shape = (1, 1, 16, 24, 32)
    x = torch.rand(shape, generator=g) * 2 - 1
    gy, gr = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    xa = x.clone().requires_grad_()
    ya = shadow(xa)
    ra = shadow(ya, inverse=True)
    ((ya * gy).sum() + (ra * gr).sum()).backward()
    res = {}
    for name, ops in (("hip", hip_ops), ("emu", RefOps(act_dtype=torch.bfloat16))):
        backend.set_ops(ops)
        try:
            net = Piresnet3D(1, 1, "instance", 3, 16)
            net.load_state_dict(sd)
            xb = x.clone().to(ops.device).requires_grad_()
            yb = net(xb)
            rb = net(yb, inverse=True)
            ((yb * gy.to(ops.device)).sum() + (rb * gr.to(ops.device)).sum()).backward()
            if ops.device.type == "cuda":
                torch.cuda.synchronize()
            res[name] = (yb.detach().cpu(), rb.detach().cpu(), xb.grad.cpu(),
                         {k: v.float().cpu() for k, v in net.grads_state_dict().items()})
        finally:
            backend.set_ops(hip_ops)
    yh, rh, gxh, gh = res["hip"]
    assert rel_l2(yh, ya.detach()) <= 3e-2 and rel_l2(rh, ra.detach()) <= 4e-2, (rel_l2(yh, ya.detach()), rel_l2(rh, ra.detach()))
    assert rel_l2(gxh, xa.grad) <= 0.35 and cosine(gxh, xa.grad) >= 0.94, (rel_l2(gxh, xa.grad), cosine(gxh, xa.grad))
    bad = []
    for n, p in shadow.named_parameters():
        if p.dim() == 1:
            continue
        a, e = gh[n], res["emu"][3][n]
        if rel_l2(a, e) > 0.33 or cosine(a, e) < 0.94 or rel_l2(a, p.grad) > 0.42 or cosine(a, p.grad) < 0.91:
            bad.append((n, round(rel_l2(a, e), 3), round(cosine(a, e), 3), round(rel_l2(a, p.grad), 3),
                        round(cosine(a, p.grad), 3)))
    assert not bad, bad
