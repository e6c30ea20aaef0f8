"""Host-side lowering (taps, parity classes, weight packs, pad fold) checked against torch.nn.functional and
autograd on the CPU through the op-level oracle (oracle/ops_ref.py). fp32 packs/activations -> tight tolerance."""
import pytest
import torch
import torch.nn.functional as F

from ganslate_amd.nn.native.spec import ConvSpec, lower, pad8
from oracle.ops_ref import RefOps, _fold

SPECS = [
    ConvSpec("conv", 5, 6, 3, 1, 1, pad_mode="reflect"),      # resnet2d.py:80-87 residual conv
    ConvSpec("conv", 3, 12, 7, 1, 3, pad_mode="reflect"),     # resnet2d.py:24-25 stem
    ConvSpec("conv", 8, 16, 3, 2, 1),                         # resnet2d.py:35 down-sampling
    ConvSpec("conv", 3, 8, 4, 2, 1),                          # patchgan2d.py:29
    ConvSpec("conv", 16, 9, 4, 1, 1),                         # patchgan2d.py:50-62 stride-1 k4
    ConvSpec("conv", 16, 1, 4, 1, 1),
    ConvSpec("convT", 16, 8, 3, 2, 1, 1),                     # resnet2d.py:52-57
    ConvSpec("convT", 8, 3, 4, 2, 1, 0),                      # unet2d.py:122
    ConvSpec("conv", 3, 8, 4, 3, 1),                          # stride 3 (selfattention_patchgan3d.py:33-36, 2-D twin)
]


SPECS_3D = [
    ConvSpec("conv", 4, 6, 3, 1, 1, pad_mode="replicate", dims=3),   # resnet3d.py:78-84 residual conv
    ConvSpec("conv", 1, 5, 7, 1, 3, pad_mode="replicate", dims=3),   # resnet3d.py:24-25 stem
    ConvSpec("conv", 8, 9, 3, 2, 1, dims=3),                         # resnet3d.py:36 down-sampling
    ConvSpec("conv", 2, 8, 4, 2, 1, dims=3),                         # patchgan3d.py:28
    ConvSpec("conv", 8, 1, 4, 1, 1, dims=3),                         # patchgan3d.py:57-60
    ConvSpec("convT", 9, 4, 3, 2, 1, 1, dims=3),                     # resnet3d.py:50-55
    ConvSpec("conv", 2, 8, 4, 3, 1, dims=3),                         # selfattention_patchgan3d.py:33-36: k4 stride 3
]


def torch_forward(spec, x, w, b):
    conv, convT = (F.conv2d, F.conv_transpose2d) if spec.dims == 2 else (F.conv3d, F.conv_transpose3d)
    if spec.kind == "conv":
        if spec.pad_mode in ("reflect", "replicate"):
            return conv(F.pad(x, (spec.pad,) * (2 * spec.dims), mode=spec.pad_mode), w, b, stride=spec.stride)
        return conv(x, w, b, stride=spec.stride, padding=spec.pad)
    return convT(x, w, b, stride=spec.stride, padding=spec.pad, output_padding=spec.out_pad)


_ID = lambda s: f"{s.kind}{s.k}s{s.stride}{s.pad_mode}{s.cin}x{s.cout}" if isinstance(s, ConvSpec) else None


@pytest.mark.parametrize("spec,size", [(s, hw) for s in SPECS for hw in [(10, 12), (9, 11)]] +
                         [(s, dhw) for s in SPECS_3D for dhw in [(8, 6, 10), (7, 9, 8)]], ids=_ID)
def test_lowering_matches_torch(spec, size):
    torch.manual_seed(0)
    ops = RefOps()
    N = 2
    x = torch.randn(N, spec.cin, *size, requires_grad=True)
    w = torch.randn(spec.torch_weight_shape(), requires_grad=True) * 0.2
    w.retain_grad()
    b = torch.randn(spec.cout, requires_grad=True)
    y = torch_forward(spec, x, w, b)
    gy = torch.randn_like(y)
    y.backward(gy)

    low = lower(spec, *size)
    assert low.out_dims == tuple(y.shape[2:])
    master = spec.master_from_torch(w.detach())
    fpack = torch.empty(low.fwd_index.size, dtype=torch.float32)
    ops.repack(master, torch.from_numpy(low.fwd_index), fpack)
    dpack = torch.empty(low.dgrad_index.size, dtype=torch.float32)
    ops.repack(master, torch.from_numpy(low.dgrad_index), dpack)
    bias = torch.zeros(spec.cout_p); bias[:spec.cout] = b.detach()

    xa = torch.zeros(N, *size, spec.cin_p); ops.image_to_act(x.detach(), xa)
    ya = torch.full((N, *low.out_dims, spec.cout_p), float("nan"))
    for g in low.fwd:
        ops.gconv(g, xa, fpack, bias, ya)
    assert torch.allclose(ya[..., :spec.cout].movedim(-1, 1), y.detach(), atol=1e-4, rtol=1e-4)
    assert torch.all(ya[..., spec.cout:] == 0)

    gya = torch.zeros(N, *low.out_dims, spec.cout_p); ops.image_to_act(gy, gya)
    f = low.dgrad_fold
    gxa = torch.full((N, *low.dgrad_dims, spec.cin_p), float("nan"))
    for g in low.dgrad:
        ops.gconv(g, gya, dpack, None, gxa)
    gx = _fold(gxa, size, f, spec.pad_mode)[..., :spec.cin].movedim(-1, 1)
    assert torch.allclose(gx, x.grad, atol=1e-4, rtol=1e-4)
    gimg = torch.empty_like(x.detach())
    ops.image_to_act_backward(gxa, gimg, fold=f, fold_mode=spec.pad_mode)
    assert torch.allclose(gimg, x.grad, atol=1e-4, rtol=1e-4)

    dw = torch.zeros(spec.P, spec.T, spec.Q)
    a, gth = (gya, xa) if spec.kind == "conv" else (xa, gya)
    ops.wgrad(low.wgrad, a, gth, dw)
    assert torch.allclose(spec.torch_from_master(dw), w.grad, atol=2e-3, rtol=1e-4)
    db = torch.zeros(spec.cout_p); ops.bias_grad(gya, spec.cout_p, db)
    assert torch.allclose(db[:spec.cout], b.grad, atol=2e-3, rtol=1e-4)


WFOLD_SPECS = [
    (ConvSpec("conv", 3, 12, 7, 1, 3, pad_mode="reflect", wfold="in"), (10, 12)),              # resnet2d.py:24-25
    (ConvSpec("conv", 12, 3, 7, 1, 3, pad_mode="reflect", wfold="out"), (10, 12)),             # resnet2d.py:64-65
    (ConvSpec("conv", 1, 6, 7, 1, 3, pad_mode="replicate", dims=3, wfold="in"), (8, 6, 10)),   # resnet3d.py:24-25
    (ConvSpec("conv", 6, 1, 7, 1, 3, pad_mode="replicate", dims=3, wfold="out"), (7, 9, 8)),   # resnet3d.py:64
    (ConvSpec("conv", 2, 5, 5, 1, 2, dims=3, wfold="in"), (6, 7, 8)),                          # zero padding
    (ConvSpec("conv", 5, 2, 5, 1, 2, dims=3, wfold="out"), (6, 7, 8)),
]


@pytest.mark.parametrize("spec,size", WFOLD_SPECS, ids=lambda v: _ID(v) + v.wfold if isinstance(v, ConvSpec) else None)
def test_wfold_lowering_matches_torch(spec, size):
    """W taps folded into channels (csrc/wfold.hip): unfold / shift-add boundary transforms + the k x [k x] 1 conv
    reproduce the plain convolution and all of its gradients"""
    torch.manual_seed(1)
    ops = RefOps(act_dtype=torch.float32)
    N = 2
    x = torch.randn(N, spec.cin, *size, requires_grad=True)
    w = torch.randn(spec.torch_weight_shape(), requires_grad=True) * 0.2
    w.retain_grad()
    b = torch.randn(spec.cout, requires_grad=True)
    y = torch_forward(spec, x, w, b)
    gy = torch.randn_like(y)
    y.backward(gy)

    low = lower(spec, *size)
    master = spec.master_from_torch(w.detach())
    assert torch.equal(spec.torch_from_master(master), w.detach())
    fpack = torch.empty(low.fwd_index.size, dtype=torch.float32)
    ops.repack(master, torch.from_numpy(low.fwd_index), fpack)
    dpack = torch.empty(low.dgrad_index.size, dtype=torch.float32)
    ops.repack(master, torch.from_numpy(low.dgrad_index), dpack)
    bias = torch.zeros(spec.cout_p); bias[:spec.cout] = b.detach()
    f = low.dgrad_fold

    if spec.wfold == "in":
        xa = torch.full((N, *size, spec.cin_p), float("nan"))
        ops.image_unfold(x.detach(), xa, spec.k, spec.pad, spec.pad_mode)
        ya = torch.full((N, *low.out_dims, spec.cout_p), float("nan"))
        for g in low.fwd:
            ops.gconv(g, xa, fpack, bias, ya)
        assert torch.allclose(ya[..., :spec.cout].movedim(-1, 1), y.detach(), atol=1e-4, rtol=1e-4)
        gya = torch.zeros(N, *low.out_dims, spec.cout_p); ops.image_to_act(gy, gya)
        gxa = torch.full((N, *low.dgrad_dims, spec.cin_p), float("nan"))
        for g in low.dgrad:
            ops.gconv(g, gya, dpack, None, gxa)
        gimg = torch.empty_like(x.detach())
        ops.image_unfold_backward(gxa, gimg, spec.k, spec.pad, f, spec.pad_mode)
        assert torch.allclose(gimg, x.grad, atol=1e-4, rtol=1e-4)
        db = torch.zeros(spec.cout_p); ops.bias_grad(gya, spec.cout_p, db)
    else:
        xa = torch.zeros(N, *size, spec.cin_p); ops.image_to_act(x.detach(), xa)
        za = torch.full((N, *low.out_dims, spec.cout_p), float("nan"))
        assert low.out_dims[-1] == size[-1] + 2 * spec.pad
        for g in low.fwd:
            ops.gconv(g, xa, fpack, None, za)
        img = torch.empty_like(y.detach())
        ops.shiftadd_to_image(za, bias, img, spec.k)
        assert torch.allclose(img, y.detach(), atol=1e-4, rtol=1e-4)
        gya = torch.full((N, *low.out_dims, spec.cout_p), float("nan"))
        ops.shiftadd_to_image_backward(gy, None, gya, spec.k)
        gxa = torch.full((N, *low.dgrad_dims, spec.cin_p), float("nan"))
        for g in low.dgrad:
            ops.gconv(g, gya, dpack, None, gxa)
        gx = _fold(gxa, size, f, spec.pad_mode)[..., :spec.cin].movedim(-1, 1)
        assert torch.allclose(gx, x.grad, atol=1e-4, rtol=1e-4)
        db = torch.zeros(spec.cout_p); ops.bias_grad(gya, spec.cout_p, db)   # first `cout` entries = the dw = 0 slice
    dw = torch.zeros(spec.P, spec.T, spec.Q)
    ops.wgrad(low.wgrad, gya, xa, dw)
    assert torch.allclose(spec.torch_from_master(dw), w.grad, atol=2e-3, rtol=1e-4)
    assert torch.allclose(db[:spec.cout], b.grad, atol=2e-3, rtol=1e-4)


def test_master_roundtrip():
    for spec in SPECS + SPECS_3D:
        w = torch.randn(spec.torch_weight_shape())
        assert torch.equal(spec.torch_from_master(spec.master_from_torch(w)), w)


def test_every_pack_of_the_headline_networks_has_a_group_index():
    """NativeNet._repack_plan for the layer types of Resnet2D-9 / PatchGAN2D / Unet2D: every group of 8 pack elements is either
    8 consecutive master elements along k, padding, or part of a transposed segment that is regular along its rows — no group
    needs the element-wise table — and expanding both group indices gives back the element-wise tables"""
    import numpy as np
    from ganslate_amd.nn.native.net import NativeNet
    from ganslate_amd.nn.native.spec import ConvSpec, lower
    specs = [ConvSpec("conv", 3, 64, 7, 1, 3, pad_mode="reflect", wfold="in"), ConvSpec("conv", 64, 128, 3, 2, 1),
             ConvSpec("conv", 256, 256, 3, 1, 1, pad_mode="reflect"), ConvSpec("convT", 256, 128, 3, 2, 1, 1),
             ConvSpec("conv", 64, 3, 7, 1, 3, pad_mode="reflect", wfold="out"), ConvSpec("conv", 6, 64, 4, 2, 1),
             ConvSpec("conv", 256, 512, 4, 1, 1), ConvSpec("convT", 1024, 512, 4, 2, 1),
             ConvSpec("conv", 16, 16, 5, 1, 2, dims=3), ConvSpec("conv", 1, 16, 5, 1, 2, dims=3)]
    lows = [lower(s, *((64, 64) if s.dims == 2 else (8, 8, 8))) for s in specs]
    w_off = np.cumsum([0] + [s.master_numel + 24 for s in specs])        # odd gaps: biases / slopes between the weights
    for which in ("fwd", "dgrad"):
        idx, offs, o = [], [], 0
        for i, lw in enumerate(lows):
            t = getattr(lw, which + "_index").astype(np.int64)
            t[t >= 0] += w_off[i]
            idx.append(t); offs.append(o); o += t.size
        plan = NativeNet._repack_plan(lows, idx, offs, which)
        flat = np.concatenate(idx)
        assert not plan["need_index"] and len(plan["seg"]) >= 4, (which, plan["need_index"], len(plan["seg"]))
        g = plan["groups"].astype(np.int64)[:, None]
        full = np.where(g >= 0, g + np.arange(8)[None, :], -1)
        full[plan["groups"] == -3] = -7
        full = full.reshape(-1)
        tiles = 0
        for off, goff, rows, kp, first in plan["seg"].tolist():
            assert first == tiles and (full[off:off + rows * kp] == -7).all()
            tg = plan["tgroups"][goff:goff + rows // 8 * kp].astype(np.int64).reshape(rows // 8, 1, kp)
            full[off:off + rows * kp] = np.where(tg >= 0, tg + np.arange(8).reshape(1, 8, 1), -1).reshape(-1)
            tiles += (rows + 63) // 64 * (kp // 64)
        assert tiles == plan["tiles"]
        assert np.array_equal(full, flat)
