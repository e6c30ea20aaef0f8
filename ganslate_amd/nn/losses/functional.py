"""Autograd entry points of the fused loss kernels (libganslate_hip: gs_l1, gs_mse_const, gs_mean,
gs_ssim_distance). Each forward is one wavefront-reduced kernel writing a 0-d fp32 tensor; each backward is one
elementwise kernel that folds the upstream scalar gradient in (passed as a device pointer — no host sync)."""
import torch

from ..native.backend import get_ops


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        get_ops().l1(a, b, loss=loss)
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        get_ops().l1(a, b, grad_a=ga, grad_scale=g.contiguous())
        return (ga if ctx.needs_input_grad[0] else None), (-ga if ctx.needs_input_grad[1] else None)


class _MSEConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        x = x.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        get_ops().mse_const(x, target, loss=loss)
        ctx.save_for_backward(x)
        ctx.target = target
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        get_ops().mse_const(x, ctx.target, grad=gx, grad_scale=g.contiguous())
        return gx, None


def l1_loss(a, b):
    """mean(|a - b|)  — nn.L1Loss (cyclegan_losses.py:64,97-101)"""
    return _L1.apply(a.float(), b.float())


def mse_const_loss(x, target: float):
    """mean((x - target)^2) — nn.MSELoss against an expanded constant (adversarial_loss.py:28-29,60-62)"""
    return _MSEConst.apply(x.float(), float(target))


def mean_nograd(x):
    out = torch.empty((), dtype=torch.float32, device=x.device)
    get_ops().mean(x.detach().contiguous().float(), out)
    return out


def ssim_distance_nograd(x, y):
    """SSIMLoss.forward on (x+1)/2, (y+1)/2 (ssim.py:65-99), no gradient (training metric)."""
    out = torch.empty((), dtype=torch.float32, device=x.device)
    get_ops().ssim_distance(x.detach().contiguous().float(), y.detach().contiguous().float(), out)
    return out


def ssim_distance_autograd(X, Y):
    """Differentiable SSIM distance for `proportion_ssim > 0` (cyclegan_losses.py:78-90). The backward of the
    fused SSIM kernel is not written yet, so the loss form is composed from torch device ops."""
    import torch.nn.functional as F
    X, Y = (X + 1) / 2, (Y + 1) / 2
    if X.ndim == 5:
        X, Y = X.reshape(-1, *X.shape[2:]), Y.reshape(-1, *Y.shape[2:])
    ch = X.shape[1]
    coords = torch.arange(11, dtype=torch.float32, device=X.device) - 5
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum()).view(1, 1, 1, 11).repeat(ch, 1, 1, 1)

    def blur(t):
        return F.conv2d(F.conv2d(t, g, groups=ch), g.transpose(2, 3), groups=ch)

    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = blur(X), blur(Y)
    s1, s2, s12 = blur(X * X) - mu1 ** 2, blur(Y * Y) - mu2 ** 2, blur(X * Y) - mu1 * mu2
    S1 = (2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)
    S2 = (2 * s12 + C2) / (s1 + s2 + C2)
    return torch.sqrt(torch.relu(2 - (S1 + S2))).mean()
