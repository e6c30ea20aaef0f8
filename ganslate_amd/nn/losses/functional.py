"""Autograd entry points of the fused loss kernels (libganslate_hip: gs_l1, gs_mse_const, gs_adv_loss, gs_mean,
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


class _Adversarial(torch.autograd.Function):
    """gs_adv_loss: vanilla / wgangp give a 0-d loss, nonsaturating one loss per sample"""
    @staticmethod
    def forward(ctx, x, mode, target_is_real, label):
        x = x.contiguous()
        loss = torch.empty((x.shape[0],) if mode == "nonsaturating" else (), dtype=torch.float32, device=x.device)
        get_ops().adv_loss(x, mode, target_is_real, label, loss=loss)
        ctx.save_for_backward(x)
        ctx.args = (mode, target_is_real, label)
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        get_ops().adv_loss(x, *ctx.args, grad=gx, grad_scale=g.contiguous().float())
        return gx, None, None, None


def adversarial_loss(x, mode: str, target_is_real: bool, label: float):
    """AdversarialLoss.calculate_loss for the modes other than lsgan (adversarial_loss.py:52-73)"""
    return _Adversarial.apply(x.float(), mode, bool(target_is_real), float(label))


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


class _SSIMDistance(torch.autograd.Function):
    """SSIM distance as a loss term: fused forward (gs_ssim_distance) and hand-written backward
    (gs_ssim_distance_backward: gradient maps + transposed separable Gaussian); the distance is symmetric, so the
    gradient w.r.t. the first image is the same kernel with the arguments swapped."""

    @staticmethod
    def forward(ctx, X, Y):
        X, Y = X.contiguous(), Y.contiguous()
        out = torch.empty((), dtype=torch.float32, device=X.device)
        get_ops().ssim_distance(X, Y, out)
        ctx.save_for_backward(X, Y)
        return out

    @staticmethod
    def backward(ctx, g):
        X, Y = ctx.saved_tensors
        g = g.contiguous().float()
        gx = gy = None
        if ctx.needs_input_grad[1]:
            gy = torch.empty_like(Y)
            get_ops().ssim_distance_backward(X, Y, gy, grad_scale=g)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(X)
            get_ops().ssim_distance_backward(Y, X, gx, grad_scale=g)
        return gx, gy


def ssim_distance_autograd(X, Y):
    """Differentiable SSIM distance for `proportion_ssim > 0` (cyclegan_losses.py:78-90; SSIMLoss on (x+1)/2)."""
    return _SSIMDistance.apply(X.float(), Y.float())


class _ScalarAffine(torch.autograd.Function):
    """out_r = c_r + sum_k rows[r][k] * x_k over 0-d losses, one launch (gs_scalar_affine); the backward is the same launch
    with the transposed matrix over the rows' upstream gradients (an unused row contributes nothing)."""

    @staticmethod
    def forward(ctx, rows, consts, *xs):
        out = get_ops().scalar_affine([x.detach() for x in xs], rows, consts)
        ctx.rows = rows
        ctx.set_materialize_grads(False)
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        rows, K = ctx.rows, len(ctx.rows[0])
        gs = [None if g is None else g.contiguous().float() for g in gs]
        gx = get_ops().scalar_affine(gs, [[rows[r][k] for r in range(len(rows))] for k in range(K)]).unbind(0)
        return (None, None) + tuple(gx[k] if ctx.needs_input_grad[2 + k] else None for k in range(K))


def scalar_affine(xs, rows, consts=None):
    """[c_r + sum_k rows[r][k] * xs[k] for r] for 0-d fp32 losses `xs` — the scalar algebra of a loss assembly as one kernel
    (forward) and one kernel (backward) instead of one torch elementwise launch per operator. Falls back to torch
    operators in the written order when a term is not a 0-d fp32 tensor on the compute device (a user criterion returning
    something else), or for more than 16 terms / 8 rows."""
    ok = 0 < len(xs) <= 16 and 0 < len(rows) <= 8 and all(
        torch.is_tensor(x) and x.dim() == 0 and x.dtype == torch.float32 and x.device == xs[0].device for x in xs)
    if not ok:
        out = []
        for r, row in enumerate(rows):
            acc = None
            for w, x in zip(row, xs):
                if w != 0:
                    term = x if w == 1 else w * x
                    acc = term if acc is None else acc + term
            if consts is not None and consts[r] != 0:
                acc = consts[r] if acc is None else acc + consts[r]
            out.append(acc)
        return out
    rows = tuple(tuple(float(w) for w in row) for row in rows)
    return list(_ScalarAffine.apply(rows, None if consts is None else tuple(float(c) for c in consts), *xs))


def scalar_sum(xs):
    """sum of 0-d losses (one launch)"""
    return scalar_affine(list(xs), [[1.0] * len(xs)])[0]


class _Fanout(torch.autograd.Function):
    """Two aliases of one tensor whose gradients are joined by the library's own kernel (gs_sum2_f32): a generated image
    that feeds both a discriminator and the other generator would otherwise have its two gradients added by autograd's
    accumulation (a torch kernel)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view(x.shape), x.view(x.shape)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        g1, g2 = g1.contiguous(), g2.contiguous()
        if g1.dtype != torch.float32 or g2.dtype != torch.float32 or g1.data_ptr() % 16 or g2.data_ptr() % 16:
            return g1 + g2
        return get_ops().sum2(g1, g2)


def fanout(x):
    """(x, x) for two consumers that both send a gradient back"""
    return _Fanout.apply(x) if (torch.is_tensor(x) and x.requires_grad) else (x, x)
