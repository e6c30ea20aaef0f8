"""Fused flat Adam over the executor's master buffers (gs_adam_step). Subclasses torch.optim.Optimizer so the
reference's LambdaLR schedule (nn/utils.py:83-99) and `param_groups[0]['lr']` logging (base.py:318-319) work
unchanged; constructor signature mirrors torch.optim.Adam(params, lr, betas) as used in cyclegan.py:81-82."""
import torch

from .native.backend import get_ops


class NativeAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        for p in params:
            if getattr(p, "_owner_net", None) is None:
                raise TypeError("NativeAdam only optimises the flat master parameters of NativeNet instances")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        ops = get_ops()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                net = p._owner_net
                if p.grad is None:
                    continue
                scale = net.finish_grad_reduction()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                ops.adam_step(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                              st["step"], grad_scale=scale, zero_grad=True)
                net.grad_dirty = False
                net.mark_packs_dirty()

    def zero_grad(self, set_to_none=True):
        """The update kernel already cleared the gradient it consumed; only buffers written since are cleared."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None and p._owner_net.grad_dirty:
                    p.grad.zero_()
                    p._owner_net.grad_dirty = False
