"""Unet2D generator (pix2pix U-Net) on the HIP kernels — constructor, channel plan, block order, bias rule, dropout
placement and state_dict names of ganslate/nn/generators/unet/unet2d.py:17-157:

  level k = 1..D (D = num_downs), inner channels c_k = ngf * min(2^(k-1), 8):
    down_k : [LeakyReLU(0.2)] -> Conv2d(k4, s2, p1) -> [InstanceNorm]      (no act for k=1; no norm for k=1 and k=D)
    up_k   : ReLU -> ConvTranspose2d(k4, s2, p1) -> [InstanceNorm] [-> Dropout(0.5)]   (k=1: bias, Tanh, no norm)
    every non-outermost block returns torch.cat([x, up(sub(down(x)))], 1)

The skip tensor h_k is read through two activations (LeakyReLU by down_{k+1}, ReLU by up_k as the first half of the
concat), so it is materialised twice by one kernel (gs_norm_act_forward_ex); the concat never exists as a copy —
producers write channel slices of the buffer up_k reads, and the gradient of that buffer is consumed as two slices.
Dropout masks are a hash of (seed, image, element) regenerated in the backward pass."""
import random
from dataclasses import dataclass

import torch

from .... import configs
from ...native.net import NativeNet, Node
from ...native.spec import ConvSpec, lower
from ...utils import is_bias_before_norm, require_instance_norm


@dataclass
class Unet2DConfig(configs.base.BaseGeneratorConfig):
    num_downs: int = 7
    ngf: int = 64
    use_dropout: bool = False


class _Saved:
    pass


class Unet2D(NativeNet):
    dims = 2          # Unet3D (unet3d.py:17-156) is the same graph on Conv3d / ConvTranspose3d / InstanceNorm3d

    def __init__(self, in_channels, out_channels, num_downs, norm_type, ngf=64, use_dropout=False):
        require_instance_norm(norm_type)
        dims = type(self).dims
        use_bias = is_bias_before_norm(norm_type)
        assert num_downs >= 5, "Unet2D needs num_downs >= 5 (unet2d.py:36-66)"
        assert ngf % 8 == 0, "ngf must be a multiple of 8"
        D = self.D = num_downs
        self.c = [in_channels] + [ngf * min(2 ** (k - 1), 8) for k in range(1, D + 1)]   # c[0] = image channels
        self.dropout_levels = set(range(5, D)) if use_dropout else set()                  # the ngf*8 middle blocks
        # reference module paths: outermost Sequential [downconv, sub, uprelu, upconv, tanh]; middle
        # [downrelu, downconv, downnorm, sub, uprelu, upconv, upnorm(, dropout)]; innermost [downrelu, downconv,
        # uprelu, upconv, upnorm]
        prefix = {1: "model.model"}
        for k in range(2, D + 1):
            prefix[k] = prefix[k - 1] + (".1" if k == 2 else ".3") + ".model"
        down_name = lambda k: prefix[k] + (".0" if k == 1 else ".1")
        up_name = lambda k: prefix[k] + (".3" if k in (1, D) else ".5")
        nodes = []
        for k in range(1, D + 1):
            nodes.append(Node(ConvSpec("conv", self.c[k - 1], self.c[k], 4, 2, 1, bias=use_bias, dims=dims),
                              norm=(1 < k < D), act="lrelu", name=down_name(k)))
        for k in range(D, 0, -1):
            cin = self.c[k] if k == D else 2 * self.c[k]
            cout = out_channels if k == 1 else self.c[k - 1]
            nodes.append(Node(ConvSpec("convT", cin, cout, 4, 2, 1, 0, bias=True if k == 1 else use_bias, dims=dims),
                              norm=(k > 1), act="none", name=up_name(k)))
        super().__init__(nodes, in_channels, out_channels, out_act="tanh")
        self.external_draw = False       # True while a captured step owns the launches: the recipe draws before replay
        self._seed_dev = torch.zeros(2, dtype=torch.int32, device=self.device) if self.dropout_levels else None

    def prepare_host_state(self):
        """host side of one training forward: a fresh 64-bit dropout seed from Python's RNG (seeded by the Trainer like
        the reference's global seeds), uploaded to the device words the norm_ex kernels read"""
        if self._seed_dev is None or not self.training:
            return
        bits = random.getrandbits(62)
        words = torch.tensor([bits & 0xFFFFFFFF, bits >> 32], dtype=torch.int64).to(torch.int32)   # wraps to two's complement
        if self._seed_dev.is_cuda:
            words = words.pin_memory()
        self._seed_dev.copy_(words, non_blocking=True)

    def _down(self, k):
        return k - 1

    def _up(self, k):
        return 2 * self.D - k

    def _lowered(self, *sizes):
        key = tuple(sizes)
        if key not in self._low_cache:
            D = self.D
            assert all(x % (1 << D) == 0 for x in key), f"input {key} must be divisible by 2^{D}"
            lows = [None] * (2 * D)
            for k in range(1, D + 1):
                lows[self._down(k)] = lower(self.nodes[self._down(k)].spec, *(x >> (k - 1) for x in key))
                lows[self._up(k)] = lower(self.nodes[self._up(k)].spec, *(x >> k for x in key))
            self._low_cache[key] = lows
        return self._low_cache[key]

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _conv(self, i, lw, pk, x, y, act="none", stats=False):
        """forward of node i (all parity classes); returns mean/rstd when stats are requested"""
        ops, sp, N = self.ops, self.nodes[i].spec, x.shape[0]
        m = self.master.detach()
        bias = m[self.b_off[i]:self.b_off[i] + sp.cout_p]
        fpack = pk["fpack"][pk["f_off"][i]:]
        if not stats:
            ops.gconv_classes(lw.fwd, x, fpack, bias, y, act=act, slope=0.2)
            return None
        slots, offs = 0, []
        for g in lw.fwd:
            offs.append(slots)
            slots += ops.stat_slots(g, N)
        part = torch.empty(N * slots * 2 * sp.cout_p, dtype=torch.float32, device=self.device)
        ops.gconv_classes(lw.fwd, x, fpack, bias, y, stats=part, stats_slots=slots, stats_slot0s=offs)
        mr = torch.empty(N * 2 * sp.cout_p, dtype=torch.float32, device=self.device)
        ops.inorm_finalize(part, N, slots, sp.cout_p, lw.out_pixels, mr)
        return mr

    def _new(self, N, sizes, C):
        return torch.empty(N, *sizes, C, dtype=self.ops.act_dtype, device=self.device)

    # ---- forward -----------------------------------------------------------------------------------------------------
    def _forward(self, x, save):
        ops, D, c = self.ops, self.D, self.c
        N, sizes = x.shape[0], tuple(x.shape[2:])
        lows, pk = self._lowered(*sizes), self._get_packs(*sizes)
        s = _Saved()
        s.x_img, s.N, s.sizes, s.lows = x, N, sizes, lows
        # nn.Dropout draws a new mask per forward (unet2d.py:146-147). The 64-bit mask seed lives in device memory
        # (gs_norm_ex_desc.seed_dev): a captured step replays these launches with identical arguments, so the recipe
        # uploads a fresh seed before every replay (prepare_host_state); launch by launch the pass draws it itself.
        s.seed = 1 if (self.training and self.dropout_levels) else 0     # host part: flag + per-level offset
        s.seed_dev = None
        if s.seed:
            if not self.external_draw:
                self.prepare_host_state()
            s.seed_dev = self._seed_dev
        a0 = self._new(N, sizes, self.nodes[0].spec.cin_p)
        ops.image_to_act(x, a0)
        s.L = {0: a0}          # L[k]: LeakyReLU(h_k), the input of down_{k+1}
        s.cat, s.yd, s.mrd, s.yu, s.mru = {}, {}, {}, {}, {}
        for k in range(1, D + 1):
            i, lw = self._down(k), lows[self._down(k)]
            sk = tuple(x >> k for x in sizes)
            if k < D:
                s.cat[k] = self._new(N, sk, 2 * c[k])
            if k == 1:            # conv + bias, no norm; LeakyReLU in the epilogue, ReLU copy into the concat buffer
                s.L[1] = self._new(N, sk, c[1])
                self._conv(i, lw, pk, s.L[0], s.L[1], act="lrelu")
                ops.norm_act_forward_ex(s.L[1], None, s.cat[1], None, act1="relu")
            elif k < D:
                s.yd[k] = self._new(N, sk, c[k])
                s.mrd[k] = self._conv(i, lw, pk, s.L[k - 1], s.yd[k], stats=True)
                s.L[k] = self._new(N, sk, c[k])
                ops.norm_act_forward_ex(s.yd[k], s.mrd[k], s.L[k], s.cat[k], act1="lrelu", act2="relu")
            else:                 # innermost: no norm; its only consumer is up_D through ReLU
                s.R = self._new(N, sk, c[D])
                self._conv(i, lw, pk, s.L[D - 1], s.R, act="relu")
        for k in range(D, 0, -1):
            i, lw = self._up(k), lows[self._up(k)]
            xin = s.R if k == D else s.cat[k]
            y = self._new(N, tuple(x >> (k - 1) for x in sizes), self.nodes[i].spec.cout_p)
            if k > 1:
                s.yu[k] = y
                s.mru[k] = self._conv(i, lw, pk, xin, y, stats=True)
                p = 0.5 if (k in self.dropout_levels and self.training) else 0.0
                ops.norm_act_forward_ex(y, s.mru[k], s.cat[k - 1], None, act1="relu", x1_co=c[k - 1], drop_p=p,
                                        seed=s.seed + k, seed_dev=s.seed_dev if p else None)
            else:
                self._conv(i, lw, pk, xin, y)
                s.y1 = y
        out = torch.empty(N, self.out_channels, *sizes, dtype=torch.float32, device=self.device)
        ops.act_to_image(s.y1, out, act="tanh")
        s.out_img = out
        return out, (s if save else None)

    # ---- backward ----------------------------------------------------------------------------------------------------
    def _param_grads(self, i, lw, x_in, dy, has_norm):
        ops, sp, grad = self.ops, self.nodes[i].spec, self.master.grad
        a_t, g_t = (dy, x_in) if sp.kind == "conv" else (x_in, dy)
        fresh, fuse = self.wgrad_fresh(i), getattr(self, "_early_fuse", None)
        # one backward pass per optimiser step (NativeAdam.arm_early) and a layer of few pixels: gradient + update in one launch
        args, which = fuse(self, i) if (fuse is not None and fresh) else (None, None)
        if args is not None and ops.wgrad_adam(lw.wgrad, a_t, g_t, *args):
            self._early_fused.append((self.w_off[i], self.w_off[i] + sp.master_numel))
            if which is not None:                          # (its transposed pack came with the launch)
                self._early_tr.append((i, which))
        else:
            ops.wgrad(lw.wgrad, a_t, g_t, grad[self.w_off[i]:self.w_off[i] + sp.master_numel], fresh=fresh)
        if sp.bias and not has_norm:
            ops.bias_grad(dy, sp.cout_p, grad[self.b_off[i]:self.b_off[i] + sp.cout_p])
        self.grad_dirty = True

    def _dgrad(self, i, lw, pk, dy, N):
        sp = self.nodes[i].spec
        gx = self._new(N, lw.in_dims, sp.cin_p)
        dpack = pk["dpack"][pk["d_off"][i]:]
        self.ops.gconv_classes(lw.dgrad, dy, dpack, None, gx)
        return gx

    def _backward(self, s, g_img, need_input_grad, want_w):
        ops, D, c, N = self.ops, self.D, self.c, s.N
        lows, pk = s.lows, self._get_packs(*s.sizes)
        if self.master.grad is None:
            self.master.grad = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        grad = self.master.grad
        final_pass = want_w and self._dist is not None and self._fw_pending == 0 and not self.external_reduce
        bias_slice = lambda i: grad[self.b_off[i]:self.b_off[i] + self.nodes[i].spec.cout_p] if want_w else None
        dy = torch.empty_like(s.y1)
        ops.act_to_image_backward(g_img.contiguous().float(), s.out_img, dy, act="tanh")
        gcat = {}
        # a layer whose weight gradient launch also updates its weights (gs_wgrad_adam) must have launched its data gradient:
        # parameter gradients after the data gradient then
        pg_last = want_w and getattr(self, "_early_fuse", None) is not None
        # ---- up path, outermost first ----
        for k in range(1, D + 1):
            i, lw = self._up(k), lows[self._up(k)]
            xin = s.R if k == D else s.cat[k]
            if want_w and not pg_last:
                self._param_grads(i, lw, xin, dy, has_norm=(k > 1))
                if final_pass:
                    self._maybe_reduce_bucket(i)
            gcat[k] = self._dgrad(i, lw, pk, dy, N)        # gradient w.r.t. ReLU(cat([h_k, u_{k+1}])) (or ReLU(h_D))
            if want_w:                                     # (node up(k)'s bias gradient came with iteration k - 1 for k > 1)
                if pg_last:
                    self._param_grads(i, lw, xin, dy, has_norm=(k > 1))
                self._early_step_at(i)
            if k < D:                                      # second half -> u_{k+1} = drop(IN(up_{k+1} raw))
                dy = torch.empty_like(s.yu[k + 1])
                p = 0.5 if (k + 1 in self.dropout_levels and s.seed) else 0.0
                ops.norm_act_backward_ex(gcat[k], None, s.yu[k + 1], s.mru[k + 1], dy, act1="relu", g1_co=c[k],
                                         drop_p=p, seed=s.seed + k + 1, seed_dev=s.seed_dev if p else None,
                                         bias_grad=bias_slice(self._up(k + 1)))
        # ---- down path, innermost first ----
        gL = None
        for k in range(D, 0, -1):
            i, lw = self._down(k), lows[self._down(k)]
            if k == D:
                dy = torch.empty_like(s.R)
                ops.norm_act_backward_ex(gcat[D], None, s.R, None, dy, act1="relu")
            elif k > 1:
                dy = torch.empty_like(s.yd[k])
                ops.norm_act_backward_ex(gL, gcat[k], s.yd[k], s.mrd[k], dy, act1="lrelu", act2="relu",
                                         bias_grad=bias_slice(i))
            else:
                dy = torch.empty_like(s.L[1])
                ops.norm_act_backward_ex(gL, gcat[1], s.L[1], None, dy, act1="lrelu", act2="relu")
            if want_w and not pg_last:
                self._param_grads(i, lw, s.L[k - 1], dy, has_norm=(1 < k < D))
                if final_pass:
                    self._maybe_reduce_bucket(i)
            if k > 1 or need_input_grad:
                gL = self._dgrad(i, lw, pk, dy, N)
            if want_w:
                if pg_last:
                    self._param_grads(i, lw, s.L[k - 1], dy, has_norm=(1 < k < D))
                self._early_step_at(i)
        if not need_input_grad:
            return None
        g_in = torch.empty_like(s.x_img)
        ops.image_to_act_backward(gL, g_in, fold=0)
        return g_in
