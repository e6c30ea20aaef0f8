"""Piresnet3D — the partially-invertible ResNet generator RevGAN was published with — on the HIP kernels: constructor,
channel plan, state_dict names and both directions of ganslate/nn/generators/resnet/piresnet3d.py:29-108.

  downconv : ReplicationPad(2) -> conv5 (cin -> c) -> IN -> ReLU -> conv3 s2 p1 (c -> 2c) -> IN -> ReLU        (:60-75)
  core     : `depth` additive couplings on the channel halves (h = c), y1 = x1 + F(x2), y2 = x2 + G(y1) and their inverse
             (invertible.py:8-48), F = G-shaped  IN -> ReplicationPad(1) -> conv3 (h -> h) -> IN -> ReLU       (:104-108)
  upconv   : convT3 s2 p1 op1 (2c -> c) -> IN -> ReLU -> ReplicationPad(2) -> conv5 (c -> cout) -> tanh       (:77-88)
  forward(x, inverse=True) uses downconv_ba / upconv_ba and runs the core backwards                             (:90-101)

The halves of a coupling are kept as separate h-channel tensors here (the core has no skip connection to share a buffer
with), so every norm is a whole-tensor InstanceNorm: the generic kernels with their padding fold (gs_inorm_act_backward
folds the replicate-padded data gradients of the k3 / k5 convs) do all the work, the coupling function's leading norm takes
its statistics from gs_slice_stats, and the inverse's subtraction is gs_pnorm_forward's res_mode 3. `use_memory_saving`
rebuilds coupling inputs from outputs in the backward pass like the V-Nets (vnet3d.py in this package)."""
from dataclasses import dataclass

import torch

from .... import configs
from ...native.net import NativeNet, Node
from ...native.spec import ConvSpec, lower
from ...utils import is_bias_before_norm, require_instance_norm
from ..vnet.vnet3d import Vnet3D, _Saved


@dataclass
class Piresnet3DConfig(configs.base.BaseGeneratorConfig):
    """Partially-invertible Resnet generator - a version of ResNet compatible with RevGAN."""
    use_memory_saving: bool = True
    use_inverse: bool = True
    first_layer_channels: int = 32
    depth: int = configs.base.MISSING


class Piresnet3D(Vnet3D):
    """(inherits the executor helpers of the V-Net: conv + statistics, weight / data gradient launches, the
    forward(x, inverse) entry; the network itself is built and walked here)"""
    dims = 3

    def __init__(self, in_channels, out_channels, norm_type, depth, first_layer_channels=64, use_memory_saving=True,
                 use_inverse=True):
        require_instance_norm(norm_type)
        c = first_layer_channels
        assert c >= 8 and c % 8 == 0, "first_layer_channels must be a multiple of 8 (16-byte channel granule)"
        use_bias = is_bias_before_norm(norm_type)
        self.use_inverse, self.use_memory_saving = bool(use_inverse), bool(use_memory_saving)
        self.c, self.depth = c, int(depth)
        conv = lambda *a, **k: ConvSpec(*a, dims=3, **k)
        nodes, order = [], []

        def add(spec, norm, name):
            nodes.append(Node(spec, norm, "none", name=name))
            order.append(f"{name}.weight")
            if spec.bias:
                order.append(f"{name}.bias")
            return len(nodes) - 1

        def down(tag):
            return (add(conv("conv", in_channels, c, 5, 1, 2, pad_mode="replicate", bias=use_bias), True, f"downconv_{tag}.1"),
                    add(conv("conv", c, 2 * c, 3, 2, 1, bias=use_bias), True, f"downconv_{tag}.4"))

        def up(tag):
            return (add(conv("convT", 2 * c, c, 3, 2, 1, 1, bias=use_bias), True, f"upconv_{tag}.0"),
                    add(conv("conv", c, out_channels, 5, 1, 2, pad_mode="replicate"), False, f"upconv_{tag}.4"))
        # (node order = layout of the flat master buffer: first node reads the image, last node writes it; the reference's
        # registration order — downconv_ab, upconv_ab, downconv_ba, upconv_ba, core, piresnet3d.py:49-58 — is kept in
        # reference_parameter_order for checkpoints)
        d_ab = down("ab")
        self.couplings = []
        for j in range(self.depth):
            ent = []
            for fn in ("Fm", "Gm"):
                ent.append(add(conv("conv", c, c, 3, 1, 1, pad_mode="replicate", bias=use_bias), True,
                               f"core.sequence.{j}.invertible_block._fn.{fn}.2"))
            self.couplings.append(tuple(ent))
        self.ends = {"ab": d_ab + up("ab")}
        if self.use_inverse:
            if in_channels != out_channels:
                raise ValueError("use_inverse needs in_channels == out_channels")
            self.ends["ba"] = down("ba") + up("ba")
        NativeNet.__init__(self, nodes, in_channels, out_channels, out_act="tanh")
        rank = lambda k: {"downconv_ab": 0, "upconv_ab": 1, "downconv_ba": 2, "upconv_ba": 3, "core": 4}[k.split(".")[0]]
        self._param_order = sorted(order, key=rank)
        self._zeros = {}

    # ---- lowering ------------------------------------------------------------------------------------------------------
    def _lowered(self, *sizes):
        key = tuple(sizes)
        if key not in self._low_cache:
            assert all(x % 2 == 0 for x in key), f"input {key} must have even sides"
            half = tuple(x // 2 for x in key)
            lows = [None] * len(self.nodes)
            for d1, d2, u1, u2 in self.ends.values():
                lows[d1] = lower(self.nodes[d1].spec, *key)
                lows[d2] = lower(self.nodes[d2].spec, *key)
                lows[u1] = lower(self.nodes[u1].spec, *half)
                lows[u2] = lower(self.nodes[u2].spec, *key)
            for nf, ng in self.couplings:
                lows[nf] = lower(self.nodes[nf].spec, *half)
                lows[ng] = lower(self.nodes[ng].spec, *half)
            self._low_cache[key] = lows
        return self._low_cache[key]

    def _zero_slope(self, C):
        z = self._zeros.get(C)
        if z is None or z.device != self.device:
            z = self._zeros[C] = torch.zeros(C, dtype=torch.float32, device=self.device)
        return z

    # ---- the coupling function  IN -> pad -> conv3 -> IN -> ReLU  and one coupling in either direction -----------------
    def _fn_forward(self, s, node, x):
        """returns (n = IN(x), mr0, raw conv output, its mean / rstd)"""
        ops, h = self.ops, self.c
        mr0 = torch.empty(s.N * 2 * h, dtype=torch.float32, device=self.device)
        ops.slice_stats(x, 0, h, mr0)
        n = torch.empty_like(x)
        ops.inorm_act_forward(x, mr0, None, n, act="none")
        y, mr = self._conv(s, node, n)
        return n, mr0, y, mr

    def _pair_forward(self, s, cp, x1, x2):
        """y1 = x1 + F(x2), y2 = x2 + G(y1)"""
        ops = self.ops
        nf, ng = cp
        na, mr0a, ya, mra = self._fn_forward(s, nf, x2)
        y1 = torch.empty_like(x1)
        ops.inorm_act_forward(ya, mra, x1, y1, act="relu")
        nb, mr0b, yb, mrb = self._fn_forward(s, ng, y1)
        y2 = torch.empty_like(x2)
        ops.inorm_act_forward(yb, mrb, x2, y2, act="relu")
        return (y1, y2), (na, mr0a, ya, mra, nb, mr0b, yb, mrb)

    def _pair_inverse(self, s, cp, y1, y2):
        """x2 = y2 - G(y1), x1 = y1 - F(x2)"""
        ops, h = self.ops, self.c
        nf, ng = cp
        z = self._zero_slope(h)
        nb, mr0b, yb, mrb = self._fn_forward(s, ng, y1)
        x2 = torch.empty_like(y2)
        ops.pnorm_forward(yb, mrb, x2, C=h, slope=z, res=y2, res_mode=3)
        na, mr0a, ya, mra = self._fn_forward(s, nf, x2)
        x1 = torch.empty_like(y1)
        ops.pnorm_forward(ya, mra, x1, C=h, slope=z, res=y1, res_mode=3)
        return (x1, x2), (na, mr0a, ya, mra, nb, mr0b, yb, mrb)

    def _fn_backward(self, s, node, g_out, sign, x, rec, want_w, g_into):
        """gradient of +-F(x) w.r.t. x added into g_into; g_out is the gradient w.r.t. the coupling output that holds the
        branch; rec = (n, mr0, y, mr) of this function"""
        ops, h = self.ops, self.c
        n, mr0, y, mr = rec
        dy = torch.empty_like(y)
        if sign > 0:
            ops.inorm_act_backward(g_out, None, y, mr, dy, None, act="relu", bias_grad=self._bias_slice(node, want_w))
        else:
            ops.pnorm_backward(g_out, y, mr, dy, C=h, slope=self._zero_slope(h), res_mode=3,
                               bias_grad=self._bias_slice(node, want_w))
        if want_w:
            self._wgrad(s, node, n, dy)
        lw, sp = s.lows[node], self.nodes[node].spec
        gn_pad = self._new(s.N, lw.dgrad_dims, sp.cin_p)
        self.ops.gconv_classes(lw.dgrad, dy, s.pk["dpack"][s.pk["d_off"][node]:], None, gn_pad)
        dx = torch.empty_like(x)
        ops.inorm_act_backward(gn_pad, None, x, mr0, dx, None, fold=lw.dgrad_fold, fold_mode=sp.pad_mode, act="none")
        ops.add_views(g_into, dx, h, accumulate=True)

    # ---- forward ----------------------------------------------------------------------------------------------------
    def _forward(self, x, save, stop=None):
        assert stop is None, "feature taps are not implemented for Piresnet3D"
        ops, c = self.ops, self.c
        N, sizes = x.shape[0], tuple(x.shape[2:])
        half = tuple(v // 2 for v in sizes)
        s = _Saved()
        inv = s.inverse = bool(getattr(self, "_next_inverse", False))
        s.recompute = bool(save and self.use_memory_saving)
        d1, d2, u1, u2 = self.ends["ba" if inv else "ab"]
        s.x_img, s.N, s.sizes = x, N, sizes
        s.lows, s.pk = self._lowered(*sizes), self._get_packs(*sizes)
        a0 = self._new(N, sizes, self.nodes[d1].spec.cin_p)
        ops.image_to_act(x, a0)
        s.a0 = a0
        s.y1, s.mr1 = self._conv(s, d1, a0)
        s.t1 = self._new(N, sizes, c)
        ops.inorm_act_forward(s.y1, s.mr1, None, s.t1, act="relu")
        s.y2, s.mr2 = self._conv(s, d2, s.t1)
        D0 = self._new(N, half, 2 * c)
        ops.inorm_act_forward(s.y2, s.mr2, None, D0, act="relu")
        a, b = self._new(N, half, c), self._new(N, half, c)
        ops.add_views(a, D0, c, dst_co=0, src_co=0, accumulate=False)
        ops.add_views(b, D0, c, dst_co=0, src_co=c, accumulate=False)
        s.coup = []
        for cp in (reversed(self.couplings) if inv else self.couplings):
            (na_, nb_), rec = (self._pair_inverse if inv else self._pair_forward)(s, cp, a, b)
            s.coup.append(None if s.recompute else ((a, b), (na_, nb_), rec))
            a, b = na_, nb_
        s.core_out = (a, b)
        Xn = self._new(N, half, 2 * c)
        ops.add_views(Xn, a, c, dst_co=0, src_co=0, accumulate=False)
        ops.add_views(Xn, b, c, dst_co=c, src_co=0, accumulate=False)
        s.Xn = Xn
        s.y3, s.mr3 = self._conv(s, u1, Xn)
        s.t3 = self._new(N, sizes, c)
        ops.inorm_act_forward(s.y3, s.mr3, None, s.t3, act="relu")
        s.z, _ = self._conv(s, u2, s.t3, stats=False)
        out = torch.empty(N, self.out_channels, *sizes, dtype=torch.float32, device=self.device)
        ops.act_to_image(s.z, out, act="tanh")
        s.out_img = out
        return out, (s if save else None)

    # ---- backward ------------------------------------------------------------------------------------------------------
    def _backward(self, s, g_img, need_input_grad, want_w, start=None, inj_x=None, inj_y=None):
        assert start is None and not inj_x and not inj_y, "feature taps are not implemented for Piresnet3D"
        ops, c, N = self.ops, self.c, s.N
        if self.master.grad is None:
            self.master.grad = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        grad = self.master.grad
        inv = s.inverse
        d1, d2, u1, u2 = self.ends["ba" if inv else "ab"]
        dpack = lambda i: s.pk["dpack"][s.pk["d_off"][i]:]
        # upconv: tanh, conv5 (replicate-padded: its data gradient lives on the padded domain and is folded by the norm)
        gz = torch.empty_like(s.z)
        ops.act_to_image_backward(g_img.contiguous().float(), s.out_img, gz, act="tanh")
        sp = self.nodes[u2].spec
        if want_w:
            self._wgrad(s, u2, s.t3, gz)
            ops.bias_grad(gz, sp.cout_p, grad[self.b_off[u2]:self.b_off[u2] + sp.cout_p])
        lw = s.lows[u2]
        g_pad = self._new(N, lw.dgrad_dims, sp.cin_p)
        ops.gconv_classes(lw.dgrad, gz, dpack(u2), None, g_pad)
        dy3 = torch.empty_like(s.y3)
        ops.inorm_act_backward(g_pad, None, s.y3, s.mr3, dy3, None, fold=lw.dgrad_fold, fold_mode=sp.pad_mode, act="relu",
                               bias_grad=self._bias_slice(u1, want_w))
        if want_w:
            self._wgrad(s, u1, s.Xn, dy3)
        gX = self._dgrad(s, u1, dy3)                       # w.r.t. the core output [.., 2c]
        half = gX.shape[1:-1]
        ga, gb = self._new(N, half, c), self._new(N, half, c)
        ops.add_views(ga, gX, c, dst_co=0, src_co=0, accumulate=False)
        ops.add_views(gb, gX, c, dst_co=0, src_co=c, accumulate=False)
        # core, last applied coupling first. (ga, gb) is the gradient w.r.t. the coupling's outputs and becomes the gradient
        # w.r.t. its inputs in place
        cur = s.core_out
        order = list(reversed(self.couplings)) if inv else list(self.couplings)
        for cp, kept in zip(reversed(order), reversed(s.coup)):
            nf, ng = cp
            if kept is None:                             # memory saving: rebuild the inputs from the outputs
                ins, rec = (self._pair_forward if inv else self._pair_inverse)(s, cp, *cur)
                outs, cur = cur, ins
            else:
                ins, outs, rec = kept
            na, mr0a, ya, mra, nb, mr0b, yb, mrb = rec
            if not inv:      # y1 = x1 + F(x2), y2 = x2 + G(y1): ga = dL/dy1, gb = dL/dy2
                (x1, x2), (y1, y2) = ins, outs
                self._fn_backward(s, ng, gb, +1, y1, (nb, mr0b, yb, mrb), want_w, ga)      # y1 also fed G
                self._fn_backward(s, nf, ga, +1, x2, (na, mr0a, ya, mra), want_w, gb)      # x2 also fed F
            else:            # x2 = y2 - G(y1), x1 = y1 - F(x2): ga = dL/dx1, gb = dL/dx2 -> dL/dy1, dL/dy2
                (y1, y2), (x1, x2) = ins, outs
                self._fn_backward(s, nf, ga, -1, x2, (na, mr0a, ya, mra), want_w, gb)      # x2 also fed F
                self._fn_backward(s, ng, gb, -1, y1, (nb, mr0b, yb, mrb), want_w, ga)      # y1 also fed G
        gD0 = self._new(N, half, 2 * c)
        ops.add_views(gD0, ga, c, dst_co=0, src_co=0, accumulate=False)
        ops.add_views(gD0, gb, c, dst_co=c, src_co=0, accumulate=False)
        # downconv
        dy2 = torch.empty_like(s.y2)
        ops.inorm_act_backward(gD0, None, s.y2, s.mr2, dy2, None, act="relu", bias_grad=self._bias_slice(d2, want_w))
        if want_w:
            self._wgrad(s, d2, s.t1, dy2)
        gt1 = self._dgrad(s, d2, dy2)
        dy1 = torch.empty_like(s.y1)
        ops.inorm_act_backward(gt1, None, s.y1, s.mr1, dy1, None, act="relu", bias_grad=self._bias_slice(d1, want_w))
        if want_w:
            self._wgrad(s, d1, s.a0, dy1)
        if not need_input_grad:
            return None
        lw, sp = s.lows[d1], self.nodes[d1].spec
        gx_pad = self._new(N, lw.dgrad_dims, sp.cin_p)
        ops.gconv_classes(lw.dgrad, dy1, dpack(d1), None, gx_pad)
        g_in = torch.empty_like(s.x_img)
        ops.image_to_act_backward(gx_pad, g_in, fold=lw.dgrad_fold, fold_mode=sp.pad_mode)
        return g_in
