"""Vnet3D generator (partially-invertible V-Net) on the HIP kernels — constructor, channel plan, block order, bias rule
and state_dict names of ganslate/nn/generators/vnet/vnet3d.py:27-267 with ganslate/nn/invertible.py:8-48:

  InputBlock : out1 = PReLU(IN(conv5(x)) + x.repeat(c / in_channels))                               (vnet3d.py:155-168)
  DownBlock i: down = PReLU(IN(conv k2 s2 (C -> 2C)));  out = PReLU(core(down) + down)               (vnet3d.py:171-202)
  UpBlock i  : up = PReLU(IN(convT k2 s2 (Cin -> Cout/2)));  xcat = cat(up, skip);  out = PReLU(core(xcat) + xcat)
  core       : n additive couplings on channel halves, y1 = x1 + F(x2), y2 = x2 + G(y1), F = G-shaped
               PReLU(IN(conv5(h -> h)))  (memcnn.AdditiveCoupling; `disable=True` wrapper = plain autograd, the
               brats yaml's use_memory_saving=False / use_inverse=False, invertible.py:15-19)
  OutBlock   : tanh(conv1(PReLU(IN(conv5(x)))))                                                    (vnet3d.py:245-259)

Execution: a coupling never splits or concatenates — convs read / write channel slices of the 2h-channel block
buffers (in_co / out_co of gs_gconv_desc), the norm + PReLU + residual chain is one kernel per half
(gs_pnorm_forward), and in the backward pass the gradient buffer of [y1|y2] is turned into the gradient of [x1|x2] in
place by data-gradient launches that accumulate into a slice (gs_gconv_desc.accumulate). PReLU slopes live in the flat
master buffer behind the conv weights (`Extra`), their gradients come out of the norm-backward reductions.

`use_inverse=True` (RevGAN, ganslate/nn/gans/unpaired/revgan.py:120-146) adds the B->A copies of the non-invertible layers
(in_ba, out_ba, down_conv_ba, up_conv_ba: vnet3d.py:60-69,179-181,213-214) and `forward(x, inverse=True)`: the same walk
with those layers and every core run backwards, x2 = y2 - G(y1), x1 = y1 - F(x2) per coupling in reversed block order
(memcnn AdditiveCoupling.inverse through invertible.py:21-24,36-48) — gs_pnorm_forward's res_mode 3; the tail PReLUs and
the couplings' weights are shared by both directions.

`use_memory_saving=True` is memcnn's activation recompute (InvertibleModuleWrapper(keep_input=False), the point of the
RevGAN paper): the forward pass keeps only each core's OUTPUT; the backward pass walks a core from its last coupling to its
first and rebuilds every coupling's input from its output with the inverse equations — which produces exactly the conv
outputs F(x2), G(y1) the gradient needs — before differentiating it. Per core the live activations drop from
O(couplings) to O(1) at the price of one extra evaluation of F and G per coupling (+1/3 of the core's work). The rebuilt
input equals the original up to storage rounding (bf16 here, fp32 in memcnn)."""
from dataclasses import dataclass
from typing import Tuple

import torch

from .... import configs
from ...native.net import Extra, NativeNet, Node, attention_extras
from ...native.twin import Twin
from ...native.spec import ConvSpec, lower
from ...utils import is_bias_before_norm, require_instance_norm


@dataclass
class Vnet3DConfig(configs.base.BaseGeneratorConfig):
    """Partially-invertible V-Net generator."""
    use_memory_saving: bool = False
    use_inverse: bool = False
    first_layer_channels: int = 16
    down_blocks: Tuple[int] = (1, 2, 3, 2)
    up_blocks: Tuple[int] = (2, 2, 1, 1)
    is_separable: bool = False


class _Saved:
    pass


class _Block:
    """bookkeeping of one Down/Up block: node indices and PReLU names"""

    def __init__(self):
        self.conv = None            # strided conv / transposed conv node
        self.conv_slope = None
        self.couplings = []         # [(node_F, slope_F, node_G, slope_G)]
        self.tail_slope = None
        self.C = 0                  # channels of the block (2h)
        self.level = 0              # resolution level of the block's tensors (0 = input resolution)


class Vnet3D(NativeNet):
    dims = 3
    twin_extras_ok = True        # TwinNet: the PReLU slopes (extras) are handled per network (see _tw below)
    twin_default = False         # ... but recipes pair V-Nets only on request (GS_TWIN=all): measured slower on the brats recipe
    _tw = None                   # the second network of a twin pass while that pass is being launched: every per-network
                                 # tensor (master, packs, gradient buffer, slopes) is then a Twin and the batch holds both
                                 # networks' images, first this network's (nn/native/twin.py)

    def __init__(self, in_channels, out_channels, norm_type, first_layer_channels=16, down_blocks=(1, 2, 3, 2),
                 up_blocks=(2, 2, 1, 1), use_memory_saving=True, use_inverse=True, is_separable=False, attention=()):
        """attention: per down block, whether a SelfAttentionBlock follows it (SelfAttentionVnet3D,
        selfattention_vnet3d.py:90-104); empty = the plain V-Net"""
        require_instance_norm(norm_type)
        self.attention = tuple(bool(a) for a in attention) + (False,) * (len(down_blocks) - len(attention))
        self.use_inverse = bool(use_inverse)
        self.use_memory_saving = bool(use_memory_saving)
        if is_separable:
            raise NotImplementedError("separable convolutions are not implemented")
        if first_layer_channels % in_channels:
            raise ValueError("`first_layer_channels` has to be divisible by `in_channels`.")
        if len(down_blocks) != len(up_blocks):
            raise ValueError("Number of `down_blocks` and `up_blocks` has to be equal.")
        c = first_layer_channels
        assert c >= 8 and (c & (c - 1)) == 0, "first_layer_channels must be a power of two >= 8 (16-byte channel slices)"
        use_bias = is_bias_before_norm(norm_type)
        dims = type(self).dims
        L = self.L = len(down_blocks)
        self.c = c
        conv = lambda *a, **k: ConvSpec(*a, dims=dims, **k)
        nodes, extras = [], []

        order = []           # parameter keys in the order the blocks register them

        def add_conv(spec, norm, name, aliases=()):
            nodes.append(Node(spec, norm, "none", name=name, aliases=tuple(aliases)))
            order.append(f"{name}.weight")
            if spec.bias:
                order.append(f"{name}.bias")
            return len(nodes) - 1

        def add_slope(name, size, aliases=()):
            extras.append(Extra(name, size, 0.25, tuple(aliases)))
            order.append(name)
            return name

        enc = lambda i, rest: (f"encoder.{i}.{rest}",)       # `encoder` = [in_ab] + downs (vnet3d.py:88)
        self.n_in = add_conv(conv("conv", in_channels, c, 5, 1, 2, bias=use_bias), True, "in_ab.conv1",
                             enc(0, "conv1"))
        self.s_in = add_slope("in_ab.relu.weight", c, enc(0, "relu.weight"))
        inv = self.use_inverse
        if inv:      # B -> A copies of the non-invertible layers, built with the same widths as the A -> B ones
            # (vnet3d.py:60-62,67-69)
            self.n_in_ba = add_conv(conv("conv", in_channels, c, 5, 1, 2, bias=use_bias), True, "in_ba.conv1")
            self.s_in_ba = add_slope("in_ba.relu.weight", c)

        def couplings(blk, prefix, h, n, alias_prefix=None):
            for j in range(n):
                ent = []
                for fn in ("Fm", "Gm"):
                    base = f"{prefix}.core.sequence.{j}.invertible_block._fn.{fn}"
                    al = (lambda r: (f"{alias_prefix}.core.sequence.{j}.invertible_block._fn.{fn}.{r}",)) \
                        if alias_prefix else (lambda r: ())
                    ent.append(add_conv(conv("conv", h, h, 5, 1, 2, bias=use_bias), True, base + ".0", al("0")))
                    ent.append(add_slope(base + ".2.weight", h, al("2.weight")))
                blk.couplings.append(tuple(ent))

        self.downs = []
        for i, n in enumerate(down_blocks):
            blk = _Block()
            cin = c * 2 ** i
            blk.C, blk.level = 2 * cin, i + 1
            blk.conv = add_conv(conv("conv", cin, 2 * cin, 2, 2, 0, bias=use_bias), True,
                                f"downs.{i}.down_conv_ab.0", enc(i + 1, "down_conv_ab.0"))
            blk.conv_slope = add_slope(f"downs.{i}.down_conv_ab.2.weight", 2 * cin, enc(i + 1, "down_conv_ab.2.weight"))
            if inv:
                blk.conv_ba = add_conv(conv("conv", cin, 2 * cin, 2, 2, 0, bias=use_bias), True,
                                       f"downs.{i}.down_conv_ba.0", enc(i + 1, "down_conv_ba.0"))
                blk.conv_slope_ba = add_slope(f"downs.{i}.down_conv_ba.2.weight", 2 * cin,
                                              enc(i + 1, "down_conv_ba.2.weight"))
            couplings(blk, f"downs.{i}", cin, n, f"encoder.{i + 1}")
            blk.tail_slope = add_slope(f"downs.{i}.relu.weight", 2 * cin, enc(i + 1, "relu.weight"))
            self.downs.append(blk)
        for i, on in enumerate(self.attention):          # registered after all down blocks (selfattention_vnet3d.py:107-108)
            if on:
                ex = attention_extras(f"attn_blocks.{i}", 2 * c * 2 ** i, dims=dims)
                extras.extend(ex)
                order.extend(e.name for e in ex)
        ucf = [2 * 2 ** i for i in reversed(range(L))]
        self.ups = []
        for i, n in enumerate(up_blocks):
            blk = _Block()
            cin, cout = (c * ucf[0], c * ucf[0]) if i == 0 else (c * ucf[i - 1], c * ucf[i])
            blk.C, blk.level, blk.cin = cout, L - 1 - i, cin
            blk.conv = add_conv(conv("convT", cin, cout // 2, 2, 2, 0, 0, bias=use_bias), True,
                                f"ups.{i}.up_conv_ab.0")
            blk.conv_slope = add_slope(f"ups.{i}.up_conv_ab.2.weight", cout // 2)
            if inv:
                blk.conv_ba = add_conv(conv("convT", cin, cout // 2, 2, 2, 0, 0, bias=use_bias), True,
                                       f"ups.{i}.up_conv_ba.0")
                blk.conv_slope_ba = add_slope(f"ups.{i}.up_conv_ba.2.weight", cout // 2)
            couplings(blk, f"ups.{i}", cout // 2, n)
            blk.tail_slope = add_slope(f"ups.{i}.relu.weight", cout)
            self.ups.append(blk)
        self.n_o1 = add_conv(conv("conv", 2 * c, 2 * c, 5, 1, 2, bias=use_bias), True, "out_ab.conv1")
        self.s_o1 = add_slope("out_ab.relu1.weight", 2 * c)
        self.n_o2 = add_conv(conv("conv", 2 * c, out_channels, 1, 1, 0), False, "out_ab.conv2")
        if inv:
            self.n_o1_ba = add_conv(conv("conv", 2 * c, 2 * c, 5, 1, 2, bias=use_bias), True, "out_ba.conv1")
            self.s_o1_ba = add_slope("out_ba.relu1.weight", 2 * c)
            self.n_o2_ba = add_conv(conv("conv", 2 * c, out_channels, 1, 1, 0), False, "out_ba.conv2")
            if in_channels != out_channels:
                raise ValueError("use_inverse needs in_channels == out_channels (both directions share in / out widths)")
        super().__init__(nodes, in_channels, out_channels, out_act="tanh", extras=extras)
        # the reference registers in_ab, in_ba, out_ab, out_ba, downs, ups in that order (vnet3d.py:60-104); inside a block
        # the order above
        rank = {"in_ab": 0, "in_ba": 1, "out_ab": 2, "out_ba": 3, "downs": 4, "attn_blocks": 5, "ups": 6}
        self._param_order = sorted(order, key=lambda k: rank[k.split(".")[0]])       # stable

    def reference_parameter_order(self):
        return list(self._param_order)

    # ---- lowering: every node at the resolution level it runs at -------------------------------------------------
    def _lowered(self, *sizes):
        key = tuple(sizes)
        if key not in self._low_cache:
            assert all(x % (1 << self.L) == 0 for x in key), f"input {key} must be divisible by 2^{self.L}"
            lv = lambda k: tuple(x >> k for x in key)
            lows = [None] * len(self.nodes)
            lows[self.n_in] = lower(self.nodes[self.n_in].spec, *key)
            if self.use_inverse:
                lows[self.n_in_ba] = lower(self.nodes[self.n_in_ba].spec, *key)
                lows[self.n_o1_ba] = lower(self.nodes[self.n_o1_ba].spec, *key)
                lows[self.n_o2_ba] = lower(self.nodes[self.n_o2_ba].spec, *key)
                for blk in self.downs:
                    lows[blk.conv_ba] = lower(self.nodes[blk.conv_ba].spec, *lv(blk.level - 1))
                for blk in self.ups:
                    lows[blk.conv_ba] = lower(self.nodes[blk.conv_ba].spec, *lv(blk.level + 1))
            for blk in self.downs:
                lows[blk.conv] = lower(self.nodes[blk.conv].spec, *lv(blk.level - 1))
                for nf, _, ng, _ in blk.couplings:
                    lows[nf] = lower(self.nodes[nf].spec, *lv(blk.level))
                    lows[ng] = lower(self.nodes[ng].spec, *lv(blk.level))
            for blk in self.ups:
                lows[blk.conv] = lower(self.nodes[blk.conv].spec, *lv(blk.level + 1))
                for nf, _, ng, _ in blk.couplings:
                    lows[nf] = lower(self.nodes[nf].spec, *lv(blk.level))
                    lows[ng] = lower(self.nodes[ng].spec, *lv(blk.level))
            lows[self.n_o1] = lower(self.nodes[self.n_o1].spec, *key)
            lows[self.n_o2] = lower(self.nodes[self.n_o2].spec, *key)
            self._low_cache[key] = lows
        return self._low_cache[key]

    # ---- helpers -------------------------------------------------------------------------------------------------------
    def _new(self, N, sizes, C):
        return torch.empty(N, *sizes, C, dtype=self.ops.act_dtype, device=self.device)

    def _conv(self, s, i, x, in_co=0, stats=True, out=None):
        """raw output of node i (+ mean/rstd of its InstanceNorm) reading channels [in_co, in_co + cin) of x"""
        ops, sp, lw, N = self.ops, self.nodes[i].spec, s.lows[i], s.N
        tw = self._tw
        m = self.master.detach() if tw is None else Twin(self.master.detach(), tw.master.detach())
        bias = m[self.b_off[i]:self.b_off[i] + sp.cout_p]
        fpack = (s.pk["fpack"] if tw is None else Twin(s.pk["fpack"], s.pk_tw["fpack"]))[s.pk["f_off"][i]:]
        y = out if out is not None else self._new(N, lw.out_dims, sp.cout_p)
        if not stats:
            ops.gconv_classes(lw.fwd, x, fpack, bias, y, in_co=in_co)
            return y, None
        slots, offs = 0, []
        for g in lw.fwd:
            offs.append(slots)
            slots += ops.stat_slots(g, N, twin=tw is not None, multi=lw.fwd if len(lw.fwd) > 1 else None)
        part = torch.empty(N * slots * 2 * sp.cout_p, dtype=torch.float32, device=self.device)
        ops.gconv_classes(lw.fwd, x, fpack, bias, y, in_co=in_co, stats=part, stats_slots=slots, stats_slot0s=offs)
        mr = torch.empty(N * 2 * sp.cout_p, dtype=torch.float32, device=self.device)
        ops.inorm_finalize(part, N, slots, sp.cout_p, lw.out_pixels, mr)
        return y, mr

    def _slope(self, name, grad=False):
        t = self.extra(name, grad=grad)
        return t if self._tw is None else Twin(t, self._tw.extra(name, grad=grad))

    def _grad_buf(self):
        return self.master.grad if self._tw is None else Twin(self.master.grad, self._tw.master.grad)

    # ---- forward -----------------------------------------------------------------------------------------------------
    def _couplings_forward(self, s, blk, X):
        """X [.., 2h] -> core(X); saves what the backward pass needs (nothing per coupling with memory saving)"""
        saved = []
        for cp in blk.couplings:
            Y, ya, mra, yb, mrb = self._coupling_fwd_one(s, blk, cp, X)
            saved.append(None if s.recompute else (X, Y, ya, mra, yb, mrb))
            X = Y
        return X, saved

    def _coupling_fwd_one(self, s, blk, cp, X):
        """y1 = x1 + F(x2), y2 = x2 + G(y1); returns Y and the raw conv outputs (+ statistics) of F and G"""
        ops, h = self.ops, blk.C // 2
        nf, sf, ng, sg = cp
        Y = self._new(s.N, X.shape[1:-1], blk.C)
        ya, mra = self._conv(s, nf, X, in_co=h)                             # F reads x2
        ops.pnorm_forward(ya, mra, Y, C=h, slope=self._slope(sf), res=X, res_mode=2, res_co=0, out_co=0)
        yb, mrb = self._conv(s, ng, Y, in_co=0)                             # G reads y1
        ops.pnorm_forward(yb, mrb, Y, C=h, slope=self._slope(sg), res=X, res_mode=2, res_co=h, out_co=h)
        return Y, ya, mra, yb, mrb

    def _coupling_inv_one(self, s, blk, cp, Y):
        """x2 = y2 - G(y1), x1 = y1 - F(x2); returns X and the raw conv outputs (+ statistics) of F and G"""
        ops, h = self.ops, blk.C // 2
        nf, sf, ng, sg = cp
        X = self._new(s.N, Y.shape[1:-1], blk.C)
        yb, mrb = self._conv(s, ng, Y, in_co=0)                             # G reads y1
        ops.pnorm_forward(yb, mrb, X, C=h, slope=self._slope(sg), res=Y, res_mode=3, res_co=h, out_co=h)
        ya, mra = self._conv(s, nf, X, in_co=h)                             # F reads x2
        ops.pnorm_forward(ya, mra, X, C=h, slope=self._slope(sf), res=Y, res_mode=3, res_co=0, out_co=0)
        return X, ya, mra, yb, mrb

    def _couplings_inverse(self, s, blk, Y):
        """Y [.., 2h] -> core.inverse(Y): couplings in reversed order, x2 = y2 - G(y1), x1 = y1 - F(x2)"""
        saved = []
        for cp in reversed(blk.couplings):
            X, ya, mra, yb, mrb = self._coupling_inv_one(s, blk, cp, Y)
            saved.append(None if s.recompute else (X, Y, ya, mra, yb, mrb))
            Y = X
        return Y, saved

    def forward(self, x, inverse=False):
        """forward(x) = A -> B; forward(x, inverse=True) = B -> A through the *_ba layers and the inverted cores
        (vnet3d.py:107-150); needs use_inverse=True"""
        if inverse and not self.use_inverse:
            raise ValueError("Trying to perform inverse forward while `use_inverse` flag is turned off.")
        self._next_inverse = bool(inverse)
        try:
            return super().forward(x)
        finally:
            self._next_inverse = False

    __call__ = forward

    def _forward(self, x, save, stop=None, tw=None):
        """tw: a second Vnet3D of identical architecture (TwinNet) — x is then ((this network's batches), (tw's batches)) and
        the pass runs as ONE batch, this network's images first"""
        assert stop is None, "feature taps are not implemented for Vnet3D"
        self._tw = tw
        try:
            return self._forward_impl(x, save)
        finally:
            self._tw = None

    def _forward_impl(self, x, save):
        ops, c, L = self.ops, self.c, self.L
        tw = self._tw
        xs = (tuple(x[0]) + tuple(x[1])) if tw is not None else (x,)
        N, sizes = sum(t.shape[0] for t in xs), tuple(xs[0].shape[2:])
        s = _Saved()
        s.tw = tw
        assert tw is None or not getattr(self, "_next_inverse", False), "twin passes run A -> B"
        inv = s.inverse = bool(getattr(self, "_next_inverse", False))
        s.recompute = bool(save and self.use_memory_saving)
        n_in, s_in = (self.n_in_ba, self.s_in_ba) if inv else (self.n_in, self.s_in)
        n_o1, s_o1, n_o2 = (self.n_o1_ba, self.s_o1_ba, self.n_o2_ba) if inv else (self.n_o1, self.s_o1, self.n_o2)
        bconv = (lambda b: (b.conv_ba, b.conv_slope_ba)) if inv else (lambda b: (b.conv, b.conv_slope))
        core = self._couplings_inverse if inv else self._couplings_forward
        s.x_img, s.N, s.sizes = x, N, sizes
        s.lows, s.pk = self._lowered(*sizes), self._get_packs(*sizes)
        s.pk_tw = tw._get_packs(*sizes) if tw is not None else None
        lv = lambda k: tuple(v >> k for v in sizes)
        a0 = self._new(N, sizes, self.nodes[0].spec.cin_p)
        n0 = 0
        for xh in xs:
            ops.image_to_act(xh, a0[n0:n0 + xh.shape[0]])
            n0 += xh.shape[0]
        s.xs = xs
        s.a0 = a0
        # InputBlock
        s.y_in, s.mr_in = self._conv(s, n_in, a0)
        out1 = self._new(N, sizes, c)
        ops.pnorm_forward(s.y_in, s.mr_in, out1, C=c, slope=self._slope(s_in), res=a0, res_mode=1,
                          res_mod=self.in_channels)
        s.out1 = out1
        # DownBlocks
        s.down = []
        cur = out1
        for blk in self.downs:
            rec = _Saved()
            rec.x_in = cur
            nconv, sconv = bconv(blk)
            rec.y, rec.mr = self._conv(s, nconv, cur)
            rec.D0 = self._new(N, lv(blk.level), blk.C)
            ops.pnorm_forward(rec.y, rec.mr, rec.D0, C=blk.C, slope=self._slope(sconv))
            rec.Xn, rec.coup = core(s, blk, rec.D0)
            rec.out = self._new(N, lv(blk.level), blk.C)
            ops.pnorm_forward(rec.Xn, None, rec.out, C=blk.C, slope=self._slope(blk.tail_slope), res=rec.D0, res_mode=1)
            rec.attn = None
            if self.attention[len(s.down)]:      # the attended map feeds the next down block AND the skip connection
                rec.out, rec.attn = ops.attn_forward(rec.out, self.attn_tensors(f"attn_blocks.{len(s.down)}"), need_backward=bool(save))
            s.down.append(rec)
            cur = rec.out
        # UpBlocks
        s.up = []
        skips = [s.down[L - 2 - i].out if i < L - 1 else out1 for i in range(L)]
        for i, blk in enumerate(self.ups):
            rec = _Saved()
            rec.x_in = cur
            h = blk.C // 2
            nconv, sconv = bconv(blk)
            rec.y, rec.mr = self._conv(s, nconv, cur)
            rec.D0 = self._new(N, lv(blk.level), blk.C)                      # xcat = [up | skip]
            ops.pnorm_forward(rec.y, rec.mr, rec.D0, C=h, slope=self._slope(sconv), out_co=0)
            ops.add_views(rec.D0, skips[i], h, dst_co=h, src_co=0, accumulate=False)
            rec.Xn, rec.coup = core(s, blk, rec.D0)
            rec.out = self._new(N, lv(blk.level), blk.C)
            ops.pnorm_forward(rec.Xn, None, rec.out, C=blk.C, slope=self._slope(blk.tail_slope), res=rec.D0, res_mode=1)
            s.up.append(rec)
            cur = rec.out
        # OutBlock
        s.o_in = cur
        s.y_o1, s.mr_o1 = self._conv(s, n_o1, cur)
        s.t = self._new(N, sizes, 2 * c)
        ops.pnorm_forward(s.y_o1, s.mr_o1, s.t, C=2 * c, slope=self._slope(s_o1))
        s.z, _ = self._conv(s, n_o2, s.t, stats=False)
        out = torch.empty(N, self.out_channels, *sizes, dtype=torch.float32, device=self.device)
        ops.act_to_image(s.z, out, act="tanh")
        s.out_img = out
        return out, (s if save else None)

    # ---- backward ----------------------------------------------------------------------------------------------------
    def _wgrad(self, s, i, x_in, dy, x_co=0):
        """parameter gradients of node i: dense side / gathered side by layer kind; x_in may be a channel slice"""
        ops, sp, lw, grad = self.ops, self.nodes[i].spec, s.lows[i], self._grad_buf()
        dw = grad[self.w_off[i]:self.w_off[i] + sp.master_numel]
        self._wgrad_written(i, self._tw)    # (every weight-gradient launch is noted, NativeNet._wgrad_written; no fresh= hint here)
        if sp.kind == "conv":
            ops.wgrad(lw.wgrad, dy, x_in, dw, g_co=x_co)
        else:
            ops.wgrad(lw.wgrad, x_in, dy, dw, a_co=x_co)
        self.grad_dirty = True
        if self._tw is not None:
            self._tw.grad_dirty = True

    def _bias_slice(self, i, want_w):
        sp = self.nodes[i].spec
        return self._grad_buf()[self.b_off[i]:self.b_off[i] + sp.cout_p] if (want_w and sp.bias) else None

    def _dgrad(self, s, i, dy, out=None, out_co=0, accumulate=False):
        sp, lw = self.nodes[i].spec, s.lows[i]
        gx = out if out is not None else self._new(s.N, lw.in_dims, sp.cin_p)
        dpack = (s.pk["dpack"] if self._tw is None else Twin(s.pk["dpack"], s.pk_tw["dpack"]))[s.pk["d_off"][i]:]
        self.ops.gconv_classes(lw.dgrad, dy, dpack, None, gx, out_co=out_co, accumulate=accumulate)
        return gx

    def _block_backward(self, s, blk, rec, g, g2, g2_co, want_w):
        """gradient w.r.t. the block output (g [+ g2 slice]) -> total gradient w.r.t. D0 (= down, or xcat)"""
        ops, h = self.ops, blk.C // 2
        dsl = (lambda name: self._slope(name, grad=True)) if want_w else (lambda name: None)
        gu = torch.empty_like(rec.out)
        # G (= gu, the residual's gradient: second output of the same pass, not a copy launch behind it) becomes the gradient
        # w.r.t. core's input, in place
        G = torch.empty_like(rec.out)
        ops.pnorm_backward(g, rec.Xn, None, gu, C=blk.C, slope=self._slope(blk.tail_slope), dslope=dsl(blk.tail_slope),
                           g2=g2, g2_co=g2_co, res=rec.D0, res_mode=1, gres=G)
        cur = rec.Xn                                     # memory saving: the core's output, walked back coupling by coupling
        if s.inverse:
            # rec.coup was recorded over reversed(blk.couplings): walk it back. Per coupling x1 = y1 - F(x2), x2 = y2 - G(y1):
            # G holds the gradient w.r.t. [x1 | x2] and ends as the gradient w.r.t. [y1 | y2]
            for cp, kept in zip(blk.couplings, reversed(rec.coup)):
                nf, sf, ng, sg = cp
                if kept is None:                         # rebuild [y1 | y2] from [x1 | x2] with the forward equations
                    X = cur
                    Y, ya, mra, yb, mrb = self._coupling_fwd_one(s, blk, cp, X)
                    cur = Y
                else:
                    X, Y, ya, mra, yb, mrb = kept
                dya = torch.empty_like(ya)
                ops.pnorm_backward(G, ya, mra, dya, C=h, slope=self._slope(sf), dslope=dsl(sf), g_co=0, res_mode=3,
                                   bias_grad=self._bias_slice(nf, want_w))
                if want_w:
                    self._wgrad(s, nf, X, dya, x_co=h)
                self._dgrad(s, nf, dya, out=G, out_co=h, accumulate=True)      # x2 also fed F
                dyb = torch.empty_like(yb)
                ops.pnorm_backward(G, yb, mrb, dyb, C=h, slope=self._slope(sg), dslope=dsl(sg), g_co=h, res_mode=3,
                                   bias_grad=self._bias_slice(ng, want_w))
                if want_w:
                    self._wgrad(s, ng, Y, dyb, x_co=0)
                self._dgrad(s, ng, dyb, out=G, out_co=0, accumulate=True)      # y1 also fed G
            ops.add_views(G, gu, blk.C, accumulate=True)     # out = core.inverse(D0) + D0
            return G
        for cp, kept in zip(reversed(blk.couplings), reversed(rec.coup)):
            nf, sf, ng, sg = cp
            if kept is None:                             # rebuild [x1 | x2] from [y1 | y2] with the inverse equations
                Y = cur
                X, ya, mra, yb, mrb = self._coupling_inv_one(s, blk, cp, Y)
                cur = X
            else:
                X, Y, ya, mra, yb, mrb = kept
            # y2 = x2 + G(y1): gradient of G's conv output from the y2 half, its data gradient joins the y1 half
            dyb = torch.empty_like(yb)
            ops.pnorm_backward(G, yb, mrb, dyb, C=h, slope=self._slope(sg), dslope=dsl(sg), g_co=h,
                               bias_grad=self._bias_slice(ng, want_w))
            if want_w:
                self._wgrad(s, ng, Y, dyb, x_co=0)
            self._dgrad(s, ng, dyb, out=G, out_co=0, accumulate=True)
            # y1 = x1 + F(x2)
            dya = torch.empty_like(ya)
            ops.pnorm_backward(G, ya, mra, dya, C=h, slope=self._slope(sf), dslope=dsl(sf), g_co=0,
                               bias_grad=self._bias_slice(nf, want_w))
            if want_w:
                self._wgrad(s, nf, X, dya, x_co=h)
            self._dgrad(s, nf, dya, out=G, out_co=h, accumulate=True)
        ops.add_views(G, gu, blk.C, accumulate=True)     # out = core(D0) + D0
        return G

    def _backward(self, s, g_img, need_input_grad, want_w, start=None, inj_x=None, inj_y=None, tw=None):
        """tw: the pass recorded by _forward(..., tw) — g_img holds one gradient per input batch (None: that output took no part
        in the loss) and the input gradients come back as a tuple"""
        assert start is None and not inj_x and not inj_y, "feature taps are not implemented for Vnet3D"
        assert tw is s.tw, "twin pass: backward with the partner the forward pass ran with"
        self._tw = tw
        try:
            return self._backward_impl(s, g_img, need_input_grad, want_w)
        finally:
            self._tw = None

    def _backward_impl(self, s, g_img, need_input_grad, want_w):
        ops, c, L, N = self.ops, self.c, self.L, s.N
        for net in (self, self._tw):
            if net is not None and net.master.grad is None:
                net.master.grad = torch.zeros(net.numel, dtype=torch.float32, device=self.device)
        grad = self._grad_buf()
        dsl = (lambda name: self._slope(name, grad=True)) if want_w else (lambda name: None)
        inv = s.inverse
        n_in, s_in = (self.n_in_ba, self.s_in_ba) if inv else (self.n_in, self.s_in)
        n_o1, s_o1, n_o2 = (self.n_o1_ba, self.s_o1_ba, self.n_o2_ba) if inv else (self.n_o1, self.s_o1, self.n_o2)
        bconv = (lambda b: (b.conv_ba, b.conv_slope_ba)) if inv else (lambda b: (b.conv, b.conv_slope))
        # OutBlock
        gz = torch.empty_like(s.z)
        if self._tw is None:
            ops.act_to_image_backward(g_img.contiguous().float(), s.out_img, gz, act="tanh")
        else:
            n0 = 0
            for gh, xh in zip(g_img, s.xs):      # one gradient per input batch
                n1 = n0 + xh.shape[0]
                if gh is None:
                    gz[n0:n1].zero_()
                else:
                    ops.act_to_image_backward(gh.contiguous().float(), s.out_img[n0:n1], gz[n0:n1], act="tanh")
                n0 = n1
        if want_w:
            self._wgrad(s, n_o2, s.t, gz)
            sp = self.nodes[n_o2].spec
            ops.bias_grad(gz, sp.cout_p, grad[self.b_off[n_o2]:self.b_off[n_o2] + sp.cout_p])
        gt = self._dgrad(s, n_o2, gz)
        dy = torch.empty_like(s.y_o1)
        ops.pnorm_backward(gt, s.y_o1, s.mr_o1, dy, C=2 * c, slope=self._slope(s_o1), dslope=dsl(s_o1),
                           bias_grad=self._bias_slice(n_o1, want_w))
        if want_w:
            self._wgrad(s, n_o1, s.o_in, dy)
        g_cur = self._dgrad(s, n_o1, dy)            # gradient w.r.t. the last UpBlock's output
        # UpBlocks, last first; skip gradients are slices of the blocks' xcat gradients
        skip_grad = {}                                   # forward skip index -> (tensor, channel offset)
        for i in range(L - 1, -1, -1):
            blk, rec = self.ups[i], s.up[i]
            h = blk.C // 2
            G = self._block_backward(s, blk, rec, g_cur, None, 0, want_w)
            skip_grad[i] = (G, h)
            nconv, sconv = bconv(blk)
            dy = torch.empty_like(rec.y)
            ops.pnorm_backward(G, rec.y, rec.mr, dy, C=h, slope=self._slope(sconv),
                               dslope=dsl(sconv), g_co=0, bias_grad=self._bias_slice(nconv, want_w))
            if want_w:
                self._wgrad(s, nconv, rec.x_in, dy)
            g_cur = self._dgrad(s, nconv, dy)         # w.r.t. the previous UpBlock's output / the last DownBlock's
        # DownBlocks, deepest first: output k also fed UpBlock L-1-k as its skip (k < L-1)
        for k in range(L - 1, -1, -1):
            blk, rec = self.downs[k], s.down[k]
            g2, g2_co = skip_grad[L - 2 - k] if k < L - 1 else (None, 0)
            if rec.attn is not None:
                # gradient w.r.t. the attended map = next stage's data gradient + the skip's slice; the block's backward
                # returns the gradient w.r.t. the down block's own output and adds the block's parameter gradients
                tot = g_cur
                if g2 is not None:
                    tot = g_cur.clone()
                    ops.add_views(tot, g2, blk.C, dst_co=0, src_co=g2_co, accumulate=True)
                prefix = f"attn_blocks.{k}"
                g_cur = ops.attn_backward(rec.attn, tot, self.attn_tensors(prefix),
                                          self.attn_tensors(prefix, grad=True) if want_w else None)
                g2, g2_co = None, 0
                if want_w:
                    self.grad_dirty = True
            G = self._block_backward(s, blk, rec, g_cur, g2, g2_co, want_w)
            nconv, sconv = bconv(blk)
            dy = torch.empty_like(rec.y)
            ops.pnorm_backward(G, rec.y, rec.mr, dy, C=blk.C, slope=self._slope(sconv),
                               dslope=dsl(sconv), bias_grad=self._bias_slice(nconv, want_w))
            if want_w:
                self._wgrad(s, nconv, rec.x_in, dy)
            g_cur = self._dgrad(s, nconv, dy)
        # InputBlock: out1 also was the skip of the last UpBlock
        g2, g2_co = skip_grad[L - 1]
        dy = torch.empty_like(s.y_in)
        gres = torch.empty_like(s.y_in) if need_input_grad else None
        ops.pnorm_backward(g_cur, s.y_in, s.mr_in, dy, C=c, slope=self._slope(s_in), dslope=dsl(s_in),
                           g2=g2, g2_co=g2_co, res=s.a0, res_mode=1, res_mod=self.in_channels, gres=gres,
                           bias_grad=self._bias_slice(n_in, want_w))
        if want_w:
            self._wgrad(s, n_in, s.a0, dy)
        if not need_input_grad:
            return None
        gx = self._dgrad(s, n_in, dy)
        if self._tw is None:
            g_in = torch.empty_like(s.x_img)
            ops.image_to_act_backward(gx, g_in, fold=0)
            ops.repeat_backward(gres, g_in, c)               # adjoint of x.repeat (vnet3d.py:165-166)
            return g_in
        g_ins, n0 = [], 0
        for xh in s.xs:
            n1 = n0 + xh.shape[0]
            g_in = torch.empty_like(xh)
            ops.image_to_act_backward(gx[n0:n1], g_in, fold=0)
            ops.repeat_backward(gres[n0:n1], g_in, c)
            g_ins.append(g_in)
            n0 = n1
        return tuple(g_ins)


@dataclass
class Vnet2DConfig(configs.base.BaseGeneratorConfig):
    """Partially-invertible V-Net generator, 2-D (ganslate/nn/generators/vnet/vnet2d.py:14-19: the block counts are not
    configurable there; the defaults build the RevGAN inverse path, which raises here — set both to False)"""
    use_memory_saving: bool = True
    use_inverse: bool = True
    first_layer_channels: int = 16


class Vnet2D(Vnet3D):
    """ganslate/nn/generators/vnet/vnet2d.py:22-248: Vnet3D's constructor, channel plan and state_dict names on 2-D
    layers — the executor above treats an image as the depth-1 volume, so only the lowering dimension changes"""
    dims = 2

    def __init__(self, in_channels, out_channels, norm_type, first_layer_channels=16, down_blocks=(1, 2, 3, 2),
                 up_blocks=(2, 2, 1, 1), use_memory_saving=True, use_inverse=True):
        super().__init__(in_channels, out_channels, norm_type, first_layer_channels, down_blocks, up_blocks,
                         use_memory_saving, use_inverse, False)
