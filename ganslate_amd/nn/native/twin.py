"""Twin execution: two networks of IDENTICAL architecture run lock-step as one batch.

A CycleGAN step runs every generator layer twice per phase on independent data — G_AB(real_A) next to G_BA(real_B), then
G_AB(fake_A) next to G_BA(fake_B), and the same pairs in the backward pass (ganslate/nn/gans/unpaired/cyclegan.py:126-152);
likewise D_B next to D_A (cyclegan.py:154-189). With per-sample InstanceNorm nothing couples the images of a batch, so the
two passes are ONE pass over a batch of 2N images in which images [0, N) use the first network's weights and images
[N, 2N) the second's: every weight-free kernel (norms, activations, boundary transforms) simply sees a larger batch, and
the conv / weight-gradient kernels pick the weight set from the image index (`gs_twin` in include/ganslate_hip.h). That
halves the launch count of the generator phases and gives the residual-conv kernels 512 tiles — two per workgroup — so the
prologue of the second tile hides under the first one's K loop (csrc/hconvw.hip).

`Twin(a, b)` pairs the same tensor of the two networks (weight pack, bias vector, gradient slice); it slices like a
tensor. Ops that have no native twin form run the two halves one after the other (`TwinSplit`), which is what two
separate passes would do: same arithmetic, same summation order.
"""
import torch


class Twin:
    """the same tensor of two networks of identical architecture"""
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        self.a, self.b = a, b

    def __getitem__(self, sl):
        return Twin(self.a[sl], self.b[sl])

    def half(self, h):
        return self.b if h else self.a

    def delta(self):
        """byte distance from the first network's tensor to the second's (identical layouts: one number per buffer)"""
        return self.b.data_ptr() - self.a.data_ptr()


def pick(t, h):
    """half h of a per-network argument (Twin) — anything else is shared by both halves"""
    return t.half(h) if isinstance(t, Twin) else t


def is_twin(*args):
    return any(isinstance(a, Twin) for a in args)


def bhalf(t, h):
    """images [h*N, (h+1)*N) of a batched tensor [2N, ...] (None stays None)"""
    if t is None:
        return None
    n = t.shape[0] // 2
    return t[h * n:(h + 1) * n]


def fhalf(t, h):
    """half h of a flat per-image buffer [2N * k]"""
    if t is None:
        return None
    n = t.numel() // 2
    return t[h * n:(h + 1) * n]


class TwinSplit:
    """Mixin for an ops backend: the weight-carrying ops on a batch of 2N images with per-network (Twin) weights, as two
    calls on the halves. A backend overrides / short-cuts the ops it has a native twin form for."""

    @staticmethod
    def _fuse_half(fuse, h, N, C_):
        if fuse is None:
            return None
        f = dict(fuse)
        f["y"], f["g2"] = bhalf(fuse["y"], h), bhalf(fuse.get("g2"), h)
        f["mean_rstd"] = fhalf(fuse["mean_rstd"], h)
        # scratch = [2N][slots][3][C] partial sums, then [2N][3][C] totals: the launch only writes its images' partial rows
        part = fuse["partial"]
        slots = part.numel() // (2 * N * 3 * C_) - 1
        f["partial"] = part[h * N * slots * 3 * C_:]
        return f

    def run_halves(self, f):
        """f(0), f(1): the two networks' halves of one op, one after the other. (Measured: the second half on a companion
        stream — a parallel branch of the captured step — is no faster, 442.5 against 445 img/s: these launches fill the
        chip; and a second companion next to the discriminators' side stream crashes hipStreamEndCapture on ROCm 7.2.)"""
        f(0)
        f(1)

    def twin_gconv(self, fn, x, wpack, bias, out, *, stats=None, fuse=None, C_=None, **kw):
        """fn = self.gconv or self.gconv_classes bound to its class argument"""
        N = x.shape[0] // 2
        self.run_halves(lambda h: fn(bhalf(x, h), pick(wpack, h), pick(bias, h), bhalf(out, h), stats=fhalf(stats, h),
                                     fuse=self._fuse_half(fuse, h, N, C_), **kw))

    def twin_wgrad(self, w, a, g, dw, *, pair=None, **kw):
        self.run_halves(lambda h: self.wgrad(w, bhalf(a, h), bhalf(g, h), pick(dw, h),
                                             pair=None if pair is None else (bhalf(pair[0], h), bhalf(pair[1], h)), **kw))

    def twin_bias_grad(self, dy, C_, db, **kw):
        self.run_halves(lambda h: self.bias_grad(bhalf(dy, h), C_, pick(db, h), **kw))

    def twin_shiftadd_to_image(self, z, bias, img, k, act="none"):
        self.run_halves(lambda h: self.shiftadd_to_image(bhalf(z, h), pick(bias, h), bhalf(img, h), k, act=act))

    # InstanceNorm + PReLU of the V-Nets: the slopes (and their gradients, and the bias gradient of the conv in front) are
    # per network — the two halves of the batch as two launches, every batched operand sliced alike
    def twin_pnorm_forward(self, y, mean_rstd, out, *, C, slope=None, res=None, **kw):
        self.run_halves(lambda h: self.pnorm_forward(bhalf(y, h), fhalf(mean_rstd, h), bhalf(out, h), C=C, slope=pick(slope, h),
                                                     res=bhalf(res, h), **kw))

    def twin_pnorm_backward(self, g, y, mean_rstd, dy, *, C, slope=None, dslope=None, g2=None, res=None, gres=None,
                            bias_grad=None, **kw):
        self.run_halves(lambda h: self.pnorm_backward(bhalf(g, h), bhalf(y, h), fhalf(mean_rstd, h), bhalf(dy, h), C=C,
                                                      slope=pick(slope, h), dslope=pick(dslope, h), g2=bhalf(g2, h),
                                                      res=bhalf(res, h), gres=bhalf(gres, h), bias_grad=pick(bias_grad, h),
                                                      **kw))


class TwinNet:
    """Two `NativeNet`s of identical architecture behind one call: `(ya, yb) = twin(xa, xb)` = `(a(xa), b(xb))`, recorded
    as ONE autograd node whose backward pass runs both networks' gradients as one batch."""

    def __init__(self, a, b):
        from .net import NativeNet
        assert isinstance(a, NativeNet) and isinstance(b, NativeNet) and a is not b
        if not self.compatible(a, b):
            raise ValueError("TwinNet: the two networks must have the same layer list")
        self.a, self.b = a, b
        b._twin_lead = a          # weight gradients held back for a merged launch sit in a's table (flush_deferred_wgrads)

    @staticmethod
    def compatible(a, b):
        from .net import NativeNet
        if not (isinstance(a, NativeNet) and isinstance(b, NativeNet)) or a is b:
            return False
        if type(a) is not type(b) or a.numel != b.numel or len(a.nodes) != len(b.nodes):
            return False
        if a.extras or b.extras:      # extra parameter vectors: only executors that run them per network (Vnet3D's PReLU slopes)
            if not getattr(a, "twin_extras_ok", False) or [(e.name, e.size) for e in a.extras] != [(e.name, e.size) for e in b.extras]:
                return False
            if getattr(a, "use_inverse", False) or getattr(b, "use_inverse", False) or any(getattr(a, "attention", ())) \
                    or any(getattr(b, "attention", ())):
                return False
        if a.out_act != b.out_act or a.in_channels != b.in_channels or a.out_channels != b.out_channels:
            return False
        for na, nb in zip(a.nodes, b.nodes):
            if (na.spec != nb.spec or na.norm != nb.norm or na.act != nb.act or na.slope != nb.slope or na.res != nb.res
                    or na.attn or nb.attn):
                return False
        return True

    def __call__(self, xa, xb):
        """xa / xb: an image batch, or a tuple of batches that follow each other in the network's batch (D(real) and
        D(fake) as one pass). Returns the outputs in the same structure."""
        a, b = self.a, self.b
        pa = tuple(xa) if isinstance(xa, (tuple, list)) else (xa,)
        pb = tuple(xb) if isinstance(xb, (tuple, list)) else (xb,)
        pa = tuple(t.contiguous().float() for t in pa)
        pb = tuple(t.contiguous().float() for t in pb)
        assert sum(t.shape[0] for t in pa) == sum(t.shape[0] for t in pb) and \
            all(t.shape[1:] == pa[0].shape[1:] for t in pa + pb), "TwinNet: both networks take batches of one shape"
        record = torch.is_grad_enabled() and (a.requires_grad or b.requires_grad or any(t.requires_grad for t in pa + pb))
        if record and a.requires_grad != b.requires_grad:
            raise RuntimeError("TwinNet: both networks must be trainable or both frozen in one pass")
        if record:
            outs = _TwinFn.apply(a._token, self, len(pa), *pa, *pb)
        else:
            out, _ = a._forward((tuple(t.detach() for t in pa), tuple(t.detach() for t in pb)), save=False, tw=b)
            outs = _split_parts(out, pa + pb)
        oa, ob = outs[:len(pa)], outs[len(pa):]
        return (oa if isinstance(xa, (tuple, list)) else oa[0]), (ob if isinstance(xb, (tuple, list)) else ob[0])


def _split_parts(out, parts):
    outs, n0 = [], 0
    for t in parts:
        outs.append(out[n0:n0 + t.shape[0]])
        n0 += t.shape[0]
    return tuple(outs)


class _TwinFn(torch.autograd.Function):
    """both networks of a TwinNet as one autograd node (one input and one output per batch part)"""

    @staticmethod
    def forward(ctx, token, twin, na, *parts):
        a, b = twin.a, twin.b
        parts = tuple(t.detach() for t in parts)
        out, saved = a._forward((parts[:na], parts[na:]), save=True, tw=b)
        ctx.twin, ctx.saved = twin, saved
        ctx.need = tuple(ctx.needs_input_grad[3:])
        ctx.want_w = a.requires_grad
        if ctx.want_w:
            a._fw_pending += 1
            b._fw_pending += 1
        ctx.set_materialize_grads(False)
        return _split_parts(out, parts)

    @staticmethod
    def backward(ctx, *grads):
        a, b = ctx.twin.a, ctx.twin.b
        if ctx.want_w:
            a._fw_pending -= 1
            b._fw_pending -= 1
            a._order_backward_begin()
            b._order_backward_begin()
        for g in grads:
            if g is not None and g.is_cuda:
                g.record_stream(torch.cuda.current_stream())
        gx = a._backward(ctx.saved, grads, any(ctx.need), ctx.want_w, tw=b)
        if ctx.want_w:
            a._order_backward_end()
            b._order_backward_end()
        ctx.saved = None
        if gx is None:
            return (None,) * (3 + len(grads))
        return (None, None, None) + tuple(g if need else None for g, need in zip(gx, ctx.need))
