"""ORACLE (test infrastructure only — never imported by the product path).

Op-level CPU restatement of every entry point of include/ganslate_hip.h, in plain torch fp32 on the CPU, with
the same tensor conventions as ganslate_amd.hip.ops.HipOps (NHWC bf16 activations, OTI fp32 master weights,
[rows][Kp] bf16 packs). It serves two purposes:
  * `-m gpu` tests compare each HIP kernel with the function of the same name here on identical inputs;
  * `-m "not gpu"` tests run the network executor (ganslate_amd/nn/native) on this backend and compare it with
    torch.nn.functional / autograd, which pins the host-side lowering (taps, packs, folds) without a GPU.

The arithmetic restated is torch's own: nn.Conv2d / nn.ConvTranspose2d (resnet2d.py:25,35,52-57,65,80-87;
patchgan2d.py:29,36-62), nn.InstanceNorm2d(eps=1e-5, affine=False) (nn/utils.py:53-59), nn.ReflectionPad2d
(resnet2d.py:24), nn.MSELoss / nn.L1Loss (adversarial_loss.py:28-29; cyclegan_losses.py:64), SSIMLoss
(nn/losses/utils/ssim.py:65-99), torch.optim.Adam (cyclegan.py:81-82).
Pinned against torch.nn.functional in tests/test_lowering_cpu.py and, through the network-level oracle
(oracle/torch_ref.py), against golden vectors generated from the imported reference (tests/golden/).
"""
import functools
import math

import torch
import torch.nn.functional as F

from ganslate_amd.nn.native.twin import TwinSplit, is_twin      # the oracle may import the product, never the reverse


def _border(idx, n, mode):
    """index tensor -> (clamped index, validity mask)"""
    if mode == "zero":
        ok = (idx >= 0) & (idx < n)
        return idx.clamp(0, n - 1), ok
    if mode == "reflect":
        i = idx.abs()
        i = torch.where(i >= n, 2 * n - 2 - i, i)
        return i, torch.ones_like(idx, dtype=torch.bool)
    return idx.clamp(0, n - 1), torch.ones_like(idx, dtype=torch.bool)


def _act(v, act, slope):
    if act == "relu":
        return F.relu(v)
    if act == "lrelu":
        return F.leaky_relu(v, slope)
    if act == "tanh":
        return torch.tanh(v)
    return v


def _act_grad_from_out(o, act, slope):
    if act == "relu":
        return (o > 0).float()
    if act == "lrelu":
        return torch.where(o > 0, torch.ones_like(o), torch.full_like(o, slope))
    if act == "tanh":
        return 1 - o * o
    return torch.ones_like(o)


def _v5(t):
    """channels-last activation viewed as [N, D, H, W, C] (a 2-D tensor is the depth-1 volume)"""
    return t if t.dim() == 5 else t.unsqueeze(1)


def _fold(gpad, dims, fold, mode="reflect"):
    """adjoint of ReflectionPad2d / ReplicationPad3d(fold) (resnet2d.py:24, resnet3d.py:24) on a channels-last
    tensor padded by `fold` along every spatial axis; dims = unpadded spatial extents"""
    if fold == 0:
        return gpad
    out = gpad
    for ax, n in enumerate(dims, start=1):
        src, _ = _border(torch.arange(n + 2 * fold) - fold, n, mode)
        shape = list(out.shape)
        shape[ax] = n
        nxt = torch.zeros(shape, dtype=out.dtype)
        nxt.index_add_(ax, src, out)
        out = nxt
    return out


class RefOps(TwinSplit):
    """Same method names and argument conventions as HipOps, CPU tensors. Twin arguments (two networks over one batch,
    ganslate_amd/nn/native/twin.py) always run as the two halves."""
    name = "oracle"
    device = torch.device("cpu")

    def __init__(self, act_dtype=torch.bfloat16):
        # bf16 emulates the HIP storage precision; fp32 isolates the executor/lowering logic from rounding
        self.act_dtype = act_dtype
        self.ring_min_blocks = 192     # the library's grid-size rule; tests lower it to walk the ring form at small batch

    def tile_m(self, g, N=1):
        return 1 << 30  # one statistics slot per class

    def stat_slots(self, g, N=1, twin=False, multi=None):
        return 1

    def fused_norm_plan(self, g, N, C_, force=False, twin=False):
        if g.so != 1 or g.si not in (1, 2) or g.Co <= 64 or g.Co != C_:
            return None
        return 1, torch.zeros(N * 2 * 3 * C_, dtype=torch.float32)

    def ring_apply_plan(self, g, N, C_, twin=False):
        """(the in-launch norm backward of gs_gconv_ring_apply is an execution form of the HIP library: the oracle backend
        always runs the two steps)"""
        return None

    def fused_ring_plan(self, g, N, C_, twin=False):
        """the layers the library's gs_gconv_ring_slots accepts (hconvw.hip hconvw_ring_eligible), restated"""
        if g is None or g.Co != C_ or g.T != 9 or g.Ci % 64 or g.Co % 128 or g.Ho % 16 or g.Wo % 16:
            return None
        if g.Ho < 32 or g.Wo < 32 or N * (g.Ho // 16) * (g.Wo // 16) * (g.Co // 128) < self.ring_min_blocks:
            return None
        return 1, torch.zeros(N * 2 * 3 * C_, dtype=torch.float32)

    # ---- convolution family ---------------------------------------------------------------------------
    def fused_multi_plan(self, classes, N, C_, twin=False):
        """the layers gs_gconv_multi_fused_slots accepts (hconvt.hip gs_hconvt_pattern), restated: the four parity classes of
        a 2-D stride-2 layer, 64-multiple channels, class grid a multiple of 16"""
        g = classes[0]
        if len(classes) != 4 or g.Co != C_ or g.so != 2 or g.si != 1 or g.Ci % 64 or g.Co % 64 or g.Dc != 1:
            return None
        if g.Hc % 16 or g.Wc % 16 or g.Ho != 2 * g.Hc or g.Wo != 2 * g.Wc or g.border != "zero":
            return None
        if sorted(c.T for c in classes) not in ([1, 2, 2, 4], [4, 4, 4, 4]):
            return None
        return 1, torch.zeros(N * 2 * 3 * C_, dtype=torch.float32)

    def gconv_classes(self, classes, x, wpack, bias, out, *, in_co=0, out_co=0, act="none", slope=0.2, stats=None,
                      stats_slots=0, stats_slot0s=None, accumulate=False, fuse=None):
        """the output-parity classes of one layer, one after the other (what gs_gconv_forward_multi merges); with `fuse`
        (gs_gconv_forward_multi_fused) followed by the consumer's norm-backward sums over the finished gradient"""
        if is_twin(wpack, bias):
            return self.twin_gconv(functools.partial(self.gconv_classes, classes), x, wpack, bias, out, in_co=in_co,
                                   out_co=out_co, act=act, slope=slope, stats=stats, stats_slots=stats_slots,
                                   stats_slot0s=stats_slot0s, accumulate=accumulate, fuse=fuse, C_=classes[0].Co)
        for i, g in enumerate(classes):
            self.gconv(g, x, wpack, bias, out, in_co=in_co, out_co=out_co, act=act, slope=slope, stats=stats,
                       stats_slots=stats_slots, stats_slot0=(stats_slot0s[i] if stats_slot0s else 0),
                       accumulate=accumulate)
        if fuse is not None:
            N, y, Cc = x.shape[0], fuse["y"], classes[0].Co
            gf = out.float()
            if fuse.get("g2") is not None:
                gf = gf + fuse["g2"].float()
            bc = (N,) + (1,) * (y.dim() - 2) + (Cc,)
            mr = fuse["mean_rstd"].view(N, 2, Cc)
            yh = (y.float() - mr[:, 0].reshape(bc)) * mr[:, 1].reshape(bc)
            gh = gf * _act_grad_from_out(yh, fuse["act"], fuse.get("slope", 0.2))
            part = fuse["partial"][:N * 3 * Cc].view(N, 1, 3, Cc)
            part[:, 0, 0] = gh.reshape(N, -1, Cc).sum(1)
            part[:, 0, 1] = (gh * yh).reshape(N, -1, Cc).sum(1)
            part[:, 0, 2] = yh.reshape(N, -1, Cc).sum(1)

    def gconv(self, g, x, wpack, bias, out, *, in_cs=None, in_co=0, out_cs=None, out_co=0, act="none", slope=0.2,
              stats=None, stats_slots=0, stats_slot0=0, accumulate=False, fuse=None):
        if is_twin(wpack, bias):
            return self.twin_gconv(functools.partial(self.gconv, g), x, wpack, bias, out, in_cs=in_cs, in_co=in_co,
                                   out_cs=out_cs, out_co=out_co, act=act, slope=slope, stats=stats,
                                   stats_slots=stats_slots, stats_slot0=stats_slot0, accumulate=accumulate, fuse=fuse,
                                   C_=g.Co)
        N = x.shape[0]
        xin = _v5(x)[..., in_co:in_co + g.Ci].float()
        Wt = wpack[g.pack_offset:g.pack_offset + g.w_rows * g.Kp].view(g.w_rows, g.Kp).float()
        Wt = Wt[:g.Co, :g.T * g.Ci].reshape(g.Co, g.T, g.Ci)
        # unpadded form of the fused data gradient (gs_gconv_ring_slots): the zero-border conv on the domain extended by
        # the fold, folded in fp32 before the storage rounding
        ring = fuse["fold"] if (fuse is not None and fuse["fold"] > 0 and
                                tuple(fuse["y"].shape[-3:-1]) == (g.Ho, g.Wo)) else 0
        z = torch.arange(g.Dc) * g.si
        i = torch.arange(-ring, g.Hc + ring) * g.si
        j = torch.arange(-ring, g.Wc + ring) * g.si
        acc = torch.zeros(N, g.Dc, g.Hc + 2 * ring, g.Wc + 2 * ring, g.Co)
        for t in range(g.T):
            iz, okz = _border(z + g.dd[t], g.Di, g.border)
            ih, okh = _border(i + g.dh[t], g.Hi, g.border)
            iw, okw = _border(j + g.dw[t], g.Wi, g.border)
            ok = (okz[:, None, None] & okh[None, :, None] & okw[None, None, :]).float()
            patch = xin[:, iz][:, :, ih][:, :, :, iw] * ok[None, :, :, :, None]
            acc += patch @ Wt[:, t, :].t()
        if bias is not None:
            acc += bias[:g.Co].float()
        if stats_slots > 0:
            sv = stats.view(N, stats_slots, 2, g.Co)
            sv[:, stats_slot0, 0] = acc.sum((1, 2, 3))
            sv[:, stats_slot0, 1] = (acc * acc).sum((1, 2, 3))
        if ring:
            acc = _fold(acc.squeeze(1), (g.Hc, g.Wc), ring, fuse["fold_mode"]).view(N, g.Dc, g.Hc, g.Wc, g.Co)
        acc = _act(acc, act, slope)
        oz = torch.arange(g.Dc) * g.so + g.pz
        oh = torch.arange(g.Hc) * g.so + g.py
        ow = torch.arange(g.Wc) * g.so + g.px
        idx = (slice(None), oz[:, None, None], oh[None, :, None], ow[None, None, :], slice(out_co, out_co + g.Co))
        if accumulate:      # bf16 read-modify-write like the kernel epilogue
            acc = acc.to(out.dtype).float() + _v5(out)[idx].float()
        _v5(out)[idx] = acc.to(out.dtype)
        if fuse is not None:   # gs_gconv_forward_fused: reduction pass of the consumer's InstanceNorm backward
            y, Cc = fuse["y"], g.Co
            gf = out.float() if ring else _fold(out.float(), y.shape[1:-1], fuse["fold"], fuse["fold_mode"])
            if fuse.get("g2") is not None:
                gf = gf + fuse["g2"].float()
            bc = (N,) + (1,) * (y.dim() - 2) + (Cc,)
            mr = fuse["mean_rstd"].view(N, 2, Cc)
            yh = (y.float() - mr[:, 0].reshape(bc)) * mr[:, 1].reshape(bc)
            gh = gf * _act_grad_from_out(yh, fuse["act"], fuse.get("slope", 0.2))
            part = fuse["partial"][:N * 3 * Cc].view(N, 1, 3, Cc)
            part[:, 0, 0] = gh.reshape(N, -1, Cc).sum(1)
            part[:, 0, 1] = (gh * yh).reshape(N, -1, Cc).sum(1)
            part[:, 0, 2] = yh.reshape(N, -1, Cc).sum(1)

    @staticmethod
    def can_merge_wgrad(w):
        if w.si != 1 or w.P % 64 or w.Q % 64:
            return False
        if w.T == 9 and w.Da == 1:
            return True
        return (w.T == 27 and w.Da > 1 and
                all(w.dd[9 * k + t] == w.dd[9 * k] and w.dh[9 * k + t] == w.dh[t] and w.dw[9 * k + t] == w.dw[t]
                    for k in range(3) for t in range(9)))

    def wgrad(self, w, a, g, dw, *, a_cs=None, a_co=0, g_cs=None, g_co=0, pair=None, fresh=False):
        if is_twin(dw):
            return self.twin_wgrad(w, a, g, dw, a_cs=a_cs, a_co=a_co, g_cs=g_cs, g_co=g_co, pair=pair)
        if fresh:      # the caller's guarantee behind gs_wgrad_desc.dw_fresh, checked here: the host logic is under test
            assert not bool(dw.any()), "wgrad(fresh=True): the gradient slice does not hold zeros"
        if pair is not None:
            self.wgrad(w, pair[0], pair[1], dw, a_cs=a_cs, a_co=a_co, g_cs=g_cs, g_co=g_co)
        av = _v5(a)[..., a_co:a_co + w.P].float()
        gv = _v5(g)[..., g_co:g_co + w.Q].float()
        z = torch.arange(w.Da) * w.si
        i = torch.arange(w.Ha) * w.si
        j = torch.arange(w.Wa) * w.si
        d = dw.view(w.P, w.T, w.Q)
        for t in range(w.T):
            iz, okz = _border(z + w.dd[t], w.Dg, w.border)
            ih, okh = _border(i + w.dh[t], w.Hg, w.border)
            iw, okw = _border(j + w.dw[t], w.Wg, w.border)
            ok = (okz[:, None, None] & okh[None, :, None] & okw[None, None, :]).float()
            patch = gv[:, iz][:, :, ih][:, :, :, iw] * ok[None, :, :, :, None]
            d[:, t, :] += torch.einsum("nzijp,nzijq->pq", av, patch)

    def bias_grad(self, dy, C_, db, *, cs=None, co=0):
        if is_twin(db):
            return self.twin_bias_grad(dy, C_, db, cs=cs, co=co)
        db[:C_] += dy[..., co:co + C_].float().reshape(-1, C_).sum(0)

    def scalar_affine(self, xs, rows, consts=None):
        """gs_scalar_affine: [c_r + sum_k rows[r][k] * xs[k]] (None = 0), accumulated in k order in fp32"""
        out = torch.zeros(len(rows), dtype=torch.float32)
        for r, row in enumerate(rows):
            acc = torch.tensor(float(consts[r]) if consts is not None else 0.0, dtype=torch.float32)
            for w, x in zip(row, xs):
                if x is not None:
                    acc = acc + torch.tensor(float(w), dtype=torch.float32) * x.detach().float().reshape(())
            out[r] = acc
        return out

    def flip_w_if(self, x, flag):
        return x.flip(-1).contiguous() if int(flag.reshape(-1)[0]) else x.clone()

    def sum2(self, a, b):
        return a + b

    # ---- feature taps (CUT): torch indexing, as the reference does it (cut.py:262-277) -----------------------------
    def tap_gather(self, src, pid, c):
        n = src.shape[0]
        return src.reshape(n, -1, src.shape[-1])[:, pid, :c].float()

    def tap_scatter_add(self, dst, pid, g, W, f0=0):
        n, Wp = dst.shape[0], dst.shape[-2]
        y, x = torch.div(pid, W, rounding_mode="floor"), pid % W
        flat = (y + f0) * Wp + x + f0
        view = dst.view(n, -1, dst.shape[-1])
        view[:, flat, :g.shape[-1]] = (view[:, flat, :g.shape[-1]].float() + g.float()).to(dst.dtype)

    def tap_rows_sum(self, g, db):
        db[:g.shape[-1]] += g.float().reshape(-1, g.shape[-1]).sum(0)

    def zeros_like_act(self, t):
        return torch.zeros_like(t)

    def image_tap_gather(self, x, pid, pad):
        xp = torch.nn.functional.pad(x.float(), (pad,) * 4, mode="reflect")
        return xp.permute(0, 2, 3, 1).flatten(1, 2)[:, pid, :]

    def image_tap_scatter(self, g, pid, shape, pad):
        with torch.enable_grad():       # (called from inside a backward pass)
            x = torch.zeros(shape, dtype=torch.float32, requires_grad=True)
            out = torch.nn.functional.pad(x, (pad,) * 4, mode="reflect").permute(0, 2, 3, 1).flatten(1, 2)[:, pid, :]
            gx, = torch.autograd.grad(out, x, g.float())
        return gx

    # ---- InstanceNorm + activation -------------------------------------------------------------------
    def inorm_finalize(self, partial, N, slots, Cc, hw, mean_rstd, eps=1e-5):
        p = partial.view(N, slots, 2, Cc).double().sum(1)
        mean = p[:, 0] / hw
        var = (p[:, 1] / hw - mean * mean).clamp_min(0)
        mr = mean_rstd.view(N, 2, Cc)
        mr[:, 0] = mean.float()
        mr[:, 1] = (1.0 / torch.sqrt(var + eps)).float()

    def inorm_act_forward(self, y, mean_rstd, res, x, act="none", slope=0.2):
        N, Cc = y.shape[0], y.shape[-1]
        bc = (N,) + (1,) * (y.dim() - 2) + (Cc,)
        mr = mean_rstd.view(N, 2, Cc)
        v = (y.float() - mr[:, 0].reshape(bc)) * mr[:, 1].reshape(bc)
        v = _act(v, act, slope)
        if res is not None:
            v = v + res.float()
        x.copy_(v.to(x.dtype))

    def inorm_stats_act_forward(self, y, partial, slots, mean_rstd, res, x, act="none", slope=0.2, eps=1e-5):
        N, Cc = y.shape[0], y.shape[-1]
        self.inorm_finalize(partial, N, slots, Cc, y.numel() // (N * Cc), mean_rstd, eps)
        self.inorm_act_forward(y, mean_rstd, res, x, act=act, slope=slope)

    def inorm_act_backward(self, g_pad, g2, y, mean_rstd, dy, gsum, fold=0, fold_mode="reflect", act="none",
                           slope=0.2, bias_grad=None, pre=None):
        N, Cc = y.shape[0], y.shape[-1]
        sp = tuple(range(1, y.dim() - 1))
        bc = (N,) + (1,) * (y.dim() - 2) + (Cc,)
        g = _fold(g_pad.float(), y.shape[1:-1], fold, fold_mode)
        if g2 is not None:
            g = g + g2.float()
        if gsum is not None:
            gsum.copy_(g.to(gsum.dtype))
        if mean_rstd is None:
            dy.copy_((g * _act_grad_from_out(y.float(), act, slope)).to(dy.dtype))
            return
        mr = mean_rstd.view(N, 2, Cc)
        rstd = mr[:, 1].reshape(bc)
        yh = (y.float() - mr[:, 0].reshape(bc)) * rstd
        gh = g * _act_grad_from_out(yh, act, slope)
        s1 = gh.mean(sp, keepdim=True)
        s2 = (gh * yh).mean(sp, keepdim=True)
        if pre is not None:            # sums delivered by the fused data-gradient launch: use THEM (that is the test)
            hw = y.numel() // (N * Cc)
            part = pre[1][:N * pre[0] * 3 * Cc].view(N, pre[0], 3, Cc).sum(1)
            s1 = (part[:, 0] / hw).reshape(bc)
            s2 = (part[:, 1] / hw).reshape(bc)
        d = rstd * (gh - s1 - yh * s2)
        dy.copy_(d.to(dy.dtype))
        if bias_grad is not None:      # sum over pixels of dy: identically zero up to rounding
            bias_grad[:Cc] += d.reshape(-1, Cc).sum(0)
        # per-image totals [N][3][C] for a deferred norm_bias_grads call (same contract as the HIP backend)
        hw = y.numel() // (N * Cc)
        sums = torch.stack([s1.reshape(N, Cc) * hw, s2.reshape(N, Cc) * hw, yh.reshape(N, -1, Cc).sum(1)], 1).contiguous()
        return sums, 0

    def norm_bias_grads(self, items):
        """items: (sums holder, offset, mean_rstd, db, N, C, hw): db[c] += sum_n -rstd * S2 * S3 / hw"""
        for holder, off, mean_rstd, db, N, Cc, hw in items:
            sums = holder.reshape(-1)[off:off + N * 3 * Cc].view(N, 3, Cc)
            rstd = mean_rstd.view(N, 2, Cc)[:, 1]
            db[:Cc] += (-rstd * sums[:, 1] * sums[:, 2] / hw).sum(0)

    # ---- generalised norm / activation for skip-connection graphs (U-Net) -------------------------------
    @staticmethod
    def _full_seed(seed, seed_dev):
        """64-bit mask seed = host part + optional device part int32[2] (lo, hi), as in gs_norm_ex_desc.seed_dev"""
        if seed_dev is None:
            return seed
        lo, hi = (int(v) & 0xFFFFFFFF for v in seed_dev.tolist())
        return (seed + (hi << 32 | lo)) & 0xFFFFFFFFFFFFFFFF

    @staticmethod
    def _drop_scale(shape, drop_p, seed):
        """same counter-based hash as ganslate_amd/csrc/norm_ex.hip (murmur3 finaliser), evaluated with int64"""
        N, H, W, Cc = shape
        if drop_p <= 0:
            return torch.ones(shape)
        M = 0xFFFFFFFF

        def h32(x):
            x = x & M
            x = x ^ (x >> 16); x = (x * 0x85EBCA6B) & M
            x = x ^ (x >> 13); x = (x * 0xC2B2AE35) & M
            return x ^ (x >> 16)

        idx = torch.arange(H * W * Cc, dtype=torch.int64).view(1, H, W, Cc)
        n = torch.arange(1, N + 1, dtype=torch.int64).view(N, 1, 1, 1)
        inner = h32(((seed & M) + 0x9E3779B9 * n) & M)
        h = h32(idx ^ inner ^ ((seed >> 32) & M))
        u = (h >> 8).float() * (1.0 / 16777216.0)
        return torch.where(u >= drop_p, torch.full(shape, 1.0 / (1.0 - drop_p)), torch.zeros(shape))

    def norm_act_forward_ex(self, y, mean_rstd, x1, x2=None, act1="none", act2="none", slope=0.2, x1_co=0, x2_co=0,
                            drop_p=0.0, seed=0, seed_dev=None):
        seed = self._full_seed(seed, seed_dev)
        flat = lambda t: t if t is None or t.dim() == 4 else t.view(t.shape[0], -1, t.shape[-2], t.shape[-1])
        y, x1, x2 = flat(y), flat(x1), flat(x2)        # geometry-free: a volume is D*H rows
        N, H, W, Cc = y.shape
        v = y.float()
        if mean_rstd is not None:
            mr = mean_rstd.view(N, 2, Cc)
            v = (v - mr[:, 0][:, None, None, :]) * mr[:, 1][:, None, None, :]
        v = v * self._drop_scale(y.shape, drop_p, seed)
        x1[..., x1_co:x1_co + Cc] = _act(v, act1, slope).to(x1.dtype)
        if x2 is not None:
            x2[..., x2_co:x2_co + Cc] = _act(v, act2, slope).to(x2.dtype)

    def norm_act_backward_ex(self, g1, g2, y, mean_rstd, dy, act1="none", act2="none", slope=0.2, g1_co=0, g2_co=0,
                             drop_p=0.0, seed=0, bias_grad=None, seed_dev=None):
        seed = self._full_seed(seed, seed_dev)
        flat = lambda t: t if t is None or t.dim() == 4 else t.view(t.shape[0], -1, t.shape[-2], t.shape[-1])
        g1, g2, y, dy = flat(g1), flat(g2), flat(y), flat(dy)
        N, H, W, Cc = y.shape
        yh = y.float()
        if mean_rstd is not None:
            mr = mean_rstd.view(N, 2, Cc)
            rstd = mr[:, 1][:, None, None, :]
            yh = (yh - mr[:, 0][:, None, None, :]) * rstd
        gh = g1[..., g1_co:g1_co + Cc].float() * _act_grad_from_out(yh, act1, slope)
        if g2 is not None:
            gh = gh + g2[..., g2_co:g2_co + Cc].float() * _act_grad_from_out(yh, act2, slope)
        gh = gh * self._drop_scale(y.shape, drop_p, seed)
        if mean_rstd is None:
            dy.copy_(gh.to(dy.dtype))
            return
        s1 = gh.mean((1, 2), keepdim=True)
        s2 = (gh * yh).mean((1, 2), keepdim=True)
        d = rstd * (gh - s1 - yh * s2)
        dy.copy_(d.to(dy.dtype))
        if bias_grad is not None:
            bias_grad[:Cc] += d.sum((0, 1, 2))

    # ---- V-Net elementwise family: IN3d -> [+res] -> PReLU(C) -> [+res] (vnet3d.py:155-267, invertible.py:8-48) --------
    @staticmethod
    def _p_preact(y, mean_rstd, res, res_mode, res_mod, C_, y_co, res_co):
        N = y.shape[0]
        bc = (N,) + (1,) * (y.dim() - 2) + (C_,)
        yh = y[..., y_co:y_co + C_].float()
        if mean_rstd is not None:
            mr = mean_rstd.view(N, 2, C_)
            yh = (yh - mr[:, 0].reshape(bc)) * mr[:, 1].reshape(bc)
        r = None
        if res is not None:
            if res_mod > 0:
                r = res[..., res_co + (torch.arange(C_) % res_mod)].float()
            else:
                r = res[..., res_co:res_co + C_].float()
        u = yh + r if res_mode == 1 else yh
        return yh, u, r

    def pnorm_forward(self, y, mean_rstd, out, *, C, slope=None, res=None, res_mode=0, res_mod=0, y_co=0, res_co=0,
                      out_co=0):
        if is_twin(slope):
            return self.twin_pnorm_forward(y, mean_rstd, out, C=C, slope=slope, res=res, res_mode=res_mode, res_mod=res_mod,
                                           y_co=y_co, res_co=res_co, out_co=out_co)
        yh, u, r = self._p_preact(y, mean_rstd, res, res_mode, res_mod, C, y_co, res_co)
        v = torch.where(u > 0, u, u * slope[:C]) if slope is not None else u
        if res_mode == 2:
            v = v + r
        elif res_mode == 3:
            v = r - v
        out[..., out_co:out_co + C] = v.to(out.dtype)

    def pnorm_backward(self, g, y, mean_rstd, dy, *, C, slope=None, dslope=None, g2=None, res=None, res_mode=0,
                       res_mod=0, gres=None, bias_grad=None, g_co=0, g2_co=0, y_co=0, res_co=0, dy_co=0, gres_co=0):
        if is_twin(slope, dslope, bias_grad):
            return self.twin_pnorm_backward(g, y, mean_rstd, dy, C=C, slope=slope, dslope=dslope, g2=g2, res=res,
                                            res_mode=res_mode, res_mod=res_mod, gres=gres, bias_grad=bias_grad, g_co=g_co,
                                            g2_co=g2_co, y_co=y_co, res_co=res_co, dy_co=dy_co, gres_co=gres_co)
        yh, u, _ = self._p_preact(y, mean_rstd, res, res_mode, res_mod, C, y_co, res_co)
        gt = g[..., g_co:g_co + C].float()
        if g2 is not None:
            gt = gt + g2[..., g2_co:g2_co + C].float()
        if res_mode == 3:
            gt = -gt
        if slope is not None:
            gu = torch.where(u > 0, gt, gt * slope[:C])
            if dslope is not None:
                dslope[:C] += torch.where(u > 0, torch.zeros_like(u), gt * u).reshape(-1, C).sum(0)
        else:
            gu = gt
        if gres is not None:
            gres[..., gres_co:gres_co + C] = gu.to(gres.dtype)
        if mean_rstd is None:
            dy[..., dy_co:dy_co + C] = gu.to(dy.dtype)
            return
        N = y.shape[0]
        sp = tuple(range(1, y.dim() - 1))
        rstd = mean_rstd.view(N, 2, C)[:, 1].reshape((N,) + (1,) * (y.dim() - 2) + (C,))
        d = rstd * (gu - gu.mean(sp, keepdim=True) - yh * (gu * yh).mean(sp, keepdim=True))
        dy[..., dy_co:dy_co + C] = d.to(dy.dtype)
        if bias_grad is not None:
            bias_grad[:C] += d.reshape(-1, C).sum(0)

    def slice_stats(self, x, co, C, mean_rstd, eps=1e-5):
        N = x.shape[0]
        v = x[..., co:co + C].double().reshape(N, -1, C)
        mean = v.mean(1)
        var = ((v * v).mean(1) - mean * mean).clamp_min(0)
        mr = mean_rstd.view(N, 2, C)
        mr[:, 0] = mean.float()
        mr[:, 1] = (1.0 / torch.sqrt(var + eps)).float()

    def add_views(self, dst, src, C, dst_co=0, src_co=0, accumulate=True):
        v = src[..., src_co:src_co + C].float()
        if accumulate:
            v = v + dst[..., dst_co:dst_co + C].float()
        dst[..., dst_co:dst_co + C] = v.to(dst.dtype)

    def repeat_backward(self, g, g_img, C, g_co=0):
        Cin = g_img.shape[1]
        gv = g[..., g_co:g_co + C].float()
        for c0 in range(Cin):
            g_img[:, c0] += gv[..., c0::Cin].sum(-1)

    # ---- network boundary --------------------------------------------------------------------------------
    def image_to_act(self, img, act_t):
        Cc = img.shape[1]
        act_t.zero_()
        act_t[..., :Cc] = img.movedim(1, -1).to(act_t.dtype)

    def image_pair_to_act(self, a, b, act_t):
        self.image_to_act(torch.cat([a, b], dim=1), act_t)

    def image_pair_to_act_backward(self, g, ga, gb, Ca, Cb):
        gi = g.float().movedim(-1, 1)
        if ga is not None:
            ga.copy_(gi[:, :Ca])
        if gb is not None:
            gb.copy_(gi[:, Ca:Ca + Cb])

    def act_to_image(self, act_t, img, act="none"):
        Cc = img.shape[1]
        img.copy_(_act(act_t[..., :Cc].float(), act, 0.0).movedim(-1, 1))

    def act_to_image_backward(self, g_img, out_img, g_act, act="none"):
        Cc = g_img.shape[1]
        g = g_img
        if act != "none":
            g = g * _act_grad_from_out(out_img, act, 0.0)
        g_act.zero_()
        g_act[..., :Cc] = g.movedim(1, -1).to(g_act.dtype)

    def image_to_act_backward(self, g_pad, g_img, fold=0, fold_mode="reflect", accumulate=False):
        Cc = g_img.shape[1]
        g = _fold(g_pad.float(), g_img.shape[2:], fold, fold_mode)[..., :Cc].movedim(-1, 1)
        if accumulate:
            g_img += g
        else:
            g_img.copy_(g)

    # ---- W-fold boundary transforms (csrc/wfold.hip) -----------------------------------------------------------------
    def image_unfold(self, img, act_t, k, p, border):
        Cc, W = img.shape[1], img.shape[-1]
        act_t.zero_()
        xl = img.movedim(1, -1)                                   # [N, ..., W, C]
        for dw in range(k):
            j, ok = _border(torch.arange(W) + dw - p, W, border)
            v = xl.index_select(-2, j) * ok.float()[:, None]
            act_t[..., dw * Cc:(dw + 1) * Cc] = v.to(act_t.dtype)

    def image_unfold_backward(self, g, g_img, k, p, fold, border, accumulate=False):
        Cc, W = g_img.shape[1], g_img.shape[-1]
        gf = g.float()
        # (depth, row) padding adjoint on every spatial axis but W
        sp = g_img.shape[2:]
        for ax, n in enumerate(sp[:-1], start=1):
            if fold == 0 or (len(sp) == 3 and ax == 1 and n == 1):
                continue
            src, _ = _border(torch.arange(n + 2 * fold) - fold, n, border)
            shape = list(gf.shape); shape[ax] = n
            nxt = torch.zeros(shape); nxt.index_add_(ax, src, gf); gf = nxt
        out = torch.zeros(*g_img.shape[:1], *sp, Cc)             # channels-last
        for dw in range(k):
            j, ok = _border(torch.arange(W) + dw - p, W, border)
            out.index_add_(out.dim() - 2, j, gf[..., dw * Cc:(dw + 1) * Cc] * ok.float()[:, None])
        out = out.movedim(-1, 1)
        if accumulate:
            g_img += out
        else:
            g_img.copy_(out)

    def shiftadd_to_image(self, z, bias, img, k, act="none"):
        if is_twin(bias):
            return self.twin_shiftadd_to_image(z, bias, img, k, act=act)
        Co, W = img.shape[1], img.shape[-1]
        acc = torch.zeros(*img.shape[:1], *img.shape[2:], Co)
        for dw in range(k):
            acc += z[..., dw:dw + W, dw * Co:(dw + 1) * Co].float()
        if bias is not None:
            acc += bias[:Co].float()
        img.copy_(_act(acc, act, 0.0).movedim(-1, 1))

    def shiftadd_to_image_backward(self, g_img, out_img, gz, k, act="none"):
        Co, W = g_img.shape[1], g_img.shape[-1]
        g = g_img
        if act != "none":
            g = g * _act_grad_from_out(out_img, act, 0.0)
        gl = g.movedim(1, -1)
        gz.zero_()
        for dw in range(k):
            gz[..., dw:dw + W, dw * Co:(dw + 1) * Co] = gl.to(gz.dtype)

    # ---- losses -----------------------------------------------------------------------------------------
    # ---- device-side 3-D training patches: the reference's own expression (normalization.py:18-30) ---------------------
    def patch_zscore(self, volume, start, size, out, scale_to_range=(-1.0, 1.0)):
        z, y, x = start
        d, h, w = size
        t = volume[z:z + d, y:y + h, x:x + w].float()
        t = (t - t.mean()) / t.std()
        if scale_to_range:
            delta1 = t.max() - t.min()
            delta2 = scale_to_range[1] - scale_to_range[0]
            t = (delta2 * (t - t.min()) / delta1) + scale_to_range[0]
        out.copy_(t.reshape(out.shape))

    # ---- device-side image preprocessing (oracle/pil_ref.py: Pillow's resampler restated) ---------------------------
    def u8_resample_h(self, img, out, bounds, kk):
        """out[y][xx][c] = clip8((2^21 + sum_k img[y][xmin + k][c] * kk[xx][k]) >> 22) from the tables as given"""
        a = img.to(torch.int64)
        for xx in range(out.shape[1]):
            xmin, xn = int(bounds[xx, 0]), int(bounds[xx, 1])
            acc = torch.full(a[:, 0].shape, 1 << 21, dtype=torch.int64)
            for x in range(xn):
                acc += a[:, xmin + x] * int(kk[xx, x])
            out[:, xx] = (acc >> 22).clamp(0, 255).to(torch.uint8)

    def u8_resample_v(self, tmp, out, bounds, kk):
        a = tmp.to(torch.int64)
        for yy in range(out.shape[0]):
            ymin, yn = int(bounds[yy, 0]), int(bounds[yy, 1])
            acc = torch.full(a[0].shape, 1 << 21, dtype=torch.int64)
            for y in range(yn):
                acc += a[ymin + y] * int(kk[yy, y])
            out[yy] = (acc >> 22).clamp(0, 255).to(torch.uint8)

    def u8_resample_v_crop_normalize(self, tmp, out, out_h, bounds, kk, top, left, flip):
        a = tmp.to(torch.int64)
        fh, fw = out.shape[1], out.shape[2]
        rows = torch.empty((fh, a.shape[1], a.shape[2]), dtype=torch.uint8)
        for i in range(fh):
            yy = top + i
            ymin, yn = int(bounds[yy, 0]), int(bounds[yy, 1])
            acc = torch.full(a[0].shape, 1 << 21, dtype=torch.int64)
            for y in range(yn):
                acc += a[ymin + y] * int(kk[yy, y])
            rows[i] = (acc >> 22).clamp(0, 255).to(torch.uint8)
        win = rows[:, left:left + fw]
        if flip:
            win = win.flip(1)
        x = win.to(torch.float32) / 255.0
        out.copy_(((x - 0.5) / 0.5).permute(2, 0, 1))

    def adv_loss(self, x, mode, target_is_real, label, loss=None, grad=None, grad_scale=None):
        """AdversarialLoss.calculate_loss (ganslate/nn/losses/adversarial_loss.py:52-73) through torch autograd"""
        import torch.nn.functional as F
        with torch.enable_grad():
            xi = x.detach().clone().requires_grad_()
            if mode == "lsgan":
                val = F.mse_loss(xi, torch.full_like(xi, label))
            elif mode == "vanilla":
                val = F.binary_cross_entropy_with_logits(xi, torch.full_like(xi, label))
            elif mode == "wgangp":
                val = -xi.mean() if target_is_real else xi.mean()
            else:
                val = F.softplus(-xi if target_is_real else xi).view(xi.size(0), -1).mean(dim=1)
            if grad is not None:
                up = grad_scale if grad_scale is not None else torch.ones_like(val)
                (g,) = torch.autograd.grad(val, xi, up.reshape(val.shape).to(val.dtype))
                grad.copy_(g)
        if loss is not None:
            loss.copy_(val.detach().reshape(loss.shape))

    def mse_const(self, x, target, loss=None, grad=None, grad_scale=None):
        if loss is not None:
            loss.copy_(((x - target) ** 2).mean())
        if grad is not None:
            s = grad_scale if grad_scale is not None else 1.0
            grad.copy_(s * 2.0 * (x - target) / x.numel())

    def l1(self, a, b, loss=None, grad_a=None, grad_scale=None):
        if loss is not None:
            loss.copy_((a - b).abs().mean())
        if grad_a is not None:
            s = grad_scale if grad_scale is not None else 1.0
            grad_a.copy_(s * torch.sign(a - b) / a.numel())

    def mean(self, x, out):
        out.copy_(x.mean())

    def ssim_distance(self, x, y, out):
        out.copy_(ssim_distance(x, y))

    def ssim_distance_backward(self, x, y, grad_y, grad_scale=None):
        """gradient of the SSIM distance w.r.t. y by differentiating the restatement with autograd"""
        yy = y.detach().clone().requires_grad_()
        with torch.enable_grad():
            d = ssim_distance(x.detach(), yy)
        (g,) = torch.autograd.grad(d, yy)
        grad_y.copy_(g * (grad_scale if grad_scale is not None else 1.0))

    # ---- optimiser ---------------------------------------------------------------------------------------
    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, zero_grad=True):
        gi = g * grad_scale
        m.lerp_(gi, 1 - beta1)
        v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2s = math.sqrt(1 - beta2 ** step)
        p.addcdiv_(m, (v.sqrt() / bc2s).add_(eps), value=-lr / bc1)
        if zero_grad:
            g.zero_()

    def wgrad_adam(self, w, a, g, p, m, v, hyper_dev, packs=None, tr=None) -> bool:
        """gs_wgrad_adam: the layer's weight gradient, consumed by its Adam update at once (the gradient buffer is not touched);
        tr = (base[T], kp[T], pack): pack[base[t] + q * kp[t] + p] = W[p][t][q] for the taps with base[t] >= 0"""
        dw = torch.zeros(p.numel(), dtype=torch.float32, device=p.device)
        self.wgrad(w, a, g, dw)
        self.adam_step_dev(p, dw, m, v, hyper_dev, grad_scale=1.0, zero_grad=False, packs=packs)
        if tr is not None:
            base, kp, pack = tr
            P, T, Q = w.P, w.T, w.Q
            W = p.view(P, T, Q)
            for t in range(T):
                b = int(base[t])
                if b < 0:
                    continue
                dst = b + torch.arange(Q)[None, :] * int(kp[t]) + torch.arange(P)[:, None]        # [P][Q]
                pack[dst.reshape(-1)] = W[:, t, :].reshape(-1).to(pack.dtype)
        return True

    def adam_step_dev(self, p, g, m, v, hyper_dev, grad_scale=1.0, zero_grad=True, packs=None):
        """hyper_dev = float32[6]: lr, beta1, beta2, eps, 1-beta1^t, sqrt(1-beta2^t) — the update of adam_step with the
        scalars taken from the tensor (they were rounded to fp32 when it was written, like the kernel's arguments)"""
        lr, beta1, beta2, eps, bc1, bc2s = (float(h) for h in hyper_dev.tolist())
        gi = g * grad_scale
        m.lerp_(gi, 1 - beta1)
        v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
        p.addcdiv_(m, (v.sqrt() / bc2s).add_(eps), value=-lr / bc1)
        if zero_grad:
            g.zero_()
        if packs is not None:       # gs_adam_step_dev_packs: identity pack groups refreshed with the update
            for inv, pack in ((packs[0], packs[1]), (packs[2], packs[3])):
                if inv is None:
                    continue
                n8 = p.numel() // 8
                src = torch.nonzero(inv[:n8] >= 0).flatten()
                dst = inv[:n8][src].long()
                pack[:pack.numel() // 8 * 8].view(-1, 8)[dst] = p[:n8 * 8].view(-1, 8)[src].to(pack.dtype)

    def adam_step_dev_ranges(self, p, g, m, v, ranges_dev, max_len, hyper_dev, grad_scale=1.0, zero_grad=True, packs=None):
        """gs_adam_step_dev_packs_ranges: the update of every range [start, end)"""
        for a, b in ranges_dev.tolist():
            sub = None
            if packs is not None:
                inv_f, fpack, inv_d, dpack = packs
                g0, g1 = a // 8, (b + 7) // 8
                sub = (inv_f[g0:g1] if inv_f is not None else None, fpack, inv_d[g0:g1] if inv_d is not None else None, dpack)
            self.adam_step_dev(p[a:b], g[a:b], m[a:b], v[a:b], hyper_dev, grad_scale=grad_scale, zero_grad=zero_grad, packs=sub)

    def pool_query(self, pool, images, out, code_dev):
        """ganslate/data/utils/image_pool.py:31-60 with the coin flips given as codes (see gs_pool_query)"""
        for b, c in enumerate(code_dev.tolist()):
            img = images[b].clone()
            if c < 0:
                out[b] = img
            elif c & 0x40000000:
                slot = c & 0x3fffffff
                out[b] = pool[slot]
                pool[slot] = img
            else:
                pool[c] = img
                out[b] = img

    # ---- PatchNCE + patch MLP: torch autograd of the reference composition --------------------------------------
    # ---- SelfAttentionBlock (ganslate/nn/attention.py:26-47 on NDHWC activations) ------------------------------------
    def attn_forward(self, x, params, need_backward=None):
        with torch.enable_grad():
            x_ = x.detach().float().clone().requires_grad_()
            p_ = {k: v.detach().clone().requires_grad_() for k, v in params.items()}
            out = attention_reference(x_, p_)
        return out.detach().to(x.dtype), (x_, p_, out)

    def attn_backward(self, saved, dout, params, grads):
        x_, p_, out = saved
        keys = [k for k in p_ if grads is not None and grads.get(k) is not None]
        g = torch.autograd.grad(out, [x_] + [p_[k] for k in keys], dout.float())
        for k, gk in zip(keys, g[1:]):
            grads[k] += gk
        return g[0].to(dout.dtype)

    def patchnce_forward(self, xq, xk, params, *, batch, nc=256, nce_T=0.07, lambda_nce=1.0):
        with torch.enable_grad():
            xq_ = [t.detach().float().clone().requires_grad_() for t in xq]
            p_ = params.detach().clone().requires_grad_()
            loss = patchnce_reference(xq_, [t.detach() for t in xk], p_, batch, nc, nce_T, lambda_nce)
        return loss.detach(), (loss, xq_, p_)

    def patchnce_backward(self, saved, params, grads, grad_scale=None):
        loss, xq_, p_ = saved
        with torch.enable_grad():
            total = loss.sum()
        g = torch.autograd.grad(total, xq_ + [p_])
        s = float(grad_scale) if grad_scale is not None else 1.0
        grads += g[-1] * s
        return [t * s for t in g[:-1]]

    def repack(self, master, index, pack):
        idx = index.long()
        vals = master.reshape(-1)[idx.clamp_min(0)]
        pack.copy_(torch.where(idx >= 0, vals, torch.zeros_like(vals)).to(pack.dtype))

    def repack_tiled(self, master, index, pack, rows, kp):
        """same refresh for one [rows][kp] segment (the HIP side only changes the access order)"""
        self.repack(master, index, pack)

    def repack_groups(self, master, gindex, pack, index=None):
        """one base index per 8 pack elements: pack[8 g + j] = master[gindex[g] + j]; -1: zeros, -2: the group's own
        entries of the element-wise table, -3: left alone (written by repack_tiled_groups)"""
        b = gindex.long()[:, None]
        idx = torch.where(b >= 0, b + torch.arange(8)[None, :], torch.full((len(gindex), 8), -1))
        if index is not None:
            idx = torch.where(b == -2, index.long().reshape(-1, 8), idx)
        vals = master.reshape(-1)[idx.clamp_min(0)]
        vals = torch.where(idx >= 0, vals, torch.zeros_like(vals)).to(pack.dtype)
        keep = (gindex != -3)
        pack.view(-1, 8)[keep] = vals[keep]

    def repack_tiled_groups(self, master, gindex, pack, seg, tiles):
        """all transposed segments of a pack (see HipOps.repack_tiled_groups)"""
        for off, goff, rows, kp, _ in seg.tolist():
            g = gindex[goff:goff + rows // 8 * kp].long().reshape(rows // 8, 1, kp)
            idx = torch.where(g >= 0, g + torch.arange(8).reshape(1, 8, 1), torch.full((rows // 8, 8, kp), -1))
            self.repack(master, idx.reshape(-1), pack[off:off + rows * kp])


def attention_reference(x, p):
    """SelfAttentionBlock.forward (attention.py:26-47) restated on a channels-last tensor x [B, ..., C]: the 1x1x1 convs are
    linear maps over C, energy = q k^T over all voxels, softmax over the keys, out = gamma * (attention @ v) + x"""
    B, Cc = x.shape[0], x.shape[-1]
    t = x.reshape(B, -1, Cc)
    q = t @ p["wq"].t() + p["bq"]
    k = t @ p["wk"].t() + p["bk"]
    v = t @ p["wv"].t() + p["bv"]
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    return (p["gamma"] * torch.bmm(att, v) + t).reshape(x.shape)


def _nce_levels(params, channels, nc):
    """views of the flat parameter buffer: per level (W1 [nc][C], b1, W2 [nc][nc], b2)"""
    out, off = [], 0
    for c in channels:
        W1 = params[off:off + nc * c].view(nc, c); off += nc * c
        b1 = params[off:off + nc]; off += nc
        W2 = params[off:off + nc * nc].view(nc, nc); off += nc * nc
        b2 = params[off:off + nc]; off += nc
        out.append((W1, b1, W2, b2))
    return out


def patchnce_reference(xq, xk, params, batch, nc, nce_T, lambda_nce):
    """FeaturePatchMLP + PatchNCELoss exactly as the reference composes them (cut.py:218-226,229-294;
    cut_losses.py:14-43), differentiable: returns the per-level losses whose sum is CUT._calculate_nce_loss"""
    channels = [int(t.shape[-1]) for t in xq]
    losses = []
    for (W1, b1, W2, b2), q, k in zip(_nce_levels(params, channels, nc), xq, xk):
        feats = []
        for x in (q, k):
            f = torch.relu(x.flatten(0, 1).float() @ W1.t() + b1) @ W2.t() + b2
            feats.append(f / (f.pow(2).sum(1, keepdim=True).pow(0.5) + 1e-7))
        fq, fk = feats[0], feats[1].detach()
        l_pos = (fq * fk).sum(1, keepdim=True)
        qb, kb = fq.view(batch, -1, nc), fk.view(batch, -1, nc)
        n = qb.size(1)
        l_neg = torch.bmm(qb, kb.transpose(2, 1))
        l_neg = l_neg.masked_fill(torch.eye(n, dtype=torch.bool)[None], -10.0).view(-1, n)
        out = torch.cat((l_pos, l_neg), dim=1) / nce_T
        ce = torch.nn.functional.cross_entropy(out, torch.zeros(out.size(0), dtype=torch.long), reduction="none")
        losses.append((ce * lambda_nce).mean() / len(xq))
    return torch.stack(losses)


def ssim_distance(X, Y):
    """Restatement of SSIMLoss.forward (ganslate/nn/losses/utils/ssim.py:65-99) on inputs in [-1, 1]
    mapped to [0, 1] by the caller's (x+1)/2 (cyclegan_losses.py:83-84, train_metrics.py:41-42)."""
    X = (X + 1) / 2
    Y = (Y + 1) / 2
    if X.ndim == 5:
        X = X.reshape(-1, *X.shape[2:])
        Y = Y.reshape(-1, *Y.shape[2:])
    ch = X.shape[1]
    coords = torch.arange(11, dtype=torch.float32) - 5
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum()).view(1, 1, 1, 11).repeat(ch, 1, 1, 1)

    def blur(t):
        t = F.conv2d(t, g, groups=ch)
        return F.conv2d(t, g.transpose(2, 3), groups=ch)

    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = blur(X), blur(Y)
    s1 = blur(X * X) - mu1 ** 2
    s2 = blur(Y * Y) - mu2 ** 2
    s12 = blur(X * Y) - mu1 * mu2
    S1 = (2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)
    S2 = (2 * s12 + C2) / (s1 + s2 + C2)
    return torch.sqrt(torch.relu(2 - (S1 + S2))).mean()
