"""Lowering of nn.Conv2d/3d and nn.ConvTranspose2d/3d (forward, data-gradient, weight-gradient) onto the
"generalised convolution" the HIP kernels implement (include/ganslate_hip.h: gs_gconv_desc / gs_wgrad_desc).

Master weights live in the OTI layout [P][T][Q]:
  Conv2d          : P = out channels, Q = in channels   (torch OIHW  -> [O][kh*kw][I])
  ConvTranspose2d : P = in channels,  Q = out channels  (torch IOHW  -> [I][kh*kw][O])
with P and Q padded to multiples of 8 (padded entries are zero and receive zero gradients).
bf16 weight packs are [rows][Kp], K = (tap, channel) contiguous, Kp = roundup(K, 64); they are produced from
the master by a gather (`pack index`, -1 = zero).

Reference semantics restated here: ganslate/nn/generators/resnet/resnet2d.py:24-25,35,52-57,65,80-87 and
ganslate/nn/discriminators/patchgan/patchgan2d.py:29,36-62 (layer hyper-parameters), torch.nn.Conv2d /
ConvTranspose2d arithmetic; the 3-D twins ganslate/nn/generators/resnet/resnet3d.py:24-64,78-84 and
ganslate/nn/discriminators/patchgan/patchgan3d.py:28-60 lower through the same code with a real depth axis.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pad8pow2(c: int) -> int:
    """smallest 8 * 2^j >= c (the gathered-channel constraint of the conv kernels)"""
    v = 8
    while v < c:
        v *= 2
    return v


@dataclass
class ConvSpec:
    """One convolution layer of the reference network (nn.Conv2d/3d or nn.ConvTranspose2d/3d)."""
    kind: str                     # "conv" | "convT"
    cin: int
    cout: int
    k: int                        # cubic/square kernels only (all hot-path layers are)
    stride: int = 1
    pad: int = 0
    out_pad: int = 0              # ConvTranspose output_padding
    pad_mode: str = "zero"        # "zero" | "reflect" | "replicate" (explicit pad layer folded into the conv)
    bias: bool = True
    dims: int = 2                 # 2 = Conv2d family, 3 = Conv3d family (resnet3d.py, patchgan3d.py)
    # "W-fold" (stride-1 convs with very few channels on one side, i.e. the k7 stem / last layer of the ResNets): the
    # taps of the W axis become channels, the conv that runs has a k x [k x] 1 kernel.
    #   "in" : input channels (dw, ci) = k*cin, produced by gs_image_unfold
    #   "out": output channels (dw, co) = k*cout on a W + 2*pad wide domain, reduced by gs_shiftadd_to_image
    wfold: str = ""

    @property
    def cin_p(self): return pad8pow2(self.k * self.cin) if self.wfold == "in" else pad8(self.cin)
    @property
    def cout_p(self): return pad8pow2(self.k * self.cout) if self.wfold == "out" else pad8(self.cout)
    @property
    def kw(self): return 1 if self.wfold else self.k
    @property
    def T(self): return self.k ** (self.dims - 1) * self.kw
    @property
    def P(self): return self.cout_p if self.kind == "conv" else self.cin_p
    @property
    def Q(self): return self.cin_p if self.kind == "conv" else self.cout_p
    @property
    def master_numel(self): return self.P * self.T * self.Q

    def out_size(self, x: int) -> int:
        if self.kind == "conv":
            return (x + 2 * self.pad - self.k) // self.stride + 1
        return (x - 1) * self.stride - 2 * self.pad + self.k + self.out_pad

    def out_hw(self, *sizes) -> Tuple[int, ...]:
        return tuple(self.out_size(x) for x in sizes)

    # ---- torch <-> master layout ---------------------------------------------------------------
    def torch_weight_shape(self):
        io = (self.cout, self.cin) if self.kind == "conv" else (self.cin, self.cout)
        return io + (self.k,) * self.dims

    def master_from_torch(self, w):
        """torch weight (OI[D]HW for conv, IO[D]HW for convT) -> padded OTI master (numpy or torch)."""
        import torch
        p, q = w.shape[0], w.shape[1]
        m = torch.zeros(self.P, self.T, self.Q, dtype=torch.float32, device=w.device)
        if self.wfold == "in":       # [co][ci][t'][dw] -> [co][t'][dw*cin + ci]
            m[:p, :, :self.k * q] = w.reshape(p, q, self.T, self.k).permute(0, 2, 3, 1).reshape(p, self.T, self.k * q)
        elif self.wfold == "out":    # [co][ci][t'][dw] -> [dw*cout + co][t'][ci]
            m[:self.k * p, :, :q] = w.reshape(p, q, self.T, self.k).permute(3, 0, 2, 1).reshape(self.k * p, self.T, q)
        else:
            m[:p, :, :q] = w.reshape(p, q, self.T).permute(0, 2, 1)
        return m

    def torch_from_master(self, m):
        p, q = (self.cout, self.cin) if self.kind == "conv" else (self.cin, self.cout)
        m = m.reshape(self.P, self.T, self.Q)
        shape = (p, q) + (self.k,) * self.dims
        if self.wfold == "in":
            return m[:p, :, :self.k * q].reshape(p, self.T, self.k, q).permute(0, 3, 1, 2).reshape(shape).contiguous()
        if self.wfold == "out":
            return m[:self.k * p, :, :q].reshape(self.k, p, self.T, q).permute(1, 3, 2, 0).reshape(shape).contiguous()
        return m[:p, :, :q].permute(0, 2, 1).reshape(shape).contiguous()


@dataclass
class GConv:
    """One class of the generalised convolution (mirrors gs_gconv_desc, minus the batch/IO strides).
    A 2-D layer is the depth-1 case (Di = Do = Dc = 1, pz = 0, dd = 0)."""
    Hi: int; Wi: int; Ci: int
    Ho: int; Wo: int; Co: int
    Hc: int; Wc: int
    so: int; py: int; px: int; si: int
    dh: List[int]; dw: List[int]
    border: str
    pack_offset: int = 0          # element offset of this class's [w_rows][Kp] block in the layer's pack
    w_rows: int = 0
    co_real: int = 0              # a Conv's forward class: output channels before padding to 8 (1: the PatchGAN's last layer)
    Di: int = 1
    Do: int = 1
    Dc: int = 1
    pz: int = 0
    dd: Optional[List[int]] = None

    def __post_init__(self):
        if self.dd is None:
            self.dd = [0] * len(self.dh)

    @property
    def T(self): return len(self.dh)
    @property
    def Kp(self): return roundup(self.T * self.Ci, 64)
    @property
    def pixels(self): return self.Dc * self.Hc * self.Wc


@dataclass
class WGrad:
    Ha: int; Wa: int; P: int
    Hg: int; Wg: int; Q: int
    si: int
    dh: List[int]; dw: List[int]
    border: str
    Da: int = 1
    Dg: int = 1
    dd: Optional[List[int]] = None
    p_real: int = 0               # a Conv's weight gradient: rows of the dense side before padding (its output channels)

    def __post_init__(self):
        if self.dd is None:
            self.dd = [0] * len(self.dh)

    @property
    def T(self): return len(self.dh)


@dataclass
class Lowered:
    """Everything the executor needs for one conv layer at one input size."""
    spec: ConvSpec
    Hi: int; Wi: int; Ho: int; Wo: int
    fwd: List[GConv] = field(default_factory=list)
    fwd_index: Optional[np.ndarray] = None        # int32 gather table master -> forward pack
    dgrad: List[GConv] = field(default_factory=list)
    dgrad_index: Optional[np.ndarray] = None
    dgrad_fold: int = 0                           # dgrad output is padded by this much (reflect/replicate)
    dgrad_ring: Optional[GConv] = None            # same gradient on the UNPADDED domain: zero-border class whose fused launch
                                                  # applies the reflect fold itself (gs_gconv_ring_slots; same pack)
    wgrad: Optional[WGrad] = None
    Di: int = 1
    Do: int = 1
    dgrad_dims3: Optional[Tuple[int, int, int]] = None   # (D, H, W) extent of the data-gradient tensor

    @property
    def in_dims(self): return (self.Hi, self.Wi) if self.spec.dims == 2 else (self.Di, self.Hi, self.Wi)
    @property
    def out_dims(self): return (self.Ho, self.Wo) if self.spec.dims == 2 else (self.Do, self.Ho, self.Wo)
    @property
    def out_pixels(self): return self.Do * self.Ho * self.Wo
    @property
    def dgrad_dims(self):
        """extent of the data-gradient tensor (the padded domain when the pad adjoint is left to the consumer)"""
        if self.dgrad_dims3 is not None:
            return self.dgrad_dims3 if self.spec.dims == 3 else self.dgrad_dims3[1:]
        return tuple(x + 2 * self.dgrad_fold for x in self.in_dims)


def _pack_index(rows: int, taps_master: List[int], chan: int, master_idx) -> np.ndarray:
    """[rows][Kp] gather table; master_idx(row, t_master, c) -> flat master index."""
    K = len(taps_master) * chan
    Kp = roundup(K, 64)
    idx = np.full((rows, Kp), -1, dtype=np.int64)
    r = np.arange(rows)[:, None, None]
    t = np.asarray(taps_master)[None, :, None]
    c = np.arange(chan)[None, None, :]
    idx[:, :K] = master_idx(r, t, c).reshape(rows, K)
    return idx


def lower(spec: ConvSpec, *sizes) -> Lowered:
    """sizes = (H, W) for dims == 2, (D, H, W) for dims == 3. Every axis follows the same 1-D rule; a 2-D layer gets
    a dummy depth axis (extent 1, kernel 1, no padding)."""
    assert len(sizes) == spec.dims, f"expected {spec.dims} spatial sizes, got {sizes}"
    k, s, p = spec.k, spec.stride, spec.pad
    T, P, Q = spec.T, spec.P, spec.Q
    wf = spec.wfold
    assert not wf or (spec.kind == "conv" and s == 1), "W-fold applies to stride-1 convolutions"
    real = (spec.dims == 3, True, True)
    ins = ((1,) + tuple(sizes)) if spec.dims == 2 else tuple(sizes)
    ka = tuple(k if r else 1 for r in real)               # kernel extent per axis
    pa = tuple(p if r else 0 for r in real)               # padding per axis
    woff = 0                                              # constant W offset of every tap of a W-folded conv
    if wf:
        ka, pa = ka[:2] + (1,), pa[:2] + (0,)
        woff = -p if wf == "out" else 0
    outs = tuple(spec.out_size(x) if r else 1 for x, r in zip(ins, real))
    if wf == "in":
        outs = outs[:2] + (ins[2],)                       # the unfold already applied the W border
    elif wf == "out":
        outs = outs[:2] + (ins[2] + 2 * p,)               # all W + 2p shifted partial rows, reduced by the shift-add
    Di, Hi, Wi = ins
    Do, Ho, Wo = outs
    low = Lowered(spec, Hi, Wi, Ho, Wo, Di=Di, Do=Do)
    taps = [(a, b, c) for a in range(ka[0]) for b in range(ka[1]) for c in range(ka[2])]   # master tap order
    off = lambda f: ([f(t[0], 0) for t in taps], [f(t[1], 1) for t in taps], [f(t[2], 2) for t in taps])
    m_conv = lambda row, t, ch: (row * T + t) * Q + ch      # master[row][t][ch]   (row = P index)
    m_tr = lambda row, t, ch: (ch * T + t) * Q + row        # master[ch][t][row]   (transposed roles)

    def gconv(i3, ci, o3, co, c3, so, ph, si, offs, border, pack_off, rows):
        dd, dh, dw = offs
        return GConv(i3[1], i3[2], ci, o3[1], o3[2], co, c3[1], c3[2], so, ph[1], ph[2], si, dh, dw, border,
                     pack_off, rows, Di=i3[0], Do=o3[0], Dc=c3[0], pz=ph[0], dd=dd)

    def parity_classes(out3, rows, chan, midx, in3):
        """stride-s 'transposed' gather: out[s i + py] = sum_{r : (py + p - r) divisible by s} in[i + (py + p - r) / s] * W[r]
        (s = 2 everywhere in the reference nets but the stride-3 first conv of SelfAttentionPatchGAN3D,
        selfattention_patchgan3d.py:33-36)"""
        classes, tables, poff = [], [], 0
        axis_taps = lambda a, ph: [(r, (ph + pa[a] - r) // s) for r in range(ka[a]) if (ph + pa[a] - r) % s == 0]
        for pz in (range(s) if real[0] else range(1)):
            for py in range(s):
                for px in range(s):
                    ph = (pz, py, px)
                    at = [axis_taps(a, ph[a]) for a in range(3)]
                    c3 = tuple((out3[a] - ph[a] + s - 1) // s if real[a] else 1 for a in range(3))
                    if any(c <= 0 for c in c3):
                        continue
                    if any(not t for t in at):
                        # no tap reaches this class (k < s on some axis): its outputs are zero; never the case for k >= s
                        raise NotImplementedError("strided layer with a kernel smaller than its stride")
                    combos = [(a, b, c) for a in at[0] for b in at[1] for c in at[2]]
                    tm = [(a[0] * ka[1] + b[0]) * ka[2] + c[0] for a, b, c in combos]
                    offs = ([a[1] for a, _, _ in combos], [b[1] for _, b, _ in combos], [c[1] for _, _, c in combos])
                    g = gconv(in3, chan, out3, rows, c3, s, ph, 1, offs, "zero", poff, rows)
                    tab = _pack_index(rows, tm, chan, midx)
                    classes.append(g); tables.append(tab.reshape(-1)); poff += tab.size
        return classes, np.concatenate(tables)

    if spec.kind == "conv":
        assert s in (1, 2, 3), "stride 1, 2 or 3"
        # forward: out[i] = sum_r in[B(i*s + r - p)] W[r]
        fwd_offs = off(lambda r, a: r - pa[a] + (woff if a == 2 else 0))
        low.fwd = [gconv(ins, spec.cin_p, outs, spec.cout_p, outs, 1, (0, 0, 0), s, fwd_offs, spec.pad_mode, 0,
                         spec.cout_p)]
        low.fwd_index = _pack_index(spec.cout_p, list(range(T)), spec.cin_p, m_conv).reshape(-1)
        if s == 1:
            if spec.pad_mode == "zero":
                # dX[ih] = sum_r dY[ih + p - r] W[:, r]^T
                low.dgrad = [gconv(outs, spec.cout_p, ins, spec.cin_p, ins, 1, (0, 0, 0), 1,
                                   off(lambda r, a: pa[a] - r - (woff if a == 2 else 0)), "zero", 0, spec.cin_p)]
            else:
                # gradient on the padded domain; the pad adjoint ("fold") is applied by the consumer
                # (W-fold "in": W is not padded, the unfold adjoint handles it; "out": W is padded like the other axes)
                pad3 = tuple(x + 2 * q for x, q in zip(ins, pa[:2] + (p if wf == "out" else pa[2],)))
                low.dgrad_dims3 = pad3
                low.dgrad = [gconv(outs, spec.cout_p, pad3, spec.cin_p, pad3, 1, (0, 0, 0), 1,
                                   off(lambda r, a: -r), "zero", 0, spec.cin_p)]
                low.dgrad_fold = p
                if spec.pad_mode == "reflect" and spec.dims == 2 and k == 3 and p == 1 and not wf:
                    low.dgrad_ring = gconv(outs, spec.cout_p, ins, spec.cin_p, ins, 1, (0, 0, 0), 1,
                                           off(lambda r, a: pa[a] - r), "zero", 0, spec.cin_p)
            low.dgrad_index = _pack_index(spec.cin_p, list(range(T)), spec.cout_p, m_tr).reshape(-1)
        else:
            assert spec.pad_mode == "zero", "strided convs use zero padding in the reference nets"
            low.dgrad, low.dgrad_index = parity_classes(ins, spec.cin_p, spec.cout_p, m_tr, outs)
        dd, dh, dw = fwd_offs
        low.wgrad = WGrad(Ho, Wo, spec.cout_p, Hi, Wi, spec.cin_p, s, dh, dw, spec.pad_mode, Da=Do, Dg=Di, dd=dd)
        if not spec.wfold:
            low.wgrad.p_real = spec.cout
            for g in low.fwd:
                g.co_real = spec.cout
    else:
        assert s == 2 and spec.pad_mode == "zero", "ConvTranspose: stride 2, zero padding"
        # forward = parity classes; pack[co][t*cin+ci] = master[ci][t][co]
        low.fwd, low.fwd_index = parity_classes(outs, spec.cout_p, spec.cin_p, m_tr, ins)
        # dgrad: dX[i] = sum_r dY[2i - p + r] W[:, r]  -> strided gather, master used as is
        offs = off(lambda r, a: r - pa[a])
        low.dgrad = [gconv(outs, spec.cout_p, ins, spec.cin_p, ins, 1, (0, 0, 0), 2, offs, "zero", 0, spec.cin_p)]
        low.dgrad_index = _pack_index(spec.cin_p, list(range(T)), spec.cout_p, m_conv).reshape(-1)
        # wgrad: dense = X, gathered = dY at (2i - p + r)
        dd, dh, dw = offs
        low.wgrad = WGrad(Hi, Wi, spec.cin_p, Ho, Wo, spec.cout_p, 2, dh, dw, "zero", Da=Di, Dg=Do, dd=dd)
    low.fwd_index = low.fwd_index.astype(np.int32)
    low.dgrad_index = low.dgrad_index.astype(np.int32)
    return low
