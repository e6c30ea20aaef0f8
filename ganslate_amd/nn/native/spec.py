"""Lowering of nn.Conv2d / nn.ConvTranspose2d (forward, data-gradient, weight-gradient) onto the
"generalised convolution" the HIP kernels implement (include/ganslate_hip.h: gs_gconv_desc / gs_wgrad_desc).

Master weights live in the OTI layout [P][T][Q]:
  Conv2d          : P = out channels, Q = in channels   (torch OIHW  -> [O][kh*kw][I])
  ConvTranspose2d : P = in channels,  Q = out channels  (torch IOHW  -> [I][kh*kw][O])
with P and Q padded to multiples of 8 (padded entries are zero and receive zero gradients).
bf16 weight packs are [rows][Kp], K = (tap, channel) contiguous, Kp = roundup(K, 64); they are produced from
the master by a gather (`pack index`, -1 = zero).

Reference semantics restated here: ganslate/nn/generators/resnet/resnet2d.py:24-25,35,52-57,65,80-87 and
ganslate/nn/discriminators/patchgan/patchgan2d.py:29,36-62 (layer hyper-parameters), torch.nn.Conv2d /
ConvTranspose2d arithmetic.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class ConvSpec:
    """One convolution layer of the reference network."""
    kind: str                     # "conv" | "convT"
    cin: int
    cout: int
    k: int                        # square kernels only (all hot-path layers are square)
    stride: int = 1
    pad: int = 0
    out_pad: int = 0              # ConvTranspose2d output_padding
    pad_mode: str = "zero"        # "zero" | "reflect" | "replicate" (explicit pad layer folded into the conv)
    bias: bool = True

    @property
    def cin_p(self): return pad8(self.cin)
    @property
    def cout_p(self): return pad8(self.cout)
    @property
    def T(self): return self.k * self.k
    @property
    def P(self): return self.cout_p if self.kind == "conv" else self.cin_p
    @property
    def Q(self): return self.cin_p if self.kind == "conv" else self.cout_p
    @property
    def master_numel(self): return self.P * self.T * self.Q

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        if self.kind == "conv":
            f = lambda x: (x + 2 * self.pad - self.k) // self.stride + 1
        else:
            f = lambda x: (x - 1) * self.stride - 2 * self.pad + self.k + self.out_pad
        return f(H), f(W)

    # ---- torch <-> master layout ---------------------------------------------------------------
    def torch_weight_shape(self):
        return (self.cout, self.cin, self.k, self.k) if self.kind == "conv" else (self.cin, self.cout, self.k, self.k)

    def master_from_torch(self, w):
        """torch weight (OIHW for conv, IOHW for convT) -> padded OTI master (numpy or torch)."""
        import torch
        p, q = w.shape[0], w.shape[1]
        m = torch.zeros(self.P, self.T, self.Q, dtype=torch.float32, device=w.device)
        m[:p, :, :q] = w.reshape(p, q, self.T).permute(0, 2, 1)
        return m

    def torch_from_master(self, m):
        p, q = (self.cout, self.cin) if self.kind == "conv" else (self.cin, self.cout)
        return m.reshape(self.P, self.T, self.Q)[:p, :, :q].permute(0, 2, 1).reshape(p, q, self.k, self.k).contiguous()


@dataclass
class GConv:
    """One class of the generalised convolution (mirrors gs_gconv_desc, minus the batch/IO strides)."""
    Hi: int; Wi: int; Ci: int
    Ho: int; Wo: int; Co: int
    Hc: int; Wc: int
    so: int; py: int; px: int; si: int
    dh: List[int]; dw: List[int]
    border: str
    pack_offset: int = 0          # element offset of this class's [w_rows][Kp] block in the layer's pack
    w_rows: int = 0

    @property
    def T(self): return len(self.dh)
    @property
    def Kp(self): return roundup(self.T * self.Ci, 64)


@dataclass
class WGrad:
    Ha: int; Wa: int; P: int
    Hg: int; Wg: int; Q: int
    si: int
    dh: List[int]; dw: List[int]
    border: str

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
    wgrad: Optional[WGrad] = None


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


def lower(spec: ConvSpec, Hi: int, Wi: int) -> Lowered:
    k, s, p = spec.k, spec.stride, spec.pad
    T, P, Q = spec.T, spec.P, spec.Q
    Ho, Wo = spec.out_hw(Hi, Wi)
    low = Lowered(spec, Hi, Wi, Ho, Wo)
    taps = [(r, c) for r in range(k) for c in range(k)]
    m_conv = lambda row, t, ch: (row * T + t) * Q + ch      # master[row][t][ch]   (row = P index)
    m_tr = lambda row, t, ch: (ch * T + t) * Q + row        # master[ch][t][row]   (transposed roles)

    def parity_classes(Hout, Wout, rows, chan, midx, in_h, in_w):
        """stride-2 'transposed' gather: out[2i+py] = sum_{r:(py+p-r) even} in[i + (py+p-r)/2] * W[r]"""
        classes, tables, off = [], [], 0
        for py in range(2):
            for px in range(2):
                th = [(r, (py + p - r) // 2) for r in range(k) if (py + p - r) % 2 == 0]
                tw = [(c, (px + p - c) // 2) for c in range(k) if (px + p - c) % 2 == 0]
                Hc, Wc = (Hout - py + 1) // 2, (Wout - px + 1) // 2
                if not th or not tw or Hc <= 0 or Wc <= 0:
                    continue
                tm = [r * k + c for r, _ in th for c, _ in tw]
                g = GConv(in_h, in_w, chan, Hout, Wout, rows, Hc, Wc, 2, py, px, 1,
                          [d for _, d in th for _ in tw], [d for _ in th for _, d in tw], "zero", off, rows)
                tab = _pack_index(rows, tm, chan, midx)
                classes.append(g); tables.append(tab.reshape(-1)); off += tab.size
        return classes, np.concatenate(tables)

    if spec.kind == "conv":
        assert s in (1, 2), "stride 1 or 2"
        # forward: out[i] = sum_r in[B(i*s + r - p)] W[r]
        low.fwd = [GConv(Hi, Wi, spec.cin_p, Ho, Wo, spec.cout_p, Ho, Wo, 1, 0, 0, s,
                         [r - p for r, _ in taps], [c - p for _, c in taps], spec.pad_mode, 0, spec.cout_p)]
        low.fwd_index = _pack_index(spec.cout_p, list(range(T)), spec.cin_p, m_conv).reshape(-1)
        if s == 1:
            if spec.pad_mode == "zero":
                # dX[ih] = sum_r dY[ih + p - r] W[:, r]^T
                low.dgrad = [GConv(Ho, Wo, spec.cout_p, Hi, Wi, spec.cin_p, Hi, Wi, 1, 0, 0, 1,
                                   [p - r for r, _ in taps], [p - c for _, c in taps], "zero", 0, spec.cin_p)]
            else:
                # gradient on the padded domain; the pad adjoint ("fold") is applied by the consumer
                Hp, Wp = Hi + 2 * p, Wi + 2 * p
                low.dgrad = [GConv(Ho, Wo, spec.cout_p, Hp, Wp, spec.cin_p, Hp, Wp, 1, 0, 0, 1,
                                   [-r for r, _ in taps], [-c for _, c in taps], "zero", 0, spec.cin_p)]
                low.dgrad_fold = p
            low.dgrad_index = _pack_index(spec.cin_p, list(range(T)), spec.cout_p, m_tr).reshape(-1)
        else:
            assert spec.pad_mode == "zero", "strided convs use zero padding in the reference nets"
            low.dgrad, low.dgrad_index = parity_classes(Hi, Wi, spec.cin_p, spec.cout_p, m_tr, Ho, Wo)
        low.wgrad = WGrad(Ho, Wo, spec.cout_p, Hi, Wi, spec.cin_p, s,
                          [r - p for r, _ in taps], [c - p for _, c in taps], spec.pad_mode)
    else:
        assert s == 2 and spec.pad_mode == "zero", "ConvTranspose2d: stride 2, zero padding"
        # forward = parity classes; pack[co][t*cin+ci] = master[ci][t][co]
        low.fwd, low.fwd_index = parity_classes(Ho, Wo, spec.cout_p, spec.cin_p, m_tr, Hi, Wi)
        # dgrad: dX[i] = sum_r dY[2i - p + r] W[:, r]  -> strided gather, master used as is
        low.dgrad = [GConv(Ho, Wo, spec.cout_p, Hi, Wi, spec.cin_p, Hi, Wi, 1, 0, 0, 2,
                           [r - p for r, _ in taps], [c - p for _, c in taps], "zero", 0, spec.cin_p)]
        low.dgrad_index = _pack_index(spec.cin_p, list(range(T)), spec.cout_p, m_conv).reshape(-1)
        # wgrad: dense = X, gathered = dY at (2i - p + r)
        low.wgrad = WGrad(Hi, Wi, spec.cin_p, Ho, Wo, spec.cout_p, 2,
                          [r - p for r, _ in taps], [c - p for _, c in taps], "zero")
    low.fwd_index = low.fwd_index.astype(np.int32)
    low.dgrad_index = low.dgrad_index.astype(np.int32)
    return low
