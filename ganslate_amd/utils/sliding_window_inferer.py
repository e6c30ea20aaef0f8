"""Patch-wise (sliding-window) inference — ganslate/utils/sliding_window_inferer.py:8-52 without MONAI.

The reference subclasses `monai.inferers.SlidingWindowInferer` (third-party, unpinned, absent here); what that class
does is restated from MONAI's published algorithm (monai/inferers/utils.py `sliding_window_inference`,
monai/data/utils.py `dense_patch_slices` / `compute_importance_map`, MONAI 0.5-0.8):

  1. an image smaller than the window is padded symmetrically (half before, rest after) with `cval`;
  2. window starts per axis: interval = roi if roi == size else max(int(roi * (1 - overlap)), 1);
     count = ceil((size - roi) / interval) + 1; start_k = min(k * interval, size - roi); windows in row-major order;
  3. importance map: ones ("constant") or a Gaussian ("gaussian": separable erf-integrated kernel with
     sigma = 0.125 * roi, centred at roi // 2, normalised to max 1, floored at max(smallest non-zero value, 1e-3));
  4. windows are run `sw_batch_size` at a time; out[window] += map * prediction, count[window] += map;
  5. result = out / count, with the padding of step 1 cropped away.

The ganslate layer on top (kept verbatim in behaviour): a 2-D `roi_size` on a volume is broadcast to [1, H, W] and the
network then sees [N, C, H, W] slices (`network_wrapper`, sliding_window_inferer.py:36-52). The predictor is the HIP
generator's no-grad forward; windows of one batch go through it as ONE launch sequence (batch = sw_batch_size)."""
import math

import torch


def _gaussian_1d(sigma: float) -> torch.Tensor:
    """monai.networks.layers.gaussian_1d(sigma, truncated=4.0, approx='erf'), normalised"""
    tail = int(max(float(sigma) * 4.0, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float32)
    t = 0.70710678 / abs(float(sigma))
    out = 0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())
    out = out.clamp(min=0)
    return out / out.sum()


def importance_map(roi, mode, device):
    if mode == "constant":
        return torch.ones(tuple(roi), dtype=torch.float32, device=device)
    if mode != "gaussian":
        raise ValueError(f"sliding window mode `{mode}` (expected `constant` or `gaussian`)")
    m = None
    for n in roi:
        # a delta at n // 2 filtered with the (zero-padded) kernel = the kernel centred there, cropped to the axis
        k = _gaussian_1d(0.125 * n)
        tail = (k.numel() - 1) // 2
        axis = torch.zeros(n)
        for i in range(n):
            j = i - n // 2 + tail
            if 0 <= j < k.numel():
                axis[i] = k[j]
        m = axis if m is None else m[..., None] * axis
    m = m / m.max()
    floor = max(m[m != 0].min().item(), 1e-3)
    return m.clamp(min=floor).to(device)


def window_starts(size, roi, overlap):
    starts = []
    for s, r in zip(size, roi):
        interval = r if r == s else max(int(r * (1 - overlap)), 1)
        num = int(math.ceil(float(s - r) / interval)) + 1 if interval > 0 else 1
        starts.append([min(k * interval, s - r) for k in range(num)])
    out = [[]]
    for axis in starts:                      # row-major product
        out = [p + [v] for p in out for v in axis]
    return out


class SlidingWindowInferer:

    def __init__(self, roi_size, sw_batch_size=1, overlap=0.25, mode="constant", cval=0.0):
        self.roi_size = list(roi_size)
        self.sw_batch_size, self.overlap, self.mode, self.cval = int(sw_batch_size), float(overlap), mode, float(cval)
        if not 0 <= self.overlap < 1:
            raise ValueError("overlap must be >= 0 and < 1.")

    def __call__(self, inputs, network, *args, **kwargs):
        if len(self.roi_size) != len(inputs.shape[2:]):
            if len(self.roi_size) == 2:          # 2-D model over a volume: slice-wise windows
                self.roi_size = [1, *self.roi_size]
            else:
                raise RuntimeError("Unsupported roi size, cannot broadcast to volume. ")
        return self._infer(inputs, lambda x: self.network_wrapper(network, x, *args, **kwargs))

    def network_wrapper(self, network, x, *args, **kwargs):
        if len(self.roi_size) == 3 and self.roi_size[0] == 1:
            return network(x.squeeze(dim=2), *args, **kwargs).unsqueeze(dim=2)
        return network(x, *args, **kwargs)

    def _infer(self, inputs, predictor):
        nd = inputs.dim() - 2
        size0 = list(inputs.shape[2:])
        roi = [r if r > 0 else s for r, s in zip(self.roi_size, size0)]          # fall_back_tuple
        pad = []
        for k in range(nd - 1, -1, -1):                                          # F.pad order: last axis first
            diff = max(roi[k] - size0[k], 0)
            pad.extend([diff // 2, diff - diff // 2])
        if any(pad):
            inputs = torch.nn.functional.pad(inputs, pad, mode="constant", value=self.cval)
        size = list(inputs.shape[2:])
        starts = window_starts(size, roi, self.overlap)
        imap = importance_map(roi, self.mode, inputs.device)
        B = inputs.shape[0]
        windows = [(b, st) for st in starts for b in range(B)]                   # MONAI: slice_g -> (window, batch item)
        out = count = None
        for g0 in range(0, len(windows), self.sw_batch_size):
            chunk = windows[g0:g0 + self.sw_batch_size]
            sl = [(slice(b, b + 1), slice(None)) + tuple(slice(s, s + r) for s, r in zip(st, roi)) for b, st in chunk]
            pred = predictor(torch.cat([inputs[s] for s in sl], dim=0)).float()
            if out is None:
                out = torch.zeros((B, pred.shape[1], *size), dtype=torch.float32, device=inputs.device)
                count = torch.zeros((B, pred.shape[1], *size), dtype=torch.float32, device=inputs.device)
            for i, s in enumerate(sl):
                out[s] += imap * pred[i:i + 1]
                count[s] += imap
        out = out / count
        crop = [slice(None), slice(None)]
        for k in range(nd):
            before = pad[2 * (nd - 1 - k)] if pad else 0
            crop.append(slice(before, before + size0[k]))
        return out[tuple(crop)]
