"""Loop-level restatement of MONAI's sliding_window_inference (monai/inferers/utils.py, 0.5-0.8) — TEST INFRASTRUCTURE.
One window at a time, every index written out; the product's inferer (ganslate_amd/utils/sliding_window_inferer.py)
is compared with it. MONAI is absent from the container and unpinned in the reference: parity with MONAI itself is
unpinned; this file pins the published algorithm the product claims to follow."""
import math

import torch


def gaussian_map(roi):
    maps = []
    for n in roi:
        sigma = 0.125 * n
        tail = int(max(sigma * 4.0, 0.5) + 0.5)
        t = 0.70710678 / sigma
        k = [0.5 * (math.erf(t * (x + 0.5)) - math.erf(t * (x - 0.5))) for x in range(-tail, tail + 1)]
        ksum = sum(k)
        k = [v / ksum for v in k]
        axis = [0.0] * n
        for i in range(n):           # delta at n // 2 convolved with k (zero padding)
            j = i - n // 2 + tail
            if 0 <= j < len(k):
                axis[i] = k[j]
        maps.append(torch.tensor(axis, dtype=torch.float32))
    m = maps[0]
    for a in maps[1:]:
        m = m[..., None] * a
    m = m / m.max()
    return m.clamp(min=max(m[m != 0].min().item(), 1e-3))


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap, mode, cval):
    nd = inputs.dim() - 2
    size0 = list(inputs.shape[2:])
    roi = [r if r > 0 else s for r, s in zip(roi_size, size0)]
    pad = []
    for k in range(nd - 1, -1, -1):
        diff = max(roi[k] - size0[k], 0)
        pad.extend([diff // 2, diff - diff // 2])
    x = torch.nn.functional.pad(inputs, pad, value=cval) if any(pad) else inputs
    size = list(x.shape[2:])
    per_axis = []
    for s, r in zip(size, roi):
        interval = r if r == s else max(int(r * (1 - overlap)), 1)
        n = int(math.ceil((s - r) / interval)) + 1
        per_axis.append([min(k * interval, s - r) for k in range(n)])
    starts = [[]]
    for ax in per_axis:
        starts = [p + [v] for p in starts for v in ax]
    imap = torch.ones(roi) if mode == "constant" else gaussian_map(roi)
    out = cnt = None
    for st in starts:
        for b in range(x.shape[0]):
            sl = (slice(b, b + 1), slice(None)) + tuple(slice(s, s + r) for s, r in zip(st, roi))
            pred = predictor(x[sl])
            if out is None:
                out = torch.zeros((x.shape[0], pred.shape[1], *size))
                cnt = torch.zeros_like(out)
            out[sl] += imap * pred
            cnt[sl] += imap
    out = out / cnt
    crop = [slice(None), slice(None)]
    for k in range(nd):
        before = pad[2 * (nd - 1 - k)] if pad else 0
        crop.append(slice(before, before + size0[k]))
    return out[tuple(crop)]
