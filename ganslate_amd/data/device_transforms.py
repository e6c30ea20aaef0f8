"""Device-side input pipeline for the image-folder datasets.

The reference builds its per-image transform from torchvision / PIL on the host — Resize(load_size, BICUBIC),
RandomCrop(final_size), RandomHorizontalFlip, ToTensor, Normalize(0.5, 0.5)
(ganslate/data/utils/transforms.py:9-61, 64-125) — with 16 loader workers in the horse2zebra yaml; at the step rates of
this backend that is the bottleneck. With `train.dataset.device_transforms: true` the dataset workers only DECODE (PIL,
8-bit HWC) and draw the random parameters; the pixels are uploaded as bytes and `DeviceImagePipeline` runs the whole
transform in two kernels per image (csrc/imgproc.hip), writing the fp32 NCHW batch the recipes' `set_input` expects.

The resize reproduces Pillow's resampler bit for bit: `resample_tables` restates `precompute_coeffs` and
`normalize_coeffs_8bpc` of Pillow's src/libImaging/Resample.c (bicubic filter a = -0.5, support 2 x max(scale, 1),
22-bit fixed point) in double precision, operation for operation; tests pin the tables' effect against PIL.Image.resize.
"""
import math
import random
from functools import lru_cache

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=256)
def resample_tables(in_size: int, out_size: int):
    """(bounds int32 [out, 2], kk int32 [out, ksize]) of one axis; identity tables when the size does not change (Pillow
    skips the pass then: 2^21 + v * 2^22 >> 22 == v)"""
    if in_size == out_size:
        bounds = np.stack([np.arange(out_size), np.ones(out_size, np.int64)], 1).astype(np.int32)
        return bounds, np.full((out_size, 1), 1 << PRECISION_BITS, np.int32)
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x, w in enumerate(k):
            if ww != 0.0:
                w = w / ww
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


class RawImage:
    """what a dataset worker hands over with device_transforms on: decoded pixels and the drawn random parameters"""
    __slots__ = ("pixels", "crop", "flip", "zoom")

    def __init__(self, pixels, crop, flip, zoom=(1.0, 1.0)):
        # uint8 (H, W, C) tensor, (u, v) in [0, 1), bool, random_zoom's (width, height) levels
        self.pixels, self.crop, self.flip, self.zoom = pixels, crop, flip, zoom


def draw_params():
    return (random.random(), random.random()), random.random() > 0.5


def collate_raw(samples):
    """DataLoader collate_fn: images of a batch may have different sizes, so they stay a list per key"""
    return {k: [s[k] for s in samples] for k in samples[0]}


class DeviceImagePipeline:
    """callable(raw batch) -> {"A": fp32 [N, C, fh, fw] on the device, "B": ...}"""

    def __init__(self, conf, device, ops=None):
        d = conf[conf.mode].dataset
        self.pre, self.load, self.final = list(d.preprocess), tuple(d.load_size), tuple(d.final_size)
        unknown = set(self.pre) - {"resize", "scale_width", "random_zoom", "random_crop", "random_flip"}
        if unknown:
            raise NotImplementedError(f"device_transforms: preprocess steps {sorted(unknown)} have no device path")
        self.device = torch.device(device)
        self._ops = ops
        self._tables = {}

    @property
    def ops(self):
        if self._ops is None:
            from ..nn.native.backend import get_ops
            self._ops = get_ops()
        return self._ops

    def _dev_tables(self, in_size, out_size):
        key = (in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            b, k = resample_tables(in_size, out_size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).contiguous().to(self.device))
            self._tables[key] = t
        return t

    def sizes(self, H, W, zoom):
        """the resizes before the crop — image_datasets._Transform.sizes, the same integer arithmetic"""
        from .image_datasets import _Transform
        t = _Transform.__new__(_Transform)
        t.pre, t.load, t.final = self.pre, self.load, self.final
        return t.sizes(H, W, {"zoom": zoom})

    def geometry(self, H, W, crop, zoom=(1.0, 1.0)):
        """(resized H, resized W, top, left, fh, fw) of one image — the same integer arithmetic as the host transform of
        ganslate_amd/data/image_datasets.py"""
        rh, rw = (self.sizes(H, W, zoom) or [(H, W)])[-1]
        if "random_crop" in self.pre:
            fh, fw = self.final
            top, left = int(crop[0] * max(rh - fh, 0)), int(crop[1] * max(rw - fw, 0))
        else:
            fh, fw, top, left = rh, rw, 0, 0
        return rh, rw, top, left, min(fh, rh), min(fw, rw)

    def one(self, raw: RawImage, out):
        px = raw.pixels
        if px.ndim == 2:
            px = px.unsqueeze(-1)
        px = px.contiguous()
        if not px.is_cuda:
            px = px.pin_memory().to(self.device, non_blocking=True) if self.device.type == "cuda" else px
        H, W, C = px.shape
        rh, rw, top, left, fh, fw = self.geometry(H, W, raw.crop, raw.zoom)
        steps = self.sizes(H, W, raw.zoom) or [(H, W)]
        for h1, w1 in steps[:-1]:          # a complete resize with an 8-bit result in front of the last one (Pillow rounds
            bh, kh = self._dev_tables(W, w1)          # to bytes between two Image.resize calls)
            bv, kv = self._dev_tables(H, h1)
            tmp = torch.empty((H, w1, C), dtype=torch.uint8, device=px.device)
            self.ops.u8_resample_h(px, tmp, bh, kh)
            px = torch.empty((h1, w1, C), dtype=torch.uint8, device=px.device)
            self.ops.u8_resample_v(tmp, px, bv, kv)
            H, W = h1, w1
        bh, kh = self._dev_tables(W, rw)
        bv, kv = self._dev_tables(H, rh)
        tmp = torch.empty((H, rw, C), dtype=torch.uint8, device=px.device)
        self.ops.u8_resample_h(px, tmp, bh, kh)
        flip = "random_flip" in self.pre and raw.flip
        self.ops.u8_resample_v_crop_normalize(tmp, out, rh, bv, kv, top, left, flip)

    def __call__(self, batch):
        out = {}
        for key, items in batch.items():
            if not items or not isinstance(items[0], RawImage):
                out[key] = items
                continue
            r0 = items[0]
            C = 1 if r0.pixels.ndim == 2 else r0.pixels.shape[-1]
            shapes = {self.geometry(r.pixels.shape[0], r.pixels.shape[1], r.crop, r.zoom)[4:] for r in items}
            assert len(shapes) == 1, f"images of one batch end at different sizes {shapes}: add random_crop or resize"
            fh, fw = next(iter(shapes))
            dst = torch.empty((len(items), C, fh, fw), dtype=torch.float32, device=self.device)
            for n, r in enumerate(items):
                self.one(r, dst[n])
            out[key] = dst
        return out
