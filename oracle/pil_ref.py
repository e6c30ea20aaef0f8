"""Restatement of Pillow's 8-bit bicubic resampler (src/libImaging/Resample.c: bicubic_filter, precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; release 12.2.0) and of the torchvision steps the
reference composes behind it (ganslate/data/utils/transforms.py:9-61) in numpy — TEST INFRASTRUCTURE ONLY. Pillow is
present in the image, so this file is pinned against PIL.Image.resize itself (tests/test_device_transforms_cpu.py); it
exists so that the HIP kernels (csrc/imgproc.hip) are checked pass by pass and not only end to end."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x):            # Resample.c bicubic_filter, a = -0.5
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs (in0 = 0, in1 = in_size) + normalize_coeffs_8bpc"""
    scale = in_size / out_size
    filterscale = scale if scale > 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        k = [(w / ww if ww != 0.0 else w) for w in k] + [0.0] * (ksize - xmax)
        kk.append([int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in k])
        bounds.append((xmin, xmax))
    return bounds, kk


def resample_pass(img, out_size, axis):
    """one 8-bit pass along `axis` of an (H, W, C) uint8 array"""
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, kk = precompute_coeffs(a.shape[0], out_size)
    out = np.empty((out_size,) + a.shape[1:], np.int64)
    for xx, (xmin, xmax) in enumerate(bounds):
        acc = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += a[xmin + x] * kk[xx][x]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def resize_bicubic(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), BICUBIC) of an (H, W, C) uint8 array: horizontal pass first, passes that keep the
    size are skipped (ImagingResample)"""
    if img.shape[1] != out_w:
        img = resample_pass(img, out_w, 1)
    if img.shape[0] != out_h:
        img = resample_pass(img, out_h, 0)
    return img


def scale_width(img, load_w, final_w):
    """__scale_width (transforms.py:163-169)"""
    img_h, img_w = img.shape[:2]
    if img_w == load_w and img_w >= final_w:
        return img
    return resize_bicubic(img, int(max(load_w * img_h / img_w, final_w)), load_w)


def random_zoom(img, final_size, zoom_level):
    """__random_zoom (transforms.py:127-137) with the two levels (width, height) passed in"""
    img_h, img_w = img.shape[:2]
    zoom_w = max(final_size[1], img_w * zoom_level[0])
    zoom_h = max(final_size[0], img_h * zoom_level[1])
    return resize_bicubic(img, int(round(zoom_h)), int(round(zoom_w)))


def single_image_transform(img, preprocess, load_size, final_size, crop, flip, zoom=(1.0, 1.0)):
    """get_single_image_transform (transforms.py:9-61) with the random draws passed in: crop = (u, v) in [0, 1) mapped to
    top = int(u * (H - fh)), left = int(v * (W - fw)) like ganslate_amd/data/image_datasets.py, flip = bool, zoom =
    random_zoom's levels. Returns fp32 (C, fh, fw) in [-1, 1]."""
    if "resize" in preprocess:
        img = resize_bicubic(img, load_size[0], load_size[1])
    elif "scale_width" in preprocess:
        img = scale_width(img, load_size[1], final_size[1])
    if "random_zoom" in preprocess:
        img = random_zoom(img, final_size, zoom)
    if "random_crop" in preprocess:
        H, W = img.shape[:2]
        top, left = int(crop[0] * max(H - final_size[0], 0)), int(crop[1] * max(W - final_size[1], 0))
        img = img[top:top + final_size[0], left:left + final_size[1]]
    if "random_flip" in preprocess and flip:
        img = img[:, ::-1]
    x = img.astype(np.float32) / np.float32(255.0)                # ToTensor
    x = (x - np.float32(0.5)) / np.float32(0.5)                   # Normalize(0.5, 0.5)
    return np.ascontiguousarray(np.moveaxis(x, -1, 0))
