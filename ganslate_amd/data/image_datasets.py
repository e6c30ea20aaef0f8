"""Config dataclasses (and thin PIL loaders) for the reference's image-folder datasets so that
projects/*/experiments/*.yaml load unchanged (ganslate/data/unpaired_image_dataset.py:19-62,
paired_image_dataset.py:20-60). Host-side image decoding is outside the hot path: the
loaders implement resize / scale_width / random_zoom / random_crop / random_flip with PIL + torch only and [-1,1] normalisation."""
import random
from dataclasses import dataclass, field
from pathlib import Path
from typing import Tuple

import torch
from torch.utils.data import Dataset

from .. import configs

EXTENSIONS = [".jpg", ".jpeg", ".png"]


@dataclass
class UnpairedImageDatasetConfig(configs.base.BaseDatasetConfig):
    image_channels: int = 3
    preprocess: Tuple[str] = ("resize", "random_crop", "random_flip")
    load_size: Tuple[int, int] = field(default_factory=lambda: [286, 286])
    final_size: Tuple[int, int] = field(default_factory=lambda: [256, 256])
    # not in the reference: workers only decode, resize / crop / flip / normalise run on the GPU (data/device_transforms.py)
    device_transforms: bool = False


@dataclass
class PairedImageDatasetConfig(configs.base.BaseDatasetConfig):
    image_channels: int = 3
    preprocess: Tuple[str] = ("resize", "random_crop", "random_flip")
    load_size: Tuple[int, int] = field(default_factory=lambda: [286, 286])
    final_size: Tuple[int, int] = field(default_factory=lambda: [256, 256])
    device_transforms: bool = False


def _files(root):
    root = Path(root)
    assert root.is_dir(), f"{root} is not a valid directory"
    return sorted(p for p in root.rglob("*") if p.suffix.lower() in EXTENSIONS)


class _Transform:
    def __init__(self, conf):
        d = conf[conf.mode].dataset
        self.pre, self.load, self.final = list(d.preprocess), tuple(d.load_size), tuple(d.final_size)
        try:
            self.raw = bool(d["device_transforms"])
        except (KeyError, AttributeError):
            self.raw = False

    def params(self):
        """the random draws of one sample (shared by both images of a pair, transforms.py:101-121): crop position as
        fractions, flip, and random_zoom's two levels in [0.8, 1) (transforms.py:101,129)"""
        return {"crop": (random.random(), random.random()), "flip": random.random() > 0.5,
                "zoom": (random.uniform(0.8, 1.0), random.uniform(0.8, 1.0))}

    def sizes(self, H, W, prm):
        """[(h, w), ...] of the resizes a decoded H x W image goes through before the crop: `resize` -> load_size, or
        `scale_width` (transforms.py:163-169: width -> load_w, height in proportion but at least final_w; untouched when the
        width already is load_w), then `random_zoom` (transforms.py:127-137: each side zoomed by its level, at least final)"""
        out = []
        if "resize" in self.pre:
            H, W = self.load
            out.append((H, W))
        elif "scale_width" in self.pre:
            load_w, final_w = self.load[1], self.final[1]
            if not (W == load_w and W >= final_w):
                H, W = int(max(load_w * H / W, final_w)), load_w
                out.append((H, W))
        if "random_zoom" in self.pre:
            zw = max(self.final[1], W * prm["zoom"][0])
            zh = max(self.final[0], H * prm["zoom"][1])
            H, W = int(round(zh)), int(round(zw))
            out.append((H, W))
        return out

    def decode_only(self, img, prm):
        """device_transforms: hand the decoded bytes and the drawn parameters to DeviceImagePipeline"""
        import numpy as np
        from .device_transforms import RawImage
        return RawImage(torch.from_numpy(np.array(img, dtype=np.uint8)), prm["crop"], prm["flip"], prm["zoom"])

    def __call__(self, img, prm):
        import numpy as np
        from PIL import Image
        if self.raw:
            return self.decode_only(img, prm)
        for h, w in self.sizes(img.size[1], img.size[0], prm):
            img = img.resize((w, h), Image.BICUBIC)
        if "random_crop" in self.pre:
            W, H = img.size
            top = int(prm["crop"][0] * max(H - self.final[0], 0))
            left = int(prm["crop"][1] * max(W - self.final[1], 0))
            img = img.crop((left, top, left + self.final[1], top + self.final[0]))
        if "random_flip" in self.pre and prm["flip"]:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        a = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0)
        a = a.unsqueeze(0) if a.ndim == 2 else a.permute(2, 0, 1)
        return (a - 0.5) / 0.5


class _DevicePipelineMixin:
    """build_loader picks these up: raw batches are lists (images differ in size), the Trainer runs the pipeline"""

    @property
    def collate_fn(self):
        if not self.transform.raw:
            return None
        from .device_transforms import collate_raw
        return collate_raw

    def device_pipeline(self, conf, device):
        if not self.transform.raw:
            return None
        from .device_transforms import DeviceImagePipeline
        return DeviceImagePipeline(conf, device)


class UnpairedImageDataset(_DevicePipelineMixin, Dataset):

    def __init__(self, conf):
        root = Path(conf[conf.mode].dataset.root)
        self.A_paths, self.B_paths = _files(root / "A"), _files(root / "B")
        self.transform = _Transform(conf)
        self.mode = "RGB" if conf[conf.mode].dataset.image_channels == 3 else "L"

    def __getitem__(self, index):
        from PIL import Image
        A = Image.open(self.A_paths[index % len(self.A_paths)]).convert(self.mode)
        B = Image.open(self.B_paths[random.randint(0, len(self.B_paths) - 1)]).convert(self.mode)
        return {"A": self.transform(A, self.transform.params()), "B": self.transform(B, self.transform.params())}

    def __len__(self):
        return max(len(self.A_paths), len(self.B_paths))


class PairedImageDataset(_DevicePipelineMixin, Dataset):

    def __init__(self, conf):
        root = Path(conf[conf.mode].dataset.root)
        self.A_paths, self.B_paths = _files(root / "A"), _files(root / "B")
        assert len(self.A_paths) == len(self.B_paths)
        self.transform = _Transform(conf)
        self.mode = "RGB" if conf[conf.mode].dataset.image_channels == 3 else "L"

    def __getitem__(self, index):
        from PIL import Image
        prm = self.transform.params()   # same random transform for both images of the pair
        A = Image.open(self.A_paths[index]).convert(self.mode)
        B = Image.open(self.B_paths[index]).convert(self.mode)
        return {"A": self.transform(A, prm), "B": self.transform(B, prm)}

    def __len__(self):
        return len(self.A_paths)
