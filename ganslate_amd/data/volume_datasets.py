"""Unpaired 3-D volume dataset with the reference's training-patch path (second half).

The reference keeps its 3-D datasets in the projects (e.g. projects/brats_mri_sequence_translation/datasets/
train_dataset.py:47-96): load two volumes (A by index, B at random — `random.randint`), draw a pair of spatially
corresponding patches with `StochasticFocalPatchSampler`, `z_score_normalize(patch, scale_to_range=(-1, 1))` each, add
the channel axis. Their file readers are SimpleITK, which this image does not have; `UnpairedVolumeDataset` is the same
`__getitem__` over `.npy` volumes in `<root>/A` and `<root>/B` (fp32 or int16, (D, H, W)), and
`BratsDatasetConfig`'s fields (`patch_size`, `focal_region_proportion`) keep their names.

With `device_transforms: true` the volumes live in HBM (uploaded on first use and kept — a few hundred 36-MB volumes are
nothing against 288 GB) and a sample is only ("A", index, start) coordinates: `DeviceVolumePipeline` crops and
normalises on the GPU (csrc/volproc.hip) in front of `set_input`, through the same Trainer hook as the image pipeline."""
import random
from dataclasses import dataclass, field
from pathlib import Path
from typing import Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import configs
from .utils.normalization import z_score_normalize
from .utils.stochastic_focal_patching import StochasticFocalPatchSampler

EXTENSIONS = [".npy"]


@dataclass
class UnpairedVolumeDatasetConfig(configs.base.BaseDatasetConfig):
    patch_size: Tuple[int, ...] = field(default_factory=lambda: [32, 32, 32])
    # proportion of the focal region's size to the volume's (stochastic_focal_patching.py:18-19)
    focal_region_proportion: float = 0
    # not in the reference: volumes resident in HBM, crop + normalisation on the GPU (data/device_volumes.py)
    device_transforms: bool = False


class RawPatch:
    """what a worker hands over with device_transforms on: which volume, and where the patch starts"""
    __slots__ = ("domain", "index", "start")

    def __init__(self, domain, index, start):
        self.domain, self.index, self.start = domain, int(index), tuple(int(v) for v in start)


def collate_raw(samples):
    return {k: [s[k] for s in samples] for k in samples[0]}


class UnpairedVolumeDataset(Dataset):

    def __init__(self, conf):
        d = conf[conf.mode].dataset
        root = Path(d.root)
        self.paths = {k: sorted(p for p in (root / k).rglob("*") if p.suffix.lower() in EXTENSIONS) for k in ("A", "B")}
        assert self.paths["A"] and self.paths["B"], f"no .npy volumes under {root}/A or {root}/B"
        self.patch_size = np.array([int(v) for v in d.patch_size])
        self.patch_sampler = StochasticFocalPatchSampler(self.patch_size, d.focal_region_proportion)
        try:
            self.raw = bool(d["device_transforms"])
        except (KeyError, AttributeError):
            self.raw = False
        self._shapes = {}

    def load(self, domain, index):
        """(D, H, W) tensor of one volume, in the dtype it is stored in (fp32 / int16 stay as they are)"""
        v = np.load(self.paths[domain][index], mmap_mode="r")
        assert v.ndim == 3, f"{self.paths[domain][index]}: expected a (D, H, W) array, got {v.shape}"
        return torch.from_numpy(np.ascontiguousarray(v))

    def shape(self, domain, index):
        key = (domain, index)
        if key not in self._shapes:
            self._shapes[key] = tuple(np.load(self.paths[domain][index], mmap_mode="r").shape)
        return self._shapes[key]

    def __getitem__(self, index):
        index_A = index % len(self.paths["A"])
        index_B = random.randint(0, len(self.paths["B"]) - 1)
        if self.raw:
            start_A, start_B = self.patch_sampler.get_start_pair(self.shape("A", index_A), self.shape("B", index_B))
            return {"A": RawPatch("A", index_A, start_A), "B": RawPatch("B", index_B, start_B)}
        A, B = self.load("A", index_A).float(), self.load("B", index_B).float()
        A, B = self.patch_sampler.get_patch_pair(A, B)
        A = z_score_normalize(A, scale_to_range=(-1, 1))
        B = z_score_normalize(B, scale_to_range=(-1, 1))
        return {"A": A.unsqueeze(0), "B": B.unsqueeze(0)}

    def __len__(self):
        return max(len(self.paths["A"]), len(self.paths["B"]))

    # ---- picked up by build_loader / Trainer ----------------------------------------------------------------------
    @property
    def collate_fn(self):
        return collate_raw if self.raw else None

    def device_pipeline(self, conf, device):
        if not self.raw:
            return None
        from .device_volumes import DeviceVolumePipeline
        return DeviceVolumePipeline(self, device)
