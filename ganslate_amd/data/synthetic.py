"""Synthetic unpaired/paired image dataset: U(-1,1) tensors of the configured shape (the value range
`Normalize(0.5, 0.5)` gives real images, data/utils/transforms.py:54-57). Used by bench.py and the tests —
no dataset can be downloaded on the target machines."""
from dataclasses import dataclass, field
from typing import Tuple

import torch
from torch.utils.data import Dataset

from .. import configs


@dataclass
class SyntheticImageDatasetConfig(configs.base.BaseDatasetConfig):
    root: str = ""
    num_workers: int = 0
    image_channels: int = 3
    final_size: Tuple[int, ...] = field(default_factory=lambda: [256, 256])
    length: int = 1024
    seed: int = 1234


class SyntheticImageDataset(Dataset):

    def __init__(self, conf):
        d = conf[conf.mode].dataset
        self.shape = (d.image_channels, *(int(v) for v in d.final_size))   # (H, W) images or (D, H, W) volumes
        self.length, self.seed = d.length, d.seed

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed + int(index))
        return {"A": torch.rand(self.shape, generator=g) * 2 - 1, "B": torch.rand(self.shape, generator=g) * 2 - 1}

    def __len__(self):
        return self.length
