from .image_datasets import (PairedImageDataset, PairedImageDatasetConfig, UnpairedImageDataset,  # noqa: F401
                             UnpairedImageDatasetConfig)
from .synthetic import SyntheticImageDataset, SyntheticImageDatasetConfig  # noqa: F401
from .volume_datasets import UnpairedVolumeDataset, UnpairedVolumeDatasetConfig  # noqa: F401
