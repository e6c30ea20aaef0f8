"""Stochastic focal patch sampling for pairs of unregistered volumes — the reference's
ganslate/data/utils/stochastic_focal_patching.py:5-119 (used by the 3-D project datasets, e.g.
projects/brats_mri_sequence_translation/datasets/train_dataset.py:54-56,83), same class name, constructor and
`get_patch_pair`, and the same sequence of `random.randint` draws, so a seeded run picks the same patches
(tests/golden/volume_patches.json holds the reference's draws).

A patch start is drawn uniformly in volume A; its position relative to A's extent is mapped into volume B, a window of
`focal_region_proportion` x B's extent is centred there, clipped to the valid starts, and B's start is drawn inside it.
`get_start_pair` exposes the draws without touching voxel data: the device-side patch pipeline
(data/device_volumes.py) crops resident volumes on the GPU from these coordinates."""
import random

import numpy as np


class StochasticFocalPatchSampler:

    def __init__(self, patch_size, focal_region_proportion):
        self.focal_region_proportion = focal_region_proportion
        self.dims = len(patch_size)
        # a 2-D patch size means single-slice patches: depth 1, squeezed away again on return
        self.patch_size = np.array([1, *patch_size] if self.dims == 2 else list(patch_size))

    # ---- coordinates only -----------------------------------------------------------------------------------------
    def _valid_starts(self, size):
        room = size - self.patch_size
        if np.any(room < 0):
            raise RuntimeError(f"The volume, {size} provided to the sampler is smaller than the patch size: "
                               f"{self.patch_size}")
        return room

    def get_start_pair(self, shape_A, shape_B):
        """((z, x, y) in A, (z, x, y) in B) for volumes whose last three extents are shape_A / shape_B.
        Draw order (stochastic_focal_patching.py:58-62,76-96): one randint per axis of A, then one per axis of B — except on
        an axis of B whose clipped window is empty, which takes the window's upper end without a draw."""
        size_A, size_B = np.array(tuple(shape_A)[-3:]), np.array(tuple(shape_B)[-3:])
        start_A = [random.randint(0, int(v)) for v in self._valid_starts(size_A)]
        relative = np.array(start_A) / size_A
        window = (self.focal_region_proportion * size_B).astype(np.int64)
        centre = relative * size_B
        room_B = self._valid_starts(size_B)
        start_B = []
        for ax in range(3):
            lo = max(0, int(centre[ax] - window[ax] / 2))
            hi = min(int(centre[ax] + window[ax] / 2), int(room_B[ax]))
            start_B.append(hi if lo > hi else random.randint(lo, hi))
        return tuple(int(v) for v in start_A), tuple(int(v) for v in start_B)

    # ---- the reference's entry point -----------------------------------------------------------------------------
    def crop(self, volume, start):
        z, x, y = start
        d, h, w = (int(v) for v in self.patch_size)
        patch = volume[..., z:z + d, x:x + h, y:y + w]
        return patch.squeeze(-3) if self.dims == 2 else patch

    def get_patch_pair(self, volume_A, volume_B):
        start_A, start_B = self.get_start_pair(volume_A.shape, volume_B.shape)
        return self.crop(volume_A, start_A), self.crop(volume_B, start_B)
