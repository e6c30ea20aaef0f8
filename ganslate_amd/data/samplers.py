"""Infinite rank-strided index stream (ganslate/data/samplers.py:20-58): every rank walks the same seed-shared
permutation and takes indices rank, rank+world, ... — the data-parallel sharding of the training step."""
import itertools

import torch
from torch.utils.data.sampler import Sampler

from ..utils import communication


class InfiniteSampler(Sampler):

    def __init__(self, size: int, shuffle: bool = True):
        assert size > 0
        self._size, self._shuffle = size, shuffle
        self._seed = communication.shared_random_seed()
        self._rank = communication.get_rank()
        self._world_size = communication.get_world_size()

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g)
            else:
                yield from torch.arange(self._size)
