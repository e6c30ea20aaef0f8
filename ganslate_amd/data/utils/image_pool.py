"""History buffer of generated images (ganslate/data/utils/image_pool.py:5-60): the first `pool_size` images pass
through and are stored; afterwards each image is swapped with a random stored one with probability 0.5
(Python `random`, rank-local — the reference's RNG stream is kept so seeded runs line up).

The pool is one device buffer and a query is ONE kernel launch (gs_pool_query) steered by a small code vector:
`draw()` takes the coin flips on the host exactly as the reference's loop does, `apply()` enqueues the launch. The
split lets a captured step (BaseGAN graph replay) keep the launch in the graph while the flips are drawn and uploaded
before each replay."""
import random

import torch

from ...nn.native.backend import get_ops

SWAP = 0x40000000


class ImagePool:

    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.external_draw = False      # True while a captured step owns the launch: the caller draws before replay
        if self.pool_size > 0:
            self.num_imgs = 0
            self.images = None          # [pool_size, *image shape] once the first query has shown the shape
            self._codes = {}            # batch size -> device code vector (kept alive: captured launches read it)
            self._pending = None

    def draw(self, batch):
        """host side of `query` for a batch of `batch` images: consumes Python's RNG like the reference loop
        (image_pool.py:44-58) and uploads the decisions"""
        if self.pool_size == 0:
            return
        code = []
        for _ in range(batch):
            if self.num_imgs < self.pool_size:
                code.append(self.num_imgs)
                self.num_imgs += 1
            elif random.uniform(0, 1) > 0.5:
                code.append(random.randint(0, self.pool_size - 1) | SWAP)
            else:
                code.append(-1)
        self._pending = torch.tensor(code, dtype=torch.int32)
        if batch in self._codes:
            dst = self._codes[batch]
            # pinned + non_blocking: a pageable upload would make the host wait for everything enqueued so far
            dst.copy_(self._pending.pin_memory() if dst.is_cuda else self._pending, non_blocking=True)

    def apply(self, images):
        images = images.detach().contiguous()
        B = images.shape[0]
        if self.images is None:
            self.images = torch.zeros((self.pool_size,) + tuple(images.shape[1:]), dtype=images.dtype,
                                      device=images.device)
        assert self._pending is not None and self._pending.numel() == B, "ImagePool.apply without a matching draw"
        if B not in self._codes:
            self._codes[B] = self._pending.to(images.device)
        out = torch.empty_like(images)
        get_ops().pool_query(self.images, images, out, self._codes[B])
        return out

    def query(self, images):
        if self.pool_size == 0:
            return images
        if not self.external_draw:
            self.draw(images.shape[0])
        return self.apply(images)
