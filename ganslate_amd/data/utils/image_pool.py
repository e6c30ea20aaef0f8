"""History buffer of generated images (ganslate/data/utils/image_pool.py:5-60): the first `pool_size` images pass
through and are stored; afterwards each image is swapped with a random stored one with probability 0.5
(Python `random`, rank-local — the reference's RNG stream is kept so seeded runs line up). Device tensors stay
on the device; only the coin flips live on the host."""
import random

import torch


class ImagePool:

    def __init__(self, pool_size):
        self.pool_size = pool_size
        if self.pool_size > 0:
            self.num_imgs = 0
            self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for image in images:
            image = torch.unsqueeze(image.detach(), 0)
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
                out.append(image)
            elif random.uniform(0, 1) > 0.5:
                idx = random.randint(0, self.pool_size - 1)
                out.append(self.images[idx].clone())
                self.images[idx] = image
            else:
                out.append(image)
        return torch.cat(out, 0)
