"""Stand-in for `memcnn` (oracle-only): additive coupling with plain autograd.
memcnn itself is absent from the container and unpinned in the reference (setup.cfg:31), so
Vnet3D parity is 'unpinned' for the memcnn-specific parts."""
import copy
import torch
from torch import nn


class AdditiveCoupling(nn.Module):
    def __init__(self, Fm, Gm=None, split_dim=1):
        super().__init__()
        self.Fm = Fm
        self.Gm = copy.deepcopy(Fm) if Gm is None else Gm
        self.split_dim = split_dim

    def forward(self, x):
        x1, x2 = torch.chunk(x, 2, dim=self.split_dim)
        y1 = x1 + self.Fm(x2)
        y2 = x2 + self.Gm(y1)
        return torch.cat([y1, y2], dim=self.split_dim)

    def inverse(self, y):
        y1, y2 = torch.chunk(y, 2, dim=self.split_dim)
        x2 = y2 - self.Gm(y1)
        x1 = y1 - self.Fm(x2)
        return torch.cat([x1, x2], dim=self.split_dim)


class InvertibleModuleWrapper(nn.Module):
    """keep_input / keep_input_inverse / disable only steer what memcnn frees and recomputes between forward and backward;
    the values and gradients are those of the plain calls below"""

    def __init__(self, fn, keep_input=False, keep_input_inverse=False, disable=False, **kw):
        super().__init__()
        self._fn = fn
        self.keep_input, self.keep_input_inverse, self.disable = keep_input, keep_input_inverse, disable

    def forward(self, x):
        return self._fn(x)

    def inverse(self, y):
        return self._fn.inverse(y)
