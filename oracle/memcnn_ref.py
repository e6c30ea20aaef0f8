"""Restatement of the two memcnn classes the reference uses (ganslate/nn/invertible.py:15-19,23-24) — TEST
INFRASTRUCTURE ONLY, like everything under oracle/.

memcnn (github.com/silvandeleemput/memcnn) is a third-party dependency of the reference, unpinned (`memcnn` in
setup.cfg:31) and absent from this container and from /root/reference; there is no network. What is restated here is
its published algorithm and API as of release 1.5.1 (PyPI, the release current when ganslate v1 was tagged):

  * `memcnn.AdditiveCoupling(Fm, Gm=None, implementation_fwd=-1, implementation_bwd=-1, split_dim=1)`
    (memcnn/models/additive.py, class AdditiveCoupling): `Gm = copy.deepcopy(Fm)` when not given; attributes `Fm`, `Gm`;
    forward with the default implementation -1 (plain autograd):
        x1, x2 = torch.chunk(x, 2, dim=split_dim);  y1 = x1 + Fm(x2);  y2 = x2 + Gm(y1);  out = cat([y1, y2], split_dim)
    inverse:
        y1, y2 = torch.chunk(y, 2, dim=split_dim);  x2 = y2 - Gm(y1);  x1 = y1 - Fm(x2);  out = cat([x1, x2], split_dim)
    — the additive coupling of RevNet (Gomez et al. 2017, eq. 6-8), which memcnn's documentation cites.
  * `memcnn.InvertibleModuleWrapper(fn, keep_input=False, keep_input_inverse=False, num_bwd_passes=1, disable=False,
    preserve_rng_state=False)` (memcnn/models/revop.py): keeps the wrapped module as `self._fn`; with `disable=True`
    `forward(x)` is `self._fn(x)` and `inverse(y)` is `self._fn.inverse(y)` under plain autograd (no activation
    freeing, no recomputation). The reference's Vnet3D passes `disable = not use_memory_saving` and every shipped config
    sets `use_memory_saving: False` (projects/brats_mri_sequence_translation/experiments/*.yaml).

Line numbers of memcnn cannot be cited: the source is not available offline. Parity for the memcnn-specific part is
therefore UNPINNED against memcnn itself; what IS pinned: these equations against the stand-in the
golden vectors were generated with (oracle/ref_stubs/memcnn, tests/test_memcnn_semantics_cpu.py), the reference's own
`InvertibleBlock` / `Vnet3D` code running over it (tests/golden/volumes.json), and the state_dict key names the
reference's checkpoints would carry (`...invertible_block._fn.Fm.N.weight`, `..._fn.Gm.N.weight`)."""
import copy

import torch
from torch import nn


class AdditiveCoupling(nn.Module):
    def __init__(self, Fm, Gm=None, implementation_fwd=-1, implementation_bwd=-1, split_dim=1):
        super().__init__()
        if Gm is None:
            Gm = copy.deepcopy(Fm)       # independent parameters, equal values at construction
        self.Gm, self.Fm = Gm, Fm
        self.implementation_fwd, self.implementation_bwd, self.split_dim = implementation_fwd, implementation_bwd, split_dim

    def forward(self, x):
        x1, x2 = torch.chunk(x, 2, dim=self.split_dim)
        x1, x2 = x1.contiguous(), x2.contiguous()
        y1 = x1 + self.Fm(x2)
        y2 = x2 + self.Gm(y1)
        return torch.cat([y1, y2], dim=self.split_dim)

    def inverse(self, y):
        y1, y2 = torch.chunk(y, 2, dim=self.split_dim)
        y1, y2 = y1.contiguous(), y2.contiguous()
        x2 = y2 - self.Gm(y1)
        x1 = y1 - self.Fm(x2)
        return torch.cat([x1, x2], dim=self.split_dim)


class InvertibleModuleWrapper(nn.Module):
    def __init__(self, fn, keep_input=False, keep_input_inverse=False, num_bwd_passes=1, disable=False,
                 preserve_rng_state=False):
        super().__init__()
        if not disable:
            raise NotImplementedError("only the plain-autograd path (disable=True) is restated: every shipped config "
                                      "runs Vnet3D with use_memory_saving False")
        self.disable, self.keep_input, self.keep_input_inverse = disable, keep_input, keep_input_inverse
        self._fn = fn

    def forward(self, *xin):
        return self._fn(*xin)

    def inverse(self, *yin):
        return self._fn.inverse(*yin)
