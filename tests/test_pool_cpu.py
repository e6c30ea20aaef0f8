"""ImagePool (device buffer + code vector, one launch per query) against the oracle's restatement of the reference's
list-based pool (ganslate/data/utils/image_pool.py:31-60) under the same Python RNG stream, on the op-level oracle
backend: fill phase, swaps, pass-throughs and two images of a batch drawing the same slot."""
import random

import pytest
import torch

from oracle import torch_ref
from oracle.ops_ref import RefOps


@pytest.fixture()
def ref_backend():
    from ganslate_amd.nn.native import backend
    old = backend._ops
    backend.set_ops(RefOps())
    yield
    backend.set_ops(old)


@pytest.mark.parametrize("pool_size,batch", [(4, 3), (2, 4), (50, 2), (0, 2), (1, 3)])
def test_pool_matches_reference_sequence(ref_backend, pool_size, batch):
    from ganslate_amd.data.utils.image_pool import ImagePool
    g = torch.Generator().manual_seed(pool_size * 10 + batch)
    queries = [torch.rand((batch, 2, 4, 4), generator=g) for _ in range(40)]
    random.seed(5)
    ref = torch_ref.ImagePool(pool_size)
    want = [ref.query(q) for q in queries]
    random.seed(5)
    pool = ImagePool(pool_size)
    got = [pool.query(q) for q in queries]
    for s, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"query {s}"


def test_pool_external_draw_equals_query(ref_backend):
    """draw() before apply() (what a captured step does) is the same as query()"""
    from ganslate_amd.data.utils.image_pool import ImagePool
    g = torch.Generator().manual_seed(3)
    queries = [torch.rand((3, 1, 4, 4), generator=g) for _ in range(20)]
    random.seed(9)
    a = ImagePool(3)
    want = [a.query(q) for q in queries]
    random.seed(9)
    b = ImagePool(3)
    b.external_draw = True
    got = []
    for q in queries:
        b.draw(q.shape[0])
        got.append(b.query(q))
    for x, y in zip(got, want):
        assert torch.equal(x, y)
