"""MONAI-free sliding-window inference (ganslate/utils/sliding_window_inferer.py:8-52) against the
loop-level restatement of MONAI's published algorithm (oracle/monai_ref.py) and against its size-independent properties."""
import pytest
import torch

from ganslate_amd.utils.sliding_window_inferer import SlidingWindowInferer, window_starts
from oracle import monai_ref


def _predictor(cin, cout, nd, seed=0):
    torch.manual_seed(seed)
    conv = (torch.nn.Conv3d if nd == 3 else torch.nn.Conv2d)(cin, cout, 3, padding=1)
    return lambda x: torch.tanh(conv(x)).detach()


@pytest.mark.parametrize("shape,roi,overlap,mode,sw", [
    ((1, 1, 20, 24, 28), (8, 16, 16), 0.25, "gaussian", 1),
    ((2, 2, 17, 19, 23), (8, 8, 8), 0.5, "gaussian", 3),
    ((1, 1, 12, 12, 12), (16, 8, 8), 0.25, "constant", 2),        # volume smaller than the window on one axis: cval pad
    ((2, 3, 40, 56), (16, 32), 0.25, "gaussian", 4),              # images
    ((1, 1, 16, 16, 16), (16, 16, 16), 0.25, "gaussian", 1),      # exactly one window
    ((1, 1, 30, 33, 35), (16, 16, 16), 0.0, "constant", 5),
])
def test_matches_loop_restatement(shape, roi, overlap, mode, sw):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(shape, generator=g) * 2 - 1
    pred = _predictor(shape[1], 2, len(shape) - 2)
    got = SlidingWindowInferer(roi, sw, overlap, mode, cval=-1)(x, pred)
    want = monai_ref.sliding_window_inference(x, list(roi), sw, pred, overlap, mode, -1.0)
    assert got.shape == want.shape == (shape[0], 2, *shape[2:])
    assert torch.allclose(got, want, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("mode", ["constant", "gaussian"])
@pytest.mark.parametrize("overlap", [0.0, 0.25, 0.6])
def test_identity_predictor_returns_the_input(mode, overlap):
    """a weighted average of identical values: out == in for every overlap / importance map; every voxel is covered"""
    x = torch.rand(1, 1, 37, 41, 29) * 2 - 1
    out = SlidingWindowInferer((16, 16, 16), 4, overlap, mode, cval=-1)(x, lambda w: w)
    assert torch.allclose(out, x, atol=1e-6)
    for s, r in ((37, 16), (41, 16), (29, 16)):
        st = [v[0] for v in window_starts([s], [r], overlap)]
        assert st[0] == 0 and st[-1] == s - r and all(b - a <= r for a, b in zip(st, st[1:]))


def test_two_d_model_over_a_volume():
    """roi [H, W] on [N, C, D, H, W]: broadcast to [1, H, W], the network sees [N, C, H, W] slices (network_wrapper)"""
    x = torch.rand(1, 2, 5, 24, 24)
    seen = []

    def net(w):
        seen.append(tuple(w.shape))
        return w * 2
    inf = SlidingWindowInferer((16, 16), 2, 0.25, "gaussian", cval=-1)
    out = inf(x, net)
    assert inf.roi_size == [1, 16, 16] and all(len(s) == 4 and s[1:] == (2, 16, 16) for s in seen)
    assert torch.allclose(out, 2 * x, atol=1e-6)
