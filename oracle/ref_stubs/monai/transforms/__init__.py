"""Stand-in for `monai.transforms` (oracle-only; monai is not installed here and has no place on the hot path).

RandSpatialCrop(roi_size, random_center=True, random_size=False) as the reference's MultiScalePatchGAN3D uses it
(ganslate/nn/discriminators/patchgan/multiscale_patchgan3d.py:25-29): called on a (B, C, D, H, W) tensor with
roi_size = (C, D/s, H/s, W/s), i.e. the first axis is taken as channels and a window of roi_size is cut from the other
four, start uniform over the valid starts of every axis whose extent exceeds the window's. The real class draws from a
numpy RandomState that nothing seeds (one fresh instance per call), so its windows cannot be reproduced; this stand-in
draws from Python's `random` — one randint per shrinking axis, in axis order — which is what the product does, so the
goldens recorded through it pin the networks and the window arithmetic, not monai's RNG stream."""
import random


class RandSpatialCrop:

    def __init__(self, roi_size, random_center=True, random_size=False):
        assert random_center and not random_size, "only the form multiscale_patchgan3d.py uses"
        self.roi_size = tuple(int(v) for v in roi_size)

    def __call__(self, img):
        dims = tuple(img.shape[1:])
        assert len(dims) == len(self.roi_size)
        starts = [random.randint(0, ms - ps) if ms > ps else 0 for ms, ps in zip(dims, self.roi_size)]
        idx = (slice(None),) + tuple(slice(s, s + min(ps, ms)) for s, ps, ms in zip(starts, self.roi_size, dims))
        return img[idx]
