from .patchgan.patchgan2d import PatchGAN2D, PatchGAN2DConfig  # noqa: F401
