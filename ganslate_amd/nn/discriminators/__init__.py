from .patchgan.patchgan2d import PatchGAN2D, PatchGAN2DConfig  # noqa: F401
from .patchgan.patchgan3d import PatchGAN3D, PatchGAN3DConfig  # noqa: F401
from .patchgan.multiscale_patchgan3d import MultiScalePatchGAN3D, MultiScalePatchGAN3DConfig  # noqa: F401
from .patchgan.selfattention_patchgan3d import SelfAttentionPatchGAN3D, SelfAttentionPatchGAN3DConfig  # noqa: F401
