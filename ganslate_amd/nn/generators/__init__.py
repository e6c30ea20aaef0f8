from .resnet.resnet2d import Resnet2D, Resnet2DConfig  # noqa: F401
from .unet.unet2d import Unet2D, Unet2DConfig  # noqa: F401
from .resnet.resnet3d import Resnet3D, Resnet3DConfig  # noqa: F401
from .resnet.piresnet3d import Piresnet3D, Piresnet3DConfig  # noqa: F401
from .unet.unet3d import Unet3D, Unet3DConfig  # noqa: F401
from .vnet.vnet3d import Vnet2D, Vnet2DConfig, Vnet3D, Vnet3DConfig  # noqa: F401
from .vnet.selfattention_vnet3d import SelfAttentionVnet3D, SelfAttentionVnet3DConfig  # noqa: F401
