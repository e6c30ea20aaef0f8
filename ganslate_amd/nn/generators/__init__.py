from .resnet.resnet2d import Resnet2D, Resnet2DConfig  # noqa: F401
