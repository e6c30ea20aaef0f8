from .pix2pix import Pix2PixConditionalGAN, Pix2PixConditionalGANConfig  # noqa: F401
