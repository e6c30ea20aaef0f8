"""ganslate_amd — MI355X-native implementation of ganslate's GAN training step behind ganslate's own plugin
surface (engines.Trainer -> BaseGAN -> `_target_` configs). See DESIGN.md."""
__version__ = "0.1.0"
