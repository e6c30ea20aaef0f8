from . import functional
