from .trainer import Trainer  # noqa: F401
from .utils import init_engine  # noqa: F401
