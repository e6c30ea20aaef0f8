"""init_engine (ganslate/engines/utils.py:14-22). Only the training engine is in the hot-path scope."""
from ..utils import communication
from ..utils.builders import build_conf
from .trainer import Trainer

ENGINES = {"train": Trainer}


def init_engine(mode, omegaconf_args):
    if mode not in ENGINES:
        raise NotImplementedError(f"engine `{mode}` is outside the scope of the MI355X build (train only)")
    communication.init_distributed()
    conf = build_conf(omegaconf_args)
    return ENGINES[mode](conf)
