"""Minimal stand-in for `omegaconf`, ONLY so that /root/reference can be imported in the
build container by oracle/gen_golden.py (test infrastructure; never shipped, never on the GPU box).
It provides the few names the reference's hot-path modules touch at import / construction time."""
from . import dictconfig
from .dictconfig import DictConfig

MISSING = "???"


def II(path):
    return "${" + path + "}"


class OmegaConf:
    @staticmethod
    def create(obj=None):
        return DictConfig(obj or {})

    @staticmethod
    def to_yaml(conf):
        import yaml
        return yaml.safe_dump(conf.to_dict())

    @staticmethod
    def to_container(conf, resolve=True):
        return conf.to_dict()
