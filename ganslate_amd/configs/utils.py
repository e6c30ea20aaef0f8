"""YAML -> structured config (ganslate/configs/utils.py:10-61): every node carrying `_target_` is completed with
the defaults of the dataclass named `<_target_>Config`, deepest nodes first, then merged onto `Config`."""
import importlib.util
import sys
from pathlib import Path

from ..utils.io import import_attr
from .omegalite import DictConfig, OmegaConf


def init_config(conf, config_class):
    conf = conf if isinstance(conf, DictConfig) else OmegaConf.load(str(conf))
    project = conf.get("project")
    if project:
        assert isinstance(project, str), "project needs to be a str path"
        project_path = Path(project).resolve() / "__init__.py"
        # The reference asserts the file exists (utils.py:21); example YAMLs carry their authors' absolute
        # paths, so a missing project dir is only an error when a `project.*` target is actually used.
        if project_path.is_file():
            spec = importlib.util.spec_from_file_location("project", str(project_path))
            module = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(module)
            sys.modules["project"] = module
    conf = instantiate_dataclasses_from_yaml(conf)
    return OmegaConf.merge(OmegaConf.structured(config_class), conf)


def instantiate_dataclasses_from_yaml(conf):
    for key in get_all_conf_keys(conf):
        node = OmegaConf.select(conf, key)
        if is_dataclass(node):
            OmegaConf.update(conf, key, OmegaConf.merge(init_dataclass(node), node), merge=False)
    return conf


def init_dataclass(node):
    return OmegaConf.structured(import_attr(f'{node["_target_"]}Config'))


def is_dataclass(node):
    return bool(isinstance(node, DictConfig) and "_target_" in node)


def get_all_conf_keys(conf):
    keys = list(iterate_nested_dict_keys(OmegaConf.to_container(conf)))
    return keys[::-1]


def iterate_nested_dict_keys(d):
    if isinstance(d, dict):
        level = list(d.keys())
        for k in level:
            yield k
        for k in level:
            for sub in iterate_nested_dict_keys(d[k]):
                yield f"{k}.{sub}"
