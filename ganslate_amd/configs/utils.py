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
    merged = OmegaConf.merge(OmegaConf.structured(config_class), conf)
    _complete_optional_sections(config_class, merged)
    return merged


def _complete_optional_sections(dc_type, node):
    """`val: Optional[ValidationConfig] = None` and the like: when the YAML provides the section, the dataclass defaults
    of its declared type apply underneath it (OmegaConf's typed merge does this; the lite merge only sees a dict
    replacing None)."""
    import dataclasses
    import typing
    hints = typing.get_type_hints(dc_type)
    for f in dataclasses.fields(dc_type):
        t = hints.get(f.name)
        args = [a for a in typing.get_args(t) if a is not type(None)] if typing.get_origin(t) is typing.Union else []
        inner = args[0] if len(args) == 1 else (t if dataclasses.is_dataclass(t) else None)
        child = node._d.get(f.name) if isinstance(node, DictConfig) else None
        if inner is None or not dataclasses.is_dataclass(inner) or not isinstance(child, DictConfig):
            continue
        if args:        # Optional[...] section present in the YAML: defaults first, the YAML's values on top
            node._set(f.name, OmegaConf.merge(OmegaConf.structured(inner), child))
            child = node._d[f.name]
        _complete_optional_sections(inner, child)


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
