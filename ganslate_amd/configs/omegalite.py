"""A small structured-config engine with the slice of OmegaConf semantics ganslate's hot path relies on
(omegaconf is not installed on the target image): dataclass defaults, MISSING ('???'), ${a.b} interpolation
(omegaconf.II), attribute + item access, dotlist overrides, deep merge, YAML I/O.

Reference usage mirrored: ganslate/configs/utils.py:10-61 (init_config / instantiate_dataclasses_from_yaml),
ganslate/utils/builders.py:16-24 (build_conf), ganslate/configs/base.py:27-43,111-129 (II defaults).
"""
import copy
import dataclasses
import re
from typing import Any

import yaml

MISSING = "???"


def II(path: str) -> str:
    return "${" + path + "}"


_INTERP = re.compile(r"^\$\{([^}]+)\}$")


class MissingMandatoryValue(Exception):
    pass


class DictConfig:
    """Nested config node. Reading an absent key raises (struct-mode behaviour of the reference's configs);
    reading a '???' value raises MissingMandatoryValue; '${path}' values resolve against the root."""

    def __init__(self, content=None, parent=None):
        object.__setattr__(self, "_d", {})
        object.__setattr__(self, "_parent", parent)
        for k, v in (content or {}).items():
            self._set(k, v)

    # ---- construction -------------------------------------------------------------------------------------
    def _wrap(self, v):
        if isinstance(v, DictConfig):
            if v._parent is not None and v._parent is not self:
                v = DictConfig(v._raw_dict())
            object.__setattr__(v, "_parent", self)
            return v
        if dataclasses.is_dataclass(v):
            return DictConfig(_dataclass_to_dict(v), self)
        if isinstance(v, dict):
            return DictConfig(v, self)
        if isinstance(v, tuple):
            return list(v)
        return v

    def _set(self, k, v):
        self._d[k] = self._wrap(v)

    def _root(self):
        n = self
        while n._parent is not None:
            n = n._parent
        return n

    def _raw_dict(self):
        return {k: (v._raw_dict() if isinstance(v, DictConfig) else copy.deepcopy(v)) for k, v in self._d.items()}

    # ---- access -------------------------------------------------------------------------------------------------
    def _resolve(self, k, v):
        if isinstance(v, str):
            if v == MISSING:
                raise MissingMandatoryValue(f"Missing mandatory value: {k}")
            m = _INTERP.match(v)
            if m:
                return select(self._root(), m.group(1))
        return v

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self._resolve(k, self._d[k])
        except KeyError:
            raise AttributeError(f"Key '{k}' is not in the config") from None

    def __getitem__(self, k):
        try:
            return self._resolve(k, self._d[k])
        except KeyError:
            raise KeyError(f"Key '{k}' is not in the config") from None

    def get(self, k, default=None):
        return self[k] if k in self._d else default

    def __setattr__(self, k, v):
        self._set(k, v)

    def __setitem__(self, k, v):
        self._set(k, v)

    def __contains__(self, k):
        return k in self._d

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def keys(self):
        return self._d.keys()

    def items(self):
        return [(k, self[k]) for k in self._d]

    def values(self):
        return [self[k] for k in self._d]

    def pop(self, k, *default):
        if k in self._d:
            v = self[k]
            del self._d[k]
            return v
        if default:
            return default[0]
        raise KeyError(k)

    def __bool__(self):
        return True

    def __repr__(self):
        return f"DictConfig({self._raw_dict()!r})"

    def __deepcopy__(self, memo):
        root = self._root()
        if root is self:
            return DictConfig(self._raw_dict())
        # keep interpolations resolvable: deep-copy from the root and walk back down
        path = []
        n = self
        while n._parent is not None:
            for k, v in n._parent._d.items():
                if v is n:
                    path.append(k)
                    break
            n = n._parent
        c = DictConfig(root._raw_dict())
        for k in reversed(path):
            c = c._d[k]
        return c


def _dataclass_to_dict(dc) -> dict:
    """dataclass class or instance -> plain dict of defaults (nested dataclasses expanded)."""
    out = {}
    for f in dataclasses.fields(dc):
        if isinstance(dc, type):
            if f.default is not dataclasses.MISSING:
                v = f.default
            elif f.default_factory is not dataclasses.MISSING:
                v = f.default_factory()
            else:
                v = MISSING
        else:
            v = getattr(dc, f.name)
        if dataclasses.is_dataclass(v):
            v = _dataclass_to_dict(v)
        elif isinstance(v, tuple):
            v = list(v)
        out[f.name] = copy.deepcopy(v)
    return out


def select(conf: DictConfig, path: str):
    node: Any = conf
    for part in path.split("."):
        node = node[part]
    return node


def _merge_into(dst: DictConfig, src: DictConfig):
    for k, v in src._d.items():
        cur = dst._d.get(k)
        if isinstance(v, DictConfig) and isinstance(cur, DictConfig):
            _merge_into(cur, v)
        elif isinstance(v, DictConfig):
            dst._set(k, DictConfig(v._raw_dict()))
        else:
            dst._set(k, copy.deepcopy(v))


def _parse_scalar(s: str):
    try:
        return yaml.safe_load(s)
    except yaml.YAMLError:
        return s


class OmegaConf:
    """The handful of static helpers the reference calls."""

    @staticmethod
    def create(obj=None) -> DictConfig:
        return DictConfig(obj or {})

    @staticmethod
    def structured(dc) -> DictConfig:
        return DictConfig(_dataclass_to_dict(dc))

    @staticmethod
    def load(path) -> DictConfig:
        with open(path) as f:
            return DictConfig(yaml.safe_load(f) or {})

    @staticmethod
    def from_dotlist(dotlist) -> DictConfig:
        root = DictConfig()
        for item in dotlist:
            key, _, val = item.partition("=")
            node = root
            parts = key.strip().split(".")
            for p in parts[:-1]:
                if p not in node or not isinstance(node._d[p], DictConfig):
                    node._set(p, {})
                node = node._d[p]
            node._set(parts[-1], _parse_scalar(val))
        return root

    @staticmethod
    def merge(*confs) -> DictConfig:
        out = DictConfig(confs[0]._raw_dict() if isinstance(confs[0], DictConfig) else confs[0])
        for c in confs[1:]:
            if not isinstance(c, DictConfig):
                c = DictConfig(c)
            _merge_into(out, c)
        return out

    @staticmethod
    def select(conf, key):
        try:
            node = conf
            for part in key.split("."):
                node = node._d[part]
            return node
        except (KeyError, AttributeError):
            return None

    @staticmethod
    def update(conf, key, value, merge=False):
        parts = key.split(".")
        node = conf
        for p in parts[:-1]:
            node = node._d[p]
        if merge and isinstance(node._d.get(parts[-1]), DictConfig):
            _merge_into(node._d[parts[-1]], value if isinstance(value, DictConfig) else DictConfig(value))
        else:
            node._set(parts[-1], value)

    @staticmethod
    def to_container(conf, resolve=False):
        if not resolve:
            return conf._raw_dict()
        return {k: (OmegaConf.to_container(v, True) if isinstance(v, DictConfig) else v) for k, v in conf.items()}

    @staticmethod
    def to_yaml(conf) -> str:
        return yaml.safe_dump(conf._raw_dict(), sort_keys=False)
