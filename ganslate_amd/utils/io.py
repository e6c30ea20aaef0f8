"""`_target_` resolution (ganslate/utils/io.py:73-76). Reference YAMLs name classes under `ganslate.*`; those
resolve to this package's counterparts, so the YAMLs run unchanged."""
import importlib


def import_attr(module_attr: str):
    module, attr = module_attr.rsplit(".", 1)
    if module == "ganslate" or module.startswith("ganslate."):
        module = "ganslate_amd" + module[len("ganslate"):]
    return getattr(importlib.import_module(module), attr)


def mkdirs(*paths):
    from pathlib import Path
    for p in paths:
        Path(p).mkdir(parents=True, exist_ok=True)
