"""Config loading + reflection factory with the reference's contract
(reference enhancing/utils/general.py:29-40,63-76): a config node is ``{"target": "pkg.mod.Class",
"params": {...}}`` and ``initialize_from_config`` instantiates it.  OmegaConf is not available in this
environment, so yaml files are read with PyYAML into attribute-style dicts (``cfg.model.params.encoder.dim``
and ``cfg["model"]`` both work, which is all the reference code path uses)."""
from __future__ import annotations

import importlib
import random
from typing import Any

import numpy as np
import torch
import yaml


class AttrDict(dict):
    """dict with attribute access, recursively applied (stand-in for OmegaConf's DictConfig)."""

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = v

    @staticmethod
    def wrap(obj: Any) -> Any:
        if isinstance(obj, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return [AttrDict.wrap(v) for v in obj]
        return obj


def set_seed(seed: int) -> None:
    """reference general.py:22-26"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_obj_from_str(name: str, reload: bool = False):
    """reference general.py:29-36"""
    module, cls = name.rsplit(".", 1)
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def initialize_from_config(config) -> object:
    """reference general.py:39-40"""
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def _merge(base: dict, over: dict) -> dict:
    out = dict(base)
    for k, v in over.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def get_config_from_file(config_file) -> AttrDict:
    """reference general.py:63-76 (with the ``base_config: <file>.yaml`` include made to work: the reference's
    own merge iterates keys only, general.py:72)."""
    with open(str(config_file)) as f:
        cfg = yaml.safe_load(f)
    base = cfg.pop("base_config", None) if isinstance(cfg, dict) else None
    if base is not None:
        if not str(base).endswith(".yaml"):
            raise ValueError(f"unsupported base_config {base!r}")
        cfg = _merge(get_config_from_file(base), cfg)
    return AttrDict.wrap(cfg)


def setup_callbacks(exp_config, config):
    """reference general.py:43-60: [SetupCallback, (checkpointing), ImageLogger] + logger.  Checkpoints are written by the in-repo Trainer itself (one
    ``{"state_dict": ...}`` file per epoch, what ModelCheckpoint(save_top_k=-1) produces); the wandb logger needs a package that is not installable here,
    so the logger slot is None and scalar logs go to the Trainer's metrics.jsonl."""
    import os
    import pathlib
    from datetime import datetime
    from .callback import ImageLogger, SetupCallback
    now = datetime.now().strftime('%d%m%Y_%H%M%S')
    basedir = pathlib.Path("experiments", exp_config.name, now)
    os.makedirs(basedir, exist_ok=True)
    return [SetupCallback(config, exp_config, basedir), ImageLogger(exp_config.batch_frequency, exp_config.max_images)], None
