"""YAML config surface of the reference scripts (config/brain.yaml, config/pelvis.yaml; merged with the CLI
flags by OmegaConf in train.py:322-324).  OmegaConf is not available here, so this is a PyYAML loader with
OmegaConf's scalar resolution for the two cases plain YAML-1.1 gets wrong:
    lr: 1e-4                -> float (PyYAML returns the string '1e-4')
    init_train_steps: 0_800_000 -> int  (PyYAML returns the string)
CLI flags override file values, key names are unchanged."""
from __future__ import annotations

import re

import yaml

_FLOAT = re.compile(r"^[-+]?(\d+(_\d+)*\.?\d*|\.\d+)([eE][-+]?\d+)?$")
_INT = re.compile(r"^[-+]?\d+(_\d+)*$")


class Config(dict):
    """dict with attribute access (what the scripts use: args.model, args.lr, ...)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def _resolve(v):
    if isinstance(v, str):
        s = v.strip()
        if _INT.match(s):
            return int(s.replace("_", ""))
        if _FLOAT.match(s):
            return float(s.replace("_", ""))
        return v
    if isinstance(v, dict):
        return Config({k: _resolve(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_resolve(x) for x in v]
    return v


def load_config(path, overrides=None) -> Config:
    with open(path) as f:
        cfg = _resolve(yaml.safe_load(f) or {})
    for k, v in (overrides or {}).items():
        if v is not None:
            cfg[k] = v
    return cfg
