"""Dependency-free stand-in for the reference's OmegaConf usage (main.py:29-34): load a flat
YAML file, merge `key=value` CLI overrides with YAML scalar typing (Null -> None, 1e-5 -> float,
[0,1] -> list) and resolve ${key} interpolations (config.yaml:42-45).  Attribute access like
`cfg.upsample_factor` works as with OmegaConf."""
from __future__ import annotations

import re
import sys

import yaml

_FLOAT = re.compile(r'^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$')
_INTERP = re.compile(r'\$\{([^}]+)\}')


class Config(dict):
    def __getattr__(self, key):
        try:
            return self._resolve(self[key])
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def _resolve(self, v, depth=0):
        if isinstance(v, str) and depth < 8:
            def sub(m):
                return str(self._resolve(self[m.group(1)], depth + 1))
            return _INTERP.sub(sub, v)
        return v

    def to_container(self):
        return {k: self._resolve(v) for k, v in self.items()}


def _scalar(text: str):
    v = yaml.safe_load(text) if text != '' else ''
    if isinstance(v, str) and _FLOAT.match(v.strip()):
        return float(v)                        # YAML 1.1 leaves "1e-5" a string; OmegaConf makes it a float
    return v


def _retype(v):
    if isinstance(v, str) and _FLOAT.match(v.strip()):
        return float(v)
    return v


def load(path: str) -> Config:
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    return Config({k: _retype(v) for k, v in raw.items()})


def from_cli(argv=None) -> Config:
    argv = sys.argv[1:] if argv is None else argv
    out = Config()
    for a in argv:
        if '=' not in a:
            raise ValueError(f'expected key=value, got {a!r}')
        k, v = a.split('=', 1)
        out[k] = _scalar(v)
    return out


def merge(base: Config, *others: Config) -> Config:
    out = Config(base)
    for o in others:
        out.update(o)
    return out
