"""Minimal Hydra-compatible config composer (hydra / omegaconf are not installed in this image).

Accepts the reference's compose tree unchanged (mava/configs/default/rec_magpo.yaml -> logger, arch,
system, network, env [-> scenario]) and the override grammar its README uses (README.md:44-57):
  group=name              env=coordsum
  group/sub=name          env/scenario=8x15-100
  a.b.c=value             arch.num_envs=64   system.total_timesteps=~
  +a.b.c=value            +env.kwargs.num_agents=4      (add a new key)
  ~a.b.c                  delete a key
``${a.b}`` interpolations are resolved on access.  The result behaves like a DictConfig with struct mode
off (rec_magpo.py:826): attribute and item access, free assignment of new keys.
"""
from __future__ import annotations

import copy
import os
import re
from typing import Any, Dict, List, Optional

import yaml

CONFIG_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")
_INTERP = re.compile(r"\$\{([^}]+)\}")


class Config:
    def __init__(self, data: Optional[Dict[str, Any]] = None, root: Optional["Config"] = None):
        object.__setattr__(self, "_d", {})
        object.__setattr__(self, "_root", root if root is not None else self)
        for k, v in (data or {}).items():
            self[k] = v

    def _wrap(self, v):
        if isinstance(v, dict):
            return Config(v, self._root)
        if isinstance(v, Config):
            object.__setattr__(v, "_root", self._root)
            for sub in v._d.values():
                if isinstance(sub, Config):
                    v._wrap(sub)
        return v

    def __setitem__(self, k, v):
        self._d[k] = self._wrap(v)

    def __setattr__(self, k, v):
        self[k] = v

    def _resolve(self, v):
        if isinstance(v, str):
            m = _INTERP.fullmatch(v)
            if m:
                return self._root.select(m.group(1))
            return _INTERP.sub(lambda mm: str(self._root.select(mm.group(1))), v)
        return v

    def __getitem__(self, k):
        return self._resolve(self._d[k])

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __contains__(self, k):
        return k in self._d

    def get(self, k, default=None):
        return self[k] if k in self._d else default

    def keys(self):
        return self._d.keys()

    def items(self):
        return [(k, self[k]) for k in self._d]

    def values(self):
        return [self[k] for k in self._d]

    def pop(self, k, *a):
        return self._d.pop(k, *a)

    def select(self, dotted: str):
        node: Any = self
        for part in dotted.split("."):
            node = node[part]
        return node

    def set_path(self, dotted: str, value, create: bool = True):
        parts = dotted.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node:
                if not create:
                    raise KeyError(dotted)
                node[p] = {}
            node = node[p]
        node[parts[-1]] = value

    def delete_path(self, dotted: str):
        parts = dotted.split(".")
        node = self
        for p in parts[:-1]:
            node = node[p]
        node.pop(parts[-1])

    def to_container(self, resolve: bool = True) -> Dict[str, Any]:
        out = {}
        for k in self._d:
            v = self[k] if resolve else self._d[k]
            out[k] = v.to_container(resolve) if isinstance(v, Config) else copy.deepcopy(v)
        return out

    def __deepcopy__(self, memo):
        return Config(self.to_container(resolve=False))

    def __repr__(self):
        return f"Config({self.to_container(resolve=False)!r})"


def _load_yaml(path: str) -> Dict[str, Any]:
    if not os.path.exists(path):
        raise FileNotFoundError(f"config file not found: {path}")
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _merge(dst: Dict[str, Any], src: Dict[str, Any]):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def _compose_group(root: str, group: str, name: str, choices: Dict[str, str]) -> Dict[str, Any]:
    """Load <root>/<group>/<name>.yaml and its own defaults list (sub-groups nest under their key)."""
    data = _load_yaml(os.path.join(root, group, name + ".yaml"))
    defaults = data.pop("defaults", ["_self_"])
    out: Dict[str, Any] = {}
    for d in defaults:
        if d == "_self_":
            _merge(out, data)
        else:
            (sub, sub_name), = d.items()
            sub_name = choices.get(f"{group}/{sub}", sub_name)
            out[sub] = _compose_group(root, f"{group}/{sub}", str(sub_name), choices)
    return out


def _parse_value(text: str):
    return yaml.safe_load(text) if text != "" else ""


def compose(config_name: str = "rec_magpo", overrides: Optional[List[str]] = None, config_root: str = CONFIG_ROOT,
            config_path: str = "default") -> Config:
    overrides = list(overrides or [])
    choices: Dict[str, str] = {}
    value_overrides = []
    for ov in overrides:
        if ov.startswith("~"):
            value_overrides.append(("del", ov[1:], None))
            continue
        if "=" not in ov:
            raise ValueError(f"bad override {ov!r}")
        k, v = ov.split("=", 1)
        add = k.startswith("+")
        k = k.lstrip("+")
        if "." not in k and os.path.isdir(os.path.join(config_root, k)):
            choices[k] = v
        else:
            value_overrides.append(("set", k, _parse_value(v), add))
    top = _load_yaml(os.path.join(config_root, config_path, config_name + ".yaml"))
    defaults = top.pop("defaults", [])
    data: Dict[str, Any] = {}
    for d in defaults:
        if d == "_self_":
            _merge(data, top)
        else:
            (group, name), = d.items()
            name = choices.get(group, name)
            key = group.split("/")[-1]
            data[key] = _compose_group(config_root, group, str(name), choices)
    cfg = Config(data)
    for op in value_overrides:
        if op[0] == "del":
            cfg.delete_path(op[1])
        else:
            _, k, v, add = op
            if not add:
                try:
                    cfg.select(k)
                except (KeyError, TypeError):
                    raise KeyError(f"override {k!r}: key not in config (use +{k}=... to add it)") from None
            cfg.set_path(k, v)
    return cfg
