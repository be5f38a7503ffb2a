"""Minimal OmegaConf/Hydra-compatible configuration layer.

The reference drives everything from a Hydra ``DictConfig`` (``/root/reference/scripts/train_sft.py:9-15``,
``/root/reference/conf/``).  Neither ``omegaconf`` nor ``hydra`` is installed on the build/GPU image, so this module
provides the subset the hot path and its entry points use, with the same spelling:

* ``DictConfig`` .......... attribute + item access, ``.get``, ``${a.b}`` interpolation, ``???`` (mandatory) values
* ``OmegaConf`` ........... ``create``, ``to_container``, ``to_yaml``, ``missing_keys``, ``merge``, ``load``
* ``compose`` / ``main`` .. Hydra-style ``defaults`` list composition + ``key=value`` command-line overrides

If the real packages are importable the scripts use them instead (see ``scripts/train_sft.py``); the Trainer accepts either.
"""

from __future__ import annotations

import copy
import functools
import os
import re
import sys
from typing import Any, Callable, Iterator

import yaml

MISSING = "???"
_INTERP = re.compile(r"\$\{([^${}]+)\}")
_SCI_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$")  # YAML 1.1 (PyYAML) reads 2e-4 as a string; OmegaConf as a float


class MissingMandatoryValue(KeyError):
    pass


class ConfigKeyError(KeyError):
    pass


class DictConfig:
    __slots__ = ("_content", "_parent", "_resolvers")

    def __init__(self, content: dict | None = None, parent: "DictConfig | None" = None):
        object.__setattr__(self, "_content", {})
        object.__setattr__(self, "_parent", parent)
        object.__setattr__(self, "_resolvers", {})
        for k, v in (content or {}).items():
            self._content[k] = self._wrap(v)

    # ---- construction helpers -------------------------------------------------------------------------------------
    def _wrap(self, v: Any) -> Any:
        if isinstance(v, DictConfig):
            node = DictConfig(parent=self)
            for k2, v2 in v._content.items():
                node._content[k2] = node._wrap(v2)
            return node
        if isinstance(v, dict):
            return DictConfig(v, parent=self)
        if isinstance(v, (list, tuple)):
            return [self._wrap(x) for x in v]
        if isinstance(v, str) and _SCI_FLOAT.match(v):
            return float(v)
        return v

    def _root(self) -> "DictConfig":
        node = self
        while node._parent is not None:
            node = node._parent
        return node

    # ---- resolution -------------------------------------------------------------------------------------------------
    def _select(self, path: str) -> Any:
        node: Any = self._root()
        for part in path.split("."):
            if not isinstance(node, DictConfig) or part not in node._content:
                raise ConfigKeyError(f"interpolation key '{path}' not found")
            node = node._resolve(node._content[part], part)
        return node

    def _resolve(self, v: Any, key: str = "?") -> Any:
        if isinstance(v, str):
            if v == MISSING:
                raise MissingMandatoryValue(f"Missing mandatory value: {key}")
            m = _INTERP.fullmatch(v)
            if m:  # whole-string interpolation keeps the type
                return self._resolve_expr(m.group(1))
            if "${" in v:
                return _INTERP.sub(lambda mm: str(self._resolve_expr(mm.group(1))), v)
        return v

    def _resolve_expr(self, expr: str) -> Any:
        expr = expr.strip()
        if ":" in expr:  # custom resolver, e.g. ${hydra:job.config_name}
            name, arg = expr.split(":", 1)
            res = self._root()._resolvers.get(name)
            if res is None:
                raise ConfigKeyError(f"unknown resolver '{name}' in '${{{expr}}}'")
            return res(arg)
        return self._select(expr)

    # ---- mapping protocol -----------------------------------------------------------------------------------------
    def __getattr__(self, key: str) -> Any:
        if key.startswith("__"):
            raise AttributeError(key)
        try:
            return self._resolve(self._content[key], key)
        except KeyError as e:
            if isinstance(e, (MissingMandatoryValue, ConfigKeyError)):
                raise
            raise AttributeError(f"Key '{key}' is not in config") from None

    def __getitem__(self, key: str) -> Any:
        if key not in self._content:
            raise KeyError(key)
        return self._resolve(self._content[key], key)

    def __setattr__(self, key: str, value: Any) -> None:
        self._content[key] = self._wrap(value)

    __setitem__ = __setattr__

    def __delattr__(self, key: str) -> None:
        del self._content[key]

    def __contains__(self, key: str) -> bool:
        return key in self._content

    def __iter__(self) -> Iterator[str]:
        return iter(self._content)

    def __len__(self) -> int:
        return len(self._content)

    def keys(self):
        return self._content.keys()

    def items(self):
        return [(k, self[k]) for k in self._content]

    def values(self):
        return [self[k] for k in self._content]

    def get(self, key: str, default: Any = None) -> Any:
        if key not in self._content:
            return default
        v = self._content[key]
        if isinstance(v, str) and v == MISSING:
            return default
        v = self._resolve(v, key)
        return default if v is None else v

    def __repr__(self) -> str:
        return f"DictConfig({OmegaConf.to_container(self, resolve=False)!r})"

    def __eq__(self, other: Any) -> bool:
        if isinstance(other, DictConfig):
            return OmegaConf.to_container(self, resolve=False) == OmegaConf.to_container(other, resolve=False)
        if isinstance(other, dict):
            return OmegaConf.to_container(self, resolve=False) == other
        return NotImplemented

    def __deepcopy__(self, memo):
        new = DictConfig(OmegaConf.to_container(self, resolve=False))
        object.__setattr__(new, "_resolvers", dict(self._resolvers))
        return new


def _plain(v: Any, owner: DictConfig, resolve: bool) -> Any:
    if isinstance(v, DictConfig):
        return OmegaConf.to_container(v, resolve=resolve)
    if isinstance(v, list):
        return [_plain(x, owner, resolve) for x in v]
    if resolve and isinstance(v, str):
        return _plain(owner._resolve(v), owner, resolve) if "${" in v else v
    return v


class OmegaConf:
    @staticmethod
    def create(obj: Any = None) -> DictConfig:
        if obj is None:
            return DictConfig({})
        if isinstance(obj, str):
            obj = yaml.safe_load(obj) or {}
        if isinstance(obj, DictConfig):
            return copy.deepcopy(obj)
        return DictConfig(dict(obj))

    @staticmethod
    def load(path: str) -> DictConfig:
        with open(path) as f:
            return DictConfig(yaml.safe_load(f) or {})

    @staticmethod
    def to_container(cfg: Any, resolve: bool = False, **_: Any) -> Any:
        if isinstance(cfg, DictConfig):
            out = {}
            for k, v in cfg._content.items():
                if resolve and isinstance(v, str) and v == MISSING:
                    raise MissingMandatoryValue(f"Missing mandatory value: {k}")
                out[k] = _plain(v, cfg, resolve)
            return out
        if isinstance(cfg, list):
            return [OmegaConf.to_container(x, resolve=resolve) for x in cfg]
        return cfg

    @staticmethod
    def to_yaml(cfg: DictConfig, resolve: bool = False, sort_keys: bool = False) -> str:
        return yaml.safe_dump(OmegaConf.to_container(cfg, resolve=resolve), sort_keys=sort_keys, default_flow_style=False)

    @staticmethod
    def missing_keys(cfg: DictConfig) -> set[str]:
        missing: set[str] = set()

        def walk(node: Any, prefix: str) -> None:
            if isinstance(node, DictConfig):
                for k, v in node._content.items():
                    walk(v, f"{prefix}.{k}" if prefix else k)
            elif isinstance(node, list):
                for i, v in enumerate(node):
                    walk(v, f"{prefix}[{i}]")
            elif isinstance(node, str) and node == MISSING:
                missing.add(prefix)

        walk(cfg, "")
        return missing

    @staticmethod
    def is_missing(cfg: DictConfig, key: str) -> bool:
        return key in cfg._content and cfg._content[key] == MISSING

    @staticmethod
    def merge(*cfgs: Any) -> DictConfig:
        def merge_into(dst: dict, src: dict) -> dict:
            for k, v in src.items():
                if isinstance(v, dict) and isinstance(dst.get(k), dict):
                    merge_into(dst[k], v)
                else:
                    dst[k] = copy.deepcopy(v)
            return dst

        out: dict = {}
        resolvers: dict = {}
        for c in cfgs:
            if isinstance(c, DictConfig):
                resolvers.update(c._resolvers)
                c = OmegaConf.to_container(c, resolve=False)
            merge_into(out, c or {})
        cfg = DictConfig(out)
        object.__setattr__(cfg, "_resolvers", resolvers)
        return cfg

    @staticmethod
    def register_resolver(cfg: DictConfig, name: str, fn: Callable[[str], Any]) -> None:
        cfg._root()._resolvers[name] = fn

    @staticmethod
    def update(cfg: DictConfig, dotted: str, value: Any) -> None:
        parts = dotted.split(".")
        node = cfg
        for p in parts[:-1]:
            if p not in node._content or not isinstance(node._content[p], DictConfig):
                node._content[p] = DictConfig({}, parent=node)
            node = node._content[p]
        node._content[parts[-1]] = node._wrap(value)


# --------------------------------------------------------------------------------------------------------------------
# Hydra-style composition
# --------------------------------------------------------------------------------------------------------------------
def _load_yaml(path: str) -> dict:
    if not os.path.exists(path):
        raise FileNotFoundError(f"config file not found: {path}")
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _nest(package: str, body: dict) -> dict:
    for part in reversed([p for p in package.split(".") if p]):
        body = {part: body}
    return body


def _compose_file(config_dir: str, group: str, option: str, choices: dict[str, str]) -> dict:
    """Load ``<config_dir>/<group>/<option>.yaml`` and recursively merge its ``defaults`` list (earlier entries first,
    the file's own body last, like Hydra's implicit ``_self_`` at the end)."""
    rel = os.path.join(group, option) if group else option
    raw = _load_yaml(os.path.join(config_dir, rel + ".yaml"))
    defaults = raw.pop("defaults", []) or []
    merged: dict = {}
    for entry in defaults:
        if isinstance(entry, str):
            if entry == "_self_":
                continue
            sub = _compose_file(config_dir, group, entry, choices)  # same group, package of the including file
            merged = OmegaConf.to_container(OmegaConf.merge(merged, sub))
        elif isinstance(entry, dict):
            (k, v), = entry.items()
            k = k.strip()
            if k.startswith("override ") or k.startswith("hydra/"):
                continue  # hydra's own logging groups: not part of the job config
            sub_group = os.path.join(group, k) if group else k
            choice = choices.get(sub_group.replace(os.sep, "/"), v)
            if choice is None or choice == "null":
                continue
            if choice == MISSING:
                raise MissingMandatoryValue(f"You must specify '{sub_group}', e.g. {sub_group}=<option>")
            sub = _compose_file(config_dir, sub_group, choice, choices)
            merged = OmegaConf.to_container(OmegaConf.merge(merged, _nest(k, sub)))
    return OmegaConf.to_container(OmegaConf.merge(merged, raw))


def compose(config_dir: str, config_name: str, overrides: list[str] | None = None) -> DictConfig:
    overrides = list(overrides or [])
    config_dir = os.path.abspath(config_dir)
    choices: dict[str, str] = {}
    value_overrides: list[tuple[str, Any]] = []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        key, val = ov.split("=", 1)
        key = key.lstrip("+")
        if os.path.isdir(os.path.join(config_dir, key.replace(".", os.sep))) and "." not in key:
            choices[key] = val  # config-group choice, e.g. data=sft/mls-hubert_large_ll60k-layer_22
        else:
            value_overrides.append((key, yaml.safe_load(val) if val != "" else ""))
    body = _compose_file(config_dir, "", config_name, choices)
    cfg = DictConfig(body)
    OmegaConf.register_resolver(cfg, "hydra", lambda arg: {"job.config_name": config_name, "job.name": config_name}.get(arg, arg))
    OmegaConf.register_resolver(cfg, "oc.env", lambda arg: os.environ.get(arg.split(",")[0], arg.split(",")[1] if "," in arg else ""))
    for key, val in value_overrides:
        OmegaConf.update(cfg, key, val)
    return cfg


def main(config_path: str, config_name: str, version_base: Any = None) -> Callable:
    """Drop-in for ``@hydra.main(config_path=..., config_name=...)``: composes the config from ``sys.argv[1:]``."""

    def deco(fn: Callable) -> Callable:
        @functools.wraps(fn)
        def wrapper(cfg: DictConfig | None = None):
            if cfg is None:
                base = os.path.dirname(os.path.abspath(sys.modules[fn.__module__].__file__))
                cfg = compose(os.path.normpath(os.path.join(base, config_path)), config_name, sys.argv[1:])
            return fn(cfg)

        return wrapper

    return deco
