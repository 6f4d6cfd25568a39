"""Minimal re-statement of the reference's plugin contract (the drop-in boundary).

Reference: waveform_analysis/core/plugins/core/base.py -- ``Option`` (:38-275) and ``Plugin``
(:321-663).  Only the attributes and hooks the Context reads are restated, with the same names
and meaning, so an instance of a plugin defined here can be handed to the reference's
``Context.register(plugin, allow_override=True)`` (core/foundation/mixins.py:56-127) as well as
to the tiny contexts used in tests.  When the reference package is importable its own classes
are used instead, so ``isinstance(plugin, waveform_analysis...Plugin)`` holds inside a real
Context.
"""

from __future__ import annotations

import abc
from typing import Any

import numpy as np


class _Option:
    """Configuration option: default + optional type / validation (base.py:38-275)."""

    def __init__(self, default: Any = None, type: Any = None, help: str = "", validate=None,
                 track: bool = True, unit=None, internal_unit=None, choices=None, min_value=None,
                 max_value=None, deprecated: bool = False, deprecated_message: str = "", alias=None):
        self.default = default
        self.type = type
        self.help = help
        self.validate = validate
        self.track = track
        self.unit = unit
        self.internal_unit = internal_unit
        self.choices = choices
        self.min_value = min_value
        self.max_value = max_value
        self.deprecated = deprecated
        self.deprecated_message = deprecated_message
        self.alias = alias

    def validate_value(self, name: str, value: Any, plugin_name: str = "") -> Any:
        if value is None:
            return value
        if self.type is not None and not isinstance(value, self.type):
            if self.type is float and isinstance(value, int) and not isinstance(value, bool):
                value = float(value)
            else:
                try:
                    value = self.type(value)
                except Exception as exc:
                    raise TypeError(f"Option '{name}' of {plugin_name} expects {self.type}") from exc
        if self.choices is not None and value not in self.choices:
            raise ValueError(f"Option '{name}' of {plugin_name} must be one of {self.choices}")
        if self.min_value is not None and value < self.min_value:
            raise ValueError(f"Option '{name}' of {plugin_name} must be >= {self.min_value}")
        if self.max_value is not None and value > self.max_value:
            raise ValueError(f"Option '{name}' of {plugin_name} must be <= {self.max_value}")
        if self.validate is not None and not self.validate(value):
            raise ValueError(f"Option '{name}' of {plugin_name} failed validation")
        return value


class _Plugin(abc.ABC):
    """Plugin base: class attributes the Context reads + compute() (base.py:321-663)."""

    provides: str = ""
    depends_on: list = []
    options: dict = {}
    save_when: str = "never"
    output_dtype: np.dtype | None = None
    input_dtype: dict = {}
    output_kind: str = "static"
    description: str = ""
    version: str = "0.0.0"
    is_side_effect: bool = False
    uses_run_config: bool = False
    timeout: float | None = None

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        merged: dict = {}
        for base in reversed(cls.__mro__):  # base.py:460-472
            opts = base.__dict__.get("options")
            if isinstance(opts, dict):
                merged.update(opts)
        cls.options = merged

    @property
    def config_keys(self) -> list[str]:
        return list(self.options.keys())

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list:
        return list(self.depends_on) if self.depends_on else []

    def get_dependency_name(self, dep) -> str:
        return dep[0] if isinstance(dep, tuple) else dep

    @abc.abstractmethod
    def compute(self, context: Any, run_id: str, **kwargs):
        ...

    def on_error(self, context: Any, exception: Exception) -> None:
        pass

    def cleanup(self, context: Any) -> None:
        pass

    def get_lineage(self, context: Any) -> dict:
        config = {k: context.get_config(self, k) for k, o in self.options.items() if getattr(o, "track", True)}
        deps = {}
        for dep in self.resolve_depends_on(context):
            name = self.get_dependency_name(dep)
            deps[name] = context.get_lineage(name)
        return {
            "plugin_class": self.__class__.__name__,
            "plugin_version": self.version,
            "description": self.description,
            "config": config,
            "depends_on": deps,
            "dtype": np.dtype(self.output_dtype).descr if self.output_dtype is not None else None,
        }


try:  # inside a reference installation: be a real subclass of its Plugin
    from waveform_analysis.core.plugins.core.base import Option, Plugin  # type: ignore
except Exception:  # standalone (tests, GPU box): the restatement above
    Option, Plugin = _Option, _Plugin


class SimpleContext:
    """The subset of Context a hot-path plugin calls (reference test double:
    tests/utils.py:323-413): config lookup plugin-nested > namespaced > global > default,
    get_data from pre-seeded data or from registered plugins, _set_data, get_plugin."""

    def __init__(self, config: dict | None = None, data: dict | None = None, plugins=()):
        self.config = dict(config or {})
        self._data = dict(data or {})
        self._results: dict = {}
        self._plugins: dict = {}
        for p in plugins:
            self.register(p)

    def register(self, plugin, allow_override: bool = True):
        if plugin.provides in self._plugins and not allow_override:
            raise ValueError(f"plugin for '{plugin.provides}' already registered")
        self._plugins[plugin.provides] = plugin
        self._results = {k: v for k, v in self._results.items() if k[1] != plugin.provides}
        return plugin

    def get_plugin(self, name: str):
        return self._plugins[name]

    def get_config(self, plugin, name: str):
        prov = plugin.provides
        block = self.config.get(prov)
        if isinstance(block, dict) and name in block:
            return block[name]
        if f"{prov}.{name}" in self.config:
            return self.config[f"{prov}.{name}"]
        if name in self.config:
            return self.config[name]
        if name in getattr(plugin, "options", {}):
            return plugin.options[name].default
        return None

    def get_data(self, run_id: str, name: str):
        if (run_id, name) in self._results:
            return self._results[(run_id, name)]
        if name in self._data:
            return self._data[name]
        if name in self._plugins:
            plugin = self._plugins[name]
            try:
                result = plugin.compute(self, run_id)
            except Exception as exc:  # context_execution.py:169-176
                plugin.on_error(self, exc)
                raise RuntimeError(f"Plugin '{name}' failed: {exc}") from exc
            finally:
                plugin.cleanup(self)
            self._results[(run_id, name)] = result
            return result
        return None

    def _set_data(self, run_id: str, name: str, value) -> None:
        self._results[(run_id, name)] = value

    def get_lineage(self, name: str) -> dict:
        return {}

    def key_for(self, run_id: str, data_name: str) -> str:
        return f"{run_id}-{data_name}-key"


__all__ = ["Option", "Plugin", "SimpleContext"]
