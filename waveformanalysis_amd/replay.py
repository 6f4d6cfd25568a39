"""strax-shaped replay driver for BASELINE.json's config 5 ("S1/S2 peak building + event_grouping on strax-adapter
replay").

Reference: waveform_analysis/core/plugins/core/adapters.py -- ``StraxPluginAdapter`` (:28-180) wraps a strax-SHAPED
plugin class (``provides`` / ``depends_on`` tuple / ``dtype`` / ``takes_config`` / ``compute(self, <dependency arrays>,
<config values>)``; strax itself is not a dependency of the reference and is not used here either) as a ``Plugin``, and
``StraxContextAdapter`` (:251-440) puts ``register`` / ``get_array`` / ``get_df`` / ``set_config`` in front of a Context.

What this module adds:

* ``strax_shaped(plugin_factory, ...)`` -- a strax-shaped class around one plugin of the hot path.  Its ``compute`` gets the
  dependency arrays and config values the adapter resolved (adapters.py:112-170), puts them into a one-plugin context and
  runs the plugin's own ``compute(context, run_id)``.  The same wrapper takes the HIP plugin classes (the product) and
  the reference's CPU plugin classes (tests/golden/make_c5_golden.py generates the expected tables that way, through
  the reference's own adapter classes).
* ``C5_CHAIN`` -- the two config-5 chains as strax-shaped declarations:
      st_waveforms -> filtered_waveforms -> hit -> waveform_width -> s1_s2      (basic_features on the side)
      records, wave_pool -> hit_threshold -> hit_merge_clusters -> hit_merged(merge_gap_ns=20) -> hit_grouped(100 ns)
* ``ReplayContext`` -- ``StraxContextAdapter`` over a Context: the reference's classes when they are importable, the
  restatements below (same calling convention) on the GPU box.
"""

from __future__ import annotations

import inspect
from typing import Any, Callable

import numpy as np

from .plugin_api import Option, Plugin, SimpleContext

try:  # inside a reference installation: its adapter classes drive the replay
    from waveform_analysis.core.plugins.core.adapters import (  # type: ignore
        StraxContextAdapter as _RefContextAdapter,
        StraxPluginAdapter as _RefPluginAdapter,
    )
except Exception:
    _RefContextAdapter = _RefPluginAdapter = None


class _StraxPluginAdapter(Plugin):
    """adapters.py:28-180: metadata from the strax-shaped class; compute() passes a dependency positionally when the
    parameter at its place has its name, by keyword otherwise, and only the config values compute() names."""

    def __init__(self, strax_plugin_class: type):
        self.strax_plugin_class = strax_plugin_class
        self.strax_plugin = strax_plugin_class()
        p = self.strax_plugin
        self.provides = getattr(p, "provides", "unknown")
        self.depends_on = getattr(p, "depends_on", ())
        self.dtype = getattr(p, "dtype", None)
        self.version = getattr(p, "__version__", "0.1.0")
        self.options = {}
        for item in getattr(p, "takes_config", ()):
            if isinstance(item, str):
                name, default = item, None
            elif isinstance(item, tuple):
                name, default = item[0], (item[1] if len(item) > 1 else None)
            else:
                continue
            self.options[name] = Option(default=default, help=f"Strax config option: {name}", track=True)
        self.data_kind = getattr(p, "data_kind", "unknown")
        self.compressor = getattr(p, "compressor", "blosc")
        self.parallel = getattr(p, "parallel", False)

    def compute(self, context: Any, run_id: str, **kwargs) -> Any:
        params = list(inspect.signature(self.strax_plugin.compute).parameters.keys())
        if params and params[0] == "self":
            params = params[1:]
        args, kw = [], {}
        for i, dep in enumerate(self.depends_on):
            name = dep if isinstance(dep, str) else dep[0]
            data = context.get_data(run_id, name)
            if i < len(params) and params[i] == name:
                args.append(data)
            else:
                kw[name] = data
        for key in self.config_keys:
            if key in params:
                kw[key] = context.get_config(self, key)
        return self.strax_plugin.compute(*args, **kw)

    def is_compatible(self) -> bool:
        return all(hasattr(self.strax_plugin, attr) for attr in ("provides", "compute"))


class _StraxContextAdapter:
    """adapters.py:251-440 (register / get_array / get_df / set_config)."""

    def __init__(self, context: Any):
        self.context = context

    def register(self, plugin_class):
        if isinstance(plugin_class, Plugin):
            self.context.register(plugin_class)
            return
        adapter = StraxPluginAdapter(plugin_class)
        if not adapter.is_compatible():
            raise ValueError(f"Incompatible strax plugin: {plugin_class}")
        self.context.register(adapter)

    def get_array(self, run_id: str, targets, **kwargs):
        if isinstance(targets, str):
            return self.context.get_data(run_id, targets, **kwargs)
        return {t: self.context.get_data(run_id, t, **kwargs) for t in targets}

    def get_df(self, run_id: str, targets, **kwargs):
        import pandas as pd

        def to_df(arr):
            if not isinstance(arr, np.ndarray):
                return arr
            if not arr.dtype.names:
                return pd.DataFrame({"data": arr})
            return pd.DataFrame({f: (arr[f].tolist() if arr[f].ndim > 1 else arr[f]) for f in arr.dtype.names})

        arrays = self.get_array(run_id, targets, **kwargs)
        return to_df(arrays) if isinstance(targets, str) else {t: to_df(a) for t, a in arrays.items()}

    def set_config(self, config: dict):
        self.context.set_config(config)


StraxPluginAdapter = _RefPluginAdapter or _StraxPluginAdapter
StraxContextAdapter = _RefContextAdapter or _StraxContextAdapter


class _ReplaySimpleContext(SimpleContext):
    def set_config(self, config: dict, plugin_name: str | None = None):
        if plugin_name:
            self.config.setdefault(plugin_name, {}).update(config)
        else:
            self.config.update(config)


def strax_shaped(plugin_factory: Callable[[], Any], provides: str, depends_on: tuple, config: dict, dtype=None,
                 fixed: dict | None = None, version: str = "0.1.0") -> type:
    """A strax-shaped class whose compute(self, <depends_on...>, <config...>) runs `plugin_factory()`'s own
    compute(context, run_id) on exactly those inputs.  `fixed`: configuration the chain pins (e.g. wave_source)."""
    dep_names = tuple(depends_on)
    cfg_names = tuple(config)

    def compute(self, *args, **kwargs):
        bound = dict(zip(dep_names, args))
        bound.update(kwargs)
        data = {k: bound[k] for k in dep_names}
        cfg = dict(fixed or {})
        cfg.update({k: bound[k] for k in cfg_names if k in bound and bound[k] is not None})
        plugin = plugin_factory()
        ctx = _ReplaySimpleContext({plugin.provides: cfg}, data)
        return plugin.compute(ctx, "replay")

    compute.__signature__ = inspect.Signature(
        [inspect.Parameter("self", inspect.Parameter.POSITIONAL_OR_KEYWORD)]
        + [inspect.Parameter(n, inspect.Parameter.POSITIONAL_OR_KEYWORD) for n in dep_names]
        + [inspect.Parameter(n, inspect.Parameter.POSITIONAL_OR_KEYWORD, default=config[n]) for n in cfg_names])
    return type(f"Strax_{provides}", (), {
        "provides": provides, "depends_on": dep_names, "dtype": dtype, "data_kind": provides,
        "takes_config": tuple((k, v) for k, v in config.items()), "__version__": version, "compute": compute,
        "__doc__": f"strax-shaped `{provides}` <- {dep_names} (config {cfg_names})"})


def c5_chain(plugins: dict) -> list:
    """The config-5 replay as strax-shaped classes.  `plugins`: name -> plugin class (or factory) for
    filtered_waveforms, basic_features, hit, waveform_width, s1_s2, hit_threshold, hit_merge_clusters, hit_merged,
    hit_merged_components, hit_grouped -- the HIP classes (`hip_c5_plugins()`) or the reference's CPU classes."""
    P = plugins
    return [
        # S1/S2 peak building on the dense rows (filtering.py:410-536, peak_finding.py:446-614, waveform_width.py:97-374,
        # s1_s2_classifier.py:133-228)
        strax_shaped(P["filtered_waveforms"], "filtered_waveforms", ("st_waveforms",), {}, fixed={"max_workers": 1}),
        strax_shaped(P["basic_features"], "basic_features", ("st_waveforms",), {"height_range": (40, 90), "area_range": (0, None)},
                     fixed={"wave_source": "st_waveforms"}),
        strax_shaped(P["hit"], "hit", ("filtered_waveforms",),
                     {"height": 8.0, "prominence": 0.5, "width": 2, "distance": 2},
                     fixed={"use_filtered": True}),
        strax_shaped(P["waveform_width"], "waveform_width", ("hit", "st_waveforms"), {"rise_low": 0.1, "rise_high": 0.9}),
        strax_shaped(P["s1_s2"], "s1_s2", ("waveform_width", "basic_features"),
                     {"s1_width_range": (0.0, 80.0), "s2_width_range": (80.0, 5000.0), "s1_area_range": None,
                      "s2_area_range": (500.0, 1e12)}),
        # threshold hits -> merge -> cross-channel grouping (hit_finder.py:329-413, hit_merge.py:115-322,
        # event_grouping.py:286-471)
        strax_shaped(P["hit_threshold"], "hit_threshold", ("records", "wave_pool"), {"threshold": 10.0},
                     fixed={"wave_source": "records"}),
        strax_shaped(P["hit_merge_clusters"], "hit_merge_clusters", ("hit_threshold",), {"merge_gap_ns": 20.0}),
        strax_shaped(P["hit_merged"], "hit_merged", ("hit_threshold", "hit_merge_clusters"), {"merge_gap_ns": 20.0}),
        strax_shaped(P["hit_merged_components"], "hit_merged_components", ("hit_merge_clusters", "hit_merged"), {}),
        strax_shaped(P["hit_grouped"], "hit_grouped", ("hit_merged", "hit_merged_components", "hit_threshold"),
                     {"time_window_ns": 100.0}),
    ]


C5_TARGETS = ("hit", "waveform_width", "s1_s2", "hit_threshold", "hit_merged", "hit_merged_components", "hit_grouped")
GROUPED_SCALARS = ("t_min", "t_max", "n_hits")
GROUPED_LISTS = ("dt", "boards", "channels", "heights", "integrals", "timestamps", "record_ids", "sample_starts", "sample_ends")


def flatten_grouped(df) -> dict:
    """The hit_grouped DataFrame (event_grouping.py:286-471) as plain arrays: per-event scalars and the list columns
    concatenated in event order (what fixtures store: data, no pickled objects)."""
    out = {f"grouped_{c}": df[c].to_numpy(dtype=np.int64) for c in GROUPED_SCALARS}
    for c in GROUPED_LISTS:
        out[f"grouped_{c}"] = np.concatenate(list(df[c])) if len(df) else np.zeros(0)
    return out


def hip_c5_plugins() -> dict:
    from .plugins.basic_features import HipBasicFeaturesPlugin
    from .plugins.filtered_waveforms import HipFilteredWaveformsPlugin
    from .plugins.hit_finder import HipHitFinderPlugin
    from .plugins.hit_grouped import HipHitGroupedPlugin
    from .plugins.hit_merge import HipHitMergeClustersPlugin, HipHitMergedComponentsPlugin, HipHitMergePlugin
    from .plugins.s1_s2 import HipS1S2ClassifierPlugin
    from .plugins.threshold_hit import HipThresholdHitPlugin
    from .plugins.waveform_width import HipWaveformWidthPlugin

    return {"filtered_waveforms": HipFilteredWaveformsPlugin, "basic_features": HipBasicFeaturesPlugin,
            "hit": HipHitFinderPlugin, "waveform_width": HipWaveformWidthPlugin, "s1_s2": HipS1S2ClassifierPlugin,
            "hit_threshold": HipThresholdHitPlugin, "hit_merge_clusters": HipHitMergeClustersPlugin,
            "hit_merged": HipHitMergePlugin, "hit_merged_components": HipHitMergedComponentsPlugin,
            "hit_grouped": HipHitGroupedPlugin}


class ReplayContext:
    """`create_strax_context` (adapters.py:413-440) for the replay: a Context behind a StraxContextAdapter, the run's
    inputs seeded as data.  `context`: a reference Context when given, the package's SimpleContext otherwise."""

    def __init__(self, inputs: dict, context: Any = None, run_id: str = "replay"):
        self.run_id = run_id
        self.context = context if context is not None else _ReplaySimpleContext()
        for name, value in inputs.items():
            self.context._set_data(run_id, name, value)
        self.strax = StraxContextAdapter(self.context)

    def register(self, strax_classes) -> "ReplayContext":
        for cls in strax_classes:
            self.strax.register(cls)
        return self

    def get_array(self, targets):
        return self.strax.get_array(self.run_id, targets)


def st_waveforms_from_records(records: np.ndarray, wave_pool: np.ndarray) -> np.ndarray:
    """Dense ST_WAVEFORM_DTYPE rows (processing/dtypes.py:36-64) of a uniform-length run: what the reference's
    st_waveforms stage hands the dense plugins."""
    from .dtypes import create_record_dtype

    L = int(records["event_length"][0])
    if not np.all(records["event_length"] == L):
        raise ValueError("st_waveforms_from_records needs records of one length")
    st = np.zeros(len(records), dtype=create_record_dtype(L))
    for f in ("baseline", "baseline_upstream", "polarity", "timestamp", "record_id", "dt", "event_length", "board", "channel"):
        if f in st.dtype.names and f in records.dtype.names:
            st[f] = records[f]
    off = records["wave_offset"].astype(np.int64)
    if np.array_equal(off, np.arange(len(records), dtype=np.int64) * L):
        st["wave"] = wave_pool[: len(records) * L].reshape(len(records), L).astype(np.int16)
    else:
        st["wave"] = wave_pool[off[:, None] + np.arange(L)[None, :]].astype(np.int16)
    return st


def mirror_positive(records: np.ndarray, wave_pool: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """The same run with every record mirrored around its pedestal, polarity "positive", baseline re-estimated from
    the first 40 samples.  The reference's width / S1-S2 stage measures pulses ABOVE the baseline
    (waveform_width.py:238-250 drops a peak whose corrected value is <= 0), so the config-5 replay runs on the
    positive-going twin of the SURVEY 8d generator's negative pulses."""
    rec = records.copy()
    rec["polarity"] = "positive"
    L = int(rec["event_length"][0])
    w = wave_pool.reshape(-1, L).astype(np.int64)
    ped = np.rint(rec["baseline"]).astype(np.int64)[:, None]
    flipped = np.clip(2 * ped - w, 0, 16383).astype(np.uint16)
    rec["baseline"] = flipped[:, :40].sum(axis=1, dtype=np.int64) / 40.0
    return rec, flipped.reshape(-1)


def replay_c5(records: np.ndarray, wave_pool: np.ndarray, plugins: dict | None = None, context: Any = None,
              config: dict | None = None) -> dict:
    """Run both config-5 chains through the strax-shaped driver; returns {target: table}."""
    inputs = {"records": records, "wave_pool": wave_pool, "st_waveforms": st_waveforms_from_records(records, wave_pool)}
    rc = ReplayContext(inputs, context=context).register(c5_chain(plugins or hip_c5_plugins()))
    if config:
        rc.strax.set_config(config)
    return rc.get_array(list(C5_TARGETS))


__all__ = ["StraxPluginAdapter", "StraxContextAdapter", "strax_shaped", "c5_chain", "C5_TARGETS", "hip_c5_plugins",
           "ReplayContext", "st_waveforms_from_records", "mirror_positive", "replay_c5", "flatten_grouped", "GROUPED_SCALARS", "GROUPED_LISTS"]
