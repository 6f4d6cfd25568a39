"""Host glue shared by the HIP plugins: input loading, dt rules, per-channel options, residency.

Mirrors (does not import) the reference helpers:
  cpu/_wave_source.py:93-227   wave_source / use_filtered -> dependencies and pool name
  cpu/_dt_compat.py:14-81      dt resolution and validation messages
  data/records_view.py:383-400 records_view() type checks
"""

from __future__ import annotations

import threading
import warnings
from typing import Any

import numpy as np

from .. import _lib
from ..channel_config import per_record_option, resolve_channel_values, scatter_per_record
from ..device import DeviceSession, default_pool
from ..plugin_api import Plugin

WAVE_SOURCE_AUTO = "auto"
WAVE_SOURCE_RECORDS = "records"
WAVE_SOURCES = {"auto", "records", "st_waveforms", "filtered_waveforms"}


def _cfg(context: Any, plugin: Any, name: str):
    """context.get_config(plugin, name); the option's default when `context` is not a Context (the reference's error
    manager calls resolve_depends_on with its own context DICT, core/foundation/error.py:95)."""
    get = getattr(context, "get_config", None)
    if get is None:
        opt = getattr(plugin, "options", {}).get(name)
        return getattr(opt, "default", None)
    return get(plugin, name)


def normalize_wave_source(value: Any) -> str:
    if value is None:
        return WAVE_SOURCE_AUTO
    source = str(value).strip().lower()
    if source not in WAVE_SOURCES:
        raise ValueError(f"Invalid wave_source: {value!r}. Expected one of {sorted(WAVE_SOURCES)}.")
    return source


def records_dependencies(context: Any, plugin: Any) -> tuple[list[str], str]:
    """Dependencies and pool name for the records source (cpu/_wave_source.py:93-140).

    The HIP plugins are records-backed: wave_source must be "records" (their default) --
    the dense st_waveforms / filtered_waveforms sources stay with the CPU plugins.
    """
    source = normalize_wave_source(_cfg(context, plugin, "wave_source"))
    if source != WAVE_SOURCE_RECORDS:
        raise ValueError(
            f"{plugin.provides} (HIP backend) reads records + wave_pool; set wave_source='records' "
            f"(got {source!r})."
        )
    use_filtered = bool(_cfg(context, plugin, "use_filtered")) if "use_filtered" in plugin.options else False
    fused = bool(_cfg(context, plugin, "fuse_filter")) if "fuse_filter" in plugin.options else False
    if use_filtered and not fused:
        return ["records", "wave_pool_filtered"], "wave_pool_filtered"
    return ["records", "wave_pool"], "wave_pool"


WAVE_SOURCE_ST = "st_waveforms"
WAVE_SOURCE_FILTERED = "filtered_waveforms"


def resolve_wave_input(context: Any, plugin: Any) -> tuple[str, list[str], str]:
    """(kind, depends_on, data_name) for plugins that implement the records AND the dense branches
    (cpu/_wave_source.py:74-172): kind is "records" or "dense"; data_name is the pool name for records,
    the structured-array name for dense.  An explicit wave_source wins; "auto" picks the dense array by
    use_filtered."""
    source = normalize_wave_source(_cfg(context, plugin, "wave_source"))
    use_filtered = bool(_cfg(context, plugin, "use_filtered")) if "use_filtered" in plugin.options else False
    if source == WAVE_SOURCE_RECORDS:
        pool = "wave_pool_filtered" if use_filtered else "wave_pool"
        return "records", [WAVE_SOURCE_RECORDS, pool], pool
    if source in (WAVE_SOURCE_ST, WAVE_SOURCE_FILTERED):
        if use_filtered:
            warnings.warn(f"Ignoring {plugin.provides}.use_filtered because wave_source={source} explicitly "
                          "selects data source.", stacklevel=3)
        return "dense", [source], source
    name = WAVE_SOURCE_FILTERED if use_filtered else WAVE_SOURCE_ST
    return "dense", [name], name


def load_dense_input(context: Any, plugin: Any, run_id: str, data_name: str) -> np.ndarray:
    """cpu/_wave_source.py:215-227."""
    plugins = getattr(context, "_plugins", None)
    if isinstance(plugins, dict) and plugins and data_name not in plugins and data_name not in getattr(context, "_data", {}):
        message = f"{plugin.provides} requires '{data_name}' but it is not registered."
        if data_name == WAVE_SOURCE_FILTERED:
            message += f" Register FilteredWaveformsPlugin to provide '{data_name}'."
        raise KeyError(message)
    data = context.get_data(run_id, data_name)
    if not isinstance(data, np.ndarray):
        raise ValueError(f"{plugin.provides} expects {data_name} as a single structured array")
    return data


def load_records_input(context: Any, plugin: Any, run_id: str, pool_name: str):
    plugins = getattr(context, "_plugins", None)
    if isinstance(plugins, dict) and plugins:
        for name, hint in (("records", "RecordsPlugin"),
                           (pool_name, "WavePoolFilteredPlugin" if pool_name == "wave_pool_filtered" else "WavePoolPlugin")):
            if name not in plugins and name not in getattr(context, "_data", {}):
                raise KeyError(f"{plugin.provides} requires '{name}' but it is not registered. "
                               f"Register {hint} to provide '{name}'.")
    records = context.get_data(run_id, "records")
    pool = context.get_data(run_id, pool_name)
    if not isinstance(records, np.ndarray):
        raise ValueError("records_view requires formal 'records' plugin output")
    if not isinstance(pool, np.ndarray):
        raise ValueError(f"records_view requires formal '{pool_name}' plugin output")
    if records.dtype.names is None:
        raise ValueError("records must be a structured array")
    return records, pool


def raw_config(context: Any, plugin: Any, name: str) -> Any:
    prov = plugin.provides
    block = context.config.get(prov)
    if isinstance(block, dict) and name in block:
        return block[name]
    if f"{prov}.{name}" in context.config:
        return context.config[f"{prov}.{name}"]
    return context.config.get(name)


def resolve_dt_config(context: Any, plugin: Any, deprecated_keys=()) -> Any:
    """cpu/_dt_compat.py:29-52."""
    dt = raw_config(context, plugin, "dt")
    if dt is not None:
        return dt
    for old in deprecated_keys:
        legacy = raw_config(context, plugin, old)
        if legacy is None:
            continue
        warnings.warn(f"[{plugin.provides}] Config '{old}' is deprecated and will be removed in a "
                      "future release. Use 'dt' instead.", DeprecationWarning, stacklevel=3)
        return legacy
    return None


def require_dt_array(data: np.ndarray, *, explicit_dt: Any, plugin_name: str, data_name: str) -> np.ndarray:
    """cpu/_dt_compat.py:55-81."""
    names = data.dtype.names or ()
    if "dt" in names:
        dt = np.asarray(data["dt"], dtype=np.int64)
        if np.any(dt <= 0):
            raise ValueError(f"[{plugin_name}] {data_name}.dt must be positive for every row")
        if np.any(dt > np.iinfo(np.int32).max):
            raise ValueError(f"[{plugin_name}] {data_name}.dt exceeds int32 range")
        return dt.astype(np.int32)
    if explicit_dt is None:
        raise ValueError(f"[{plugin_name}] Input '{data_name}' is missing required field 'dt'; "
                         "provide explicit config 'dt' for this migration period.")
    dt_scalar = int(explicit_dt)
    if dt_scalar <= 0:
        raise ValueError(f"[{plugin_name}] dt must be > 0")
    if dt_scalar > np.iinfo(np.int32).max:
        raise ValueError(f"[{plugin_name}] dt exceeds int32 range: {dt_scalar}")
    return np.full(len(data), dt_scalar, dtype=np.int32)


def per_record_channel_option(records: np.ndarray, channel_config: Any, run_id: str, name: str,
                              base_value: Any, default: float) -> np.ndarray:
    """Resolve one option per (board, channel) and scatter it to a per-record float64 array (a 0-d array when there is
    no per-channel configuration: every record gets the base value, nothing per record happens on the host)."""
    if not channel_config:
        return np.asarray(default if base_value is None else float(base_value), dtype=np.float64)
    n = len(records)
    names = records.dtype.names or ()
    boards = records["board"] if "board" in names else np.zeros(n, dtype=np.int16)
    channels = records["channel"] if "channel" in names else np.zeros(n, dtype=np.int16)
    # one sort of a 64-bit key instead of a row-wise unique + one mask per channel
    key = np.asarray(boards, dtype=np.int64) * 65536 + (np.asarray(channels, dtype=np.int64) & 0xFFFF)
    uniq, inverse = np.unique(key, return_inverse=True)
    values = np.empty(len(uniq), dtype=np.float64)
    for k, kv in enumerate(uniq):
        b, c = int(kv >> 16), int(np.int16(np.uint16(kv & 0xFFFF)))
        v = resolve_channel_values(channel_config, run_id, b, c, {name: base_value}).get(name, base_value)
        values[k] = default if v is None else float(v)
    return values[inverse.reshape(-1)]


# ---- residency: keep the pool of a run on the GPU between plugin calls -----------------------------
def resident_session(context: Any, pool: np.ndarray, pool_filtered: np.ndarray | None = None, *,
                     cacheable: bool = True) -> DeviceSession:
    """Session of this thread with `pool` on the device.

    The upload is skipped only when the session still holds this very array OBJECT (DeviceSession.ensure_pool:
    strong reference, compared with `is`, dropped by every call that replaces a device pool -- upload_pool,
    pool_gather, the filters, close).  Pass cacheable=False for temporaries: dense `wave` fields, astype / asarray
    copies.  Arrays handed out by a Context (`get_data` memoises its results) are the cacheable case."""
    pool_obj = getattr(context, "wfa_device_pool", None) or default_pool()
    sess = note_session(pool_obj.session())
    sess.ensure_pool(pool, cacheable=cacheable)
    if pool_filtered is not None:
        sess.ensure_filtered_pool(pool_filtered, cacheable=cacheable)
    return sess


def invalidate_residency(context: Any = None) -> None:
    """Drop what this thread's session believes to be resident (kept for callers that replace a pool through
    the session's own methods -- those reset the tags themselves)."""
    from .. import device as _device

    pool_obj = (getattr(context, "wfa_device_pool", None) if context is not None else None) or _device._default_pool
    sess = getattr(pool_obj._local, "session", None) if pool_obj is not None else None
    if sess is not None:
        sess.forget_resident()


SRC_RAW, SRC_F32, SRC_SG_FUSED = _lib.SRC_RAW, _lib.SRC_F32, _lib.SRC_SG_FUSED


# ---- the reference's profiling / statistics / cleanup hooks --------------------------------------
def _device_pool(context: Any):
    return getattr(context, "wfa_device_pool", None) or default_pool()


def _hooks(context: Any):
    """(profiler, stats collector) of a reference Context when they are switched on
    (core/context_execution.py:140-149, core/foundation/utils.py:92-207, core/plugins/core/stats.py:103-520)."""
    prof = getattr(context, "profiler", None)
    if prof is not None and not hasattr(prof, "timeit"):
        prof = None
    stats = getattr(context, "stats_collector", None)
    if stats is not None and not (hasattr(stats, "is_enabled") and stats.is_enabled()):
        stats = None
    return prof, stats


def publish_device_report(context: Any, plugin: Any, report: dict, n_samples: int, n_records: int, n_rows: int) -> dict:
    """Kernel times of one compute() into the Context's Profiler under `plugin.<name>.hip.<kernel>` (seconds, launches),
    and the rates the reference has no counter for -- Gsamples/s and algorithmic HBM GB/s of the device section --
    into `plugin.device_stats` and the stats collector (`hip_metrics[<name>]`, logged like its own records)."""
    prof, stats = _hooks(context)
    name = plugin.provides
    device_s = sum(ms for ms, _n in report.values()) / 1e3
    if prof is not None and hasattr(prof, "durations") and hasattr(prof, "counts"):
        for kernel, (ms, launches) in report.items():
            key = f"plugin.{name}.hip.{kernel}"
            prof.durations[key] += ms / 1e3
            prof.counts[key] += int(launches)
    bps, bpr, bprow = getattr(plugin, "algorithmic_bytes", (2, 29, 0))
    hbm_bytes = bps * n_samples + bpr * n_records + bprow * n_rows
    stats_row = {
        "samples": int(n_samples), "records": int(n_records), "rows": int(n_rows), "device_s": device_s,
        "gsamples_per_s": (n_samples / device_s / 1e9) if device_s > 0 else 0.0,
        "hbm_GBps_algorithmic": (hbm_bytes / device_s / 1e9) if device_s > 0 else 0.0,
        "kernels_ms": {k: ms for k, (ms, _n) in report.items()},
    }
    plugin.device_stats = stats_row
    if stats is not None:
        table = getattr(stats, "hip_metrics", None)
        if table is None:
            table = stats.hip_metrics = {}
        table.setdefault(name, []).append(stats_row)
        import logging

        logging.getLogger("waveform_analysis.core.plugins.core.stats").info(
            "Plugin '%s' device section: %.3f ms, %.1f Gsamples/s, %.0f GB/s (algorithmic HBM bytes)",
            name, device_s * 1e3, stats_row["gsamples_per_s"], stats_row["hbm_GBps_algorithmic"])
    return stats_row


_tls = threading.local()


def note_session(sess):
    """Called where a plugin takes its device session: while an instrumented compute() runs on this thread, the first
    session it touches gets its kernel timers switched on (HIP events around every launch, resolved when read)."""
    frame = getattr(_tls, "frame", None)
    if frame is not None and frame["sess"] is None and hasattr(sess, "profile") and hasattr(sess, "profile_report"):
        sess.profile(True)
        frame["sess"] = sess
    return sess


def _instrument(fn):
    import functools

    @functools.wraps(fn)
    def compute(self, context, run_id, **kwargs):
        self._wfa_failed = False
        prof, stats = _hooks(context)
        if prof is None and stats is None:
            return fn(self, context, run_id, **kwargs)
        outer = getattr(_tls, "frame", None)  # a plugin that pulls a dependency through the Context nests compute() calls
        frame = _tls.frame = {"sess": None}
        try:
            if prof is not None:
                with prof.timeit(f"plugin.{self.provides}.hip"):
                    result = fn(self, context, run_id, **kwargs)
            else:
                result = fn(self, context, run_id, **kwargs)
        finally:
            _tls.frame = outer
            sess, report = frame["sess"], None
            if sess is not None:
                try:
                    report = sess.profile_report()
                    sess.profile(False)
                except Exception:  # the failure that brought us here is the one to report
                    report = None
        if report:
            n_rows = len(result) if hasattr(result, "__len__") else 0
            publish_device_report(context, self, report, getattr(sess, "n_samples", 0), getattr(sess, "n_records", 0), n_rows)
        return result

    compute._wfa_instrumented = True
    return compute


class HipPlugin(Plugin):
    """Base of the plugins that run on the device: the Context's hooks around compute().

    * profiler / stats (core/context_execution.py:146-149): when the Context has a Profiler or an enabled
      PluginStatsCollector, the kernels of this compute() are timed with HIP events and published
      (`publish_device_report`);
    * cleanup(context) (core/plugins/core/base.py:608-613, always called after compute()): the device scratch the
      next call rebuilds by itself is freed (`wfa_release_scratch`), the resident pool / records / rows stay; after a
      failed compute() (`on_error`, base.py:602-606) the thread's session is closed: its device state is not trusted.
    algorithmic_bytes = (per sample, per record, per output row) of the plugin's device pass."""

    algorithmic_bytes = (2, 29, 0)

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        fn = cls.__dict__.get("compute")
        if fn is not None and not getattr(fn, "_wfa_instrumented", False):
            cls.compute = _instrument(fn)

    def on_error(self, context: Any, exception: Exception) -> None:
        self._wfa_failed = True

    def cleanup(self, context: Any) -> None:
        from .. import device as _device

        pool_obj = getattr(context, "wfa_device_pool", None) or _device._default_pool  # never creates a pool here
        peek = getattr(pool_obj, "peek_session", None)
        sess = peek() if peek is not None else None
        if sess is None:
            return
        if getattr(self, "_wfa_failed", False):
            self._wfa_failed = False
            pool_obj.drop_session()
            return
        sess.release_scratch()
