"""HipThresholdHitPlugin -- drop-in for ThresholdHitPlugin
(reference: waveform_analysis/core/plugins/builtin/cpu/hit_finder.py:82-413)."""

from __future__ import annotations

from typing import Any

import numpy as np

from ..dtypes import THRESHOLD_HIT_DTYPE
from ..plugin_api import Option, Plugin
from ..sg_plan import normalize_window
from . import _common as K


class HipThresholdHitPlugin(Plugin):
    """Threshold-only hit detector with THRESHOLD_HIT_DTYPE output, computed on the GPU."""

    provides = "hit_threshold"
    depends_on = []  # dynamic, see resolve_depends_on
    description = "Threshold-only hit detector with THRESHOLD_HIT_DTYPE output (HIP, gfx950)."
    version = "0.11.0+hip1"
    output_dtype = THRESHOLD_HIT_DTYPE
    save_when = "always"

    options = {
        "threshold": Option(default=10.0, type=float, help="hit threshold"),
        "use_filtered": Option(default=False, type=bool, help="threshold the filtered waveform"),
        "fuse_filter": Option(
            default=False, type=bool,
            help="with use_filtered: evaluate the Savitzky-Golay filter inside the hit kernel from "
                 "wave_pool instead of reading a materialised wave_pool_filtered (same result)"),
        "fuse_baseline": Option(
            default=None,
            help="None, or (start, end): re-estimate records.baseline as the mean of samples "
                 "[start, end) inside the hit kernel (the records-builder rule)"),
        "wave_source": Option(default=K.WAVE_SOURCE_RECORDS, type=str, help="must be 'records'"),
        "left_extension": Option(default=2, type=int, help="samples added left of a hit"),
        "right_extension": Option(default=2, type=int, help="samples added right of a hit"),
        "dt": Option(default=None, type=int, help="sample interval (ns) when records lack dt"),
        "channel_config": Option(default=None, type=dict, help="per (board, channel) threshold"),
    }

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list[str]:
        deps, _pool = K.records_dependencies(context, self)
        return deps

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        threshold = float(context.get_config(self, "threshold"))
        le = max(0, int(context.get_config(self, "left_extension")))
        re = max(0, int(context.get_config(self, "right_extension")))
        explicit_dt = K.resolve_dt_config(context, self, deprecated_keys=("sampling_interval_ns", "dt_ns"))
        channel_config = context.get_config(self, "channel_config")
        use_filtered = bool(context.get_config(self, "use_filtered"))
        fused = use_filtered and bool(context.get_config(self, "fuse_filter"))
        fuse_baseline = context.get_config(self, "fuse_baseline")
        _deps, pool_name = K.records_dependencies(context, self)
        records, pool = K.load_records_input(context, self, run_id, pool_name)
        if len(records) == 0:
            return np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)

        dt_values = K.require_dt_array(records, explicit_dt=explicit_dt, plugin_name=self.provides,
                                       data_name="records")
        thresholds = K.per_record_channel_option(records, channel_config, run_id, "threshold",
                                                 threshold, threshold)
        rec = records
        if "dt" not in (records.dtype.names or ()):
            rec = _with_dt(records, dt_values)

        if use_filtered and not fused:
            if pool.dtype != np.float32:
                pool = np.asarray(pool, dtype=np.float32)
            sess = K.resident_session(context, pool)
            source = K.SRC_F32
        else:
            if pool.dtype != np.uint16:
                raise ValueError(f"wave_pool must be uint16, got {pool.dtype}")
            sess = K.resident_session(context, pool)
            source = K.SRC_SG_FUSED if fused else K.SRC_RAW
        sess.upload_records(rec, thresholds)
        if fused:
            fplugin = context.get_plugin("wave_pool_filtered") if "wave_pool_filtered" in getattr(context, "_plugins", {}) else None
            w = context.get_config(fplugin, "sg_window_size") if fplugin else 11
            p = context.get_config(fplugin, "sg_poly_order") if fplugin else 2
            sess.set_sg_plan(*normalize_window(w, p))
        if fused and fuse_baseline is not None:
            return sess.fused_baseline_filter_hits(tuple(fuse_baseline), le, re)
        if fuse_baseline is not None:
            sess.baseline_mean(int(fuse_baseline[0]), int(fuse_baseline[1]), update_records=True)
        return sess.threshold_hits(source, le, re)


def _with_dt(records: np.ndarray, dt_values: np.ndarray) -> np.ndarray:
    out = np.zeros(len(records), dtype=np.dtype(records.dtype.descr + [("dt", "i4")]))
    for n in records.dtype.names:
        out[n] = records[n]
    out["dt"] = dt_values
    return out
