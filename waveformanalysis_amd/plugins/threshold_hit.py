"""HipThresholdHitPlugin -- drop-in for ThresholdHitPlugin
(reference: waveform_analysis/core/plugins/builtin/cpu/hit_finder.py:82-413)."""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import dense
from ..dtypes import THRESHOLD_HIT_DTYPE
from ..plugin_api import Option, Plugin
from ..sg_plan import normalize_window
from . import _common as K


class HipThresholdHitPlugin(K.HipPlugin):
    """Threshold-only hit detector with THRESHOLD_HIT_DTYPE output, computed on the GPU."""

    provides = "hit_threshold"
    algorithmic_bytes = (2, 29, 60)  # device pass: bytes per sample, per record, per output row
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
        "wave_source": Option(default=K.WAVE_SOURCE_AUTO, type=str,
                              help="auto|records|st_waveforms|filtered_waveforms"),
        "left_extension": Option(default=2, type=int, help="samples added left of a hit"),
        "right_extension": Option(default=2, type=int, help="samples added right of a hit"),
        "dt": Option(default=None, type=int, help="sample interval (ns) when records lack dt"),
        "channel_config": Option(default=None, type=dict, help="per (board, channel) threshold"),
    }

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list[str]:
        kind, deps, _name = K.resolve_wave_input(context, self)
        if kind == "dense":
            return deps
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
        kind, _deps, data_name = K.resolve_wave_input(context, self)
        if kind == "dense":
            return self._compute_dense(context, run_id, data_name, threshold, le, re, explicit_dt, channel_config)
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
            converted = pool.dtype != np.float32
            if converted:
                pool = np.asarray(pool, dtype=np.float32)
            sess = K.resident_session(context, pool, cacheable=not converted)
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

    def _compute_dense(self, context, run_id, data_name, threshold, le, re, explicit_dt, channel_config) -> np.ndarray:
        """hit_finder.py:179-255: the whole row is searched; the records/wave_pool length of the same record_id
        (hit_finder.py:257-286) only clamps the reported edges."""
        data = K.load_dense_input(context, self, run_id, data_name)
        if len(data) == 0:
            return np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)
        names = data.dtype.names or ()
        pool, source, L = dense.dense_pool(data, data_name)
        rec = dense.dense_records(data, L, keep_record_id=True)
        rec["dt"] = K.require_dt_array(data, explicit_dt=explicit_dt, plugin_name=self.provides, data_name=data_name)
        if "event_length" in names:
            source_lengths = np.asarray(data["event_length"], dtype=np.int64)
        else:
            source_lengths = np.full(len(data), L, dtype=np.int64)
        record_lengths = _lengths_from_records(context, run_id, rec["record_id"], source_lengths)
        thresholds = K.per_record_channel_option(rec, channel_config, run_id, "threshold", threshold, threshold)
        sess = K.resident_session(context, pool, cacheable=False)  # temporary of the dense `wave` field
        sess.upload_records(rec, thresholds)
        if "baseline" not in names:
            if source != K.SRC_RAW:
                raise ValueError(f"hit_threshold (HIP backend) needs a 'baseline' field on float32 {data_name}")
            sess.baseline_mean(0, L, update_records=True)  # waves.mean(axis=1): exact integer sum / L
        hits = sess.threshold_hits(source, le, re)
        if len(hits) and np.any(record_lengths < L):
            order = np.argsort(rec["record_id"], kind="stable")
            row = order[np.searchsorted(rec["record_id"][order], hits["record_id"])]
            limit = np.maximum(record_lengths[row], 0).astype(np.int32)
            edge_start = np.minimum(hits["edge_start"], limit)
            edge_end = np.maximum(np.minimum(hits["edge_end"], limit), edge_start)
            hits["edge_start"], hits["edge_end"] = edge_start, edge_end
            hits["width"] = (edge_end - edge_start).astype(np.float32)
        return hits


def _lengths_from_records(context: Any, run_id: str, record_ids: np.ndarray, source_lengths: np.ndarray) -> np.ndarray:
    """hit_finder.py:257-286 (`_resolve_wave_pool_metadata`): every dense row must exist in records with the same
    length; returns the records-side event_length per row."""
    records = context.get_data(run_id, "records")
    if not isinstance(records, np.ndarray) or records.dtype.names is None:
        raise ValueError("hit_threshold needs the 'records' table to resolve record_id into records/wave_pool")
    rid = np.asarray(records["record_id"], dtype=np.int64)
    order = np.argsort(rid, kind="stable")
    # duplicates: the reference's dict keeps the LAST row of an id
    pos = np.searchsorted(rid[order], record_ids, side="right") - 1
    found = (pos >= 0) & (rid[order][np.maximum(pos, 0)] == record_ids) if len(rid) else np.zeros(len(record_ids), bool)
    if not np.all(found):
        bad = int(record_ids[np.flatnonzero(~found)[0]])
        raise ValueError(f"hit_threshold could not resolve record_id={bad} into records/wave_pool")
    lengths = np.asarray(records["event_length"], dtype=np.int64)[order][pos]
    diff = np.flatnonzero(lengths != source_lengths)
    if len(diff):
        i = int(diff[0])
        raise ValueError("hit_threshold waveform source length does not match records/wave_pool length for "
                         f"record_id={int(record_ids[i])}: source={int(source_lengths[i])}, records={int(lengths[i])}")
    return lengths


def _with_dt(records: np.ndarray, dt_values: np.ndarray) -> np.ndarray:
    out = np.zeros(len(records), dtype=np.dtype(records.dtype.descr + [("dt", "i4")]))
    for n in records.dtype.names:
        out[n] = records[n]
    out["dt"] = dt_values
    return out
