"""HipHitFinderPlugin -- drop-in for HitFinderPlugin, records source
(reference: waveform_analysis/core/plugins/builtin/cpu/peak_finding.py:49-614).

The reference runs scipy.signal.find_peaks per record on -rv.signals(record) (or its first
difference) and turns each surviving peak into a HIT_DTYPE row.  Here one GPU lane per record runs
the same algorithm (k_find_peaks); the rows come back in (record, position) order,
which is the order the reference's per-record loop appends them in.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import dense
from ..dtypes import HIT_DTYPE
from ..plugin_api import Option, Plugin
from . import _common as K


class HipHitFinderPlugin(K.HipPlugin):
    """find_peaks-based hit detector with HIT_DTYPE output, computed on the GPU."""

    provides = "hit"
    algorithmic_bytes = (4, 29, 48)  # device pass: bytes per sample, per record, per output row
    depends_on = []  # dynamic, see resolve_depends_on
    description = "Detect peaks in waveforms and extract peak features (HIP, gfx950)."
    version = "3.0.0+hip1"
    save_when = "always"
    output_dtype = HIT_DTYPE

    options = {
        "use_filtered": Option(default=True, type=bool, help="detect on wave_pool_filtered"),
        "wave_source": Option(default=K.WAVE_SOURCE_AUTO, type=str,
                              help="auto|records|st_waveforms|filtered_waveforms"),
        "use_derivative": Option(default=True, type=bool, help="detect on the first difference"),
        "height": Option(default=30.0, type=float, help="minimum peak height"),
        "distance": Option(default=2, type=int, help="minimum distance between peaks (samples)"),
        "prominence": Option(default=0.7, type=float, help="minimum prominence"),
        "width": Option(default=4, type=int, help="minimum width at half prominence (samples)"),
        "threshold": Option(default=None, help="minimum vertical distance to the neighbours (optional)"),
        "height_method": Option(default="minmax", type=str, help="'diff' or 'minmax'"),
        "height_window_extension": Option(default=4, type=int, help="samples added either side of the peak window"),
        "dt": Option(default=None, type=int, help="sample interval (ns) when records lack dt"),
        # accepted for config compatibility; the GPU pass has no thread pool to size
        "parallel": Option(default=True, type=bool, help="ignored (kept for config compatibility)"),
        "n_workers": Option(default=0, type=int, help="ignored"),
        "chunk_size": Option(default=1024, type=int, help="ignored"),
        "parallel_min_events": Option(default=20480, type=int, help="ignored"),
    }

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list[str]:
        _kind, deps, _name = K.resolve_wave_input(context, self)
        return deps

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        use_derivative = bool(context.get_config(self, "use_derivative"))
        height = float(context.get_config(self, "height"))
        distance = int(context.get_config(self, "distance"))
        prominence = float(context.get_config(self, "prominence"))
        width = int(context.get_config(self, "width"))  # peak_finding.py:200 truncates to int
        threshold = context.get_config(self, "threshold")
        height_method = str(context.get_config(self, "height_method"))
        ext = int(context.get_config(self, "height_window_extension"))
        explicit_dt = K.resolve_dt_config(context, self, deprecated_keys=("sampling_interval_ns", "dt_ns"))
        kind, _deps, pool_name = K.resolve_wave_input(context, self)
        if threshold is not None and not np.isscalar(threshold):
            raise ValueError("hit (HIP backend) takes a scalar threshold (a lower bound), or None")
        peak_kw = dict(use_derivative=use_derivative, height=height, distance=distance, prominence=prominence,
                       width=width, threshold=None if threshold is None else float(threshold),
                       height_method=height_method, height_window_extension=ext)
        if kind == "dense":
            return self._compute_dense(context, run_id, pool_name, explicit_dt, peak_kw)
        records, pool = K.load_records_input(context, self, run_id, pool_name)
        if len(records) == 0:
            return np.zeros(0, dtype=HIT_DTYPE)
        if height_method not in ("minmax", "diff"):
            raise ValueError(f"不支持的峰高计算方法: {height_method}")  # peak_finding.py:612

        names = records.dtype.names or ()
        if "dt" in names:
            dt_values = np.asarray(records["dt"], dtype=np.int64)
            if np.any(dt_values <= 0):
                raise ValueError("[hit] dt must be > 0")  # peak_finding.py:542
            if np.any(dt_values > np.iinfo(np.int32).max):
                raise ValueError(f"[hit] dt exceeds int32 range: {int(dt_values.max())}")
        elif explicit_dt is None:
            raise ValueError("[hit] records is missing required field 'dt'; provide explicit config 'dt'.")
        else:
            dt_values = K.require_dt_array(records, explicit_dt=explicit_dt, plugin_name="hit", data_name="records")

        rec = _records_for_upload(records, dt_values)
        converted = False
        if pool_name == "wave_pool_filtered":
            if pool.dtype != np.float32:
                pool, converted = np.asarray(pool, dtype=np.float32), True
            source = K.SRC_F32
        else:
            if pool.dtype != np.uint16:
                raise ValueError(f"wave_pool must be uint16, got {pool.dtype}")
            source = K.SRC_RAW
        sess = K.resident_session(context, pool, cacheable=not converted)
        sess.upload_records(rec, np.zeros(len(rec), dtype=np.float64))
        return sess.find_peaks(source, **peak_kw)

    def _compute_dense(self, context, run_id, data_name, explicit_dt, peak_kw) -> np.ndarray:
        """peak_finding.py:316-378: the row (cut at event_length) is the waveform, pulses are negative-going."""
        data = K.load_dense_input(context, self, run_id, data_name)
        if len(data) == 0:
            return np.zeros(0, dtype=HIT_DTYPE)
        if peak_kw["height_method"] not in ("minmax", "diff"):
            raise ValueError(f"不支持的峰高计算方法: {peak_kw['height_method']}")
        names = data.dtype.names or ()
        if "dt" not in names and explicit_dt is None:
            raise ValueError("[hit] st_waveforms is missing required field 'dt'; provide explicit config 'dt'.")
        dt_values = np.asarray(data["dt"], dtype=np.int64) if "dt" in names else np.full(len(data), int(explicit_dt), np.int64)
        if np.any(dt_values <= 0):
            raise ValueError("[hit] dt must be > 0")
        if np.any(dt_values > np.iinfo(np.int32).max):
            raise ValueError(f"[hit] dt exceeds int32 range: {int(dt_values.max())}")
        if "baseline" not in names and not peak_kw["use_derivative"]:
            raise ValueError(f"hit (HIP backend) needs a 'baseline' field on {data_name} when use_derivative=False")
        pool, source, L = dense.dense_pool(data, data_name)
        rec = dense.dense_records(data, L, keep_record_id=True, truncate_to_event_length=True)
        rec["dt"] = dt_values
        sess = K.resident_session(context, pool, cacheable=False)  # `pool` is a temporary of the dense `wave` field
        sess.upload_records(rec, np.zeros(len(rec), dtype=np.float64))
        return sess.find_peaks(source, dense_rows=True, **peak_kw)


def _records_for_upload(records: np.ndarray, dt_values: np.ndarray) -> np.ndarray:
    """Rows as the reference loop sees them: metadata of row i, waveform of row record_id[i].

    peak_finding.py:401-407 fetches the signal with rv.signals(record_id) -- an index into records --
    while timestamp/board/channel/dt come from records[i].  With the usual record_id == arange(n) the
    two coincide; otherwise the waveform fields are taken from the row record_id points at.
    """
    names = records.dtype.names or ()
    need_dt = "dt" not in names
    descr = records.dtype.descr + ([("dt", "i4")] if need_dt else [])
    n = len(records)
    rid = np.asarray(records["record_id"], dtype=np.int64) if "record_id" in names else np.arange(n, dtype=np.int64)
    if not need_dt and np.array_equal(rid, np.arange(n, dtype=np.int64)):
        return records
    out = np.zeros(n, dtype=np.dtype(descr))
    for name in names:
        out[name] = records[name]
    if need_dt:
        out["dt"] = dt_values
    if not np.array_equal(rid, np.arange(n, dtype=np.int64)):
        if np.any(rid < -n) or np.any(rid >= n):
            raise IndexError(f"index {int(rid[(rid < -n) | (rid >= n)][0])} is out of bounds for axis 0 with size {n}")
        for name in ("wave_offset", "event_length", "baseline", "polarity"):
            if name in names:
                out[name] = records[name][rid]
    return out
