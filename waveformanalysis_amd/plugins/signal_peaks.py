"""HipSignalPeaksStreamPlugin -- drop-in for SignalPeaksStreamPlugin
(reference: waveform_analysis/core/plugins/builtin/streaming/cpu/signal_peaks.py:36-406).

The reference walks `st_waveforms` + `filtered_waveforms` per channel, cuts each channel at dt changes and time
breaks into chunks of `chunk_size` events and runs scipy.signal.find_peaks on every filtered row (converted to
float64).  Here the chunk rule is the same host code; compute_chunk uploads the chunk's rows and runs the find_peaks
kernels in their float64-row mode (WFA_PEAK_SIGNAL_ROWS_F64), one lane per row / per candidate.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import dense
from ..chunk import Chunk, get_endtime
from ..device import default_pool
from ..dtypes import HIT_DTYPE
from ..plugin_api import Option
from ..streaming import HipStreamingPlugin
from . import _common as K

TIMESTAMP_FIELD = "timestamp"
EVENT_LENGTH_FIELD = "event_length"


class HipSignalPeaksStreamPlugin(HipStreamingPlugin):
    """Stream peak detection from filtered waveforms (HIP, gfx950)."""

    provides = "signal_peaks_stream"
    depends_on = ["filtered_waveforms", "st_waveforms"]
    description = "Stream peak detection from filtered waveforms (HIP, gfx950)."
    version = "1.2.0+hip1"
    save_when = "never"
    output_dtype = None

    output_time_field = TIMESTAMP_FIELD
    output_endtime_field = "endtime"
    output_data_kind = "peaks"
    required_halo_ns = 0
    clip_strict = False
    is_stateful = False
    chunk_size = 4096
    parallel = True
    executor_type = "thread"  # the reference uses a process pool; here workers are threads, one HIP stream each
    max_workers = None

    options = {
        "use_derivative": Option(default=True, type=bool, help="detect on the first difference"),
        "height": Option(default=30.0, type=float, help="minimum peak height"),
        "distance": Option(default=2, type=int, help="minimum distance between peaks (samples)"),
        "prominence": Option(default=0.7, type=float, help="minimum prominence"),
        "width": Option(default=4, type=int, help="minimum width (samples)"),
        "threshold": Option(default=None, help="minimum vertical distance to the neighbours (optional)"),
        "height_method": Option(default="diff", type=str, help="'diff' or 'minmax'"),
        "minmax_window_expand": Option(default=2, type=int, help="samples added either side for 'minmax'"),
        "dt": Option(default=None, type=int, help="sample interval (ns) when the input lacks dt"),
    }

    def compute(self, context: Any, run_id: str, **kwargs):
        self._load_config(context)
        return super().compute(context, run_id, **kwargs)

    def _load_config(self, context: Any) -> None:
        self.use_derivative = context.get_config(self, "use_derivative")
        self.height = context.get_config(self, "height")
        self.distance = context.get_config(self, "distance")
        self.prominence = context.get_config(self, "prominence")
        self.width = context.get_config(self, "width")
        self.threshold = context.get_config(self, "threshold")
        self.height_method = context.get_config(self, "height_method")
        self.minmax_window_expand = max(0, int(context.get_config(self, "minmax_window_expand")))
        self.explicit_dt = K.resolve_dt_config(context, self, deprecated_keys=("sampling_interval_ns", "dt_ns"))
        self.time_field = TIMESTAMP_FIELD
        self.length_field = EVENT_LENGTH_FIELD
        self.dt_field = "dt"
        self.endtime_field = "endtime"

    def _get_input_chunks(self, context: Any, run_id: str, **kwargs):
        """signal_peaks.py:113-224: per channel, per dt segment, per time segment, `chunk_size` events per chunk."""
        filtered_waveforms = context.get_data(run_id, "filtered_waveforms")
        st_waveforms = context.get_data(run_id, "st_waveforms")
        if not isinstance(filtered_waveforms, np.ndarray) or not isinstance(st_waveforms, np.ndarray):
            raise ValueError("signal_peaks_stream expects st_waveforms as a single array")
        if len(filtered_waveforms) == 0 or len(st_waveforms) == 0:
            return
        dt_values = K.require_dt_array(st_waveforms, explicit_dt=self.explicit_dt, plugin_name=self.provides,
                                       data_name="st_waveforms")
        if "channel" not in (st_waveforms.dtype.names or ()):
            raise ValueError("st_waveforms missing required 'channel' field for streaming peaks")
        channels = st_waveforms["channel"]
        for ch_idx in np.unique(channels):
            mask = channels == ch_idx
            st_ch, filtered_ch, dt_ch = st_waveforms[mask], filtered_waveforms[mask], dt_values[mask]
            if len(st_ch) == 0 or len(filtered_ch) == 0:
                continue
            if TIMESTAMP_FIELD not in st_ch.dtype.names:
                raise KeyError(f"st_waveforms 缺少时间字段: {TIMESTAMP_FIELD}")
            n_events = min(len(filtered_ch), len(st_ch))
            st_ch, filtered_ch, dt_ch = st_ch[:n_events], filtered_ch[:n_events], dt_ch[:n_events]
            times = st_ch[TIMESTAMP_FIELD]
            dt_bounds = np.concatenate([[0], np.where(dt_ch[1:] != dt_ch[:-1])[0] + 1, [n_events]])
            segment_id = 0
            for a, b in zip(dt_bounds[:-1], dt_bounds[1:]):
                if b <= a:
                    segment_id += 1
                    continue
                st_dt, filtered_dt, dt_times = st_ch[a:b], filtered_ch[a:b], times[a:b]
                chunk_dt_ps = float(int(dt_ch[a]) * 1e3)
                kw = dict(time_field=TIMESTAMP_FIELD, length_field=EVENT_LENGTH_FIELD, dt=chunk_dt_ps)
                if self.break_threshold_ps and self.break_threshold_ps > 0 and len(st_dt) > 1:
                    endtime_all = get_endtime(st_dt, **kw)
                    gaps = dt_times[1:].astype(np.int64) - endtime_all[:-1].astype(np.int64)
                    bounds = np.concatenate([[0], np.where(gaps > self.break_threshold_ps)[0] + 1, [len(st_dt)]])
                else:
                    bounds = np.array([0, len(st_dt)], dtype=np.int64)
                for s0, s1 in zip(bounds[:-1], bounds[1:]):
                    if s1 <= s0:
                        segment_id += 1
                        continue
                    for start in range(int(s0), int(s1), self.chunk_size):
                        end = min(int(s1), start + self.chunk_size)
                        st_chunk = st_dt[start:end]
                        if len(st_chunk) == 0:
                            continue
                        main_start = int(np.min(st_chunk[TIMESTAMP_FIELD]))
                        main_end = int(np.max(get_endtime(st_chunk, **kw)))
                        yield Chunk(st_chunk, main_start, main_end, run_id=run_id, data_type=self.provides,
                                    time_field=TIMESTAMP_FIELD, length_field=EVENT_LENGTH_FIELD, dt=chunk_dt_ps,
                                    metadata={"filtered_waveforms": filtered_dt[start:end],
                                              "event_offset": int(a) + start, "channel_index": ch_idx,
                                              "main_start": main_start, "main_end": main_end,
                                              "segment_id": segment_id})
                    segment_id += 1

    def compute_chunk(self, chunk: Chunk, context: Any, run_id: str, **kwargs):
        st_chunk = chunk.data
        filtered_chunk = chunk.metadata.get("filtered_waveforms")
        if filtered_chunk is None or len(st_chunk) == 0:
            return None
        if self.height_method not in ("diff", "minmax"):
            raise ValueError(f"不支持的峰高计算方法: {self.height_method}")
        names = st_chunk.dtype.names or ()
        n = min(len(filtered_chunk), len(st_chunk))  # zip(..., strict=False)
        wave = filtered_chunk["wave"] if filtered_chunk.dtype.names and "wave" in filtered_chunk.dtype.names else filtered_chunk
        wave = np.asarray(wave)[:n]
        if wave.ndim == 1:
            wave = wave.reshape(n, -1)
        pool, source, L = dense.matrix_pool(wave, "filtered_waveforms")
        rec = np.zeros(n, dtype=dense.DENSE_RECORD_DTYPE)
        rec["timestamp"] = st_chunk[TIMESTAMP_FIELD][:n]
        rec["channel"] = st_chunk["channel"][:n]
        rec["board"] = st_chunk["board"][:n] if "board" in names else 0
        if "baseline" in names:
            rec["baseline"] = st_chunk["baseline"][:n]
        elif not self.use_derivative:
            raise ValueError("signal_peaks_stream (HIP backend) needs st_waveforms.baseline when use_derivative=False")
        rec["dt"] = st_chunk["dt"][:n] if "dt" in names else int(self.explicit_dt)
        if np.any(rec["dt"] <= 0):
            raise ValueError("[signal_peaks_stream] dt must be > 0")
        event_offset = int(chunk.metadata.get("event_offset", 0))
        rec["record_id"] = st_chunk["record_id"][:n] if "record_id" in names else event_offset + np.arange(n)
        rec["polarity"] = "negative"
        rec["wave_offset"] = np.arange(n, dtype=np.int64) * L
        rec["event_length"] = L
        with self._pool(context).borrow() as sess:
            sess.upload_pool(pool)
            sess.upload_records(rec, 0.0)
            peaks = sess.find_peaks(source, use_derivative=bool(self.use_derivative), height=float(self.height),
                                    distance=int(self.distance), prominence=float(self.prominence),
                                    width=float(self.width),
                                    threshold=None if self.threshold is None else float(self.threshold),
                                    height_method=self.height_method,
                                    height_window_extension=self.minmax_window_expand, dense_rows=2)
        if len(peaks) == 0:
            return None
        return Chunk(peaks, int(np.min(peaks["timestamp"])), int(np.max(peaks["timestamp"])), run_id=run_id,
                     data_type=self.provides, data_kind=self.output_data_kind, time_field=TIMESTAMP_FIELD)
