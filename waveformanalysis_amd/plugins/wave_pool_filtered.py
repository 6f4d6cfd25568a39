"""HipWavePoolFilteredPlugin -- drop-in for WavePoolFilteredPlugin
(reference: waveform_analysis/core/plugins/builtin/cpu/records.py:334-438)."""

from __future__ import annotations

from typing import Any

import numpy as np

from ..plugin_api import Option, Plugin
from ..filter_engine import design_bw, plan_filter_groups, run_filter_groups  # noqa: F401 (design_bw re-exported)
from . import _common as K


class HipWavePoolFilteredPlugin(K.HipPlugin):
    """Build the float32 filtered wave_pool on the GPU (Savitzky-Golay, interpolated edges)."""

    provides = "wave_pool_filtered"
    algorithmic_bytes = (2 + 4, 0, 0)  # device pass: bytes per sample, per record, per output row
    depends_on = ["records", "wave_pool"]
    description = "Build filtered wave_pool from records-backed raw waveforms (HIP, gfx950)."
    version = "3.0.0+hip1"
    save_when = "always"
    output_dtype = np.dtype(np.float32)
    options = {
        "filter_type": Option(default="SG", type=str, help="'SG' (Savitzky-Golay) or 'BW' (Butterworth sosfiltfilt)"),
        "lowcut": Option(default=0.1, type=float, help="BW low cut"),
        "highcut": Option(default=0.5, type=float, help="BW high cut"),
        "fs": Option(default=0.5, type=float, help="BW sampling rate (GHz)"),
        "filter_order": Option(default=4, type=int, help="BW order"),
        "sg_window_size": Option(default=11, type=int, help="SG window (odd)"),
        "sg_poly_order": Option(default=2, type=int, help="SG polynomial order"),
        "max_workers": Option(default=None, type=int, help="ignored by the HIP backend", track=False),
        "batch_size": Option(default=0, type=int, help="ignored by the HIP backend (must be >= 0)"),
        "channel_config": Option(default=None, type=dict, help="per (board, channel) overrides"),
    }

    def compute(self, context: Any, run_id: str, **kwargs) -> np.ndarray:
        records = context.get_data(run_id, "records")
        wave_pool = context.get_data(run_id, "wave_pool")
        if not isinstance(records, np.ndarray):
            raise ValueError("wave_pool_filtered expects records as a structured array")
        if not isinstance(wave_pool, np.ndarray):
            raise ValueError("wave_pool_filtered expects wave_pool as a numpy array")
        if records.dtype.names is None:
            raise ValueError("wave_pool_filtered expects structured records input")
        missing = [n for n in ("wave_offset", "event_length") if n not in records.dtype.names]
        if missing:
            raise ValueError(f"wave_pool_filtered records missing required fields: {missing}")
        if len(records) == 0 or len(wave_pool) == 0:
            return np.zeros(len(wave_pool), dtype=np.float32)
        batch_size = int(context.get_config(self, "batch_size"))
        if batch_size < 0:
            raise ValueError(f"batch_size ({batch_size}) 必须大于等于 0")

        n = len(records)
        names = records.dtype.names
        boards = records["board"] if "board" in names else np.zeros(n, dtype=np.int16)
        channels = records["channel"] if "channel" in names else np.zeros(n, dtype=np.int16)
        groups = plan_filter_groups(context, self, run_id, boards, channels)

        off = records["wave_offset"].astype(np.int64)
        length = records["event_length"].astype(np.int64)
        bad = (length > 0) & ((off < 0) | (off + length > len(wave_pool)))
        if np.any(bad):
            i = int(np.flatnonzero(bad)[0])
            raise ValueError("wave_pool_filtered found out-of-bounds wave slice "
                             f"(offset={int(off[i])}, length={int(length[i])}, wave_pool_size={len(wave_pool)})")
        sess = K.resident_session(context, wave_pool if isinstance(wave_pool, np.ndarray) else np.asarray(wave_pool),
                                  cacheable=isinstance(wave_pool, np.ndarray))
        out = run_filter_groups(sess, _view_records(records), groups)  # (the filters drop the float32 tag themselves)
        return out


def _view_records(records: np.ndarray) -> np.ndarray:
    """WavePoolFilteredPlugin only needs offsets/lengths; fill what DeviceSession requires."""
    names = records.dtype.names
    need = ("record_id", "timestamp", "baseline")
    if all(n in names for n in need):
        return records
    out = np.zeros(len(records), dtype=[("wave_offset", "i8"), ("event_length", "i4"), ("record_id", "i8"),
                                        ("timestamp", "i8"), ("baseline", "f8")])
    out["wave_offset"] = records["wave_offset"]
    out["event_length"] = records["event_length"]
    out["record_id"] = records["record_id"] if "record_id" in names else np.arange(len(records))
    for n in ("timestamp", "baseline"):
        if n in names:
            out[n] = records[n]
    return out
