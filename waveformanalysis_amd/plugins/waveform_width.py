"""HipWaveformWidthPlugin -- drop-in for WaveformWidthPlugin
(reference: waveform_analysis/core/plugins/builtin/cpu/waveform_width.py:39-374).

Per `hit` row: baseline = mean of the first 50 samples of the waveform row the hit points at, first
crossings of the rise/fall fractions either side of the peak with linear interpolation, divided by the
sampling rate.  One GPU lane per hit (k_waveform_width); the row lookup and the id columns are table
work done here with numpy.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import dense
from ..dtypes import WAVEFORM_WIDTH_DTYPE
from ..plugin_api import Option, Plugin
from . import _common as K


def first_row_of_record_id(row_record_ids: np.ndarray, wanted: np.ndarray) -> np.ndarray:
    """Index of the FIRST row whose record_id equals each wanted id, -1 if none
    (waveform_width.py:163-167: np.flatnonzero(record_id == wanted)[0])."""
    ids, first = np.unique(np.asarray(row_record_ids, dtype=np.int64), return_index=True)
    wanted = np.asarray(wanted, dtype=np.int64)
    pos = np.searchsorted(ids, wanted)
    pos_c = np.minimum(pos, max(len(ids) - 1, 0))
    found = (pos < len(ids)) & (ids[pos_c] == wanted) if len(ids) else np.zeros(len(wanted), dtype=bool)
    return np.where(found, first[pos_c] if len(ids) else -1, -1).astype(np.int64)


class HipWaveformWidthPlugin(K.HipPlugin):
    """Rise / fall / total width per detected peak, computed on the GPU."""

    provides = "waveform_width"
    algorithmic_bytes = (0, 0, 56)  # device pass: bytes per sample, per record, per output row
    depends_on = []  # dynamic, see resolve_depends_on
    description = "Calculate rise/fall time based on peak detection results (HIP, gfx950)."
    version = "3.0.0+hip1"
    save_when = "always"
    output_dtype = WAVEFORM_WIDTH_DTYPE

    options = {
        "use_filtered": Option(default=False, type=bool, help="read filtered_waveforms instead of st_waveforms"),
        "sampling_rate": Option(default=None, type=float, help="sampling rate (GHz); 0.5 when unset"),
        "rise_low": Option(default=0.1, type=float, help="low fraction of the rise time"),
        "rise_high": Option(default=0.9, type=float, help="high fraction of the rise time"),
        "fall_high": Option(default=0.9, type=float, help="high fraction of the fall time"),
        "fall_low": Option(default=0.1, type=float, help="low fraction of the fall time"),
        "interpolation": Option(default=True, type=bool, help="linear interpolation of the crossings"),
    }

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list[str]:
        if context.get_config(self, "use_filtered"):
            return ["hit", "filtered_waveforms"]
        return ["hit", "st_waveforms"]

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        use_filtered = context.get_config(self, "use_filtered")
        sampling_rate = context.get_config(self, "sampling_rate")
        if sampling_rate is None:
            sampling_rate = 0.5
        # python floats, as Option(type=float) delivers them: numpy then keeps float32 rows in float32
        rise_low = float(context.get_config(self, "rise_low"))
        rise_high = float(context.get_config(self, "rise_high"))
        fall_high = float(context.get_config(self, "fall_high"))
        fall_low = float(context.get_config(self, "fall_low"))
        interpolation = bool(context.get_config(self, "interpolation"))

        hits = context.get_data(run_id, "hit")
        data_name = "filtered_waveforms" if use_filtered else "st_waveforms"
        waveform_data = context.get_data(run_id, data_name)
        if not isinstance(hits, np.ndarray):
            raise ValueError("waveform_width expects hit as a single structured array")
        if not isinstance(waveform_data, np.ndarray):
            raise ValueError("waveform_width expects st_waveforms as a single structured array")
        if len(hits) == 0 or len(waveform_data) == 0:
            return np.zeros(0, dtype=WAVEFORM_WIDTH_DTYPE)

        hit_names = hits.dtype.names or ()
        record_id = np.asarray(hits["record_id"] if "record_id" in hit_names else hits["event_index"], dtype=np.int64)
        position = np.asarray(hits["position"], dtype=np.int64)
        if np.any(position < 0):
            raise ValueError("waveform_width (HIP backend) requires hit positions >= 0")
        if "record_id" in (waveform_data.dtype.names or ()):
            row = first_row_of_record_id(waveform_data["record_id"], record_id)
        else:
            row = np.where((record_id >= 0) & (record_id < len(waveform_data)), record_id, -1)

        pool, source, L = dense.dense_pool(waveform_data, data_name)
        sess = K.resident_session(context, pool, cacheable=False)  # temporary of the dense `wave` field
        rows, valid = sess.waveform_width(source, position, row, len(waveform_data), L, rise_low, rise_high,
                                          fall_high, fall_low, float(sampling_rate), interpolation)
        out = rows[valid]
        sel = hits[valid]
        out["timestamp"] = sel["timestamp"]
        out["board"] = sel["board"] if "board" in hit_names else 0
        out["channel"] = sel["channel"]
        out["record_id"] = record_id[valid]
        return out
