"""Structured dtypes of the per-record waveform hot path.

These are the on-disk / in-memory layouts the reference plugins exchange; the
HIP kernels write rows with exactly these packed layouts so a result buffer can
be viewed as the structured array without a repack.

Reference layouts:
  RECORDS_DTYPE                 waveform_analysis/core/processing/dtypes.py:80-100 (102 B)
  THRESHOLD_HIT_DTYPE           waveform_analysis/core/plugins/builtin/cpu/hit_finder.py:33-49 (60 B)
  BASIC_FEATURES_DTYPE          waveform_analysis/core/plugins/builtin/cpu/basic_features.py:29-40 (36 B)
  WAVEFORM_WIDTH_INTEGRAL_DTYPE waveform_analysis/core/plugins/builtin/cpu/waveform_width_integral.py:25-39 (52 B)
"""

from __future__ import annotations

import numpy as np

RECORDS_DTYPE = np.dtype(
    [
        ("timestamp", "i8"),
        ("pid", "i4"),
        ("board", "i2"),
        ("channel", "i2"),
        ("baseline", "f8"),
        ("baseline_upstream", "f8"),
        ("polarity", "U8"),
        ("record_id", "i8"),
        ("dt", "i4"),
        ("trigger_type", "i2"),
        ("flags", "u4"),
        ("wave_offset", "i8"),
        ("event_length", "i4"),
        ("time", "i8"),
    ]
)

THRESHOLD_HIT_DTYPE = np.dtype(
    [
        ("position", "i8"),
        ("height", "f4"),
        ("integral", "f4"),
        ("edge_start", "i4"),
        ("edge_end", "i4"),
        ("width", "f4"),
        ("dt", "i4"),
        ("rise_time", "f4"),
        ("fall_time", "f4"),
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("record_id", "i8"),
    ]
)

BASIC_FEATURES_DTYPE = np.dtype(
    [
        ("height", "f4"),
        ("amp", "f4"),
        ("area", "f4"),
        ("max_abs_diff", "f4"),
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("event_index", "i8"),
    ]
)

WAVEFORM_WIDTH_INTEGRAL_DTYPE = np.dtype(
    [
        ("t_low", "f4"),
        ("t_high", "f4"),
        ("width", "f4"),
        ("t_low_samples", "f4"),
        ("t_high_samples", "f4"),
        ("width_samples", "f4"),
        ("q_total", "f8"),
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("event_index", "i8"),
    ]
)

# HIT_DTYPE of the find_peaks-based detector (reference: cpu/peak_finding.py:30-43, 48 B)
HIT_DTYPE = np.dtype(
    [
        ("position", "i8"),
        ("height", "f4"),
        ("integral", "f4"),
        ("edge_start", "f4"),
        ("edge_end", "f4"),
        ("dt", "i4"),
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("record_id", "i8"),
    ]
)

assert HIT_DTYPE.itemsize == 48
assert RECORDS_DTYPE.itemsize == 102
assert THRESHOLD_HIT_DTYPE.itemsize == 60
assert BASIC_FEATURES_DTYPE.itemsize == 36
assert WAVEFORM_WIDTH_INTEGRAL_DTYPE.itemsize == 52

__all__ = [
    "RECORDS_DTYPE",
    "THRESHOLD_HIT_DTYPE",
    "BASIC_FEATURES_DTYPE",
    "WAVEFORM_WIDTH_INTEGRAL_DTYPE",
    "HIT_DTYPE",
]
