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

# WAVEFORM_WIDTH_DTYPE (reference: cpu/waveform_width.py:21-36, 56 B)
WAVEFORM_WIDTH_DTYPE = np.dtype(
    [
        ("rise_time", "f4"),
        ("fall_time", "f4"),
        ("total_width", "f4"),
        ("rise_time_samples", "f4"),
        ("fall_time_samples", "f4"),
        ("total_width_samples", "f4"),
        ("peak_position", "i8"),
        ("peak_height", "f4"),
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("record_id", "i8"),
    ]
)

# S1_S2_CLASSIFIER_DTYPE (reference: cpu/s1_s2_classifier.py:30-43, 45 B)
S1_S2_CLASSIFIER_DTYPE = np.dtype(
    [
        ("label", "i1"),
        ("width_ns", "f4"),
        ("width_samples", "f4"),
        ("height", "f4"),
        ("area", "f4"),
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("record_id", "i8"),
        ("peak_position", "i8"),
    ]
)

# hit merge outputs (reference: cpu/hit_merge.py:17-49; 72 B, 16 B, 16 B)
HIT_MERGED_DTYPE = np.dtype(
    [
        ("position", "i8"),
        ("height", "f4"),
        ("integral", "f4"),
        ("sample_start", "i4"),
        ("sample_end", "i4"),
        ("width", "f4"),
        ("dt", "i4"),
        ("rise_time", "f4"),
        ("fall_time", "f4"),
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("record_id", "i8"),
        ("component_offset", "i8"),
        ("component_count", "i4"),
    ]
)
HIT_MERGED_COMPONENTS_DTYPE = np.dtype([("merged_index", "i8"), ("hit_index", "i8")])
HIT_MERGE_CLUSTERS_DTYPE = np.dtype([("cluster_index", "i8"), ("hit_index", "i8")])


# PEAK_DTYPE of the legacy helpers (reference: processing/dtypes.py:67-77, 30 B)
PEAK_DTYPE = np.dtype(
    [("time", "i8"), ("area", "f4"), ("height", "f4"), ("width", "f4"), ("channel", "i2"), ("event_index", "i8")]
)


def create_record_dtype(wave_length: int) -> np.dtype:
    """ST_WAVEFORM_DTYPE with a `wave_length`-sample int16 wave (reference: processing/dtypes.py:36-64)."""
    return np.dtype(
        [
            ("baseline", "f8"),
            ("baseline_upstream", "f8"),
            ("polarity", "U8"),
            ("timestamp", "i8"),
            ("record_id", "i8"),
            ("dt", "i4"),
            ("event_length", "i4"),
            ("board", "i2"),
            ("channel", "i2"),
            ("wave", "i2", (int(wave_length),)),
        ]
    )


def create_filtered_waveform_dtype(source_dtype: np.dtype) -> np.dtype:
    """Same fields as `source_dtype` with float32 wave samples (reference: cpu/filtering.py:133-158)."""
    names = source_dtype.names or ()
    if "wave" not in names:
        raise ValueError("source dtype missing required 'wave' field")
    fields = []
    for name in names:
        fdt = source_dtype.fields[name][0]
        sub = fdt.subdtype
        base, shape = (fdt, None) if sub is None else sub
        if name == "wave":
            base = np.float32
        fields.append((name, base) if shape is None else (name, base, shape))
    return np.dtype(fields)


assert HIT_DTYPE.itemsize == 48
assert WAVEFORM_WIDTH_DTYPE.itemsize == 56
assert S1_S2_CLASSIFIER_DTYPE.itemsize == 45
assert HIT_MERGED_DTYPE.itemsize == 72
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
    "WAVEFORM_WIDTH_DTYPE",
    "S1_S2_CLASSIFIER_DTYPE",
    "HIT_MERGED_DTYPE",
    "HIT_MERGED_COMPONENTS_DTYPE",
    "HIT_MERGE_CLUSTERS_DTYPE",
    "PEAK_DTYPE",
    "create_record_dtype",
    "create_filtered_waveform_dtype",
]
