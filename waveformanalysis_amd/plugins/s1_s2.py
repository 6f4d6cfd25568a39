"""HipS1S2ClassifierPlugin -- drop-in for S1S2ClassifierPlugin
(reference: waveform_analysis/core/plugins/builtin/cpu/s1_s2_classifier.py:71-228).

Range cuts on two small tables (waveform_width rows joined with basic_features rows): no waveform samples
are touched, so this stage is host table work, vectorised here where the reference loops per row and
searches the feature table linearly per peak.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from ..dtypes import S1_S2_CLASSIFIER_DTYPE
from ..plugin_api import Option, Plugin
from .waveform_width import first_row_of_record_id

LABEL_UNKNOWN, LABEL_S1, LABEL_S2 = 0, 1, 2


def _normalize_range(value):
    """s1_s2_classifier.py:46-54."""
    if value is None:
        return None
    if not isinstance(value, tuple) or len(value) != 2:
        raise ValueError("range must be a tuple of (min, max)")
    lo, hi = value
    if lo is None and hi is None:
        return None
    return (None if lo is None else float(lo), None if hi is None else float(hi))


def _in_range(values: np.ndarray, bounds) -> np.ndarray:
    """s1_s2_classifier.py:57-69 on a float64 column (NaN is never in range when bounds are set)."""
    if bounds is None:
        return np.ones(len(values), dtype=bool)
    lo, hi = bounds
    ok = ~np.isnan(values)
    if lo is not None:
        ok &= ~(values < lo)
    if hi is not None:
        ok &= ~(values > hi)
    return ok


def classify(widths: np.ndarray, features: np.ndarray, *, width_unit="ns", s1_width_range=None,
             s2_width_range=None, s1_area_range=None, s2_area_range=None, s1_height_range=None,
             s2_height_range=None, conflict_policy="unknown", strict=False) -> np.ndarray:
    s1w, s2w = _normalize_range(s1_width_range), _normalize_range(s2_width_range)
    s1a, s2a = _normalize_range(s1_area_range), _normalize_range(s2_area_range)
    s1h, s2h = _normalize_range(s1_height_range), _normalize_range(s2_height_range)
    s1_enabled = any(r is not None for r in (s1w, s1a, s1h))
    s2_enabled = any(r is not None for r in (s2w, s2a, s2h))
    if strict and not s1_enabled and not s2_enabled:
        raise ValueError("No S1/S2 criteria configured; set ranges or disable strict.")
    if not isinstance(widths, np.ndarray):
        raise ValueError("s1_s2 expects waveform_width as a single array")
    if not isinstance(features, np.ndarray):
        raise ValueError("s1_s2 expects basic_features as a single array")
    n = len(widths)
    out = np.zeros(n, dtype=S1_S2_CLASSIFIER_DTYPE)
    if n == 0:
        return out
    wnames = widths.dtype.names or ()
    record_id = np.asarray(widths["record_id"] if "record_id" in wnames else widths["event_index"], dtype=np.int64)
    width_ns = widths["total_width"].astype(np.float64)
    width_samples = widths["total_width_samples"].astype(np.float64)
    if "record_id" in (features.dtype.names or ()):
        row = first_row_of_record_id(features["record_id"], record_id)
    else:
        row = np.where((record_id >= 0) & (record_id < len(features)), record_id, -1)
    found = row >= 0
    height = np.full(n, np.nan)
    area = np.full(n, np.nan)
    if len(features):
        safe = np.where(found, row, 0)
        height = np.where(found, features["height"][safe].astype(np.float64), np.nan)
        area = np.where(found, features["area"][safe].astype(np.float64), np.nan)
    width_value = width_samples if width_unit == "samples" else width_ns
    s1_ok = (_in_range(width_value, s1w) & _in_range(area, s1a) & _in_range(height, s1h)) if s1_enabled else np.zeros(n, bool)
    s2_ok = (_in_range(width_value, s2w) & _in_range(area, s2a) & _in_range(height, s2h)) if s2_enabled else np.zeros(n, bool)
    label = np.zeros(n, dtype=np.int8)
    label[s1_ok & ~s2_ok] = LABEL_S1
    label[s2_ok & ~s1_ok] = LABEL_S2
    both = s1_ok & s2_ok
    if conflict_policy == "prefer_s1":
        label[both] = LABEL_S1
    elif conflict_policy == "prefer_s2":
        label[both] = LABEL_S2
    out["label"] = label
    out["width_ns"] = width_ns
    out["width_samples"] = width_samples
    out["height"] = height
    out["area"] = area
    out["timestamp"] = widths["timestamp"]
    out["board"] = widths["board"] if "board" in wnames else 0
    out["channel"] = widths["channel"]
    out["record_id"] = record_id
    out["peak_position"] = widths["peak_position"]
    return out


class HipS1S2ClassifierPlugin(Plugin):
    """Classify peaks into S1/S2/Unknown using waveform width + basic features."""

    provides = "s1_s2"
    depends_on = ["waveform_width", "basic_features"]
    description = "Classify peaks into S1/S2 using width/area/height ranges."
    version = "0.4.0+hip1"
    save_when = "always"
    output_dtype = S1_S2_CLASSIFIER_DTYPE

    options = {
        "width_unit": Option(default="ns", type=str, help="'ns' or 'samples'"),
        "s1_width_range": Option(default=None, type=tuple, help="S1 width range (min, max); None disables"),
        "s2_width_range": Option(default=None, type=tuple, help="S2 width range (min, max); None disables"),
        "s1_area_range": Option(default=None, type=tuple, help="S1 area range (min, max); None disables"),
        "s2_area_range": Option(default=None, type=tuple, help="S2 area range (min, max); None disables"),
        "s1_height_range": Option(default=None, type=tuple, help="S1 height range (min, max); None disables"),
        "s2_height_range": Option(default=None, type=tuple, help="S2 height range (min, max); None disables"),
        "conflict_policy": Option(default="unknown", type=str, help="unknown|prefer_s1|prefer_s2"),
        "strict": Option(default=False, type=bool, help="raise if no S1/S2 criteria are configured"),
    }

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        widths = context.get_data(run_id, "waveform_width")
        features = context.get_data(run_id, "basic_features")
        cfg = {k: context.get_config(self, k) for k in self.options}
        return classify(widths, features, **cfg)
