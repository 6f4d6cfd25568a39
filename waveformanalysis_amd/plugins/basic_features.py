"""HipBasicFeaturesPlugin -- drop-in for BasicFeaturesPlugin, records and dense (st_waveforms /
filtered_waveforms) sources (reference: waveform_analysis/core/plugins/builtin/cpu/basic_features.py:43-278)."""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import dense
from ..dtypes import BASIC_FEATURES_DTYPE
from ..plugin_api import Option, Plugin
from . import _common as K

PEAK_RANGE = (40, 90)  # reference: core/foundation/constants.py:22 (FeatureDefaults.PEAK_RANGE)


class HipBasicFeaturesPlugin(K.HipPlugin):
    """height / amp / area / max_abs_diff per record, computed on the GPU."""

    provides = "basic_features"
    algorithmic_bytes = (2, 29 + 36, 0)  # device pass: bytes per sample, per record, per output row
    depends_on = []
    description = "Compute basic height, amplitude, area, and max-abs-diff features (HIP, gfx950)."
    version = "4.0.0+hip1"
    save_when = "always"
    output_dtype = BASIC_FEATURES_DTYPE
    options = {
        "height_range": Option(default=PEAK_RANGE, type=tuple, help="(start, end) for height/amp"),
        "area_range": Option(default=(0, None), type=tuple, help="(start, end) for area; None = to the end"),
        "use_filtered": Option(default=False, type=bool, help="read wave_pool_filtered"),
        "wave_source": Option(default=K.WAVE_SOURCE_AUTO, type=str,
                              help="auto|records|st_waveforms|filtered_waveforms"),
        "fixed_baseline": Option(default=None, type=dict, help="deprecated; use channel_config"),
        "channel_config": Option(default=None, type=dict, help="per (board, channel) fixed_baseline"),
    }

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list[str]:
        _kind, deps, _name = K.resolve_wave_input(context, self)
        return deps

    def compute(self, context: Any, run_id: str, **kwargs) -> np.ndarray:
        channel_config = context.get_config(self, "channel_config")
        height_range = tuple(context.get_config(self, "height_range"))
        area_range = tuple(context.get_config(self, "area_range"))
        kind, _deps, data_name = K.resolve_wave_input(context, self)
        if kind == "dense":
            return self._compute_dense(context, run_id, data_name, channel_config, height_range, area_range)
        records, pool = K.load_records_input(context, self, run_id, data_name)
        if len(records) == 0:
            return np.zeros(0, dtype=BASIC_FEATURES_DTYPE)
        fixed = None
        if channel_config:
            fixed = K.per_record_channel_option(records, channel_config, run_id, "fixed_baseline", None, np.nan)
            if np.all(np.isnan(fixed)):
                fixed = None
        if pool.dtype == np.float32:
            source = K.SRC_F32
        elif pool.dtype == np.uint16:
            source = K.SRC_RAW
        else:
            raise ValueError(f"wave pool must be uint16 or float32, got {pool.dtype}")
        sess = K.resident_session(context, pool)
        sess.upload_records(records)
        return sess.basic_features(source, height_range, area_range, fixed)

    def _compute_dense(self, context, run_id, data_name, channel_config, height_range, area_range) -> np.ndarray:
        """basic_features.py:197-278: whole rows, wave-based formulas, sign from the literal "positive"."""
        data = K.load_dense_input(context, self, run_id, data_name)
        if len(data) == 0:
            return np.zeros(0, dtype=BASIC_FEATURES_DTYPE)
        pool, source, L = dense.dense_pool(data, data_name)
        records = dense.dense_records(data, L)
        fixed = None
        if channel_config:
            fixed = K.per_record_channel_option(records, channel_config, run_id, "fixed_baseline", None, np.nan)
            if np.all(np.isnan(fixed)):
                fixed = None
        sess = K.resident_session(context, pool, cacheable=False)  # temporary of the dense `wave` field
        sess.upload_records(records, polarity=dense.dense_polarity_wave_rule(data))
        return sess.basic_features(source, height_range, area_range, fixed)
