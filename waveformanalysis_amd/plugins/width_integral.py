"""HipWaveformWidthIntegralPlugin -- drop-in for WaveformWidthIntegralPlugin (records source)
(reference: waveform_analysis/core/plugins/builtin/cpu/waveform_width_integral.py:42-235)."""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import dense
from ..dtypes import WAVEFORM_WIDTH_INTEGRAL_DTYPE
from ..plugin_api import Option, Plugin
from . import _common as K


class HipWaveformWidthIntegralPlugin(K.HipPlugin):
    """Event-wise integral quantile width, computed on the GPU."""

    provides = "waveform_width_integral"
    algorithmic_bytes = (2, 29 + 52, 0)  # device pass: bytes per sample, per record, per output row
    depends_on = []
    description = "Event-wise integral quantile width from records + wave_pool (HIP, gfx950)."
    version = "2.7.0+hip1"
    save_when = "always"
    output_dtype = WAVEFORM_WIDTH_INTEGRAL_DTYPE
    options = {
        "q_low": Option(default=0.10, type=float, help="low quantile"),
        "q_high": Option(default=0.90, type=float, help="high quantile"),
        "use_filtered": Option(default=False, type=bool, help="read wave_pool_filtered"),
        "wave_source": Option(default=K.WAVE_SOURCE_AUTO, type=str,
                              help="auto|records|st_waveforms|filtered_waveforms"),
        "sampling_rate": Option(default=0.5, type=float, help="GHz, used when dt is None"),
        "dt": Option(default=None, type=float, help="sample interval (ns), wins over sampling_rate"),
    }

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list[str]:
        _kind, deps, _name = K.resolve_wave_input(context, self)
        return deps

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        q_low = float(context.get_config(self, "q_low"))
        q_high = float(context.get_config(self, "q_high"))
        dt = context.get_config(self, "dt")
        sampling_rate = context.get_config(self, "sampling_rate")
        kind, _deps, pool_name = K.resolve_wave_input(context, self)
        if kind == "dense":
            data = K.load_dense_input(context, self, run_id, pool_name)
            records = pool = None
        else:
            records, pool = K.load_records_input(context, self, run_id, pool_name)
        if dt is None:
            if sampling_rate <= 0:
                raise ValueError(f"sampling_rate ({sampling_rate}) 必须大于 0")
            dt = 1.0 / float(sampling_rate)
        if q_low <= 0 or q_high >= 1 or q_low >= q_high:
            raise ValueError(f"q_low/q_high 无效: q_low={q_low}, q_high={q_high}")
        if kind == "dense":
            # waveform_width_integral.py:139-189: float64 of the row minus the float64 baseline, sign from the
            # literal "positive" only
            if len(data) == 0:
                return np.zeros(0, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE)
            for name in ("baseline", "timestamp"):
                if name not in (data.dtype.names or ()):
                    raise ValueError(f"no field of name {name}")  # numpy's message for data[i][name]
            dpool, source, L = dense.dense_pool(data, pool_name)
            sess = K.resident_session(context, dpool, cacheable=False)
            sess.upload_records(dense.dense_records(data, L), polarity=dense.dense_polarity_wave_rule(data))
            return sess.width_integral(source, q_low, q_high, float(dt))
        if len(records) == 0:
            return np.zeros(0, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE)
        if pool.dtype == np.float32:
            source = K.SRC_F32
        elif pool.dtype == np.uint16:
            source = K.SRC_RAW
        else:
            raise ValueError(f"wave pool must be uint16 or float32, got {pool.dtype}")
        sess = K.resident_session(context, pool)
        sess.upload_records(records)
        return sess.width_integral(source, q_low, q_high, float(dt))
