"""HipFilteredWaveformsPlugin -- drop-in for FilteredWaveformsPlugin, the dense twin of wave_pool_filtered
(reference: waveform_analysis/core/plugins/builtin/cpu/filtering.py:410-536).

Every row of st_waveforms["wave"] (int16, n_events x n_samples) is filtered whole with the settings of its
hardware channel; the output is the same structured array with a float32 wave.  Rows are uniform-length
contiguous records of the row-major matrix, so they run through the span kernels of the records path.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import dense
from ..dtypes import create_filtered_waveform_dtype, create_record_dtype
from ..filter_engine import plan_filter_groups, run_filter_groups
from ..plugin_api import Option, Plugin
from . import _common as K


class HipFilteredWaveformsPlugin(K.HipPlugin):
    provides = "filtered_waveforms"
    algorithmic_bytes = (2 + 4, 0, 0)  # device pass: bytes per sample, per record, per output row
    depends_on = ["st_waveforms"]
    description = "Apply filtering to waveforms using Butterworth or Savitzky-Golay filters (HIP, gfx950)."
    version = "3.0.0+hip1"
    save_when = "target"
    output_dtype = create_filtered_waveform_dtype(create_record_dtype(1500))

    options = {
        "filter_type": Option(default="SG", type=str, help="'BW' or 'SG'"),
        "lowcut": Option(default=0.1, type=float, help="BW low cut"),
        "highcut": Option(default=0.5, type=float, help="BW high cut"),
        "fs": Option(default=0.5, type=float, help="BW sampling rate (GHz)"),
        "filter_order": Option(default=4, type=int, help="BW order"),
        "sg_window_size": Option(default=11, type=int, help="SG window (odd)"),
        "sg_poly_order": Option(default=2, type=int, help="SG polynomial order"),
        "max_workers": Option(default=None, type=int, help="ignored by the HIP backend", track=False),
        "batch_size": Option(default=0, type=int, help="ignored by the HIP backend (must be >= 0)"),
        "channel_config": Option(default=None, type=dict, help="per (board, channel) overrides of the filter options"),
    }

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        st = context.get_data(run_id, "st_waveforms")
        if not isinstance(st, np.ndarray):
            raise ValueError("filtered_waveforms expects st_waveforms as a single structured array")
        names = st.dtype.names or ()
        if "wave" not in names:
            raise ValueError("st_waveforms missing required 'wave' field for filtering")
        out_dtype = create_filtered_waveform_dtype(st.dtype)
        if out_dtype != self.output_dtype:
            self.output_dtype = out_dtype
        if len(st) == 0:
            return np.zeros(0, dtype=out_dtype)
        if "channel" not in names:
            raise ValueError("st_waveforms missing required 'channel' field for filtering")
        if st["wave"].ndim != 2:
            raise ValueError("st_waveforms['wave'] must be 2D (n_events, n_samples)")
        batch_size = int(context.get_config(self, "batch_size"))
        if batch_size < 0:
            raise ValueError(f"batch_size ({batch_size}) 必须大于等于 0")

        output = np.empty(len(st), dtype=out_dtype)
        for name in names:
            if name != "wave":
                output[name] = st[name]
        boards = st["board"] if "board" in names else np.zeros(len(st), dtype=np.int16)
        groups = plan_filter_groups(context, self, run_id, boards, st["channel"])
        pool, source, L = dense.dense_pool(st, "st_waveforms")
        if source != K.SRC_RAW:
            raise ValueError(f"st_waveforms['wave'] must be int16, got {st['wave'].dtype}")
        sess = K.resident_session(context, pool, cacheable=False)
        filtered = run_filter_groups(sess, dense.dense_records(st, L), groups)
        sess.forget_resident()
        output["wave"] = filtered.reshape(len(st), L)
        return output
