"""HipWavePoolFilteredPlugin -- drop-in for WavePoolFilteredPlugin
(reference: waveform_analysis/core/plugins/builtin/cpu/records.py:334-438)."""

from __future__ import annotations

from typing import Any

import numpy as np

from ..plugin_api import Option, Plugin
from ..sg_plan import normalize_window
from . import _common as K


class HipWavePoolFilteredPlugin(Plugin):
    """Build the float32 filtered wave_pool on the GPU (Savitzky-Golay, interpolated edges)."""

    provides = "wave_pool_filtered"
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

        filter_type = str(context.get_config(self, "filter_type"))
        if filter_type not in ("BW", "SG"):
            raise ValueError(f"不支持的滤波器类型: {filter_type}. 请使用 'BW' 或 'SG'.")
        if context.get_config(self, "channel_config"):
            raise NotImplementedError("HIP backend: per-channel filter overrides are not implemented yet")
        bw = None
        if filter_type == "BW":
            bw = design_bw(context.get_config(self, "lowcut"), context.get_config(self, "highcut"),
                           context.get_config(self, "fs"), context.get_config(self, "filter_order"))
        else:
            window, order = normalize_window(context.get_config(self, "sg_window_size"),
                                             context.get_config(self, "sg_poly_order"))

        off = records["wave_offset"].astype(np.int64)
        length = records["event_length"].astype(np.int64)
        bad = (length > 0) & ((off < 0) | (off + length > len(wave_pool)))
        if np.any(bad):
            i = int(np.flatnonzero(bad)[0])
            raise ValueError("wave_pool_filtered found out-of-bounds wave slice "
                             f"(offset={int(off[i])}, length={int(length[i])}, wave_pool_size={len(wave_pool)})")
        sess = K.resident_session(context, np.asarray(wave_pool))
        sess.upload_records(_view_records(records))
        if bw is not None:
            out = sess.sosfiltfilt(*bw, download=True)
        else:
            sess.set_sg_plan(window, order)
            out = sess.savgol(download=True)
        K.invalidate_residency()  # the resident float32 pool now belongs to this output
        return out


def design_bw(lowcut, highcut, fs, order):
    """Filter design on the host exactly as the reference validates and designs it
    (filtering.py:84-101) + scipy's steady-state initial conditions and the pad length of
    filtering.py:198-203.  Returns (sos, zi, padlen)."""
    from scipy.signal import butter, sosfilt_zi

    lowcut, highcut, fs, order = float(lowcut), float(highcut), float(fs), int(order)
    if fs <= 0:
        raise ValueError(f"fs ({fs}) 必须大于 0")
    if order <= 0:
        raise ValueError(f"滤波器阶数 ({order}) 必须大于 0")
    if lowcut <= 0 or highcut <= 0:
        raise ValueError("截止频率必须大于 0")
    if lowcut >= highcut:
        raise ValueError(f"lowcut ({lowcut}) 必须小于 highcut ({highcut})")
    if highcut >= fs / 2:
        raise ValueError(f"highcut ({highcut}) 必须小于奈奎斯特频率 ({fs / 2})")
    sos = butter(order, [lowcut, highcut], btype="band", output="sos", fs=fs)
    n_sections = int(sos.shape[0])
    padlen = 3 * (2 * n_sections + 1 - min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum())))
    return sos, sosfilt_zi(sos), padlen


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
