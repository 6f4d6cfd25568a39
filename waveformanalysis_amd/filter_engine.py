"""Shared filter engine of wave_pool_filtered and filtered_waveforms: option resolution per hardware
channel, validation with the reference's messages, and the GPU passes
(reference: waveform_analysis/core/plugins/builtin/cpu/filtering.py:30-131,339-374).

Records are grouped by their RESOLVED filter settings (not by channel: channels that share settings share a
kernel launch); each group is filtered straight into the one resident float32 output.
"""

from __future__ import annotations

import warnings
from typing import Any

import numpy as np

from .channel_config import per_record_option

FILTER_OPTION_NAMES = ("filter_type", "lowcut", "highcut", "fs", "filter_order", "sg_window_size", "sg_poly_order")


def design_bw(lowcut, highcut, fs, order):
    """Validation and design of filtering.py:84-101 + scipy's steady-state initial conditions and the pad
    length of filtering.py:198-203.  Returns (sos, zi, padlen)."""
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


def resolve_filter_key(values: dict[str, Any]) -> tuple:
    """Validated, hashable filter settings (filtering.py:76-124)."""
    filter_type = str(values["filter_type"])
    if filter_type not in ("BW", "SG"):
        raise ValueError(f"不支持的滤波器类型: {filter_type}. 请使用 'BW' 或 'SG'.")
    if filter_type == "BW":
        key = ("BW", float(values["lowcut"]), float(values["highcut"]), float(values["fs"]), int(values["filter_order"]))
        design_bw(*key[1:])  # raises the reference's messages for bad settings
        return key
    window, order = int(values["sg_window_size"]), int(values["sg_poly_order"])
    if window <= 0:
        raise ValueError(f"SG 窗口大小 ({window}) 必须大于 0")
    if order < 0:
        raise ValueError(f"SG 多项式阶数 ({order}) 必须大于等于 0")
    if window % 2 == 0:
        window += 1
        warnings.warn(f"SG 窗口大小已调整为奇数: {window}", stacklevel=3)
    if order >= window:
        raise ValueError(f"SG 多项式阶数 ({order}) 必须小于窗口大小 ({window})")
    return ("SG", window, order)


def plan_filter_groups(context: Any, plugin: Any, run_id: str, boards: np.ndarray, channels: np.ndarray):
    """[(filter key, bool mask of the records it applies to)], one entry per distinct resolved setting."""
    base = {name: context.get_config(plugin, name) for name in FILTER_OPTION_NAMES}
    channel_config = context.get_config(plugin, "channel_config") if "channel_config" in plugin.options else None
    per = per_record_option(boards, channels, channel_config, run_id, base)
    by_key: dict[tuple, np.ndarray] = {}
    boards = np.asarray(boards)
    channels = np.asarray(channels)
    for (b, c), values in per.items():
        key = resolve_filter_key({**base, **values})
        mask = (boards == b) & (channels == c)
        by_key[key] = mask if key not in by_key else (by_key[key] | mask)
    return list(by_key.items())


def run_filter_groups(sess, records: np.ndarray, groups) -> np.ndarray:
    """Filter every group into the session's float32 pool and return it (host copy)."""
    sess.filter_keep_output(False)
    try:
        for k, (key, mask) in enumerate(groups):
            sub = records if mask.all() else records[mask]
            sess.upload_records(sub)
            if k == 1:
                sess.filter_keep_output(True)
            if key[0] == "BW":
                sess.sosfiltfilt(*design_bw(*key[1:]), download=False)
            else:
                sess.set_sg_plan(key[1], key[2])
                sess.savgol(download=False)
        return sess.download_filtered()
    finally:
        sess.filter_keep_output(False)
