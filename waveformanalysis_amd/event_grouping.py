"""Event grouping of hit rows on the gathering rank (the one exchange step of the path).

`group_hit_windows` follows waveform_analysis/core/processing/event_grouping.py:286-471: absolute hit windows
in float64 ps, a global lexsort, gap-chained clustering and a per-event ordering.  The reference walks the
sorted hits in a Python loop; here the chain is a running maximum,

    sorted by abs_start, a hit opens a new event  <=>  abs_start > max(abs_end of all earlier hits) + gap

(the maximum over *all* earlier hits equals the maximum over the current cluster, because every earlier cluster
ends more than `gap` before the current one starts), so event ids are a prefix sum and the per-event ordering is
one more lexsort with the event id as the primary key.  The two lexsorts (stable radix passes), the max-scan
and the prefix sum run on the GPU (wfa_group_hit_windows_count / _fill); this module validates the table,
resolves the windows of merged hits that span records from their components (event_grouping.py:369-416) and
builds the ragged DataFrame.  Hits arrive from all GPUs through the RCCL gather (sharding.py).
"""

from __future__ import annotations

import numpy as np

EVENT_COLUMNS = [
    "event_id", "t_min", "t_max", "dt/ns", "n_hits", "dt", "boards", "channels", "heights", "integrals",
    "timestamps", "record_ids", "sample_starts", "sample_ends",
]


def _window_fields(names: set[str]) -> tuple[str, str]:
    if {"sample_start", "sample_end"}.issubset(names):
        return "sample_start", "sample_end"
    if {"edge_start", "edge_end"}.issubset(names):
        return "edge_start", "edge_end"
    return "sample_start", "sample_end"


def _component_windows(hits, names, invalid, component_rows, component_hits):
    """abs window of merged hits without a sample window = extent of their component hits (:369-416)."""
    if component_rows is None or component_hits is None:
        raise ValueError("component_rows and component_hits are required when hit windows contain invalid edges")
    hit_indices = np.asarray(component_rows["hit_index"], dtype=np.int64)
    c_dt_ps = np.asarray(component_hits["dt"], dtype=np.int32).astype(np.float64) * 1e3
    c_ts = np.asarray(component_hits["timestamp"], dtype=np.int64).astype(np.float64)
    c_pos = np.asarray(component_hits["position"], dtype=np.float64)
    c_abs0 = c_ts + (np.asarray(component_hits["edge_start"], dtype=np.int32) - c_pos) * c_dt_ps
    c_abs1 = c_ts + (np.asarray(component_hits["edge_end"], dtype=np.int32) - c_pos) * c_dt_ps
    fix0 = np.full(len(hits), np.nan)
    fix1 = np.full(len(hits), np.nan)
    have_table = "component_offset" in names and "component_count" in names
    merged_indices = None if have_table else np.asarray(component_rows["merged_index"], dtype=np.int64)
    for idx in np.flatnonzero(invalid):
        if have_table:
            off, cnt = int(hits["component_offset"][idx]), int(hits["component_count"][idx])
            if cnt <= 0:
                raise ValueError(f"missing hit_merged_components rows for hit_merged index {int(idx)}")
            subset = hit_indices[off : off + cnt]
        else:
            mask = merged_indices == int(idx)
            if not np.any(mask):
                raise ValueError(f"missing hit_merged_components rows for hit_merged index {int(idx)}")
            subset = hit_indices[mask]
        fix0[idx] = float(np.min(c_abs0[subset]))
        fix1[idx] = float(np.max(c_abs1[subset]))
    return fix0, fix1


def group_hit_windows_flat(hits: np.ndarray, time_window_ns: float, dt_values: np.ndarray | None = None,
                           component_rows: np.ndarray | None = None, component_hits: np.ndarray | None = None,
                           session=None) -> dict:
    """Flat form: `order` (hit indices, event-major, reference order inside an event),
    `event_start` (offsets into order, len = n_events + 1), `t_min`, `t_max` (int64 ps)."""
    if not isinstance(hits, np.ndarray):
        raise ValueError("hits must be a single structured array")
    if time_window_ns < 0:
        raise ValueError("time_window_ns must be >= 0")
    names = set(hits.dtype.names or ())
    start_name, end_name = _window_fields(names)
    required = {"timestamp", "position", "board", "channel", "height", "integral", "record_id"}
    missing = sorted(required - names)
    if missing:
        raise KeyError(f"hits missing required fields: {missing}")
    if start_name not in names or end_name not in names:
        raise KeyError(f"hits missing required fields: {[start_name, end_name]}")
    if dt_values is None:
        if "dt" not in names:
            raise KeyError("hits missing required field: dt")
        dt_values = np.asarray(hits["dt"], dtype=np.int32)
    else:
        dt_values = np.asarray(dt_values, dtype=np.int32)
    if len(dt_values) != len(hits):
        raise ValueError("dt_values length must match hits")
    if np.any(dt_values <= 0):
        raise ValueError("hit dt must be positive for every row")
    n = len(hits)
    if n == 0:
        z = np.zeros(0, dtype=np.int64)
        return {"order": z, "event_start": np.zeros(1, dtype=np.int64), "t_min": z, "t_max": z, "dt": dt_values,
                "start_name": start_name, "end_name": end_name}
    s_rel = np.asarray(hits[start_name], dtype=np.int32)
    e_rel = np.asarray(hits[end_name], dtype=np.int32)
    position = hits["position"]
    if position.dtype.kind == "f":
        raise ValueError("hit position must be an integer sample index")
    invalid = (s_rel < 0) | (e_rel < 0)
    fix0 = fix1 = None
    if np.any(invalid):
        fix0, fix1 = _component_windows(hits, names, invalid, component_rows, component_hits)
    if session is None:
        from .device import default_pool

        session = default_pool().session()
    flat = session.group_hit_windows(hits["timestamp"], position, s_rel, e_rel, dt_values, hits["board"], hits["channel"],
                                     hits["record_id"], float(time_window_ns), fix0, fix1)
    flat.update({"dt": dt_values, "start_name": start_name, "end_name": end_name})
    return flat


def group_hit_windows(hits: np.ndarray, time_window_ns: float, dt_values: np.ndarray | None = None,
                      component_rows: np.ndarray | None = None, component_hits: np.ndarray | None = None,
                      session=None):
    """Same DataFrame as the reference (ragged per-event arrays in object columns)."""
    import pandas as pd

    if isinstance(hits, np.ndarray) and len(hits) == 0:
        return pd.DataFrame(columns=EVENT_COLUMNS)
    flat = group_hit_windows_flat(hits, time_window_ns, dt_values, component_rows, component_hits, session)
    order, es = flat["order"], flat["event_start"]
    cols = {
        "dt": flat["dt"][order].astype(np.int32),
        "boards": np.asarray(hits["board"], dtype=np.int16)[order],
        "channels": np.asarray(hits["channel"], dtype=np.int16)[order],
        "heights": np.asarray(hits["height"], dtype=np.float32)[order],
        "integrals": np.asarray(hits["integral"], dtype=np.float32)[order],
        "timestamps": np.asarray(hits["timestamp"], dtype=np.int64)[order],
        "record_ids": np.asarray(hits["record_id"], dtype=np.int64)[order],
        "sample_starts": np.asarray(hits[flat["start_name"]], dtype=np.int32)[order],
        "sample_ends": np.asarray(hits[flat["end_name"]], dtype=np.int32)[order],
    }
    n_events = len(es) - 1
    rows = []
    for ev in range(n_events):
        a, b = int(es[ev]), int(es[ev + 1])
        t_min, t_max = int(flat["t_min"][ev]), int(flat["t_max"][ev])
        row = {"event_id": ev, "t_min": t_min, "t_max": t_max, "dt/ns": (t_max - t_min) / 1e3, "n_hits": b - a}
        for k, v in cols.items():
            row[k] = v[a:b].copy()
        rows.append(row)
    return pd.DataFrame(rows, columns=EVENT_COLUMNS)


def find_hits(waves: np.ndarray, baselines: np.ndarray, threshold: float, left_extension: int = 2,
              right_extension: int = 2, session=None) -> np.ndarray:
    """Legacy vectorised hit finding (event_grouping.py:46-95): contiguous regions of (baseline - wave) > threshold
    on a dense (n_events, n_samples) array; PEAK_DTYPE rows with `time` = start sample and `event_index` set
    (the extensions are accepted and unused, as in the reference)."""
    from . import _lib
    from .dtypes import PEAK_DTYPE

    waves = np.asarray(waves)
    if waves.size == 0:
        return np.zeros(0, dtype=PEAK_DTYPE)
    if waves.ndim != 2:
        raise ValueError("waves must be a 2D array (n_events, n_samples)")
    baselines = np.asarray(baselines, dtype=np.float64)
    if waves.dtype in (np.int16, np.uint16):
        flat = np.ascontiguousarray(waves).reshape(-1)
        if waves.dtype == np.int16:
            if int(flat.min()) < 0:
                raise ValueError("waves holds negative samples; the HIP backend reads unsigned ADC codes")
            flat = flat.view(np.uint16)
        source = _lib.SRC_RAW
    elif waves.dtype == np.float32:
        flat, source = np.ascontiguousarray(waves).reshape(-1), _lib.SRC_F32
    else:
        raise ValueError(f"waves must be int16 / uint16 / float32, got {waves.dtype}")
    if session is None:
        from .device import default_pool

        session = default_pool().session()
    session.upload_pool(flat)
    event_index, start = session.find_hits_legacy(source, waves.shape[0], waves.shape[1], baselines, float(threshold))
    hits = np.zeros(len(start), dtype=PEAK_DTYPE)
    hits["event_index"] = event_index
    hits["time"] = start
    return hits


def find_cluster_boundaries(ts_sorted: np.ndarray, time_window_ps: float) -> np.ndarray:
    """Fixed-window clustering of sorted timestamps (event_grouping.py:475-525): a cluster takes every hit within
    `time_window_ps` of its FIRST hit.  All jump targets come from one vectorised searchsorted; the chain of jumps
    from hit 0 is then followed (one step per event)."""
    ts_sorted = np.asarray(ts_sorted)
    n = len(ts_sorted)
    if n == 0:
        return np.array([0])
    nxt = np.searchsorted(ts_sorted, ts_sorted + time_window_ps, side="right")
    out = [0]
    cur = 0
    while cur < n:
        cur = int(nxt[cur])
        out.append(cur)
    return np.array(out)


MULTI_CHANNEL_COLUMNS = ["event_id", "t_min", "t_max", "dt/ns", "n_hits", "channels", "areas", "heights", "timestamps"]


def group_multi_channel_hits(df, time_window_ns: float, use_numba: bool = True, n_processes: int | None = None,
                             session=None):
    """Legacy DataFrame grouping (event_grouping.py:98-283): sort by timestamp, fixed windows from each cluster's
    first hit, members ordered by channel.  `use_numba` / `n_processes` are accepted and ignored.  The reference sorts
    with pandas' / numpy's default (unstable) kinds, so the order of equal timestamps, and of equal channels inside an
    event, is unspecified there; here both sorts are stable.

    Integer timestamp and channel columns go through the device (wfa_group_multi_channel_*: two stable radix sorts, the
    window chain by pointer jumping); `session=False`, or any other column type, keeps the host table code
    (`_group_multi_channel_order_host`), which is also what the device result is tested against."""
    import pandas as pd

    if not time_window_ns >= 0:  # (the reference's boundary loop does not terminate for a negative window)
        raise ValueError("time_window_ns must be >= 0")
    time_window_ps = time_window_ns * 1e3
    area_col = "area" if "area" in df.columns else "charge"
    height_col = "height" if "height" in df.columns else "peak"
    if area_col not in df.columns or height_col not in df.columns:
        raise KeyError("df must contain area/height (or charge/peak) columns")
    n = len(df)
    if n == 0:
        return pd.DataFrame(columns=MULTI_CHANNEL_COLUMNS)
    ts_in = df["timestamp"].to_numpy()
    ch_in = df["channel"].to_numpy()
    on_device = session is not False and ts_in.dtype.kind in "iu" and ch_in.dtype.kind in "iu" and \
        ts_in.dtype != np.uint64 and ch_in.dtype != np.uint64
    if on_device:
        if session is None:
            from .device import default_pool

            session = default_pool().session()
        order, bounds = session.group_multi_channel(ts_in, ch_in, float(time_window_ps))
    else:
        order, bounds = _group_multi_channel_order_host(ts_in, ch_in, time_window_ps)
    ts_o, ch_o = ts_in[order], ch_in[order]
    ar_o, he_o = df[area_col].to_numpy()[order], df[height_col].to_numpy()[order]
    n_events = len(bounds) - 1
    starts, ends = bounds[:-1], bounds[1:]
    t_min = ts_o[starts].astype(np.int64)       # the reference takes the first / last row AFTER the channel sort
    t_max = ts_o[ends - 1].astype(np.int64)
    split = bounds[1:-1]
    return pd.DataFrame({
        "event_id": np.arange(n_events, dtype=np.int64),
        "t_min": t_min,
        "t_max": t_max,
        "dt/ns": (ts_o[ends - 1] - ts_o[starts]) / 1e3,
        "n_hits": np.diff(bounds).astype(np.int32),
        "channels": np.split(ch_o, split),
        "areas": np.split(ar_o, split),
        "heights": np.split(he_o, split),
        "timestamps": np.split(ts_o, split),
    })


def _group_multi_channel_order_host(ts_in: np.ndarray, ch_in: np.ndarray, time_window_ps: float):
    """(order, bounds) of group_multi_channel_hits with numpy: stable sort by timestamp, window chain, stable sort by
    (event, channel)."""
    n = len(ts_in)
    by_ts = np.argsort(ts_in, kind="stable")
    ts_all = ts_in[by_ts]
    bounds = find_cluster_boundaries(ts_all, time_window_ps)
    n_events = len(bounds) - 1
    event_of = np.repeat(np.arange(n_events), np.diff(bounds))
    inner = np.lexsort((np.arange(n), ch_in[by_ts], event_of))  # per event: by channel, stable
    return by_ts[inner], np.asarray(bounds, dtype=np.int64)


__all__ = ["group_hit_windows", "group_hit_windows_flat", "EVENT_COLUMNS", "find_hits", "find_cluster_boundaries",
           "group_multi_channel_hits", "MULTI_CHANNEL_COLUMNS"]
