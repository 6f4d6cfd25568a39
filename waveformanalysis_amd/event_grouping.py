"""Event grouping of hit rows on the gathering rank (the one exchange step of the path).

`group_hit_windows` is a vectorised restatement of the reference's
waveform_analysis/core/processing/event_grouping.py:286-471 (`group_hit_windows`): absolute hit
windows in float64 ps, a global lexsort, gap-chained clustering, and a per-event ordering.  The
reference walks the sorted hits in a Python loop; here the chain is a running maximum:

    sorted by abs_start, a hit opens a new event  <=>  abs_start > max(abs_end of all earlier hits) + gap

(the maximum over *all* earlier hits equals the maximum over the current cluster, because every
earlier cluster ends more than `gap` before the current one starts), so event ids are a cumulative
sum and the per-event ordering is one more lexsort with the event id as the primary key.
Hits arrive from all GPUs through the RCCL gather (sharding.py / wfa_rccl_gather_rows).

Merged hits whose window spans records (sample_start/end < 0, produced by hit_merge) need the
component tables of the reference's `hit_merged_components`; that stage is not part of this backend
yet, so such input raises the same ValueError as the reference does without components.
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


def group_hit_windows_flat(hits: np.ndarray, time_window_ns: float, dt_values: np.ndarray | None = None) -> dict:
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

    timestamps = np.asarray(hits["timestamp"], dtype=np.int64)
    positions = np.asarray(hits["position"], dtype=np.float64)
    s_rel = np.asarray(hits[start_name], dtype=np.int32)
    e_rel = np.asarray(hits[end_name], dtype=np.int32)
    if np.any((s_rel < 0) | (e_rel < 0)):
        raise ValueError("component_rows and component_hits are required when hit windows contain invalid edges")
    record_ids = np.asarray(hits["record_id"], dtype=np.int64)
    dt_ps = dt_values.astype(np.float64) * 1e3
    abs_starts = timestamps.astype(np.float64) + (s_rel - positions) * dt_ps   # event_grouping.py:365-367
    abs_ends = timestamps.astype(np.float64) + (e_rel - positions) * dt_ps

    order = np.lexsort((record_ids, timestamps, dt_values, abs_starts))          # :418
    gap_ps = time_window_ns * 1e3
    ends_sorted = abs_ends[order]
    run_max = np.maximum.accumulate(ends_sorted)
    new_event = np.ones(n, dtype=bool)
    new_event[1:] = abs_starts[order][1:] > run_max[:-1] + gap_ps                # :457-470
    event_of_sorted = np.cumsum(new_event) - 1
    event_id = np.empty(n, dtype=np.int64)
    event_id[order] = event_of_sorted

    boards = np.asarray(hits["board"], dtype=np.int16)
    channels = np.asarray(hits["channel"], dtype=np.int16)
    inner = np.lexsort((record_ids, timestamps, abs_starts, dt_values, channels, boards, event_id))  # :423-432
    n_events = int(event_of_sorted[-1]) + 1
    counts = np.bincount(event_id, minlength=n_events)
    event_start = np.zeros(n_events + 1, dtype=np.int64)
    np.cumsum(counts, out=event_start[1:])
    t_min = np.minimum.reduceat(abs_starts[inner], event_start[:-1]).astype(np.int64)  # int(np.min(...)) truncation
    t_max = np.maximum.reduceat(abs_ends[inner], event_start[:-1]).astype(np.int64)
    return {"order": inner, "event_start": event_start, "t_min": t_min, "t_max": t_max, "dt": dt_values,
            "start_name": start_name, "end_name": end_name}


def group_hit_windows(hits: np.ndarray, time_window_ns: float, dt_values: np.ndarray | None = None):
    """Same DataFrame as the reference (ragged per-event arrays in object columns)."""
    import pandas as pd

    if isinstance(hits, np.ndarray) and len(hits) == 0:
        return pd.DataFrame(columns=EVENT_COLUMNS)
    flat = group_hit_windows_flat(hits, time_window_ns, dt_values)
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


__all__ = ["group_hit_windows", "group_hit_windows_flat", "EVENT_COLUMNS"]
