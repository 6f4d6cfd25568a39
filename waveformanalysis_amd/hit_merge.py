"""Hit merging: cluster membership and merged rows
(reference: waveform_analysis/core/plugins/builtin/cpu/hit_merge.py:52-322).

The chain (per hardware channel, ordered by absolute start, greedy with a total-width cap) and the per-cluster
reductions run on the GPU (wfa_hit_merge_count / _fill / _emit); this module resolves the hit table's column
names the way the reference's `_pick` does and assembles the structured outputs from the index tables.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .dtypes import HIT_MERGE_CLUSTERS_DTYPE, HIT_MERGED_COMPONENTS_DTYPE, HIT_MERGED_DTYPE


def _pick_name(names, *candidates: str) -> str:
    for name in candidates:
        if name in names:
            return name
    raise KeyError(f"Missing fields {candidates} in HIT_DTYPE")  # hit_merge.py:52-56


def _int_column(hits: np.ndarray, name: str, dtype) -> np.ndarray:
    col = hits[name]
    if col.dtype.kind == "f" and np.any(col != np.floor(col)):
        raise ValueError(f"hit field {name} holds non-integer values; hit merging works on sample indices")
    return np.ascontiguousarray(col, dtype=dtype)


def require_hit_dt(hits: np.ndarray, explicit_dt, plugin_name: str) -> np.ndarray:
    """cpu/_dt_compat.py:55-81 with the data name hit_merge uses."""
    from .plugins._common import require_dt_array

    return require_dt_array(hits, explicit_dt=explicit_dt, plugin_name=plugin_name, data_name="hit_threshold[channel]")


def compute_cluster_rows(sess, hits: np.ndarray, merge_gap_ns: float, max_total_width_ns: float, explicit_dt,
                         plugin_name: str) -> np.ndarray:
    """hit_merge.py:115-191 -> HIT_MERGE_CLUSTERS_DTYPE rows (cluster_index, hit_index)."""
    if len(hits) == 0:
        return np.zeros(0, dtype=HIT_MERGE_CLUSTERS_DTYPE)
    names = hits.dtype.names or ()
    if "channel" not in names:
        raise ValueError(f"{plugin_name} requires hit data with a 'channel' field")
    n = len(hits)
    boards = hits["board"] if "board" in names else np.zeros(n, dtype=np.int16)
    dt_values = require_hit_dt(hits, explicit_dt, plugin_name)
    ts = hits[_pick_name(names, "timestamp", "hit_timestamp_ps")]
    pos = _int_column(hits, _pick_name(names, "position", "hit_sample_idx"), np.int64)
    start = _int_column(hits, _pick_name(names, "edge_start", "sample_start", "hit_left_sample_idx"), np.int32)
    end = _int_column(hits, _pick_name(names, "edge_end", "sample_end", "hit_right_sample_idx"), np.int32)
    order, offset = sess.hit_merge_clusters(ts, pos, start, end, dt_values, boards, hits["channel"],
                                            float(merge_gap_ns), float(max_total_width_ns))
    rows = np.zeros(n, dtype=HIT_MERGE_CLUSTERS_DTYPE)
    rows["cluster_index"] = np.repeat(np.arange(len(offset) - 1, dtype=np.int64), np.diff(offset))
    rows["hit_index"] = order
    return rows


def cluster_bounds(cluster_rows: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """hit_merge.py:224-236: (cluster index of each run, offsets of the runs)."""
    ci = np.asarray(cluster_rows["cluster_index"], dtype=np.int64)
    if len(ci) == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(1, dtype=np.int64)
    cuts = np.flatnonzero(np.diff(ci) != 0) + 1
    offsets = np.concatenate(([0], cuts, [len(ci)])).astype(np.int64)
    return ci[offsets[:-1]], offsets


def compute_merged_rows(sess, hits: np.ndarray, cluster_rows: np.ndarray, explicit_dt, plugin_name: str) -> np.ndarray:
    """hit_merge.py:256-322 + 382-405: one HIT_MERGED_DTYPE row per run of cluster_rows."""
    ids, offsets = cluster_bounds(cluster_rows)
    k = len(ids)
    if k == 0:
        return np.zeros(0, dtype=HIT_MERGED_DTYPE)
    if not np.array_equal(ids, np.arange(k)):
        raise ValueError("hit_merge_clusters rows are not ordered by cluster_index without gaps")
    names = hits.dtype.names or ()
    dt_values = require_hit_dt(hits, explicit_dt, plugin_name)
    member = np.asarray(cluster_rows["hit_index"], dtype=np.int64)
    if len(member) and (member.min() < 0 or member.max() >= len(hits)):
        raise KeyError(int(member[(member < 0) | (member >= len(hits))][0]))  # enriched_lookup[...] in the reference
    # sample window of a cluster: (sample_start, sample_end) if present, else (edge_start, edge_end), else none
    if {"sample_start", "sample_end"}.issubset(names):
        w_start, w_end = "sample_start", "sample_end"
    elif {"edge_start", "edge_end"}.issubset(names):
        w_start, w_end = "edge_start", "edge_end"
    else:
        w_start = w_end = None
    n = len(hits)
    if w_start is None:
        win_s = np.full(n, -1, dtype=np.int32)
        win_e = np.full(n, -1, dtype=np.int32)
    else:
        win_s, win_e = _int_column(hits, w_start, np.int32), _int_column(hits, w_end, np.int32)
    red = sess.hit_merge_emit(hits["timestamp"], win_s, win_e, hits["record_id"], hits["height"], hits["integral"],
                              member, offsets)
    counts = np.diff(offsets)
    single = counts == 1
    anchor = red["anchor"]
    a = hits[anchor]
    out = np.zeros(k, dtype=HIT_MERGED_DTYPE)
    out["position"] = a["position"]
    out["timestamp"] = a["timestamp"]
    out["board"] = a["board"] if "board" in names else 0
    out["channel"] = a["channel"]
    out["record_id"] = a["record_id"]
    out["dt"] = a["dt"] if "dt" in names else dt_values[anchor]
    out["rise_time"] = a["rise_time"] if "rise_time" in names else 0.0
    out["fall_time"] = a["fall_time"] if "fall_time" in names else 0.0
    out["component_offset"] = offsets[:-1]
    out["component_count"] = counts
    # clusters of several hits: reductions; clusters of one hit: the hit itself (hit_merge.py:266-284)
    out["height"] = red["height"]
    out["integral"] = red["integral"]
    out["sample_start"] = red["sample_start"]
    out["sample_end"] = red["sample_end"]
    out["width"] = red["width"]
    if w_start is None:  # no window fields at all: only reached for multi-hit clusters, singles need _pick to succeed
        if np.any(single):
            _pick_name(names, "sample_start", "edge_start")
    if np.any(single):
        s_name = _pick_name(names, "sample_start", "edge_start")
        e_name = _pick_name(names, "sample_end", "edge_end")
        one = a[single]
        out["height"][single] = one["height"]
        out["integral"][single] = one["integral"]
        out["sample_start"][single] = one[s_name]
        out["sample_end"][single] = one[e_name]
        out["width"][single] = one["width"]
    return out


def compute_component_rows(merged: np.ndarray, cluster_rows: np.ndarray) -> np.ndarray:
    """hit_merge.py:437-532 (table checks + flat copy)."""
    if len(merged) == 0 or len(cluster_rows) == 0:
        return np.zeros(0, dtype=HIT_MERGED_COMPONENTS_DTYPE)
    ids, offsets = cluster_bounds(cluster_rows)
    if len(ids) != len(merged):
        raise ValueError("hit_merged_components cluster count does not match hit_merged rows: "
                         f"clusters={len(ids)}, hit_merged={len(merged)}")
    names = merged.dtype.names or ()
    starts, counts = offsets[:-1], np.diff(offsets)
    for i in range(len(ids)):  # messages name the first offending row, like the reference's loop
        if "component_offset" in names and int(merged["component_offset"][i]) != int(starts[i]):
            raise ValueError(f"hit_merged[{i}] component_offset mismatch: expected {int(starts[i])}, "
                             f"got {int(merged['component_offset'][i])}")
        if "component_count" in names and int(merged["component_count"][i]) != int(counts[i]):
            raise ValueError(f"hit_merged[{i}] component_count mismatch: expected {int(counts[i])}, "
                             f"got {int(merged['component_count'][i])}")
        if int(ids[i]) != i:
            raise ValueError("hit_merge_clusters rows are not ordered by cluster_index without gaps")
    out = np.zeros(len(cluster_rows), dtype=HIT_MERGED_COMPONENTS_DTYPE)
    out["merged_index"] = np.repeat(np.arange(len(ids), dtype=np.int64), counts)
    out["hit_index"] = cluster_rows["hit_index"]
    return out


def resolve_merge_config(context: Any, plugin: Any):
    from .plugins._common import resolve_dt_config

    return (float(context.get_config(plugin, "merge_gap_ns")), float(context.get_config(plugin, "max_total_width_ns")),
            resolve_dt_config(context, plugin, deprecated_keys=("sampling_interval_ns", "dt_ns")))
