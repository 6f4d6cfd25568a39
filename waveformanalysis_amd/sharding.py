"""Channel sharding of a run across GPUs and re-assembly of per-rank hit rows.

Records are independent for every stage up to event grouping, so a run
shards by hardware channel with no data-path collective: rank g gets the records of the channels
dealt to it, packed into its own contiguous wave_pool.  The only exchange is the gather of the
(60-byte) hit rows to one rank, where they are put back into the reference's order
(record index, start sample) -- waveform_analysis/core/plugins/builtin/cpu/hit_finder.py:354-366 --
before `group_hit_windows` (core/processing/event_grouping.py:286-471) consumes them.

The gather is `DeviceSession.rccl_gather_rows` (RCCL over xGMI, device-resident rows); the CPU tests of the
sharding logic bring their own transport (tests/dist_util.py: torch.distributed / gloo).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def channel_assignment(boards: np.ndarray, channels: np.ndarray, n_shards: int) -> np.ndarray:
    """Shard index per record: distinct (board, channel) pairs, sorted, dealt round-robin."""
    keys = np.asarray(boards, dtype=np.int64) * 65536 + (np.asarray(channels, dtype=np.int64) & 0xFFFF)
    uniq, inv = np.unique(keys, return_inverse=True)
    return (inv % int(n_shards)).astype(np.int32)


@dataclass
class Shard:
    records: np.ndarray      # records of this shard, wave_offset rewritten for `wave_pool`
    wave_pool: np.ndarray    # contiguous samples of those records, in record order
    orig_index: np.ndarray   # index of every shard record in the full records array
    max_len: int             # padded width of the FULL run (hit_finder.py:364,370 semantics)


def make_shard(records: np.ndarray, wave_pool: np.ndarray, n_shards: int, shard: int) -> Shard:
    names = records.dtype.names or ()
    boards = records["board"] if "board" in names else np.zeros(len(records), dtype=np.int16)
    channels = records["channel"] if "channel" in names else np.zeros(len(records), dtype=np.int16)
    sel = np.flatnonzero(channel_assignment(boards, channels, n_shards) == shard)
    sub = records[sel].copy()
    lengths = sub["event_length"].astype(np.int64)
    new_off = np.zeros(len(sub), dtype=np.int64)
    if len(sub):
        new_off[1:] = np.cumsum(lengths)[:-1]
    pool = np.empty(int(lengths.sum()), dtype=wave_pool.dtype)
    old_off = sub["wave_offset"].astype(np.int64)
    uniform = len(sub) > 0 and np.all(lengths == lengths[0])
    if uniform:
        L = int(lengths[0])
        idx = old_off[:, None] + np.arange(L, dtype=np.int64)[None, :]
        pool[:] = wave_pool[idx].reshape(-1)
    else:
        for i in range(len(sub)):
            pool[new_off[i] : new_off[i] + lengths[i]] = wave_pool[old_off[i] : old_off[i] + lengths[i]]
    sub["wave_offset"] = new_off
    max_len = int(records["event_length"].max()) if len(records) else 0
    return Shard(sub, pool, sel.astype(np.int64), max_len)


def merge_rows(rows_by_rank: list[np.ndarray], orig_index_by_rank: list[np.ndarray],
               shard_records_by_rank: list[np.ndarray], key_field: str = "record_id") -> np.ndarray:
    """Concatenate per-rank rows and restore the reference order (original record index, then the
    order inside the record, which every rank already produced)."""
    if not rows_by_rank:
        return np.zeros(0)
    parts, keys = [], []
    for rows, orig, recs in zip(rows_by_rank, orig_index_by_rank, shard_records_by_rank):
        if len(rows) == 0:
            continue
        rid = recs[key_field].astype(np.int64)
        order = np.argsort(rid, kind="stable")
        pos = np.searchsorted(rid[order], rows[key_field].astype(np.int64))
        keys.append(orig[order[pos]])
        parts.append(rows)
    if not parts:
        return rows_by_rank[0][:0].copy()
    rows = np.concatenate(parts)
    key = np.concatenate(keys)
    return rows[np.argsort(key, kind="stable")]


__all__ = ["channel_assignment", "Shard", "make_shard", "merge_rows"]
