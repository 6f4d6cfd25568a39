"""Streaming chunk interface for the HIP hit finder, with a device pool behind it.

Mirrors the part of the reference's streaming contract a chunk plugin needs
(waveform_analysis/core/plugins/core/streaming.py:119-176,362-378,740-860 and
core/processing/chunk.py:77-206): a `Chunk` carries a time-contiguous slice of a structured array
with [start, end) bounds in ps; `compute_chunk(chunk, context, run_id)` maps one chunk to one chunk and
may be called concurrently from worker threads, so it keeps no per-call state on `self`.  Where the
reference hands chunks to an ExecutorManager thread pool (`_compute_parallel`), `run_chunks` hands
them to worker threads that each borrow a `DeviceSession` (one HIP stream on one GPU) from a
`DevicePool`, chunk k -> GPU (k mod n_gpus).
"""

from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Any

import numpy as np

from . import _lib
from .chunk import Chunk
from .device import DevicePool, default_pool
from .dtypes import THRESHOLD_HIT_DTYPE
from .sg_plan import normalize_window


def records_to_chunks(records: np.ndarray, chunk_size: int, run_id: str = "") -> list[Chunk]:
    """Time-ordered records -> chunks of at most chunk_size records (streaming.py:592-691, the
    `chunk_size` slicing; records are already sorted by timestamp)."""
    out = []
    for lo in range(0, len(records), int(chunk_size)):
        sub = records[lo : lo + int(chunk_size)]
        start = int(sub["timestamp"][0])
        # endtime = time + dt * length (core/processing/chunk.py:388-431), in ps
        endtime = sub["timestamp"].astype(np.int64) + sub["dt"].astype(np.int64) * 1000 * sub["event_length"]
        end = int(endtime.max()) + 1
        out.append(Chunk(sub, start, end, run_id=run_id, data_type="records", data_kind="records",
                         time_field="timestamp", length_field="event_length", metadata={"first_record": lo}))
    return out


class HipThresholdHitStream:
    """compute_chunk() = threshold hits of one chunk of records (fused SG filter optional)."""

    provides = "hit_threshold_stream"
    depends_on = ["records", "wave_pool"]
    output_dtype = THRESHOLD_HIT_DTYPE
    output_kind = "stream"
    chunk_size = 50_000          # reference default (streaming.py:153-176)
    parallel = True
    is_stateful = False

    def __init__(self, threshold: float = 10.0, left_extension: int = 2, right_extension: int = 2,
                 use_filtered: bool = False, sg_window_size: int = 11, sg_poly_order: int = 2,
                 max_len: int = 0, device_pool: DevicePool | None = None):
        self.threshold = float(threshold)
        self.le, self.re = max(0, int(left_extension)), max(0, int(right_extension))
        self.use_filtered = bool(use_filtered)
        self.sg = normalize_window(sg_window_size, sg_poly_order)
        self.max_len = int(max_len)   # padded width of the whole run (hit_finder.py:364), 0 = per chunk
        self.device_pool = device_pool

    def compute_chunk(self, chunk: Chunk, context: Any, run_id: str, **_kw) -> Chunk:
        wave_pool = context.get_data(run_id, "wave_pool")
        recs = chunk.data
        if len(recs) == 0:
            return Chunk(np.zeros(0, dtype=THRESHOLD_HIT_DTYPE), chunk.start, chunk.end, run_id, self.provides,
                         data_kind="hits", time_field="timestamp")
        # the chunk's samples are one contiguous slice of the pool for time-sorted records
        lo = int(recs["wave_offset"].min())
        hi = int((recs["wave_offset"].astype(np.int64) + recs["event_length"]).max())
        sub = recs.copy()
        sub["wave_offset"] -= lo
        sess = (self.device_pool or default_pool()).session()
        sess.upload_pool(np.ascontiguousarray(wave_pool[lo:hi]))
        sess.upload_records(sub, self.threshold)
        if self.use_filtered:
            sess.set_sg_plan(*self.sg)
            hits = sess.threshold_hits(_lib.SRC_SG_FUSED, self.le, self.re, self.max_len)
        else:
            hits = sess.threshold_hits(_lib.SRC_RAW, self.le, self.re, self.max_len)
        return Chunk(hits, chunk.start, chunk.end, run_id, self.provides, data_kind="hits", time_field="timestamp",
                     metadata=dict(chunk.metadata))

    def run_chunks(self, chunks: list[Chunk], context: Any, run_id: str, max_workers: int = 4) -> list[Chunk]:
        """Ordered results, chunks processed concurrently (one session / HIP stream per worker)."""
        if not self.parallel or max_workers <= 1 or len(chunks) <= 1:
            return [self.compute_chunk(c, context, run_id) for c in chunks]
        with ThreadPoolExecutor(max_workers=max_workers) as ex:
            return list(ex.map(lambda c: self.compute_chunk(c, context, run_id), chunks))


__all__ = ["Chunk", "records_to_chunks", "HipThresholdHitStream"]
