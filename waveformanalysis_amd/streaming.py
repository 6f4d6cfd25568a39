"""Streaming chunk interface for the HIP hit finder, with a device pool behind it.

Mirrors the part of the reference's streaming contract a chunk plugin needs
(waveform_analysis/core/plugins/core/streaming.py:119-176,362-378,740-860 and
core/processing/chunk.py:77-206): a `Chunk` carries a time-contiguous slice of a structured array
with [start, end) bounds in ps; `compute_chunk(chunk, context, run_id)` maps one chunk to one chunk and
may be called concurrently from worker threads, so it keeps no per-call state on `self`.  Where the
reference hands chunks to an ExecutorManager thread pool (`_compute_parallel`, streaming.py:740-860), the
classes here override `_compute_parallel` to hand them to `DeviceSession`s (one HIP stream on one GPU)
borrowed from a `DevicePool`.

Inside a reference installation `HipStreamingPlugin` IS a subclass of the reference's `StreamingPlugin` and the
chunks are the reference's `Chunk` objects, so `get_streaming_context(ctx, run).get_stream("hit_threshold_stream")`
(streaming.py:977-1068) drives it like any of its own streaming plugins; standalone (tests, GPU box) the same
class sits on the restatement below.
"""

from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Any

import numpy as np

from . import _lib
from .chunk import (
    DEFAULT_BREAK_THRESHOLD_PS,
    DT_FIELD,
    ENDTIME_FIELD,
    LENGTH_FIELD,
    TIME_FIELD,
    TIMESTAMP_FIELD,
    Chunk,
    check_chunk_boundaries,
    get_endtime,
    select_time_range,
    split_by_breaks,
)
from .device import DevicePool, default_pool
from .dtypes import THRESHOLD_HIT_DTYPE
from .plugin_api import Option
from .sg_plan import normalize_window

try:  # inside a reference installation: be a real subclass of its StreamingPlugin, speak its Chunk
    from waveform_analysis.core.plugins.core.streaming import StreamingPlugin as _RefStreamingPlugin  # type: ignore
    from waveform_analysis.core.processing.chunk import Chunk  # type: ignore  # noqa: F811
except Exception:  # standalone: the restatements (same constructor, same fields)
    _RefStreamingPlugin = None


def _pick_time_field(data, preferred: str):
    """streaming.py:105-116."""
    if not hasattr(data, "dtype") or data.dtype.names is None:
        return None
    if preferred in data.dtype.names:
        return preferred
    if preferred == TIME_FIELD and TIMESTAMP_FIELD in data.dtype.names:
        return TIMESTAMP_FIELD
    if preferred == TIMESTAMP_FIELD and TIME_FIELD in data.dtype.names:
        return TIME_FIELD
    return None


class _StandaloneStreamingPlugin:
    """Chunk-stream driver with the knobs and the chunking rules of the reference's StreamingPlugin
    (core/plugins/core/streaming.py:119-176, 318-360, 380-445, 463-545, 592-691, 882-910):

    * a static structured array is cut into segments at time breaks (`break_threshold_ps`), segments into chunks of
      `chunk_size` rows, each chunk widened by the halo and carrying its core range as metadata main_start/main_end;
    * `compute_chunk` results are wrapped, clipped back to the core range (`clip_strict`) and boundary-checked;
    * `compute()` is a generator that yields the results in input order; parallel runs go through
      `_compute_parallel` (streaming.py:740-860), serial ones (stateful plugins, `reset_state()` at every new segment)
      through the loop below.
    """

    provides = "stream"
    depends_on: list = []
    output_kind = "stream"
    chunk_size: int = 50000
    parallel: bool = True
    parallel_batch_size = None
    executor_type: str = "thread"
    max_workers = None
    time_field: str = TIMESTAMP_FIELD
    dt_field: str = DT_FIELD
    length_field: str = LENGTH_FIELD
    endtime_field: str = ENDTIME_FIELD
    dt = None
    output_time_field: str = TIMESTAMP_FIELD
    output_endtime_field: str = ENDTIME_FIELD
    output_data_kind: str = "stream"
    required_halo_ns: int = 0
    required_halo_left_ns: int = 0
    required_halo_right_ns: int = 0
    clip_strict: bool = False
    is_stateful: bool = False
    reset_on_break: bool = True
    break_threshold_ps: int = DEFAULT_BREAK_THRESHOLD_PS

    # -- to override --------------------------------------------------------------------------------------
    def compute_chunk(self, chunk: Chunk, context: Any, run_id: str, **kwargs):
        return chunk

    def reset_state(self) -> None:
        return None

    # -- chunking -----------------------------------------------------------------------------------------
    def _get_required_halo(self) -> tuple[int, int]:
        left = self.required_halo_left_ns or 0
        right = self.required_halo_right_ns or 0
        if self.required_halo_ns:
            left, right = max(left, self.required_halo_ns), max(right, self.required_halo_ns)
        return int(left), int(right)

    def _time_kw(self) -> dict:
        return dict(endtime_field=self.endtime_field, dt_field=self.dt_field, length_field=self.length_field, dt=self.dt)

    def _endtime_of(self, data: np.ndarray, time_field: str) -> np.ndarray:
        """End of every row in the units of `time_field`; the reference's rule is time + dt * length (chunk.py:263-306).
        Subclasses whose time field and dt use different units override this (see HipThresholdHitStream)."""
        return get_endtime(data, time_field=time_field, **self._time_kw())

    def _iter_segments(self, data: np.ndarray, time_field: str):
        if self.break_threshold_ps and self.break_threshold_ps > 0:
            for seg_id, (seg, info) in enumerate(split_by_breaks(data, break_threshold_ps=self.break_threshold_ps,
                                                                 min_chunk_size=1, time_field=time_field,
                                                                 **self._time_kw())):
                yield seg, int(info.start_time), int(np.max(self._endtime_of(seg, time_field))), seg_id
            return
        if len(data) == 0:
            return
        yield data, int(np.min(data[time_field])), int(np.max(self._endtime_of(data, time_field))), 0

    def _data_to_chunks(self, data: Any, run_id: str):
        if not isinstance(data, np.ndarray):
            raise TypeError("the HIP streaming driver chunks numpy arrays (or takes a chunk iterator as it is)")
        tf = _pick_time_field(data, self.time_field)
        if tf:
            halo_left, halo_right = self._get_required_halo()
            for seg, seg_start, seg_end, seg_id in self._iter_segments(data, tf):
                for i in range(0, len(seg), self.chunk_size):
                    main = seg[i : i + self.chunk_size]
                    if len(main) == 0:
                        continue
                    main_start = int(np.min(main[tf]))
                    main_end = int(np.max(self._endtime_of(main, tf)))
                    ext_start = max(seg_start, main_start - halo_left)
                    ext_end = min(seg_end, main_end + halo_right)
                    ext = select_time_range(seg, start=ext_start, end=ext_end, strict=False, time_field=tf,
                                            **self._time_kw())
                    yield Chunk(ext, ext_start, ext_end, run_id=run_id, data_type=self.provides, time_field=tf,
                                metadata={"main_start": main_start, "main_end": main_end, "segment_id": seg_id},
                                **self._time_kw())
            return
        for i in range(0, len(data), self.chunk_size):
            part = data[i : i + self.chunk_size]
            if len(part) == 0:
                continue
            yield Chunk(part, i, i + len(part), run_id=run_id, data_type=self.provides, time_field=self.time_field,
                        metadata={"main_start": i, "main_end": i + len(part), "segment_id": 0}, **self._time_kw())

    def _get_input_chunks(self, context: Any, run_id: str, **kwargs):
        deps = self.resolve_depends_on(context, run_id=run_id) if hasattr(self, "resolve_depends_on") else self.depends_on
        if not deps:
            return iter([])
        dep = context.get_data(run_id, deps[0])
        if hasattr(dep, "__next__") or (hasattr(dep, "__iter__") and not isinstance(dep, (np.ndarray, list))):
            return dep
        return self._data_to_chunks(dep, run_id)

    # -- results ------------------------------------------------------------------------------------------
    def _postprocess_result(self, result: Any, input_chunk: Chunk):
        if result is None:
            return None
        if not isinstance(result, Chunk):
            result = Chunk(np.asarray(result), input_chunk.metadata.get("main_start", input_chunk.start),
                           input_chunk.metadata.get("main_end", input_chunk.end), run_id=input_chunk.run_id,
                           data_type=self.provides, data_kind=self.output_data_kind, time_field=self.output_time_field,
                           dt_field=self.dt_field, length_field=self.length_field,
                           endtime_field=self.output_endtime_field, dt=self.dt,
                           metadata={"segment_id": input_chunk.metadata.get("segment_id")})
        main_start, main_end = input_chunk.metadata.get("main_start"), input_chunk.metadata.get("main_end")
        if main_start is None or main_end is None:
            return result
        if not hasattr(result.data, "dtype") or result.data.dtype.names is None:
            return result
        clipped = select_time_range(result.data, start=main_start, end=main_end, strict=self.clip_strict,
                                    time_field=result.time_field, endtime_field=result.endtime_field,
                                    dt_field=result.dt_field, length_field=result.length_field, dt=result.dt)
        if len(clipped) == 0:
            return None
        meta = dict(result.metadata)
        meta.update({"main_start": main_start, "main_end": main_end,
                     "segment_id": input_chunk.metadata.get("segment_id")})
        return Chunk(clipped, int(main_start), int(main_end), run_id=result.run_id, data_type=result.data_type,
                     data_kind=result.data_kind, time_field=result.time_field, dt_field=result.dt_field,
                     length_field=result.length_field, endtime_field=result.endtime_field, dt=result.dt, metadata=meta)

    def _validate_chunk(self, chunk: Chunk) -> None:
        if len(chunk.data) == 0 or not hasattr(chunk.data, "dtype") or chunk.data.dtype.names is None:
            return
        v = check_chunk_boundaries(chunk.data, chunk.start, chunk.end, time_field=chunk.time_field,
                                   endtime_field=chunk.endtime_field, dt_field=chunk.dt_field,
                                   length_field=chunk.length_field, dt=chunk.dt)
        if not v.is_valid:
            raise ValueError(f"Chunk boundary violation in {self.provides}: {v.errors}")

    # -- driver -------------------------------------------------------------------------------------------
    STREAMING_CONFIG_KEYS = ("chunk_size", "parallel", "executor_type", "max_workers", "parallel_batch_size",
                             "break_threshold_ps", "required_halo_ns", "required_halo_left_ns",
                             "required_halo_right_ns", "clip_strict")

    def _apply_streaming_config(self, streaming_config) -> None:
        """streaming.py:264-316: every run starts from the CLASS defaults, then `streaming_config` (a dict) wins;
        unknown keys are reported and dropped."""
        if streaming_config is not None and not isinstance(streaming_config, dict):
            raise TypeError("streaming_config must be a dict")
        merged = {k: getattr(type(self), k, getattr(self, k, None)) for k in self.STREAMING_CONFIG_KEYS}
        unknown = []
        for key, value in (streaming_config or {}).items():
            if key in merged:
                merged[key] = value
            elif key != "executor_config":
                unknown.append(key)
        if unknown:
            import warnings

            warnings.warn(f"Unknown streaming_config keys for {self.provides}: {sorted(unknown)}", UserWarning, stacklevel=3)
        for key, value in merged.items():
            setattr(self, key, value)

    def compute(self, context: Any, run_id: str, **kwargs):
        self._apply_streaming_config(kwargs.pop("streaming_config", None))
        chunks = self._get_input_chunks(context, run_id, **kwargs)
        parallel = self.parallel and not self.is_stateful and (self.max_workers is None or self.max_workers > 1)
        if not parallel:
            last_segment = None
            for chunk in chunks:
                if self.is_stateful and self.reset_on_break:
                    seg = chunk.metadata.get("segment_id")
                    if seg is not None and seg != last_segment:
                        self.reset_state()
                        last_segment = seg
                result = self._process(chunk, context, run_id, kwargs)
                if result is not None:
                    yield result
            return
        for result in self._compute_parallel(chunks, context, run_id, **kwargs):
            if result is not None:
                yield result


class _DevicePoolMixin:
    """What both bases get on top: results through `_process`, and `_compute_parallel` (streaming.py:740-860) handing
    chunks to worker threads that each borrow a DeviceSession from a DevicePool -- chunk k on GPU (k mod n_gpus) --
    where the reference submits to its ExecutorManager.  Results come back in input order; an exception cancels
    what has not started and is re-raised."""

    device_pool = None

    def _pool(self, context: Any = None) -> DevicePool:
        return self.device_pool or getattr(context, "wfa_device_pool", None) or default_pool()

    def _process(self, chunk: Chunk, context: Any, run_id: str, kwargs: dict):
        result = self._postprocess_result(self.compute_chunk(chunk, context, run_id, **kwargs), chunk)
        if result is not None:
            self._validate_chunk(result)
        return result

    def _compute_parallel(self, input_chunks, context: Any, run_id: str, executor_config: dict | None = None, **kwargs):
        workers = int((executor_config or {}).get("max_workers") or self.max_workers or 4)
        batch = int(self.parallel_batch_size or 2 * workers)
        with ThreadPoolExecutor(max_workers=workers) as ex:
            pending: list = []
            try:
                for chunk in input_chunks:
                    pending.append(ex.submit(self._process, chunk, context, run_id, kwargs))
                    if len(pending) >= batch:
                        result = pending.pop(0).result()
                        if result is not None:
                            yield result
                while pending:
                    result = pending.pop(0).result()
                    if result is not None:
                        yield result
            finally:
                for fut in pending:
                    fut.cancel()


class HipStreamingPlugin(_DevicePoolMixin, _RefStreamingPlugin or _StandaloneStreamingPlugin):
    """Base of the HIP streaming plugins: the reference's StreamingPlugin when that is importable, the restatement above
    otherwise, with the device pool behind `_compute_parallel`."""

    provides = "stream"


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


_UNSET = object()


class HipThresholdHitStream(HipStreamingPlugin):
    """compute_chunk() = threshold hits of one chunk of records (fused SG filter optional).

    `compute(context, run_id)` streams over the static `records` array (breaks at 10^13 ps, 50 000 records per chunk);
    hits are instantaneous rows (time = `timestamp`).  Parallel runs (`_compute_parallel`) drive two borrowed sessions as
    a double buffer: chunk k + 1 goes through the pinned staging ring of one session while the kernels of chunk k run on
    the other.  Options are read from the Context like those of HipThresholdHitPlugin (hit_finder.py:82-130); a value
    given to the constructor wins."""

    provides = "hit_threshold_stream"
    depends_on = ["records", "wave_pool"]
    description = "Threshold hits of the records stream, chunk by chunk (HIP, gfx950)."
    version = "0.11.0+hip1"
    output_dtype = THRESHOLD_HIT_DTYPE
    save_when = "never"
    chunk_size = 50_000          # reference default (streaming.py:153-176)
    length_field = "event_length"
    output_data_kind = "hits"

    options = {
        "threshold": Option(default=10.0, type=float, help="hit threshold"),
        "left_extension": Option(default=2, type=int, help="samples added left of a hit"),
        "right_extension": Option(default=2, type=int, help="samples added right of a hit"),
        "use_filtered": Option(default=False, type=bool, help="threshold the Savitzky-Golay filtered waveform (fused)"),
        "sg_window_size": Option(default=11, type=int, help="Savitzky-Golay window"),
        "sg_poly_order": Option(default=2, type=int, help="Savitzky-Golay polynomial order"),
    }

    def __init__(self, threshold=_UNSET, left_extension=_UNSET, right_extension=_UNSET, use_filtered=_UNSET,
                 sg_window_size=_UNSET, sg_poly_order=_UNSET, max_len: int = 0, device_pool: DevicePool | None = None):
        super().__init__()
        given = dict(threshold=threshold, left_extension=left_extension, right_extension=right_extension,
                     use_filtered=use_filtered, sg_window_size=sg_window_size, sg_poly_order=sg_poly_order)
        self._given = {k: v for k, v in given.items() if v is not _UNSET}
        self.max_len = int(max_len)   # padded width of the whole run (hit_finder.py:364), 0 = per chunk
        self.device_pool = device_pool
        self._configure(None)

    def _configure(self, context: Any) -> None:
        """Constructor argument > Context configuration > option default."""
        def value(name):
            if name in self._given:
                return self._given[name]
            if context is not None and hasattr(context, "get_config"):
                v = context.get_config(self, name)
                if v is not None:
                    return v
            return self.options[name].default

        self.threshold = float(value("threshold"))
        self.le, self.re = max(0, int(value("left_extension"))), max(0, int(value("right_extension")))
        self.use_filtered = bool(value("use_filtered"))
        self.sg = normalize_window(value("sg_window_size"), value("sg_poly_order"))

    def _endtime_of(self, data: np.ndarray, time_field: str) -> np.ndarray:
        # records: timestamp in ps, dt in ns per sample -> the end of a record in ps
        return data[time_field].astype(np.int64) + data["dt"].astype(np.int64) * 1000 * data["event_length"].astype(np.int64)

    def _data_to_chunks(self, data: Any, run_id: str):
        """Records of different channels overlap in time, so time-range chunks (the base rule) share records at
        their borders and every shared record's hits would come out twice.  Hits depend on their own record only:
        a chunk is `chunk_size` consecutive records of a time segment, nothing is widened and nothing clipped
        (no main_start / main_end in the metadata)."""
        if not isinstance(data, np.ndarray) or data.dtype.names is None:
            raise TypeError("hit_threshold_stream chunks the structured `records` array")
        if len(data) == 0:
            return
        if self.break_threshold_ps and self.break_threshold_ps > 0:
            segments = [seg for seg, _info in split_by_breaks(
                data, break_threshold_ps=self.break_threshold_ps, min_chunk_size=1, time_field="timestamp",
                endtime_field=self.endtime_field, dt_field=self.dt_field, length_field=self.length_field, dt=self.dt)]
        else:
            segments = [data]
        for seg_id, seg in enumerate(segments):
            for ch in records_to_chunks(seg, self.chunk_size, run_id):
                ch.metadata["segment_id"] = seg_id
                yield ch

    def _empty(self, chunk: Chunk, run_id: str) -> Chunk:
        return Chunk(np.zeros(0, dtype=THRESHOLD_HIT_DTYPE), chunk.start, chunk.end, run_id, self.provides,
                     data_kind="hits", time_field="timestamp")

    def _stage(self, sess, chunk: Chunk, wave_pool: np.ndarray) -> None:
        """Upload the chunk's samples + records and queue its hit pass on `sess` (returns without waiting for the kernels
        once the session has sized its row buffers: wfa_hits_enqueue)."""
        recs = chunk.data
        # the chunk's samples are one contiguous slice of the pool for time-sorted records
        lo = int(recs["wave_offset"].min())
        hi = int((recs["wave_offset"].astype(np.int64) + recs["event_length"]).max())
        sub = recs.copy()
        sub["wave_offset"] -= lo
        sess.upload_pool(np.ascontiguousarray(wave_pool[lo:hi]))   # pinned staging ring: chunk by chunk onto the wire
        sess.upload_records(sub, self.threshold)
        if self.use_filtered:
            sess.set_sg_plan(*self.sg)
        sess.hits_enqueue(_lib.SRC_SG_FUSED if self.use_filtered else _lib.SRC_RAW, (0, 0), self.le, self.re, self.max_len)

    def _collect(self, sess, chunk: Chunk, run_id: str) -> Chunk:
        hits = sess._fill_hits(sess.hits_wait())
        # a record that overlaps the chunk's end (non-strict halo selection) has hits behind it: the result chunk
        # spans every record it was computed from; the driver clips it back to the core range
        end = max(int(chunk.end), int(self._endtime_of(chunk.data, "timestamp").max()) + 1)
        return Chunk(hits, chunk.start, end, run_id, self.provides, data_kind="hits", time_field="timestamp",
                     metadata=dict(chunk.metadata))

    def compute_chunk(self, chunk: Chunk, context: Any, run_id: str, **_kw) -> Chunk:
        if len(chunk.data) == 0:
            return self._empty(chunk, run_id)
        self._configure(context)
        wave_pool = context.get_data(run_id, "wave_pool")
        # worker threads come and go with every compute(): the session is borrowed for this chunk only
        with self._pool(context).borrow() as sess:
            self._stage(sess, chunk, wave_pool)
            return self._collect(sess, chunk, run_id)

    def _pipeline(self, input_chunks, context: Any, run_id: str, timeline: list | None = None,
                  max_workers: int | None = None):
        """(input chunk, raw result) pairs in input order.  One host thread drives a ring of sessions -- two per device
        of the pool (each its own HIP stream and device buffers), so chunk k runs on device k mod n_devices: while the
        kernels of chunks k - n + 1 .. k run, chunk k + 1 is uploaded through the next session's pinned staging ring;
        nobody waits for chunk k before chunk k + 1 is queued.  `max_workers` (the reference's executor knob,
        streaming.py:740-860) caps the ring.  `timeline` (optional) receives (k, t_stage_begin, t_queued, t_collected)
        host times per chunk."""
        import time
        from collections import deque

        self._configure(context)
        wave_pool = context.get_data(run_id, "wave_pool")
        pool = self._pool(context)
        n = max(2, 2 * len(pool.device_ids))
        if max_workers:
            n = max(2, min(n, int(max_workers)))
        n = min(n, pool.max_sessions)
        if n < 2:
            raise ValueError("the streaming hit pipeline needs a device pool with max_sessions >= 2")
        with pool.borrow_many(n) as ring:
            pending: deque = deque()  # (k, chunk, session, stamps) of the chunks whose kernels are queued, oldest first

            def finish():
                pk, pchunk, ps, stamps = pending.popleft()
                result = self._collect(ps, pchunk, run_id)
                if timeline is not None:
                    timeline.append((pk, *stamps, time.perf_counter()))
                return pchunk, result

            staged = 0
            for k, chunk in enumerate(input_chunks):
                if len(chunk.data) == 0:
                    while pending:
                        yield finish()
                    yield chunk, self._empty(chunk, run_id)
                    continue
                sess = ring[staged % n]  # free: results are collected in order, at most n - 1 are pending here
                staged += 1
                t0 = time.perf_counter()
                self._stage(sess, chunk, wave_pool)
                t1 = time.perf_counter()
                pending.append((k, chunk, sess, (t0, t1)))
                while len(pending) > n - 1:
                    yield finish()
            while pending:
                yield finish()

    def _compute_parallel(self, input_chunks, context: Any, run_id: str, executor_config: dict | None = None, **kwargs):
        """streaming.py:740-860, with the device pool in the place of the executor."""
        workers = (executor_config or {}).get("max_workers") or self.max_workers
        for chunk, raw in self._pipeline(input_chunks, context, run_id, max_workers=workers):
            result = self._postprocess_result(raw, chunk)
            if result is not None:
                self._validate_chunk(result)
                yield result

    def run_chunks(self, chunks: list[Chunk], context: Any, run_id: str, max_workers: int = 4,
                   timeline: list | None = None) -> list[Chunk]:
        """The raw results of `chunks`, in order (the bench and the parity tests call this with ready-made chunks)."""
        if not self.parallel or max_workers <= 1 or len(chunks) <= 1:
            return [self.compute_chunk(c, context, run_id) for c in chunks]
        return [raw for _chunk, raw in self._pipeline(chunks, context, run_id, timeline, max_workers=max_workers)]


__all__ = ["Chunk", "records_to_chunks", "HipStreamingPlugin", "HipThresholdHitStream"]
