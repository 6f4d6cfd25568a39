"""Chunk container and the time-range helpers the streaming path uses
(reference: waveform_analysis/core/processing/chunk.py:27-206, 211-257, 263-306, 388-431, 620-672, 857-928,
1134-1203, 1274-1295).

Host bookkeeping only (no samples touched): a chunk is a structured array plus [start, end) bounds; the
constructor refuses data outside its bounds, `endtime = time + dt * length`, with the reference's field
fallbacks (`time` -> `timestamp`, `length` -> `event_length`) and its formulas, units included.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Generator, Optional

import numpy as np

TIME_FIELD = "time"
DT_FIELD = "dt"
LENGTH_FIELD = "length"
ENDTIME_FIELD = "endtime"
CHANNEL_FIELD = "channel"
TIMESTAMP_FIELD = "timestamp"
EVENT_LENGTH_FIELD = "event_length"
DEFAULT_CHUNK_SIZE = 500_000
DEFAULT_BREAK_THRESHOLD_PS = 10_000_000_000_000


def _structured(data) -> None:
    if not hasattr(data, "dtype") or data.dtype.names is None:
        raise TypeError("Data must be a structured numpy array")


def resolve_time_field(data: np.ndarray, time_field: str) -> str:
    _structured(data)
    if time_field in data.dtype.names:
        return time_field
    if time_field == TIME_FIELD and TIMESTAMP_FIELD in data.dtype.names:
        return TIMESTAMP_FIELD
    return time_field


def resolve_length_field(data: np.ndarray, length_field: str) -> str:
    _structured(data)
    if length_field in data.dtype.names:
        return length_field
    if length_field == LENGTH_FIELD and EVENT_LENGTH_FIELD in data.dtype.names:
        return EVENT_LENGTH_FIELD
    return length_field


def validate_time_fields(data, require_length=True, time_field=TIME_FIELD, dt_field=DT_FIELD,
                         length_field=LENGTH_FIELD, dt=None) -> None:
    _structured(data)
    required = [resolve_time_field(data, time_field)]
    if dt is None:
        required.append(dt_field)
    if require_length:
        required.append(resolve_length_field(data, length_field))
    missing = [f for f in required if f not in data.dtype.names]
    if missing:
        raise KeyError(f"Missing required fields: {missing}")


def compute_endtime(data, time_field=TIME_FIELD, dt_field=DT_FIELD, length_field=LENGTH_FIELD,
                    dt: Optional[float] = None) -> np.ndarray:
    """endtime = time + dt * length as int64 (chunk.py:263-306)."""
    tf = resolve_time_field(data, time_field)
    lf = resolve_length_field(data, length_field)
    validate_time_fields(data, True, tf, dt_field, lf, dt)
    time = data[tf].astype(np.int64)
    length = data[lf].astype(np.int64)
    if dt is None:
        endtime = time + data[dt_field].astype(np.int64) * length
    else:
        endtime = time + (np.asarray(dt, dtype=np.float64) * length)
    return endtime.astype(np.int64)


def get_endtime(data, time_field=TIME_FIELD, endtime_field=ENDTIME_FIELD, dt_field=DT_FIELD,
                length_field=LENGTH_FIELD, dt: Optional[float] = None) -> np.ndarray:
    """The endtime field if present, else computed; rows without length / dt are instantaneous (chunk.py:388-431)."""
    _structured(data)
    if endtime_field in data.dtype.names:
        return data[endtime_field]
    tf = resolve_time_field(data, time_field)
    lf = resolve_length_field(data, length_field)
    if lf not in data.dtype.names:
        return data[tf]
    if dt is None and dt_field not in data.dtype.names:
        return data[tf]
    return compute_endtime(data, tf, dt_field, lf, dt)


class Chunk:
    """Data plus its [start, end) range (chunk.py:77-206)."""

    def __init__(self, data: np.ndarray, start: int, end: int, run_id: str = "unknown", data_type: str = "raw",
                 data_kind: str = "waveforms", time_field: str = TIME_FIELD, dt_field: str = DT_FIELD,
                 length_field: str = LENGTH_FIELD, endtime_field: str = ENDTIME_FIELD, dt: Optional[float] = None,
                 metadata: Optional[dict] = None):
        self.data = data
        self.start = int(start)
        self.end = int(end)
        self.run_id = run_id
        self.data_type = data_type
        self.data_kind = data_kind
        self.dtype = data.dtype
        self.time_field = time_field
        self.dt_field = dt_field
        self.length_field = length_field
        self.endtime_field = endtime_field
        self.dt = dt
        self.metadata = metadata or {}
        if len(data) > 0:
            tf = resolve_time_field(data, self.time_field)
            lf = resolve_length_field(data, self.length_field)
            data_start = int(np.min(data[tf]))
            if data_start < self.start:
                raise ValueError(f"Chunk data starts at {data_start}, before chunk start {self.start}")
            data_end = get_endtime(data, time_field=tf, endtime_field=self.endtime_field, dt_field=self.dt_field,
                                   length_field=lf, dt=self.dt).max()
            if data_end > self.end:
                raise ValueError(f"Chunk data ends at {data_end}, after chunk end {self.end}")

    def __len__(self):
        return len(self.data)

    @property
    def duration(self):
        return self.end - self.start

    @property
    def nbytes(self):
        return self.data.nbytes

    def __repr__(self):
        return f"Chunk({self.run_id}.{self.data_type}: {self.start} - {self.end}, {len(self)} items)"

    def _like(self, data, start, end) -> "Chunk":
        return Chunk(data, start, end, self.run_id, self.data_type, self.data_kind, time_field=self.time_field,
                     dt_field=self.dt_field, length_field=self.length_field, endtime_field=self.endtime_field,
                     dt=self.dt, metadata=self.metadata)

    def split(self, t: int):
        """Two chunks cut at time t (rows by their start time)."""
        t = max(min(t, self.end), self.start)
        mask = self.data[resolve_time_field(self.data, self.time_field)] < t
        return self._like(self.data[mask], self.start, t), self._like(self.data[~mask], t, self.end)


@dataclass
class ChunkInfo:
    start_time: int
    end_time: int
    n_records: int
    chunk_i: int = 0
    run_id: str = ""

    @property
    def duration(self) -> int:
        return self.end_time - self.start_time

    def overlaps(self, other: "ChunkInfo") -> bool:
        return self.start_time < other.end_time and other.start_time < self.end_time

    def contains(self, time: int) -> bool:
        return self.start_time <= time < self.end_time

    def __repr__(self) -> str:
        return (f"ChunkInfo(start={self.start_time}, end={self.end_time}, n={self.n_records}, "
                f"duration={self.duration}ns)")


@dataclass
class ValidationResult:
    is_valid: bool
    errors: list = field(default_factory=list)
    warnings: list = field(default_factory=list)
    stats: dict = field(default_factory=dict)

    def __bool__(self) -> bool:
        return self.is_valid

    def raise_if_invalid(self, prefix: str = ""):
        if not self.is_valid:
            msg = "; ".join(self.errors)
            raise ValueError(f"{prefix}{msg}" if prefix else msg)


def select_time_range(data, start: Optional[int] = None, end: Optional[int] = None, strict: bool = False,
                      time_field=TIME_FIELD, endtime_field=ENDTIME_FIELD, dt_field=DT_FIELD,
                      length_field=LENGTH_FIELD, dt: Optional[float] = None) -> np.ndarray:
    """Rows inside [start, end): fully inside when strict, overlapping otherwise (chunk.py:620-672)."""
    if len(data) == 0:
        return data
    tf = resolve_time_field(data, time_field)
    time = data[tf]
    endtime = get_endtime(data, tf, endtime_field, dt_field, length_field, dt)
    mask = np.ones(len(data), dtype=bool)
    if strict:
        if start is not None:
            mask &= time >= start
        if end is not None:
            mask &= endtime <= end
    else:
        if start is not None:
            mask &= endtime > start
        if end is not None:
            mask &= time < end
    return data[mask]


def split_by_breaks(data, break_threshold_ps: int = DEFAULT_BREAK_THRESHOLD_PS, min_chunk_size: int = 1,
                    time_field=TIME_FIELD, endtime_field=ENDTIME_FIELD, dt_field=DT_FIELD, length_field=LENGTH_FIELD,
                    dt: Optional[float] = None) -> Generator[tuple, None, None]:
    """Cut time-sorted data where the gap to the previous row's end exceeds the threshold (chunk.py:857-928)."""
    if len(data) == 0:
        return
    tf = resolve_time_field(data, time_field)
    time = data[tf]
    endtime = get_endtime(data, tf, endtime_field, dt_field, length_field, dt)
    gaps = time[1:].astype(np.int64) - endtime[:-1].astype(np.int64)
    cuts = np.concatenate([[0], np.where(gaps > break_threshold_ps)[0] + 1, [len(data)]])
    chunk_i = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b - a < min_chunk_size:
            continue
        part = data[a:b]
        info = ChunkInfo(start_time=int(np.min(part[tf])),
                         end_time=int(np.max(get_endtime(part, tf, endtime_field, dt_field, length_field, dt))),
                         n_records=len(part), chunk_i=chunk_i)
        yield part, info
        chunk_i += 1


def check_chunk_boundaries(data, chunk_start: int, chunk_end: int, time_field=TIME_FIELD,
                           endtime_field=ENDTIME_FIELD, dt_field=DT_FIELD, length_field=LENGTH_FIELD,
                           dt: Optional[float] = None) -> ValidationResult:
    """Rows starting before chunk_start or ending after chunk_end (chunk.py:1134-1203)."""
    result = ValidationResult(is_valid=True)
    if len(data) == 0:
        result.stats = {"n_records": 0, "violations": 0}
        return result
    tf = resolve_time_field(data, time_field)
    time = data[tf]
    endtime = get_endtime(data, tf, endtime_field, dt_field, length_field, dt)
    before = time < chunk_start
    n_before = np.sum(before)
    if n_before > 0:
        result.is_valid = False
        result.errors.append(f"{n_before} records start before chunk boundary "
                             f"(earliest: {np.min(time[before])} < {chunk_start})")
    after = endtime > chunk_end
    n_after = np.sum(after)
    if n_after > 0:
        result.is_valid = False
        result.errors.append(f"{n_after} records extend beyond chunk boundary "
                             f"(latest: {np.max(endtime[after])} > {chunk_end})")
    result.stats = {"n_records": len(data), "n_before_start": int(n_before), "n_after_end": int(n_after),
                    "violations": int(n_before + n_after)}
    return result


__all__ = ["Chunk", "ChunkInfo", "ValidationResult", "compute_endtime", "get_endtime", "select_time_range",
           "split_by_breaks", "check_chunk_boundaries", "TIME_FIELD", "DT_FIELD", "LENGTH_FIELD", "ENDTIME_FIELD",
           "TIMESTAMP_FIELD", "EVENT_LENGTH_FIELD", "DEFAULT_CHUNK_SIZE", "DEFAULT_BREAK_THRESHOLD_PS"]
