"""Records builder: global record order and wave_pool packing on the GPU
(reference: waveform_analysis/core/processing/records_builder.py:115-120 `_records_sort_order`, 645-794
`_build_records_from_channels` / `build_records_from_st_waveforms`, 869-945 `merge_records_parts`).

The reference sorts with np.lexsort and then copies every wave slice in a Python loop (or pops a heap per
record when merging parts).  Here the order is a stable device radix sort (wfa_records_sort) and the packing is
one gather kernel (wfa_pool_gather) that leaves the packed pool resident on the GPU, so the hit / feature passes
that follow read it without another upload.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import numpy as np

from .dtypes import RECORDS_DTYPE


@dataclass
class RecordsBundle:
    """records + wave_pool (records_builder.py:35-43)."""

    records: np.ndarray
    wave_pool: np.ndarray


def _session(session):
    if session is not None:
        return session
    from .device import default_pool

    return default_pool().session()


def _empty() -> RecordsBundle:
    return RecordsBundle(np.zeros(0, dtype=RECORDS_DTYPE), np.zeros(0, dtype=np.uint16))


def records_sort_order(records: np.ndarray, session=None) -> np.ndarray:
    """Stable global order by (timestamp, pid, board, channel) (records_builder.py:115-120)."""
    if len(records) == 0:
        return np.zeros(0, dtype=np.int64)
    return _session(session).records_sort_order(records["timestamp"], records["pid"], records["board"], records["channel"])


def build_records_from_st_waveforms(st_waveforms: np.ndarray, default_dt_ns: int = 1, session=None) -> RecordsBundle:
    """records + wave_pool from a dense st_waveforms array, globally sorted (records_builder.py:645-794).

    The reference concatenates the rows per hardware channel and sorts by (timestamp, pid, board, channel, seq);
    board and channel are sort keys ahead of seq, so that is the stable (timestamp, pid, board, channel) order
    of the original rows."""
    if st_waveforms is None or len(st_waveforms) == 0:
        return _empty()
    if not isinstance(st_waveforms, np.ndarray) or st_waveforms.dtype.names is None:
        raise ValueError("st_waveforms must be a structured numpy array")
    names = st_waveforms.dtype.names
    if "board" not in names or "channel" not in names:
        raise ValueError("st_waveforms missing required 'board'/'channel' fields")
    n = len(st_waveforms)
    rec = np.zeros(n, dtype=RECORDS_DTYPE)
    rec["timestamp"] = st_waveforms["timestamp"] if "timestamp" in names else 0
    rec["pid"] = 0
    rec["channel"] = st_waveforms["channel"]
    rec["board"] = st_waveforms["board"].astype(np.int16, copy=False)
    rec["baseline"] = st_waveforms["baseline"] if "baseline" in names else 0.0
    rec["baseline_upstream"] = st_waveforms["baseline_upstream"] if "baseline_upstream" in names else np.nan
    rec["polarity"] = st_waveforms["polarity"] if "polarity" in names else "unknown"
    if "event_length" in names:
        lengths = st_waveforms["event_length"].astype(np.int64, copy=False)
        if lengths.size and lengths.max() > np.iinfo(np.int32).max:
            raise ValueError("event_length exceeds int32 range")
        rec["event_length"] = lengths.astype(np.int32, copy=False)
    elif "wave" in names:
        rec["event_length"] = np.int32(st_waveforms["wave"].shape[1])
    rec["dt"] = st_waveforms["dt"].astype(np.int32, copy=False) if "dt" in names else np.int32(default_dt_ns)
    rec["trigger_type"] = st_waveforms["trigger_type"].astype(np.int16, copy=False) if "trigger_type" in names else 0
    rec["flags"] = st_waveforms["flags"].astype(np.uint32, copy=False) if "flags" in names else 0
    rec["time"] = st_waveforms["time"] if "time" in names else rec["timestamp"] // 1000
    source_record_id = (st_waveforms["record_id"].astype(np.int64, copy=False) if "record_id" in names
                        else np.full(n, -1, dtype=np.int64))

    sess = _session(session)
    order = records_sort_order(rec, sess)
    rec = rec[order]
    source_record_id = source_record_id[order]
    rec["record_id"] = source_record_id if np.all(source_record_id >= 0) else np.arange(n, dtype=np.int64)
    if "wave" not in names:
        raise ValueError("st_waveforms missing 'wave' field required for wave_pool")
    wave = st_waveforms["wave"]
    width = int(wave.shape[-1])
    rec["event_length"] = np.clip(rec["event_length"].astype(np.int64), 0, width).astype(np.int32)
    flat = np.ascontiguousarray(wave).reshape(-1)
    if flat.dtype not in (np.int16, np.uint16):
        flat = flat.astype(np.uint16)  # _clip_wave_to_uint16
    out_off, pool = sess.pool_gather(order * width, rec["event_length"], flat)
    rec["wave_offset"] = out_off
    return RecordsBundle(records=rec, wave_pool=pool)


def _part_is_sorted(records: np.ndarray) -> bool:
    if len(records) < 2:
        return True
    keys = [records[k].astype(np.int64) for k in ("timestamp", "pid", "board", "channel")]
    later = np.zeros(len(records) - 1, dtype=bool)   # strictly greater decided by an earlier key
    equal = np.ones(len(records) - 1, dtype=bool)
    for k in keys:
        d = np.diff(k)
        later |= equal & (d > 0)
        if np.any(equal & ~later & (d < 0)):
            return False
        equal &= d == 0
    return True


def merge_records_parts(parts: Sequence[RecordsBundle], session=None) -> RecordsBundle:
    """Merge sorted parts into one globally sorted bundle (records_builder.py:869-945).

    A k-way heap merge with the tie-break (part index, row index) is the stable sort of the concatenated parts
    by (timestamp, pid, board, channel) when every part is sorted, which is the documented precondition."""
    if not parts:
        return _empty()
    total_records = sum(len(p.records) for p in parts)
    if total_records == 0:
        return _empty()
    for k, p in enumerate(parts):
        if not _part_is_sorted(p.records):
            raise ValueError(f"records part {k} is not sorted by (timestamp, pid, board, channel)")
    live = [p for p in parts if len(p.records)]
    records = np.concatenate([p.records for p in live]).astype(RECORDS_DTYPE, copy=False)
    pool_sizes = np.array([len(p.wave_pool) for p in live], dtype=np.int64)
    pool_base = np.concatenate(([0], np.cumsum(pool_sizes)[:-1]))
    src_offset = records["wave_offset"].astype(np.int64) + np.repeat(pool_base, [len(p.records) for p in live])
    src_pool = np.concatenate([np.asarray(p.wave_pool, dtype=np.uint16) for p in live])
    sess = _session(session)
    order = records_sort_order(records, sess)
    out = records[order]
    lengths = np.maximum(out["event_length"], 0).astype(np.int32)
    # the reference slices part.wave_pool[offset:offset+length]: a slice past the part's pool would raise there
    part_of = np.repeat(np.arange(len(live)), [len(p.records) for p in live])[order]
    end = out["wave_offset"].astype(np.int64) + lengths
    bad = (lengths > 0) & ((out["wave_offset"] < 0) | (end > pool_sizes[part_of]))
    if np.any(bad):
        i = int(np.flatnonzero(bad)[0])
        raise ValueError(f"could not broadcast input array: record {i} wave slice outside its part's wave_pool")
    out_off, pool = sess.pool_gather(src_offset[order], lengths, src_pool)
    out["wave_offset"] = out_off
    record_ids = out["record_id"].astype(np.int64, copy=False)
    if len(np.unique(record_ids)) != len(record_ids):
        out["record_id"] = np.arange(total_records, dtype=np.int64)
    return RecordsBundle(records=out, wave_pool=pool)


def v1725_index(blob: np.ndarray) -> dict:
    """Per-wave header fields and payload positions of a V1725 DAW_DEMO byte stream (utils/formats/v1725.py:66-114)."""
    import ctypes as C

    from . import _lib

    buf = np.ascontiguousarray(blob, dtype=np.uint8)
    lib = _lib.load()
    n = C.c_int64(0)
    ptr = buf.ctypes.data_as(C.c_void_p)
    _lib.check(lib.wfa_v1725_index(ptr, buf.size, 0, None, None, None, None, None, None, C.byref(n)))
    k = int(n.value)
    out = {"channel": np.empty(k, np.int16), "timestamp": np.empty(k, np.int64), "trunc": np.empty(k, np.uint8),
           "baseline": np.empty(k, np.uint16), "payload_offset": np.empty(k, np.int64), "n_samples": np.empty(k, np.int32)}
    _lib.check(lib.wfa_v1725_index(ptr, buf.size, k, *[out[f].ctypes.data_as(C.c_void_p) for f in (
        "channel", "timestamp", "trunc", "baseline", "payload_offset", "n_samples")], C.byref(n)))
    return out


def _board_from_path(path) -> int:
    import os
    import re

    m = re.search(r"_b(\d+)", os.path.basename(str(path)), flags=re.IGNORECASE)
    return int(m.group(1)) if m else 0


def build_records_from_v1725_blob(blob: np.ndarray, board: int, dt_ns: int, session=None) -> RecordsBundle:
    """One file's worth of waves -> sorted records + packed pool (records_builder.py:164-209 after the reader):
    header walk on the host, order and payload packing on the GPU straight from the file bytes."""
    idx = v1725_index(blob)
    n = len(idx["channel"])
    if n == 0:
        return _empty()
    rec = np.zeros(n, dtype=RECORDS_DTYPE)
    rec["timestamp"] = idx["timestamp"] * np.int64(int(dt_ns) * 1000)   # SAMPLE_INDEX mode (formats/base.py:177-185)
    rec["pid"] = 0
    rec["board"] = board
    rec["channel"] = idx["channel"]
    rec["baseline"] = idx["baseline"].astype(np.float64)
    rec["baseline_upstream"] = np.nan
    rec["polarity"] = "unknown"
    rec["dt"] = np.int32(dt_ns)
    rec["trigger_type"] = 0
    rec["flags"] = idx["trunc"].astype(np.uint32)
    rec["event_length"] = idx["n_samples"]
    rec["time"] = rec["timestamp"] // 1000
    sess = _session(session)
    order = records_sort_order(rec, sess)
    rec = rec[order]
    buf = np.ascontiguousarray(blob, dtype=np.uint8)
    if buf.size % 2:
        buf = np.concatenate([buf, np.zeros(1, dtype=np.uint8)])
    out_off, pool = sess.pool_gather(idx["payload_offset"][order] // 2, rec["event_length"], buf.view(np.uint16))
    rec["wave_offset"] = out_off
    rec["record_id"] = np.arange(n, dtype=np.int64)
    return RecordsBundle(records=rec, wave_pool=pool)


def build_records_from_v1725_files(file_paths, dt_ns: int, session=None) -> RecordsBundle:
    """records_builder.py:797-830: one sorted part per file, parts merged."""
    import os

    if not file_paths:
        return _empty()
    sess = _session(session)
    parts = []
    for path in file_paths:
        if not os.path.exists(path):   # the reference's reader logs a warning and goes on
            continue
        part = build_records_from_v1725_blob(np.fromfile(path, dtype=np.uint8), _board_from_path(path), dt_ns, sess)
        if len(part.records):
            parts.append(part)
    if not parts:
        return _empty()
    if len(parts) == 1:
        return parts[0]
    return merge_records_parts(parts, sess)


# ---- CAEN VX2730 CSV (utils/formats/vx2730.py) ------------------------------------------------------------------
VX2730_DELIMITER = ";"
VX2730_SAMPLES_START = 7          # BOARD;CHANNEL;TIMETAG;ENERGY;ENERGYSHORT;FLAGS;PROBE_CODE;samples...
VX2730_BASELINE_COLUMNS = (7, 47)  # VX2730_SPEC: the first 40 samples (vx2730.py:90-93)


def _looks_like_vx2730_header(line: bytes) -> bool:
    fields = [f.strip().upper() for f in line.decode("utf-8", errors="ignore").strip().split(VX2730_DELIMITER)]
    return len(fields) >= 3 and tuple(fields[:3]) == ("BOARD", "CHANNEL", "TIMETAG")


def vx2730_skiprows(data: bytes, is_first_file: bool) -> int:
    """vx2730.py:165-191 `_resolve_skiprows`: a header in the first or second line wins; otherwise the first file of a
    channel carries the legacy two header rows and the others none."""
    first_end = data.find(b"\n")
    first = data if first_end < 0 else data[: first_end + 1]
    if _looks_like_vx2730_header(first):
        return 1
    if first_end >= 0:
        second_end = data.find(b"\n", first_end + 1)
        second = data[first_end + 1 :] if second_end < 0 else data[first_end + 1 : second_end + 1]
        if second and _looks_like_vx2730_header(second):
            return 2
    return 2 if is_first_file else 0


def _strip_rows(data: bytes, n: int) -> bytes:
    pos = 0
    for _ in range(n):
        nxt = data.find(b"\n", pos)
        if nxt < 0:
            return b""
        pos = nxt + 1
    return data[pos:]


def _baseline_window(baseline_samples, samples_start: int, baseline_start: int, baseline_end: int) -> tuple[int, int]:
    """records_builder.py:53-105: None -> the adapter's columns; int n -> the first n samples; (s, e) -> samples[s:e]
    (same validation messages)."""
    if isinstance(baseline_samples, list):
        baseline_samples = tuple(baseline_samples)
    if baseline_samples is None:
        return baseline_start, baseline_end
    if isinstance(baseline_samples, tuple):
        if len(baseline_samples) != 2:
            raise ValueError("baseline_samples tuple must have 2 elements (start, end), "
                             f"got {len(baseline_samples)}")
        s0, s1 = baseline_samples
        if not isinstance(s0, int) or not isinstance(s1, int):
            raise TypeError("baseline_samples tuple elements must be int, "
                            f"got ({type(s0).__name__}, {type(s1).__name__})")
        if s0 < 0 or s1 < 0:
            raise ValueError(f"baseline_samples indices must be non-negative, got ({s0}, {s1})")
        if s0 >= s1:
            raise ValueError(f"baseline_samples start must be less than end, got ({s0}, {s1})")
        return samples_start + s0, samples_start + s1
    if isinstance(baseline_samples, int):
        if baseline_samples <= 0:
            raise ValueError(f"baseline_samples must be positive, got {baseline_samples}")
        return baseline_start, baseline_start + int(baseline_samples)
    raise TypeError("baseline_samples must be int or tuple (start, end), "
                    f"got {type(baseline_samples).__name__}")


def build_records_from_vx2730_files(raw_files, default_dt_ns: int = 1, baseline_samples=None, epoch_ns=None,
                                    session=None) -> RecordsBundle:
    """`build_records_from_raw_files(raw_files, adapter_name="vx2730", ...)` (records_builder.py:524-642, 834-867;
    per file `_build_records_part_from_raw_array` 212-302): raw_files is a list of per-channel file lists.

    The text of all files goes to the GPU in one piece (header rows cut off on the host), is decoded there
    (wfa_csv_decode_*), the records are ordered with the device sort and the decoded samples are packed into the
    final wave_pool without leaving the device; the baselines are means over the packed pool.  The reference sorts
    every file's part and heap-merges the parts with (part, row) as tie-break, which is the stable sort of all rows
    in (channel list, file, row) order -- one sort here."""
    import os

    b0, b1 = _baseline_window(baseline_samples, VX2730_SAMPLES_START, *VX2730_BASELINE_COLUMNS)
    chunks, file_rows_base, file_channel = [], [], []
    total = 0
    for channel_idx, files in enumerate(raw_files or []):
        for k, path in enumerate(files or []):
            if not os.path.exists(path) or os.path.getsize(path) == 0:   # the reader skips both
                continue
            with open(path, "rb") as fh:
                data = fh.read()
            body = _strip_rows(data, vx2730_skiprows(data, is_first_file=(k == 0)))
            if not body:
                continue
            if not body.endswith(b"\n"):
                body += b"\n"
            chunks.append(body)
            file_rows_base.append(total)
            file_channel.append(channel_idx)
            total += len(body)
    if not chunks:
        return _empty()
    if total >= 2**31:
        raise ValueError(f"{total} bytes of CSV text; one device decode call takes < 2^31 bytes -- build the run in "
                         "several calls and merge_records_parts the results")
    sess = _session(session)
    dec = sess.csv_decode(b"".join(chunks), VX2730_DELIMITER, VX2730_SAMPLES_START, (0, 1, 2))
    keep = dec["n_fields"] > 0            # blank lines
    file_of = np.searchsorted(np.asarray(file_rows_base, dtype=np.int64), dec["row_offset"], side="right") - 1
    nf = dec["n_fields"]
    for f in range(len(chunks)):          # a file is one 2-D array in the reference: its rows have one width
        widths = np.unique(nf[(file_of == f) & keep])
        if len(widths) > 1:
            raise ValueError(f"CSV file {f} of channel list {file_channel[f]}: rows with {int(widths[0])} and "
                             f"{int(widths[1])} fields")
    rows = np.flatnonzero(keep)
    n = len(rows)
    if n == 0:
        return _empty()
    lengths = np.maximum(nf[rows] - VX2730_SAMPLES_START, 0).astype(np.int32)
    rec = np.zeros(n, dtype=RECORDS_DTYPE)
    meta = dec["meta"][rows]
    rec["timestamp"] = meta[:, 2]                       # TimestampUnit.PICOSECONDS: already ps
    rec["pid"] = 0
    rec["board"] = meta[:, 0].astype(np.int16)
    rec["channel"] = meta[:, 1].astype(np.int16)
    rec["baseline_upstream"] = np.nan
    rec["polarity"] = "unknown"
    rec["dt"] = np.int32(default_dt_ns)
    rec["trigger_type"] = 0
    rec["flags"] = np.uint32(0)
    rec["event_length"] = lengths
    rec["time"] = rec["timestamp"] // 1000 if epoch_ns is None else np.int64(epoch_ns) + rec["timestamp"] // 1000
    order = records_sort_order(rec, sess)
    rec = rec[order]
    out_off, pool = sess.pool_gather(dec["sample_offset"][rows][order], rec["event_length"], None,
                                     src_samples=dec["n_samples"])
    rec["wave_offset"] = out_off
    rec["record_id"] = np.arange(n, dtype=np.int64)
    sess.upload_records(rec)
    rec["baseline"] = sess.baseline_mean(b0 - VX2730_SAMPLES_START, b1 - VX2730_SAMPLES_START)
    return RecordsBundle(records=rec, wave_pool=pool)


__all__ = ["RecordsBundle", "records_sort_order", "build_records_from_st_waveforms", "merge_records_parts",
           "v1725_index", "build_records_from_v1725_blob", "build_records_from_v1725_files",
           "build_records_from_vx2730_files", "vx2730_skiprows"]
