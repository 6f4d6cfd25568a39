"""Chunk container and time-range helpers against verdicts / selections produced by the reference's chunk.py."""

import os

import numpy as np
import pytest

from tests import golden_util as G
from waveformanalysis_amd import chunk as C

KW = dict(time_field="timestamp", length_field="event_length")


def load():
    z = np.load(os.path.join(G.GOLDEN, "chunk_helpers.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_endtime_and_selection():
    case = load()
    rec = case["records"]
    np.testing.assert_array_equal(C.get_endtime(rec, **KW), case["endtime"])
    np.testing.assert_array_equal(C.get_endtime(rec, dt=2.5, **KW), case["endtime_dt"])
    t0, t1 = (int(v) for v in case["sel_bounds"])
    np.testing.assert_array_equal(C.select_time_range(rec, t0, t1, **KW)["record_id"], case["sel_loose"])
    np.testing.assert_array_equal(C.select_time_range(rec, t0, t1, strict=True, **KW)["record_id"], case["sel_strict"])
    np.testing.assert_array_equal(C.select_time_range(rec, None, t1, **KW)["record_id"], case["sel_open"])
    # default field names fall back: time -> timestamp is NOT taken when a `time` field exists (records have both)
    assert C.resolve_time_field(rec, "time") == "time"
    assert C.resolve_length_field(rec, "length") == "event_length"
    with pytest.raises(KeyError, match="Missing required fields"):
        C.compute_endtime(rec[["timestamp", "event_length"]], **KW)
    with pytest.raises(TypeError, match="structured"):
        C.get_endtime(np.zeros(3))


def test_split_by_breaks_and_boundaries():
    case = load()
    rec = case["records"]
    parts = list(C.split_by_breaks(rec, **KW))
    np.testing.assert_array_equal([p[0]["record_id"][0] for p in parts], case["break_first"])
    np.testing.assert_array_equal([(i.start_time, i.end_time, i.n_records, i.chunk_i) for _p, i in parts], case["break_info"])
    parts = list(C.split_by_breaks(rec, break_threshold_ps=5 * 10**6, min_chunk_size=3, **KW))
    np.testing.assert_array_equal([(i.start_time, i.end_time, i.n_records, i.chunk_i) for _p, i in parts], case["break2_info"])
    t0, t1 = (int(v) for v in case["sel_bounds"])
    r = C.check_chunk_boundaries(rec, t0, t1, **KW)
    got = [r.stats["n_records"], r.stats["n_before_start"], r.stats["n_after_end"], r.stats["violations"], int(r.is_valid)]
    np.testing.assert_array_equal(got, case["bounds_stats"])
    assert "\n".join(r.errors) == bytes(case["bounds_errors"]).decode()
    with pytest.raises(ValueError, match="records start before chunk boundary"):
        r.raise_if_invalid()
    assert C.check_chunk_boundaries(rec[:0], 0, 1).stats == {"n_records": 0, "violations": 0}


def test_chunk_constructor_verdicts_and_split():
    case = load()
    rec = case["records"]
    end = int(case["endtime"].max())
    want = bytes(case["chunk_verdicts"]).decode().split("\n")
    got = []
    for start, stop in ((int(rec["timestamp"].min()), end), (int(rec["timestamp"].min()) + 1, 10**18), (0, end - 1)):
        try:
            C.Chunk(rec, start, stop, **KW)
            got.append("ok")
        except ValueError as exc:
            got.append(str(exc))
    assert got == want
    ch = C.Chunk(rec, 0, end, run_id="r", data_type="records", **KW)
    left, right = ch.split(int(rec["timestamp"][200]))
    assert len(left) + len(right) == len(ch) and left.end == right.start and repr(ch).startswith("Chunk(r.records:")
    assert ch.duration == end and ch.nbytes == rec.nbytes
    info = C.ChunkInfo(0, 10, 3)
    assert info.contains(9) and not info.contains(10) and info.overlaps(C.ChunkInfo(9, 20, 1)) and info.duration == 10
