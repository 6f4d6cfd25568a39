"""Records builder (global order + wave_pool packing): oracle restatement and the GPU implementation against a
fixture produced by the reference's build_records_from_st_waveforms / merge_records_parts."""

import os

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import synth
from waveformanalysis_amd.dtypes import RECORDS_DTYPE, create_record_dtype


def load():
    z = np.load(os.path.join(G.GOLDEN, "sort_mixed.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_oracle_matches_reference():
    case = load()
    rec, pool = O.build_records_from_st_waveforms(case["st_waveforms"], default_dt_ns=4)
    G.assert_struct_equal(rec, case["records"])
    np.testing.assert_array_equal(pool, case["wave_pool"])
    parts = [(case[f"part{p}_records"], case[f"part{p}_pool"]) for p in range(3)]
    rec, pool = O.merge_records_parts(parts)
    G.assert_struct_equal(rec, case["merged_records"])
    np.testing.assert_array_equal(pool, case["merged_pool"])
    np.testing.assert_array_equal(O.records_sort_order(case["merged_records"]), np.arange(len(rec)))


@pytest.mark.gpu
def test_gpu_matches_reference():
    from waveformanalysis_amd.records_builder import RecordsBundle, build_records_from_st_waveforms, merge_records_parts

    case = load()
    b = build_records_from_st_waveforms(case["st_waveforms"], default_dt_ns=4)
    G.assert_struct_equal(b.records, case["records"])
    np.testing.assert_array_equal(b.wave_pool, case["wave_pool"])
    parts = [RecordsBundle(case[f"part{p}_records"], case[f"part{p}_pool"]) for p in range(3)]
    m = merge_records_parts(parts)
    G.assert_struct_equal(m.records, case["merged_records"])
    np.testing.assert_array_equal(m.wave_pool, case["merged_pool"])
    assert len(merge_records_parts([]).records) == 0
    unsorted = RecordsBundle(case["part0_records"][::-1].copy(), case["part0_pool"])
    with pytest.raises(ValueError, match="not sorted"):
        merge_records_parts([unsorted, parts[1]])


@pytest.mark.gpu
def test_gpu_against_oracle_large_and_resident_pool():
    """20 000 records in 7 parts with ragged lengths and unaligned offsets; the packed pool stays resident and
    feeds the hit pass without another upload."""
    from waveformanalysis_amd import _lib
    from waveformanalysis_amd.device import DeviceSession
    from waveformanalysis_amd.records_builder import RecordsBundle, merge_records_parts, records_sort_order

    rng = np.random.default_rng(12)
    rec, pool = synth.make_run(20000, "v1725", cfg=23)
    rec["timestamp"] = (rec["timestamp"] // 10**6) * 10**6
    rec["pid"] = rng.integers(-2, 3, len(rec))
    rec["event_length"] = rng.choice([800, 800, 797, 13, 0], len(rec))
    which = rng.integers(0, 7, len(rec))
    parts, oparts = [], []
    for p in range(7):
        r = rec[which == p].copy()
        r = r[O.records_sort_order(r)]
        w = np.concatenate([pool[o : o + n] for o, n in zip(r["wave_offset"], r["event_length"])] + [np.zeros(3, np.uint16)])
        r["wave_offset"] = np.concatenate(([0], np.cumsum(r["event_length"][:-1])))
        parts.append(RecordsBundle(r, w))
        oparts.append((r, w))
    sess = DeviceSession(0)
    try:
        got = merge_records_parts(parts, session=sess)
        want_rec, want_pool = O.merge_records_parts(oparts)
        G.assert_struct_equal(got.records, want_rec)
        np.testing.assert_array_equal(got.wave_pool, want_pool)
        np.testing.assert_array_equal(records_sort_order(rec, sess), O.records_sort_order(rec))
        # resident: no upload_pool between the merge and the hit pass
        sess.upload_records(got.records, 10.0)
        hits = sess.threshold_hits(_lib.SRC_RAW)
        G.assert_struct_equal(hits, O.threshold_hits_chunked(want_rec, want_pool), float_rtol=1e-6)
    finally:
        sess.close()


@pytest.mark.gpu
def test_gpu_dense_rows_to_records_roundtrip():
    from waveformanalysis_amd.records_builder import build_records_from_st_waveforms

    rec, pool = synth.make_run(3000, "vx2730", cfg=24)
    rng = np.random.default_rng(2)
    perm = rng.permutation(len(rec))
    st = np.zeros(len(rec), dtype=create_record_dtype(1500))
    for f in ("timestamp", "board", "channel", "baseline", "dt", "polarity", "event_length"):
        st[f] = rec[f][perm]
    st["record_id"] = -1                       # -> renumbered
    st["wave"] = pool.reshape(-1, 1500)[perm].astype(np.int16)
    b = build_records_from_st_waveforms(st)
    want_rec, want_pool = O.build_records_from_st_waveforms(st)
    G.assert_struct_equal(b.records, want_rec)
    np.testing.assert_array_equal(b.wave_pool, want_pool)
    assert b.records.dtype == RECORDS_DTYPE
