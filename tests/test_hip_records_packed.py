"""Records as packed rows, unpacked on the device (wfa_upload_records_packed; reference row layout
core/processing/dtypes.py:80-100, input contract data/records_view.py:16-56,383-400) against the column route
(wfa_upload_records_soa) and the oracle: same device tables (seen through every kernel family that reads them), same
errors with the same wording."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession
from waveformanalysis_amd.dtypes import RECORDS_DTYPE

pytestmark = pytest.mark.gpu


def both_routes(sess, rec, thresholds, fn, polarity=None):
    out = []
    for packed in (True, False):
        sess.packed_records = packed
        sess.upload_records(rec, thresholds, polarity=polarity)
        out.append(fn(sess))
    sess.packed_records = True
    return out


def test_packed_rows_give_the_same_tables_as_columns():
    rec, pool = synth.make_run(3000, "v1725", cfg=41)
    rng = np.random.default_rng(3)
    rec["polarity"] = rng.choice(["unknown", "negative", "positive", "", "Positive", "negativ"], size=len(rec))
    thr = rng.choice([8.0, 10.0, 12.5], size=len(rec))
    assert DeviceSession._packed_layout(rec) is not None
    filt = O.filter_wave_pool(rec, pool)
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        for thresholds in (10.0, thr):
            want = O.threshold_hits_chunked(rec, filt, thresholds=None if np.isscalar(thresholds) else thresholds)
            a, b = both_routes(sess, rec, thresholds, lambda s: s.threshold_hits(_lib.SRC_SG_FUSED, 2, 2))
            G.assert_struct_equal(a, b, what="hits: packed rows vs columns")
            G.assert_struct_equal(a, want, float_rtol=1e-6, what="hits: packed rows vs oracle")
        a, b = both_routes(sess, rec, 10.0, lambda s: s.basic_features(_lib.SRC_RAW, (40, 90), (0, None)))
        G.assert_struct_equal(a, b, what="basic features")
        G.assert_struct_equal(a, O.basic_features(rec, pool), what="basic features vs oracle")
        a, b = both_routes(sess, rec, 10.0, lambda s: s.width_integral(_lib.SRC_RAW, 0.1, 0.9, 2.0))
        G.assert_struct_equal(a, b, what="width integral")
        # caller's polarity codes override the text field
        codes = rng.integers(0, 3, size=len(rec)).astype(np.int8)
        a, b = both_routes(sess, rec, 10.0, lambda s: s.threshold_hits(_lib.SRC_RAW, 2, 2), polarity=codes)
        G.assert_struct_equal(a, b, what="polarity override")
        assert sess.max_len == 800 and sess.n_records == len(rec)


def test_uniform_layouts_are_recognised_on_the_device():
    """Span mode / padded shadow eligibility comes out of the unpack kernel: the streaming kernel runs for both presets."""
    for preset, n in (("v1725", 2000), ("vx2730", 1000)):
        rec, pool = synth.make_run(n, preset, cfg=7)
        want = O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool))
        with DeviceSession(0) as sess:
            sess.upload_pool(pool)
            sess.set_sg_plan(11, 2)
            sess.upload_records(rec, 10.0)
            sess.profile(True)
            got = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
            names = sess.profile_report()
            assert "k_sg_runs32" in names and "k_unpack_records + bitmap offsets" not in names, sorted(names)
            G.assert_struct_equal(got, want, float_rtol=1e-6, what=preset)
            # a gap between two records: not uniform any more, the per-record kernels take over, same rows
            ragged = rec.copy()
            keep = np.r_[0:n // 2, n // 2 + 1:n]
            sess.upload_records(np.ascontiguousarray(ragged[keep]), 10.0)
            sess.profile(True)
            got2 = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
            assert "k_sg_runs32" not in sess.profile_report()
            G.assert_struct_equal(got2, want[want["record_id"] != rec["record_id"][n // 2]], float_rtol=1e-6, what=preset + " gap")


def test_sparse_tables_and_record_id_rules():
    rec, pool = synth.make_run(500, "v1725", cfg=9)
    slim = np.zeros(len(rec), dtype=[("record_id", "<i8"), ("baseline", "<f8"), ("wave_offset", "<i8"),
                                     ("timestamp", "<i8"), ("event_length", "<i4")])
    for f in slim.dtype.names:
        slim[f] = rec[f]
    assert DeviceSession._packed_layout(slim) is not None                      # no polarity / dt / board / channel
    want = O.threshold_hits_chunked(rec, pool)
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        a, b = both_routes(sess, slim, 10.0, lambda s: s.threshold_hits(_lib.SRC_RAW, 2, 2))
        G.assert_struct_equal(a, b, what="slim table")
        for f in ("position", "edge_start", "edge_end", "record_id", "height", "integral"):
            np.testing.assert_array_equal(a[f], want[f], err_msg=f)
        assert np.all(a["dt"] == 1) and np.all(a["board"] == 0) and np.all(a["channel"] == 0)
        # ids out of order but unique: fine; a duplicate: the reference's message (records_view.py:383-400)
        shuffled = rec.copy()
        shuffled["record_id"] = shuffled["record_id"][::-1]
        sess.upload_records(shuffled, 10.0)
        dup = rec.copy()
        dup["record_id"][17] = dup["record_id"][400]
        with pytest.raises(ValueError, match=f"record_id must be unique, got duplicate {int(dup['record_id'][17])}"):
            sess.upload_records(dup, 10.0)
        # views the device cannot take as rows fall back to columns: same result
        sess.upload_records(rec[::2], 10.0)
        half = sess.threshold_hits(_lib.SRC_RAW, 2, 2)
        G.assert_struct_equal(half, want[np.isin(want["record_id"], rec["record_id"][::2])], what="strided view")


@pytest.mark.parametrize("field,value,message", [
    ("wave_offset", -8, "negative wave_offset"), ("event_length", -1, "negative event_length"),
    ("wave_offset", 10**9, "outside wave_pool bounds"), ("event_length", 130_433, "outside wave_pool bounds|at most 130432"),
])
def test_validation_messages_match_the_column_route(field, value, message):
    rec, pool = synth.make_run(300, "v1725", cfg=2)
    bad = rec.copy()
    bad[field][123] = value
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        errors = []
        for packed in (True, False):
            sess.packed_records = packed
            with pytest.raises(Exception, match=message) as e:
                sess.upload_records(bad, 10.0)
            errors.append(str(e.value))
        assert errors[0] == errors[1] or "at most 130432" in errors[0], errors
        sess.packed_records = True
        sess.upload_records(rec, 10.0)                                      # the session is usable afterwards
        assert len(sess.threshold_hits(_lib.SRC_RAW, 2, 2)) > 100


def test_empty_table():
    rec, pool = synth.make_run(10, "v1725", cfg=2)
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.upload_records(rec[:0], 10.0)
        assert len(sess.threshold_hits(_lib.SRC_RAW, 2, 2)) == 0 and sess.max_len == 0
