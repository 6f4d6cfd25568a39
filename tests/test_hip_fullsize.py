"""The fused baseline + filter + hit pass at BASELINE.json's full single-GPU size (1.25e6 records x 800 samples =
1e9 samples) through properties that do not need the oracle to finish a 10^9-sample run:

* order: rows come in (record, start) order, windows of one record do not overlap, every field is in range;
* idempotence: a second pass over the resident chunk gives byte-identical rows (the speculative row launch of the
  second pass and the exact launch of the first take different code paths);
* restriction: records are independent, so the pass over the first k records alone equals the rows of the full pass
  with record_id < k -- at k = 200 000 (span boundaries inside) and against the oracle at k = 20 000;
* a checksum of per-record checksums: the materialised filter + plain threshold pass over the same chunk gives the
  same rows as the fused pass (two different kernels families agreeing on 2.1e6 rows).
"""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession

pytestmark = pytest.mark.gpu

N_RECORDS = 1_250_000


@pytest.fixture(scope="module")
def chunk():
    rec, pool = synth.make_run(N_RECORDS, "v1725", cfg=200)
    rec_in = rec.copy()
    rec_in["baseline"] = np.nan  # the pass estimates it from the first 40 samples
    return rec, rec_in, pool


def _row_digest(rows):
    """Order-sensitive 64-bit digest per row block, summed per record -> one number per record."""
    raw = rows.view(np.uint8).reshape(len(rows), rows.dtype.itemsize).astype(np.uint64)
    weights = (np.arange(rows.dtype.itemsize, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    per_row = (raw * weights).sum(axis=1, dtype=np.uint64)
    out = np.zeros(N_RECORDS, dtype=np.uint64)
    np.add.at(out, rows["record_id"], per_row)
    return out


def test_full_chunk_properties(chunk):
    rec, rec_in, pool = chunk
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.upload_records(rec_in, 10.0)
        sess.set_sg_plan(11, 2)
        rows = sess.fused_baseline_filter_hits((0, synth.BASELINE_SAMPLES), 2, 2)
        again = sess.fused_baseline_filter_hits((0, synth.BASELINE_SAMPLES), 2, 2)
        assert rows.tobytes() == again.tobytes()                      # idempotent, both launch paths
        assert len(rows) > N_RECORDS                                   # ~1.7 hits per record on this generator

        rid, pos = rows["record_id"], rows["position"]
        assert np.all(np.diff(rid) >= 0)                               # record order
        same = np.diff(rid) == 0
        assert np.all(rows["edge_start"][1:][same] >= rows["edge_end"][:-1][same] - 4)  # windows: runs are disjoint, extensions (2 + 2) may touch
        assert np.all(np.diff(rows["edge_start"])[same] > 0)           # start order inside a record
        assert np.all((rows["edge_start"] >= 0) & (rows["edge_end"] <= 800) & (rows["edge_start"] < rows["edge_end"]))
        assert np.all((pos >= rows["edge_start"]) & (pos < rows["edge_end"]))
        assert np.all(rows["width"] == (rows["edge_end"] - rows["edge_start"]).astype(np.float32))
        assert np.all(rows["height"] >= 10.0) and np.all(rows["integral"] >= rows["height"] * np.float32(0.999999))
        ts_rec = rec["timestamp"][rid]
        np.testing.assert_array_equal(rows["timestamp"], (ts_rec.astype(np.float64) + pos * 4000.0).astype(np.int64))
        np.testing.assert_array_equal(rows["channel"], rec["channel"][rid])

        # restriction to a prefix of the records: another pass, other span / grid shapes
        k = 200_000
        sess.upload_records(rec_in[:k], 10.0)
        part = sess.fused_baseline_filter_hits((0, synth.BASELINE_SAMPLES), 2, 2)
        assert part.tobytes() == rows[rid < k].tobytes()

        # the same rows from the two-kernel route: materialised float32 filter, then the plain threshold pass
        sess.upload_records(rec, 10.0)                                 # baselines as the builder computed them
        sess.savgol(download=False)
        two_step = sess.threshold_hits(_lib.SRC_F32, 2, 2)
        assert len(two_step) == len(rows)
        for f in ("position", "edge_start", "edge_end", "timestamp", "record_id"):
            np.testing.assert_array_equal(two_step[f], rows[f], err_msg=f)
        np.testing.assert_allclose(two_step["height"], rows["height"], rtol=1e-6)
        np.testing.assert_allclose(two_step["integral"], rows["integral"], rtol=1e-6)
        np.testing.assert_array_equal(_row_digest(two_step[["position", "edge_start", "edge_end", "timestamp", "record_id"]].copy()),
                                      _row_digest(rows[["position", "edge_start", "edge_end", "timestamp", "record_id"]].copy()))

    # the oracle on a prefix it finishes in seconds
    k = 20_000
    want = O.threshold_hits_chunked(rec[:k], O.filter_wave_pool_uniform(pool[: k * 800], 800))
    G.assert_struct_equal(rows[rid < k], want, float_rtol=1e-6, what="prefix vs oracle")


def test_full_chunk_features(chunk):
    """Basic features and width/integral rows over all 1e9 samples (the records branch of the C3 plugins).

    * two kernel families: the lane-per-leaf kernels on the uniform layout and the general lane-per-record kernels
      (`no_span`) walk the same additions in numpy's order by different routes -- byte-identical rows;
    * restriction: the rows of a prefix uploaded alone (other group and grid shapes, a ragged last group) are the
      first rows of the full result;
    * the oracle on a prefix it finishes in seconds, bit for bit;
    * every quantile position lies inside its record and low <= high.
    """
    rec, _rec_in, pool = chunk
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.upload_records(rec, 10.0)
        sess.profile(True)
        bf = sess.basic_features(_lib.SRC_RAW, (40, 90), (0, None))
        wi = sess.width_integral(_lib.SRC_RAW, 0.1, 0.9, 2.0)
        names = set(sess.profile_report())
        assert {"k_basic_features_leaf", "k_width_integral_leaf"} <= names, names   # the fast kernels ran
        np.testing.assert_array_equal(bf["event_index"], np.arange(N_RECORDS))
        np.testing.assert_array_equal(bf["timestamp"], rec["timestamp"])
        np.testing.assert_array_equal(wi["timestamp"], rec["timestamp"])
        lo, hi = wi["t_low_samples"], wi["t_high_samples"]
        assert np.all((lo >= 0) & (hi <= 800) & (lo <= hi))
        assert np.all(bf["max_abs_diff"] >= 0) and np.all(np.isfinite(bf["area"]))

        sess.profile(True)                                            # one read of the pool for both tables
        bf_both, wi_both = sess.features_both((40, 90), (0, None), 0.1, 0.9, 2.0)
        assert "k_features_both_leaf" in sess.profile_report()
        assert bf_both.tobytes() == bf.tobytes() and wi_both.tobytes() == wi.tobytes()

        sess.set_option("no_span", True)                              # the general kernels
        bf_general = sess.basic_features(_lib.SRC_RAW, (40, 90), (0, None))
        wi_general = sess.width_integral(_lib.SRC_RAW, 0.1, 0.9, 2.0)
        sess.set_option("no_span", False)
        assert bf_general.tobytes() == bf.tobytes()
        assert wi_general.tobytes() == wi.tobytes()

        k = 200_003                                                    # not a multiple of the 8-record group
        sess.upload_records(rec[:k], 10.0)
        assert sess.basic_features(_lib.SRC_RAW, (40, 90), (0, None)).tobytes() == bf[:k].tobytes()
        assert sess.width_integral(_lib.SRC_RAW, 0.1, 0.9, 2.0).tobytes() == wi[:k].tobytes()

    k = 20_000
    G.assert_struct_equal(bf[:k], O.basic_features(rec[:k], pool[: k * 800]), what="basic features prefix vs oracle (bit-exact)")
    G.assert_struct_equal(wi[:k], O.width_integral(rec[:k], pool[: k * 800], dt=2.0), what="width integral prefix vs oracle (bit-exact)")


def test_full_chunk_find_peaks(chunk):
    """The find_peaks hit detector over the filtered 1e9-sample pool (reference defaults: derivative, height 30).

    * three candidate routes -- the height prefilter (default), the plateau machine over every sample (`no_peak_hot`) and the
      lane-per-record walk of arbitrary layouts (`no_span`) -- give byte-identical rows;
    * order and range: rows in (record, position) order, positions inside the record, edge_start <= position <= edge_end;
    * restriction: a prefix uploaded alone gives the first rows of the full result; the oracle on a prefix, bit for bit.
    """
    rec, _rec_in, pool = chunk
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        sess.upload_records(rec, 10.0)
        sess.savgol(download=False)
        sess.profile(True)
        rows = sess.find_peaks(_lib.SRC_F32)
        assert "k_find_peaks_hot" in sess.profile_report()
        assert len(rows) > 400_000
        rid = rows["record_id"]
        assert np.all(np.diff(rid) >= 0)
        assert np.all(np.diff(rows["position"])[np.diff(rid) == 0] > 0)
        assert np.all((rows["position"] >= 0) & (rows["position"] < 799))
        assert np.all((rows["edge_start"] <= rows["position"]) & (rows["position"] <= rows["edge_end"]))
        for opt in ("no_peak_hot", "no_span"):
            sess.set_option(opt, True)
            assert sess.find_peaks(_lib.SRC_F32).tobytes() == rows.tobytes(), opt
            sess.set_option(opt, False)
        k = 200_003
        sess.upload_records(rec[:k], 10.0)
        assert sess.find_peaks(_lib.SRC_F32).tobytes() == rows[rid < k].tobytes()
    k = 20_000
    filt = O.filter_wave_pool_uniform(pool[: k * 800], 800)
    G.assert_struct_equal(rows[rid < k], O.find_peak_hits(rec[:k], filt), what="find_peaks prefix vs oracle (bit-exact)")
