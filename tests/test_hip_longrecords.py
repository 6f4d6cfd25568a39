"""Records far longer than a digitiser window (up to WFA_MAX_RECORD_SAMPLES = 130 432): every per-record kernel family
against the oracle on a ragged run whose records are 100 000, 70 001, 40 000, 33 000 and a few hundred samples long
(the previous build refused anything above 32 760)."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib
from waveformanalysis_amd.device import DeviceSession
from waveformanalysis_amd.dtypes import RECORDS_DTYPE

pytestmark = pytest.mark.gpu

LENGTHS = [100_000, 70_001, 300, 40_000, 33_000, 800, 130_432]


def long_run(seed=11):
    rng = np.random.default_rng(seed)
    rec = np.zeros(len(LENGTHS), dtype=RECORDS_DTYPE)
    waves, off = [], 0
    for i, L in enumerate(LENGTHS):
        ped = int(rng.integers(7800, 8200))
        w = ped + np.rint(rng.normal(0, 3, L))
        for _ in range(max(1, L // 4000)):                       # a pulse every few thousand samples
            t0 = int(rng.integers(60, L - 120))
            amp = 10 ** rng.uniform(1.3, 3.3)
            t = np.arange(min(L - t0, 400))
            w[t0:t0 + len(t)] -= amp * (np.exp(-t / rng.uniform(10, 60)) - np.exp(-t / 4.0))
        w = np.clip(w, 0, 16383).astype(np.uint16)
        waves.append(w)
        rec[i]["wave_offset"], rec[i]["event_length"] = off, L
        rec[i]["baseline"] = w[:40].astype(np.float64).mean()
        rec[i]["timestamp"] = 10**12 + i * 10**9
        rec[i]["dt"], rec[i]["board"], rec[i]["channel"], rec[i]["record_id"] = 4, 0, i % 4, i
        rec[i]["polarity"] = "unknown"
        off += L + (-L) % 8                                       # 16-byte aligned starts, like the records builder
        waves.append(np.zeros((-L) % 8, dtype=np.uint16))
    return rec, np.concatenate(waves)


def test_long_records_all_kernel_families():
    rec, pool = long_run()
    filt = O.filter_wave_pool(rec, pool)
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.upload_records(rec, 10.0)
        G.assert_struct_equal(sess.threshold_hits(_lib.SRC_RAW, 2, 2), O.threshold_hits(rec, pool), float_rtol=1e-6, what="raw hits")
        sess.set_sg_plan(11, 2)
        want_f = O.threshold_hits(rec, filt)
        G.assert_struct_equal(sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2), want_f, float_rtol=1e-6, what="fused hits")
        rec_nan = rec.copy()
        rec_nan["baseline"] = np.nan
        sess.upload_records(rec_nan, 10.0)
        # the whole record as baseline window: 130 432 x 65 535 needs more than 31 bits
        got_bl = sess.baseline_mean(0, 200_000)
        want_bl = np.array([pool[o:o + n].astype(np.float64).mean() for o, n in zip(rec["wave_offset"], rec["event_length"])])
        np.testing.assert_array_equal(got_bl, want_bl)
        rec_bl = rec.copy()
        rec_bl["baseline"] = want_bl
        fused = sess.fused_baseline_filter_hits((0, 200_000), 2, 2)
        G.assert_struct_equal(fused, O.threshold_hits(rec_bl, O.filter_wave_pool(rec_bl, pool)), float_rtol=1e-6, what="fused baseline")
        sess.upload_records(rec, 10.0)
        np.testing.assert_array_equal(sess.savgol(), filt)
        G.assert_struct_equal(sess.basic_features(_lib.SRC_RAW, (40, 90), (0, None)), O.basic_features(rec, pool), what="basic features")
        G.assert_struct_equal(sess.width_integral(_lib.SRC_RAW, 0.1, 0.9, 2.0), O.width_integral(rec, pool, dt=2.0), what="width integral")
        sess.upload_filtered_pool(filt)
        cfg = dict(height=8.0, prominence=0.5, width=2)
        G.assert_struct_equal(sess.find_peaks(_lib.SRC_F32, **cfg), O.find_peak_hits(rec, filt, **cfg), what="find_peaks")

        too_long = rec[:1].copy()
        too_long["event_length"] = 130_433
        with pytest.raises(Exception, match="at most 130432"):
            sess.upload_records(too_long, 10.0)
