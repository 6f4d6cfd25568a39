"""The lane-per-leaf feature kernels (wfa_features.hip) over the shapes their plan depends on: record lengths whose numpy
pairwise tree is a single short leaf (24), one leaf (64, 104), unbalanced (520: 5 leaves, 1000: 8 unequal leaves),
balanced (800, 2048), many leaves per record with few records per wave (4000, 8192), a last group that is not full,
records of different polarity codes inside one wave, per-record fixed baselines, and area / height ranges that move the
reduction off the record start.  Every case: bit-exact against the oracle (numpy's own summation order) and
byte-identical to the general lane-per-record kernels (`no_span`)."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    with DeviceSession(0) as s:
        yield s


def _run(L, n, seed, mixed_polarity=False):
    rec, pool = synth.make_run(n, "v1725", cfg=seed, L=L)
    if mixed_polarity:
        rng = np.random.default_rng(seed)
        codes = np.array(["unknown", "negative", "positive"])
        rec["polarity"] = codes[rng.integers(0, 3, n)]
    return rec, pool


@pytest.mark.parametrize("L,n", [(24, 1000), (64, 700), (104, 333), (520, 257), (800, 1003), (1000, 130), (2048, 67),
                                 (4000, 35), (8192, 9)])
def test_leaf_kernels_over_record_lengths(sess, L, n):
    rec, pool = _run(L, n, 40 + L % 7)
    sess.upload_pool(pool)
    sess.upload_records(rec, 10.0)
    hr = (min(40, L // 3), min(90, L - 2))
    for ar in ((0, None), (8, L - 8), (16, 16 + max(8, L // 2))):
        sess.profile(True)
        got = sess.basic_features(_lib.SRC_RAW, hr, ar)
        assert "k_basic_features_leaf" in set(sess.profile_report()), (L, ar)
        G.assert_struct_equal(got, O.basic_features(rec, pool, height_range=hr, area_range=ar), what=f"L {L} area {ar}")
        sess.set_option("no_span", True)
        assert sess.basic_features(_lib.SRC_RAW, hr, ar).tobytes() == got.tobytes()
        sess.set_option("no_span", False)
    for q in ((0.1, 0.9), (0.05, 0.95), (0.3, 0.6)):
        sess.profile(True)
        got = sess.width_integral(_lib.SRC_RAW, q[0], q[1], 2.0)
        assert "k_width_integral_leaf" in set(sess.profile_report()), (L, q)
        G.assert_struct_equal(got, O.width_integral(rec, pool, q_low=q[0], q_high=q[1], dt=2.0), what=f"L {L} q {q}")
        sess.set_option("no_span", True)
        assert sess.width_integral(_lib.SRC_RAW, q[0], q[1], 2.0).tobytes() == got.tobytes()
        sess.set_option("no_span", False)
    sess.profile(False)


def test_leaf_kernels_mixed_polarity_and_fixed_baselines(sess):
    rec, pool = _run(800, 2001, 77, mixed_polarity=True)
    rng = np.random.default_rng(3)
    fixed = np.where(rng.random(len(rec)) < 0.4, rng.uniform(7900, 8100, len(rec)), np.nan)
    flat = pool.reshape(-1, 800)
    flat[5] = 8000                                   # constant records: every term zero (sign of zero, q_total = 0)
    rec["baseline"][5] = 8000.0
    flat[6, :] = 8000
    flat[6, 400] = 8500                              # a single sample above the baseline
    rec["baseline"][6] = 8000.0
    sess.upload_pool(pool)
    sess.upload_records(rec, 10.0)
    got = sess.basic_features(_lib.SRC_RAW, (40, 90), (0, None), fixed)
    G.assert_struct_equal(got, O.basic_features(rec, pool, fixed_baseline=fixed), what="mixed polarity, fixed baselines")
    got = sess.width_integral(_lib.SRC_RAW, 0.1, 0.9, 2.0)
    G.assert_struct_equal(got, O.width_integral(rec, pool, dt=2.0), what="mixed polarity width")


def test_unaligned_area_start_takes_the_general_kernel(sess):
    rec, pool = _run(800, 300, 12)
    sess.upload_pool(pool)
    sess.upload_records(rec, 10.0)
    sess.profile(True)
    got = sess.basic_features(_lib.SRC_RAW, (40, 90), (3, 500))
    assert "k_basic_features_leaf" not in set(sess.profile_report())
    sess.profile(False)
    G.assert_struct_equal(got, O.basic_features(rec, pool, area_range=(3, 500)), what="area start 3")
