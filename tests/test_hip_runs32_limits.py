"""Where the streaming kernel k_sg_runs32 must hand over to the general (bitmap) route, and that it does so without
losing or corrupting a hit (reference: hit_finder.py:329-413, the runs of `sig >= thr` per record):

* its run events are (record in span << 16) | sample: uniform records whose stride does not fit 16 bits are not
  taken at all (the build before refused nothing above L = 64 and silently mis-sorted events from L = 65 536 on);
* a span's events go to its 2048-event slot of the event buffer: a span with more raises flag 1, the pass is redone on
  the bitmap route for this upload, and the next upload tries the streaming kernel again.
"""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession
from waveformanalysis_amd.dtypes import RECORDS_DTYPE

pytestmark = pytest.mark.gpu


def uniform_long_run(n_rec, L, seed):
    rng = np.random.default_rng(seed)
    ped = rng.integers(7800, 8200, size=n_rec)
    w = ped[:, None] + np.rint(rng.normal(0, 3, (n_rec, L)))
    t = np.arange(400)
    for r in range(n_rec):
        starts = rng.integers(60, L - 520, size=max(1, L // 6000))
        for t0 in starts:
            amp = 10 ** rng.uniform(1.3, 3.3)
            w[r, t0:t0 + 400] -= amp * (np.exp(-t / rng.uniform(10, 60)) - np.exp(-t / 4.0))
        if r % 3 == 0:                                    # pulses on both record edges: head / tail events
            w[r, :30] -= 200.0 * np.exp(-np.arange(30) / 12.0)
            w[r, L - 25:] -= 150.0
    pool = np.clip(w, 0, 16383).astype(np.uint16)
    rec = np.zeros(n_rec, dtype=RECORDS_DTYPE)
    rec["wave_offset"] = np.arange(n_rec, dtype=np.int64) * L
    rec["event_length"] = L
    rec["baseline"] = pool[:, :40].astype(np.float64).mean(axis=1)
    rec["timestamp"] = 10**12 + np.arange(n_rec, dtype=np.int64) * 10**9
    rec["dt"], rec["board"], rec["channel"] = 4, 0, np.arange(n_rec) % 16
    rec["record_id"] = np.arange(n_rec)
    rec["polarity"] = "unknown"
    return rec, pool.reshape(-1)


@pytest.mark.parametrize("L", [65_504, 65_536, 70_016])
@pytest.mark.parametrize("fused_baseline", [False, True])
def test_uniform_records_around_the_16_bit_event_limit(L, fused_baseline):
    """128 back-to-back records of L samples (L % 32 == 0: the layout the streaming kernel takes).  65 504 is its
    largest stride; the two longer ones must come out of the bitmap route -- same rows as the oracle either way, and
    the same rows as with the streaming kernel switched off."""
    n_rec = 128 if L <= 65_536 else 96
    rec, pool = uniform_long_run(n_rec, L, seed=L)
    want = O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool))
    assert len(want) > 500
    assert np.any(want["edge_end"] == L) and np.any(want["edge_start"] == 0)
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        up = rec.copy()
        if fused_baseline:
            up["baseline"] = np.nan
        sess.upload_records(up, 10.0)
        sess.profile(True)
        run = (lambda: sess.fused_baseline_filter_hits((0, 40), 2, 2)) if fused_baseline else \
            (lambda: sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2))
        got = run()
        names = sess.profile_report()
        streamed = any(k.startswith("k_sg_runs32") for k in names)
        assert streamed == (L <= 65_504), sorted(names)
        G.assert_struct_equal(got, want, float_rtol=1e-6, what=f"uniform L={L}")
        sess.set_option("no_runs32", True)
        sess.upload_records(up, 10.0)
        G.assert_struct_equal(run(), got, what=f"uniform L={L}: bitmap route == default route")


def test_event_overflow_of_a_span_falls_back_and_recovers():
    """thr = 1.0 on sigma = 3 noise: ~4000 runs per 64-record span, four times what a span's event slot holds."""
    rec, pool = synth.make_run(640, "v1725", cfg=5)
    filt = O.filter_wave_pool(rec, pool)
    want_low = O.threshold_hits_chunked(rec, filt, threshold=1.0)
    per_span = np.bincount(np.searchsorted(rec["record_id"], want_low["record_id"]) // 64, minlength=10)
    assert per_span.min() > 1024, per_span   # > 1024 hits = more than the 2048 events of a slot, in every span
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        sess.upload_records(rec, 1.0)
        sess.profile(True)
        got = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
        first = sess.profile_report()
        # the streaming kernel ran, raised the overflow flag, and the bitmap route produced the rows
        assert first.get("k_sg_runs32", (0, 0))[1] == 1, sorted(first)
        assert any(k.startswith("k_sg_mask") for k in first) and "k_hit_runs" in first, sorted(first)
        G.assert_struct_equal(got, want_low, float_rtol=1e-6, what="overflowing spans")
        # same upload, second pass: straight to the bitmap route (the flag is remembered per upload)
        sess.profile(True)
        G.assert_struct_equal(sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2), got, what="second pass after overflow")
        assert "k_sg_runs32" not in sess.profile_report()
        # fused baseline variant overflows the same way
        blank = rec.copy()
        blank["baseline"] = np.nan
        sess.upload_records(blank, 1.0)
        sess.profile(True)
        got_bl = sess.fused_baseline_filter_hits((0, 40), 2, 2)
        assert sess.profile_report().get("k_sg_runs32<baseline>", (0, 0))[1] == 1
        G.assert_struct_equal(got_bl, want_low, float_rtol=1e-6, what="overflowing spans, fused baseline")
        # the next upload takes the streaming kernel again
        sess.upload_records(rec, 10.0)
        sess.profile(True)
        got_hi = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
        again = sess.profile_report()
        assert "k_sg_runs32" in again and not any(k.startswith("k_sg_mask") for k in again), sorted(again)
        G.assert_struct_equal(got_hi, O.threshold_hits_chunked(rec, filt), float_rtol=1e-6, what="after recovery")


def test_one_busy_span_among_quiet_ones():
    """Only span 3 overflows (its records carry a square wave around the threshold): the whole upload is redone on the
    bitmap route and still equals the oracle."""
    rec, pool = synth.make_run(512, "v1725", cfg=6)
    pool = pool.copy()
    w = pool.reshape(512, 800)
    sq = (np.arange(800) // 6 % 2).astype(np.int64) * 40
    w[192:256] = np.clip(w[192:256].astype(np.int64) - sq[None, :], 0, 16383).astype(np.uint16)
    rec["baseline"] = w[:, :40].astype(np.float64).mean(axis=1)
    want = O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool))
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        sess.upload_records(rec, 10.0)
        sess.profile(True)
        got = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
        names = sess.profile_report()
        assert "k_sg_runs32" in names and "k_hit_runs" in names, sorted(names)
        G.assert_struct_equal(got, want, float_rtol=1e-6, what="one overflowing span")


@pytest.mark.parametrize("preset,n", [("v1725", 6000), ("vx2730", 3000)])
def test_flat_rows_kernel_equals_grouped_rows_kernel(preset, n):
    """The chunk-per-lane row kernel (default) against the 8-lanes-per-hit kernel of rounds 1-2 (`rows_grouped`): same
    extremum (float32 order, first index), same rows; the float64 window sums are added in another order (tolerance of
    the float fields 1e-6, north_star), everything else is bit-identical.  Both against the oracle."""
    rec, pool = synth.make_run(n, preset, cfg=31)
    want = O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool))
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        sess.upload_records(rec, 10.0)
        sess.profile(True)
        flat = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
        assert "k_hit_rows_flat" in sess.profile_report()
        sess.set_option("rows_grouped", True)
        sess.profile(True)
        grouped = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
        assert "k_hit_rows_grp" in sess.profile_report()
        G.assert_struct_equal(flat, want, float_rtol=1e-6, what="flat rows vs oracle")
        G.assert_struct_equal(grouped, want, float_rtol=1e-6, what="grouped rows vs oracle")
        for name in flat.dtype.names:
            if name != "integral":
                np.testing.assert_array_equal(flat[name], grouped[name], err_msg=name)
        # wide extensions: windows that reach both record edges and the zero padding
        sess.set_option("rows_grouped", False)
        wide = sess.threshold_hits(_lib.SRC_SG_FUSED, 900, 900)
        G.assert_struct_equal(wide, O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool), left_extension=900,
                                                             right_extension=900), float_rtol=1e-6, what="wide windows")


@pytest.mark.parametrize("preset,n,L", [("v1725", 3000, None), ("vx2730", 1500, None), ("v1725", 900, 1024), ("v1725", 700, 1184)])
@pytest.mark.parametrize("fused_baseline", [False, True])
def test_edge_samples_from_lds_equal_edge_samples_from_memory(preset, n, L, fused_baseline):
    """The flush evaluates the 2H edge samples of every record with the polynomial-fit rows.  Their inputs -- the first 12
    samples (kept by the span prologue) and the 12 samples around L - W (left by the lane of the tile loop that holds the
    record's end) -- come out of LDS by default; `no_deposit` reads the tail from memory again, and so do layouts whose
    span image leaves no room (L = 1024) or whose padding pushes those samples out of the last lane.  Same rows."""
    rec, pool = synth.make_run(n, preset, cfg=41) if L is None else synth.make_run(n, preset, cfg=41, L=L)
    rng = np.random.default_rng(n)
    w = pool.reshape(len(rec), -1).copy()
    for r in rng.choice(len(rec), size=len(rec) // 3, replace=False):   # pulses that sit on a record's first / last samples
        if r % 2:
            w[r, -int(rng.integers(3, 14)):] -= np.uint16(rng.integers(60, 900))
        else:
            w[r, :int(rng.integers(1, 9))] -= np.uint16(rng.integers(60, 900))
    pool = w.reshape(-1)
    rec["baseline"] = w[:, :40].astype(np.float64).mean(axis=1)
    want = O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool))
    assert np.any(want["edge_start"] == 0) and np.any(want["edge_end"] == w.shape[1])
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        up = rec.copy()
        if fused_baseline:
            up["baseline"] = np.nan
        run = (lambda: sess.fused_baseline_filter_hits((0, 40), 2, 2)) if fused_baseline else \
            (lambda: sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2))
        sess.upload_records(up, 10.0)
        sess.profile(True)
        got = run()
        assert any(k.startswith("k_sg_runs32") for k in sess.profile_report())
        G.assert_struct_equal(got, want, float_rtol=1e-6, what=f"{preset} L={L}")
        sess.set_option("no_deposit", True)
        sess.upload_records(up, 10.0)
        assert run().tobytes() == got.tobytes()
        sess.set_option("no_deposit", False)
        sess.set_option("span_records", 51)     # spans of fewer records than a wave has lanes (measurement option)
        sess.upload_records(up, 10.0)
        assert run().tobytes() == got.tobytes()


def test_queued_passes_overflow_regrow_and_control_words():
    """Queued passes (wfa_hits_enqueue) leave no host round trip between their kernels: the last kernel of a pass writes the
    row count and the control words to pinned memory and clears control words and group sums for the next pass.  Three
    ways such a pass has to be redone or followed up: a span overflows its event slot (general route, then recovery), the
    speculative row launch was sized for fewer rows than the pass finds (exact route), and several queued passes in a row
    with a plain (waited) pass in between -- rows equal the oracle's every time."""
    rec, pool = synth.make_run(20_000, "v1725", cfg=5)
    filt = O.filter_wave_pool(rec, pool)
    want = {thr: O.threshold_hits_chunked(rec, filt, threshold=thr) for thr in (6.0, 10.0, 40.0)}
    assert len(want[40.0]) * 1.125 + 4096 < len(want[6.0])          # the speculative launch after the 40.0 pass is too small
    small = rec[:640]
    want[1.0] = O.threshold_hits_chunked(small, filt[: 640 * 800], threshold=1.0)
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)

        def queued(thr, n_queue=1):
            sess.upload_records(small if thr == 1.0 else rec, thr)
            for _ in range(n_queue):
                sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, 0), 2, 2)
            return sess._fill_hits(sess.hits_wait())

        sess.upload_records(rec, 10.0)
        first = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)             # sizes the row buffers: later passes may speculate
        G.assert_struct_equal(first, want[10.0], float_rtol=1e-6, what="waited pass")
        G.assert_struct_equal(queued(10.0, 3), want[10.0], float_rtol=1e-6, what="three queued passes")
        G.assert_struct_equal(queued(40.0), want[40.0], float_rtol=1e-6, what="fewer rows than the last pass")
        G.assert_struct_equal(queued(6.0), want[6.0], float_rtol=1e-6, what="more rows than the speculative launch held")
        G.assert_struct_equal(queued(6.0, 2), want[6.0], float_rtol=1e-6, what="queued again after the regrow")
        sess.profile(True)
        G.assert_struct_equal(queued(1.0), want[1.0], float_rtol=1e-6, what="a span overflows its event slot")
        names = sess.profile_report()
        assert "k_sg_runs32" in names and "k_hit_runs" in names, sorted(names)   # streaming pass, then the bitmap route
        sess.profile(True)
        G.assert_struct_equal(queued(10.0, 2), want[10.0], float_rtol=1e-6, what="next upload: streaming route again")
        assert "k_sg_runs32" in sess.profile_report() and "k_hit_runs" not in sess.profile_report()
        G.assert_struct_equal(sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2), want[10.0], float_rtol=1e-6, what="waited pass after queued ones")
