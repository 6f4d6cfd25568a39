"""find_peaks-based `hit` detector (k_find_peaks) through the C ABI vs the reference fixtures and the oracle.

Bar: position / timestamp / ids bit-exact; height / edge_start / edge_end are float32 roundings of the same
float64 expressions scipy evaluates (same operation order, no contraction) -> compared exactly as well.
"""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import HipHitFinderPlugin, HipWavePoolFilteredPlugin

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = DeviceSession(0)
    yield s
    s.close()


def _run(sess, case, cfg):
    cfg = dict(cfg)
    filt = cfg.pop("use_filtered", True)
    pool = case["wave_pool_filtered"] if filt else case["wave_pool"]
    sess.upload_pool(pool)
    sess.upload_records(case["records"], 0.0)
    return sess.find_peaks(_lib.SRC_F32 if filt else _lib.SRC_RAW, **cfg)


@pytest.mark.parametrize("name", G.peaks_case_names())
def test_session_matches_reference(sess, name):
    case = G.load_peaks(name)
    for k, cfg in enumerate(case["configs"]):
        G.assert_struct_equal(_run(sess, case, cfg), case[f"hit_{k}"], what=f"{name} cfg {k}")


@pytest.mark.parametrize("name", G.peaks_case_names())
def test_plugin_matches_reference(name):
    case = G.load_peaks(name)
    for k, cfg in enumerate(case["configs"]):
        ctx = SimpleContext({"wave_source": "records", "hit": dict(cfg)},
                            {"records": case["records"], "wave_pool": case["wave_pool"],
                             "wave_pool_filtered": case["wave_pool_filtered"]},
                            plugins=[HipHitFinderPlugin()])
        G.assert_struct_equal(ctx.get_data("run", "hit"), case[f"hit_{k}"], what=f"{name} cfg {k}")


def test_plugin_chain_from_raw_pool():
    """hit <- wave_pool_filtered <- wave_pool, both stages on the GPU."""
    case = G.load_peaks("peaks_v1725")
    ctx = SimpleContext({"wave_source": "records", "hit": dict(case["configs"][0])},
                        {"records": case["records"], "wave_pool": case["wave_pool"]},
                        plugins=[HipWavePoolFilteredPlugin(), HipHitFinderPlugin()])
    G.assert_struct_equal(ctx.get_data("run", "hit"), case["hit_0"])


@pytest.mark.parametrize("cfg", [
    dict(),
    dict(use_derivative=False, height=12.0, prominence=3.0, width=2),
    dict(height=4.0, prominence=0.5, width=1, distance=7),
    dict(use_derivative=False, height=8.0, prominence=1.0, width=1, distance=25, threshold=0.5),
    dict(height=2.0, prominence=0.1, width=1, distance=1, height_window_extension=1),
    dict(height=3.0, prominence=0.2, width=1, distance=60, height_method="diff"),
    dict(use_derivative=False, height=5.0, prominence=0.5, width=30, height_method="diff"),
])
def test_against_oracle_medium(sess, cfg):
    rec, pool = synth.make_run(1500, "v1725", cfg=33)
    filt = O.filter_wave_pool_uniform(pool, 800)
    for src, p in ((_lib.SRC_F32, filt), (_lib.SRC_RAW, pool)):
        sess.upload_pool(p)
        sess.upload_records(rec, 0.0)
        got = sess.find_peaks(src, **cfg)
        want = O.find_peak_hits(rec, p, **cfg)
        assert len(want) > 0
        G.assert_struct_equal(got, want, what=f"src {src} cfg {cfg}")


def test_one_walk_and_two_walk_candidate_paths(sess):
    """The candidate list comes from one walk into per-record slots (k_find_peaks_slots + k_peak_compact) unless a
    record has more candidates than slots, then from the count + fill pair; `no_peak_slots` forces the pair.  Same rows."""
    rec, pool = synth.make_run(4000, "v1725", cfg=34)
    filt = O.filter_wave_pool_uniform(pool, 800)
    sess.upload_pool(filt)
    sess.upload_records(rec, 0.0)
    for cfg, route in ((dict(), "k_peak_compact"), (dict(height=1.0, prominence=0.1, width=1), "k_find_peaks<fill candidates>")):
        sess.profile(True)
        got = sess.find_peaks(_lib.SRC_F32, **cfg)
        names = set(sess.profile_report())
        assert "k_find_peaks_hot" in names and route in names, names         # uniform records: LDS-staged, height prefilter
        sess.set_option("no_peak_hot", True)
        sess.profile(True)
        every_sample = sess.find_peaks(_lib.SRC_F32, **cfg)
        assert "k_find_peaks_staged" in set(sess.profile_report())            # the plateau machine over every sample
        sess.set_option("no_peak_hot", False)
        assert every_sample.tobytes() == got.tobytes()
        sess.set_option("no_span", True)
        sess.profile(True)
        per_record = sess.find_peaks(_lib.SRC_F32, **cfg)
        assert "k_find_peaks_slots" in set(sess.profile_report())             # any layout: one lane per record
        sess.set_option("no_span", False)
        assert per_record.tobytes() == got.tobytes()
        sess.set_option("no_peak_slots", True)
        sess.profile(True)
        two_walks = sess.find_peaks(_lib.SRC_F32, **cfg)
        assert "k_find_peaks<count candidates>" in set(sess.profile_report())
        sess.set_option("no_peak_slots", False)
        assert len(got) > 0 and got.tobytes() == two_walks.tobytes()
        G.assert_struct_equal(got, O.find_peak_hits(rec, filt, **cfg), what=f"{cfg}")
    sess.profile(False)


def test_edge_cases(sess):
    rec, pool = synth.make_run(8, "v1725", cfg=5)
    # records too short for any peak, and an empty record list
    short = rec.copy()
    short["event_length"] = [0, 1, 2, 3, 4, 5, 800, 800]
    sess.upload_pool(pool)
    sess.upload_records(short, 0.0)
    cfg = dict(height=1.0, prominence=0.1, width=0)
    G.assert_struct_equal(sess.find_peaks(_lib.SRC_RAW, **cfg), O.find_peak_hits(short, pool, **cfg))
    sess.upload_records(rec[:0], 0.0)
    assert len(sess.find_peaks(_lib.SRC_RAW)) == 0
    sess.upload_records(rec, 0.0)
    with pytest.raises(ValueError, match="distance"):
        sess.find_peaks(_lib.SRC_RAW, distance=0)
    with pytest.raises(ValueError, match="峰高计算方法"):
        sess.find_peaks(_lib.SRC_RAW, height_method="nope")
    # no cap on the number of peaks: every local maximum of every record
    cfg = dict(height=-1e9, prominence=0.0, width=0, height_window_extension=1)
    got = sess.find_peaks(_lib.SRC_RAW, **cfg)
    assert len(got) > 100 * len(rec)
    G.assert_struct_equal(got, O.find_peak_hits(rec, pool, **cfg))
    # an empty minmax window (ext = 0 around a zero-width peak) is numpy's ValueError in the reference
    with pytest.raises(ValueError, match="zero-size array"):
        sess.find_peaks(_lib.SRC_RAW, height=-1e9, prominence=0.0, width=0, height_window_extension=0)


def test_plugin_record_id_indirection_and_errors():
    """peak_finding.py:401-407: the waveform is fetched by record_id, the metadata by row."""
    case = G.load_peaks("peaks_v1725")
    rec = case["records"].copy()
    rec["record_id"] = rec["record_id"][::-1]
    data = {"records": rec, "wave_pool": case["wave_pool"], "wave_pool_filtered": case["wave_pool_filtered"]}
    ctx = SimpleContext({"wave_source": "records", "hit": dict(case["configs"][0])}, data, plugins=[HipHitFinderPlugin()])
    got = ctx.get_data("run", "hit")
    mixed = rec.copy()
    for f in ("wave_offset", "event_length", "baseline", "polarity"):
        mixed[f] = rec[f][rec["record_id"]]
    cfg = dict(case["configs"][0])
    cfg.pop("use_filtered", None)
    G.assert_struct_equal(got, O.find_peak_hits(mixed, case["wave_pool_filtered"], **cfg))
    with pytest.raises(RuntimeError, match="requires 'st_waveforms'"):
        SimpleContext({"hit": {"wave_source": "st_waveforms"}}, data, plugins=[HipHitFinderPlugin()]).get_data("run", "hit")
    with pytest.raises(RuntimeError, match="峰高计算方法"):
        SimpleContext({"hit": {"wave_source": "records", "height_method": "nope"}}, data,
                      plugins=[HipHitFinderPlugin()]).get_data("run", "hit")


@pytest.mark.parametrize("L,n", [(24, 900), (64, 500), (104, 300), (520, 200), (1000, 130), (2048, 66), (4000, 33), (8192, 9)])
def test_staged_candidate_walk_over_record_lengths(sess, L, n):
    """The LDS-staged candidate walk (k_find_peaks_staged) for 1..64 lanes per record, both sources, with and without
    the derivative: the oracle's rows, and the same bytes as the lane-per-record walk (`no_span`)."""
    rec, pool = synth.make_run(n, "v1725", cfg=50 + L % 11, L=L)
    filt = O.filter_wave_pool_uniform(pool, L) if L >= 11 else pool.astype(np.float32)
    for src, p in ((_lib.SRC_F32, filt), (_lib.SRC_RAW, pool)):
        sess.upload_pool(p)
        sess.upload_records(rec, 0.0)
        for cfg in (dict(height=6.0, prominence=0.5, width=1), dict(use_derivative=False, height=8.0, prominence=1.0, width=2),
                    dict(height=1.0, prominence=0.1, width=0, distance=5)):
            got = sess.find_peaks(src, **cfg)
            G.assert_struct_equal(got, O.find_peak_hits(rec, p, **cfg), what=f"L {L} src {src} {cfg}")
            sess.set_option("no_span", True)
            assert sess.find_peaks(src, **cfg).tobytes() == got.tobytes()
            sess.set_option("no_span", False)
            sess.set_option("no_peak_hot", True)
            assert sess.find_peaks(src, **cfg).tobytes() == got.tobytes()
            sess.set_option("no_peak_hot", False)


@pytest.mark.parametrize("dense,polarity", [(0, "negative"), (0, "positive"), (1, "negative"), (2, "negative")])
def test_height_prefilter_at_the_boundary(sess, dense, polarity):
    """k_find_peaks_hot runs the plateau machine only on chunks that may hold a detection value >= `height`, decided in
    float32.  Raw uint16 samples give integer detection values, long plateaus and ties: heights that ARE detection values
    (>= must keep them), heights half a step off, a height no value reaches, and plateaus that cross chunk and record
    ends.  Rows equal the oracle's and the bytes of the every-sample walk, for the records branch and both dense-row forms."""
    rng = np.random.default_rng(77 + dense)
    L, n = 160, 400
    rec, _ = synth.make_run(n, "v1725", cfg=3, L=L)
    w = np.full((n, L), 8000, dtype=np.int64) + rng.integers(-1, 2, size=(n, L))
    for r in range(n):
        for _ in range(3):
            t0, wid, amp = int(rng.integers(2, L - 40)), int(rng.integers(1, 30)), int(rng.integers(2, 40))
            w[r, t0:t0 + wid] -= amp                      # flat-bottomed pulses: plateaus of the signal and of its slope
        if r % 5 == 0:
            w[r, L - 9:] -= 25                            # a step that stays open at the record's end
        if r % 7 == 0:
            w[r, :3] -= 30
    pool = w.astype(np.uint16).reshape(-1)
    rec["baseline"] = 8000.0 + (np.arange(n) % 3) * 0.25
    rec["polarity"] = polarity                             # one polarity class per upload: the uniform-layout routes need it
    f32 = pool.astype(np.float32)
    for src, p in ((_lib.SRC_RAW, pool), (_lib.SRC_F32, f32)):
        sess.upload_pool(p)
        sess.upload_records(rec, 0.0)
        for cfg in (dict(height=25.0, prominence=0.0, width=0), dict(height=24.5, prominence=0.0, width=0),
                    dict(use_derivative=False, height=25.0, prominence=0.0, width=0),
                    dict(use_derivative=False, height=24.75, prominence=1.0, width=1),
                    dict(use_derivative=False, height=2.0, prominence=0.0, width=0, threshold=1.0),
                    dict(height=1.0, prominence=0.0, width=0, threshold=1.0),
                    dict(height=1e7, prominence=0.0, width=0)):
            sess.profile(True)
            got = sess.find_peaks(src, dense_rows=dense, **cfg)
            assert "k_find_peaks_hot" in set(sess.profile_report())
            sess.set_option("no_peak_hot", True)
            assert sess.find_peaks(src, dense_rows=dense, **cfg).tobytes() == got.tobytes(), (src, cfg)
            sess.set_option("no_peak_hot", False)
            if dense == 2:      # the streaming detector's float64 rows: covered against its oracle in test_hip_streaming
                continue
            if dense:
                st = np.zeros(n, dtype=[("wave", p.dtype if src == _lib.SRC_F32 else np.int16, (L,)), ("baseline", "f8"),
                                        ("dt", "i4"), ("timestamp", "i8"), ("board", "i2"), ("channel", "i2"),
                                        ("record_id", "i8"), ("event_length", "i4")])
                st["wave"] = p.reshape(n, L)
                for f in ("baseline", "dt", "timestamp", "board", "channel", "record_id", "event_length"):
                    st[f] = rec[f]
                want = O.find_peak_hits_dense(st, **cfg)
            else:
                want = O.find_peak_hits(rec, p, **cfg)
            assert len(want) == 0 if cfg["height"] > 1e6 else (len(want) > 0 or polarity == "positive")
            G.assert_struct_equal(got, want, what=f"dense {dense} src {src} {cfg}")
    sess.profile(False)
