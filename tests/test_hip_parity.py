"""HIP kernels through the C ABI vs the golden fixtures (reference outputs) and the oracle.

Integer fields (positions, edges, timestamps, ids) must be bit-exact.  Float fields are compared
bit-exact where the kernel reproduces the reference's evaluation order, else within
FLOAT_RTOL = 1e-6 (north_star tolerance for areas / widths), stated per test.
"""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import (
    HipBasicFeaturesPlugin,
    HipThresholdHitPlugin,
    HipWaveformWidthIntegralPlugin,
    HipWavePoolFilteredPlugin,
)

pytestmark = pytest.mark.gpu

FLOAT_RTOL = 1e-6
SG_CASES = [n for n in G.case_names() if n != "v1725_bw"]


@pytest.fixture(scope="module")
def sess():
    s = DeviceSession(0)
    yield s
    s.close()


def _sg(case):
    fp = G.filter_params(case)
    return fp["sg_window_size"], fp["sg_poly_order"]


@pytest.mark.parametrize("name", SG_CASES)
def test_savgol_materialised_bit_exact(sess, name):
    case = G.load_case(name)
    sess.upload_pool(case["wave_pool"])
    sess.upload_records(case["records"])
    sess.set_sg_plan(*_sg(case))
    got = sess.savgol()
    np.testing.assert_array_equal(got, case["wave_pool_filtered"])


@pytest.mark.parametrize("name", G.case_names())
def test_hits_raw_and_prefiltered(sess, name):
    case = G.load_case(name)
    hp = G.hit_params(case)
    if "hits_raw" in case:
        sess.upload_pool(case["wave_pool"])
        sess.upload_records(case["records"], hp["thresholds"])
        got = sess.threshold_hits(_lib.SRC_RAW, hp["left_extension"], hp["right_extension"])
        G.assert_struct_equal(got, case["hits_raw"], float_rtol=FLOAT_RTOL, what=f"{name} hits_raw")
    if "hits_filt" in case:
        sess.upload_pool(case["wave_pool"])
        sess.upload_filtered_pool(case["wave_pool_filtered"])
        sess.upload_records(case["records"], hp["thresholds"])
        got = sess.threshold_hits(_lib.SRC_F32, hp["left_extension"], hp["right_extension"])
        G.assert_struct_equal(got, case["hits_filt"], float_rtol=FLOAT_RTOL, what=f"{name} hits_filt")


@pytest.mark.parametrize("name", SG_CASES)
def test_hits_fused_filter(sess, name):
    case = G.load_case(name)
    if "hits_filt" not in case:
        pytest.skip("no filtered hits in fixture")
    hp = G.hit_params(case)
    sess.upload_pool(case["wave_pool"])
    sess.upload_records(case["records"], hp["thresholds"])
    sess.set_sg_plan(*_sg(case))
    got = sess.threshold_hits(_lib.SRC_SG_FUSED, hp["left_extension"], hp["right_extension"])
    G.assert_struct_equal(got, case["hits_filt"], float_rtol=FLOAT_RTOL, what=f"{name} fused")


def test_fused_baseline_filter_hits(sess):
    case = G.load_case("v1725_default")
    rec = case["records"].copy()
    want_bl = rec["baseline"].copy()
    rec["baseline"] = -1.0  # must be recomputed by the kernel
    sess.upload_pool(case["wave_pool"])
    sess.upload_records(rec, 10.0)
    sess.set_sg_plan(11, 2)
    got = sess.fused_baseline_filter_hits((0, 40))
    G.assert_struct_equal(got, case["hits_filt"], float_rtol=FLOAT_RTOL, what="fused baseline")
    np.testing.assert_array_equal(sess.baseline_mean(0, 40), want_bl)
    assert np.all(np.isnan(sess.baseline_mean(7, 7)))


@pytest.mark.parametrize("name", G.case_names())
def test_basic_features(sess, name):
    case = G.load_case(name)
    bp = G.bf_params(case)
    fixed = None if np.all(np.isnan(bp["fixed_baseline"])) else bp["fixed_baseline"]
    for key, pool, src in (("bf_raw", "wave_pool", _lib.SRC_RAW), ("bf_filt", "wave_pool_filtered", _lib.SRC_F32)):
        if key not in case:
            continue
        sess.upload_pool(case["wave_pool"])
        if src == _lib.SRC_F32:
            sess.upload_filtered_pool(case[pool])
        sess.upload_records(case["records"])
        got = sess.basic_features(src, bp["height_range"], bp["area_range"], fixed)
        G.assert_struct_equal(got, case[key], what=f"{name} {key} (bit-exact)")


@pytest.mark.parametrize("name", G.case_names())
def test_features_both_in_one_read(sess, name):
    """wfa_features_both: both feature tables of the raw pool from one call, against the reference's two tables.  Uniform
    records with the plugins' default area range take the fused kernel (one staging of every record group), everything
    else the two kernels in turn -- bit-exact either way."""
    case = G.load_case(name)
    if "bf_raw" not in case or "wi_raw" not in case:
        pytest.skip("no raw feature tables in this fixture")
    bp, wp = G.bf_params(case), G.wi_params(case)
    if not np.all(np.isnan(bp["fixed_baseline"])):
        pytest.skip("per-channel fixed baselines: the separate basic_features call covers them")
    dt = wp["dt"] if wp["dt"] is not None else 1.0 / wp["sampling_rate"]
    sess.upload_pool(case["wave_pool"])
    sess.upload_records(case["records"])
    sess.profile(True)
    bf, wi = sess.features_both(bp["height_range"], bp["area_range"], wp["q_low"], wp["q_high"], float(dt))
    names = sess.profile_report()
    rec = case["records"]
    uniform = len(rec) > 0 and np.all(rec["event_length"] == rec["event_length"][0]) and \
        np.array_equal(rec["wave_offset"], rec["wave_offset"][0] + np.arange(len(rec)) * int(rec["event_length"][0]))
    if uniform and tuple(bp["area_range"]) == (0, None) and rec["event_length"][0] % 8 == 0 and rec["wave_offset"][0] % 8 == 0 \
            and rec["event_length"][0] >= 24:                          # (span mode of the upload: wfa_capi.hip)
        assert "k_features_both_leaf" in names, sorted(names)
    G.assert_struct_equal(bf, case["bf_raw"], what=f"{name} fused basic features (bit-exact)")
    G.assert_struct_equal(wi, case["wi_raw"], what=f"{name} fused width integral (bit-exact)")


@pytest.mark.parametrize("name", G.case_names())
def test_width_integral(sess, name):
    case = G.load_case(name)
    wp = G.wi_params(case)
    dt = wp["dt"] if wp["dt"] is not None else 1.0 / wp["sampling_rate"]
    for key, pool, src in (("wi_raw", "wave_pool", _lib.SRC_RAW), ("wi_filt", "wave_pool_filtered", _lib.SRC_F32)):
        if key not in case:
            continue
        sess.upload_pool(case["wave_pool"])
        if src == _lib.SRC_F32:
            sess.upload_filtered_pool(case[pool])
        sess.upload_records(case["records"])
        got = sess.width_integral(src, wp["q_low"], wp["q_high"], float(dt))
        G.assert_struct_equal(got, case[key], what=f"{name} {key} (bit-exact)")


def test_against_oracle_medium(sess):
    """2*10^6 samples, uniform records: every stage against the oracle on the same seeded input."""
    rec, pool = synth.make_run(2500, "v1725", cfg=21)
    sess.upload_pool(pool)
    sess.upload_records(rec, 10.0)
    sess.set_sg_plan(11, 2)
    filt = sess.savgol()
    want_filt = O.filter_wave_pool_uniform(pool, 800)
    np.testing.assert_array_equal(filt, want_filt)
    G.assert_struct_equal(sess.threshold_hits(_lib.SRC_RAW), O.threshold_hits_chunked(rec, pool),
                          float_rtol=FLOAT_RTOL, what="raw hits")
    want = O.threshold_hits_chunked(rec, want_filt)
    G.assert_struct_equal(sess.threshold_hits(_lib.SRC_F32), want, float_rtol=FLOAT_RTOL, what="f32 hits")
    G.assert_struct_equal(sess.threshold_hits(_lib.SRC_SG_FUSED), want, float_rtol=FLOAT_RTOL, what="fused hits")
    # features reproduce numpy's summation order: bit-exact floats, raw and filtered pools
    G.assert_struct_equal(sess.basic_features(_lib.SRC_RAW), O.basic_features(rec, pool), what="bf raw")
    G.assert_struct_equal(sess.width_integral(_lib.SRC_RAW, dt=2.0), O.width_integral(rec, pool), what="wi raw")
    G.assert_struct_equal(sess.basic_features(_lib.SRC_F32), O.basic_features(rec, want_filt), what="bf filt")
    G.assert_struct_equal(sess.width_integral(_lib.SRC_F32, dt=2.0), O.width_integral(rec, want_filt), what="wi filt")


def test_plugins_drop_in_chain():
    """The plugin classes behind a context, chained like the reference profile."""
    case = G.load_case("v1725_channel_cfg")
    opt = case["options"]
    ctx = SimpleContext(
        {"wave_source": "records", "hit_threshold": {**opt["hit"], "use_filtered": True}, "basic_features": opt["bf"]},
        {"records": case["records"], "wave_pool": case["wave_pool"]},
        plugins=[HipWavePoolFilteredPlugin(), HipThresholdHitPlugin(), HipBasicFeaturesPlugin(),
                 HipWaveformWidthIntegralPlugin()],
    )
    np.testing.assert_array_equal(ctx.get_data("run", "wave_pool_filtered"), case["wave_pool_filtered"])
    G.assert_struct_equal(ctx.get_data("run", "hit_threshold"), case["hits_filt"], float_rtol=FLOAT_RTOL)
    G.assert_struct_equal(ctx.get_data("run", "basic_features"), case["bf_raw"])
    G.assert_struct_equal(ctx.get_data("run", "waveform_width_integral"), case["wi_raw"])
    # fused variant gives the same hits without materialising the filtered pool
    ctx2 = SimpleContext({"wave_source": "records",
                          "hit_threshold": {**opt["hit"], "use_filtered": True, "fuse_filter": True}},
                         {"records": case["records"], "wave_pool": case["wave_pool"]},
                         plugins=[HipWavePoolFilteredPlugin(), HipThresholdHitPlugin()])
    G.assert_struct_equal(ctx2.get_data("run", "hit_threshold"), case["hits_filt"], float_rtol=FLOAT_RTOL)


def test_error_behaviour(sess):
    case = G.load_case("kat_records_view")
    sess.upload_pool(case["wave_pool"])
    bad = case["records"].copy()
    bad["wave_offset"] = 4
    with pytest.raises(ValueError, match="outside wave_pool bounds"):
        sess.upload_records(bad)
    bad["wave_offset"] = -1
    with pytest.raises(ValueError, match="negative wave_offset"):
        sess.upload_records(bad)
    with pytest.raises(ValueError, match="q_low/q_high"):
        sess.upload_records(case["records"])
        sess.width_integral(_lib.SRC_RAW, 0.9, 0.1, 1.0)


def test_features_long_and_positive_records(sess):
    """Records longer than numpy's 8192-element reduce buffer and known polarities (float32 signal path)."""
    rng = np.random.default_rng(11)
    lens = [20000, 8193, 8192, 129, 128, 127, 9, 8, 7, 1]
    rec = np.zeros(len(lens), dtype=O.RECORDS_DTYPE)
    parts, cur = [], 3
    parts.append(np.zeros(3, dtype=np.uint16))
    for i, L in enumerate(lens):
        w = np.clip(np.rint(8000 + rng.normal(0, 40, L)), 0, 65535).astype(np.uint16)
        w[L // 3 : L // 3 + max(L // 10, 1)] -= 300
        rec["wave_offset"][i], rec["event_length"][i] = cur, L
        rec["baseline"][i] = float(np.mean(w[: min(40, L)].astype(float)))
        parts.append(w)
        cur += L
    rec["record_id"] = np.arange(len(lens))
    rec["dt"] = 2
    rec["timestamp"] = np.arange(len(lens)) * 10**8
    rec["polarity"] = ["unknown", "negative", "positive", "unknown", "negative", "positive", "unknown", "negative",
                       "positive", "unknown"]
    pool = np.concatenate(parts)
    sess.upload_pool(pool)
    sess.upload_records(rec)
    G.assert_struct_equal(sess.basic_features(_lib.SRC_RAW, (40, 90), (0, None)),
                          O.basic_features(rec, pool), what="bf long")
    G.assert_struct_equal(sess.basic_features(_lib.SRC_RAW, (-50, None), (5, -3)),
                          O.basic_features(rec, pool, height_range=(-50, None), area_range=(5, -3)), what="bf slices")
    G.assert_struct_equal(sess.width_integral(_lib.SRC_RAW, 0.1, 0.9, 2.0), O.width_integral(rec, pool, dt=2.0),
                          what="wi long")


def test_sharded_equals_unsharded(sess):
    """Channel shards processed one after the other on this GPU, merged on the host == one pass."""
    from waveformanalysis_amd import sharding

    rec, pool = synth.make_run(640, "v1725", cfg=31)
    sess.upload_pool(pool)
    sess.upload_records(rec, 10.0)
    sess.set_sg_plan(11, 2)
    want = sess.threshold_hits(_lib.SRC_SG_FUSED)
    rows, shards = [], [sharding.make_shard(rec, pool, 4, r) for r in range(4)]
    for s in shards:
        sess.upload_pool(s.wave_pool)
        sess.upload_records(s.records, 10.0)
        rows.append(sess.threshold_hits(_lib.SRC_SG_FUSED, max_len=s.max_len))
    merged = sharding.merge_rows(rows, [s.orig_index for s in shards], [s.records for s in shards])
    G.assert_struct_equal(merged, want, what="sharded == unsharded")


def test_streaming_chunks_with_device_pool():
    """compute_chunk over a thread pool (one session / stream per worker) == the static plugin."""
    from waveformanalysis_amd.device import DevicePool
    from waveformanalysis_amd.streaming import HipThresholdHitStream, records_to_chunks

    rec, pool = synth.make_run(3000, "v1725", cfg=32)
    ctx = SimpleContext({"wave_source": "records"}, {"records": rec, "wave_pool": pool})
    dp = DevicePool([0])
    try:
        for use_filtered in (False, True):
            plugin = HipThresholdHitStream(use_filtered=use_filtered, max_len=800, device_pool=dp)
            chunks = records_to_chunks(rec, 700, "run")
            timeline = []
            outs = plugin.run_chunks(chunks, ctx, "run", max_workers=3, timeline=timeline)
            # double buffer: chunk k + 1 is staged (uploaded and queued) before anybody waits for chunk k, on the
            # other of the two sessions; results are collected in input order
            assert [t[0] for t in timeline] == list(range(len(chunks)))
            for (k, _b, queued, collected), (_k1, begin1, queued1, _c1) in zip(timeline, timeline[1:]):
                assert queued <= begin1 <= queued1 <= collected, (k, timeline)
            got = np.concatenate([c.data for c in outs])
            want = O.threshold_hits_chunked(rec, O.filter_wave_pool_uniform(pool, 800) if use_filtered else pool)
            G.assert_struct_equal(got, want, float_rtol=FLOAT_RTOL, what=f"stream filtered={use_filtered}")
            assert all(o.start == c.start and o.end == c.end for o, c in zip(outs, chunks))
            # the generator entry point: the base class cuts `records` at time breaks / chunk_size, halo records are
            # processed twice and clipped away again, results arrive in input order
            rec2 = rec.copy()
            rec2["timestamp"][1500:] += 3 * 10**13
            ctx2 = SimpleContext({"wave_source": "records"}, {"records": rec2, "wave_pool": pool})
            want2 = O.threshold_hits_chunked(rec2, O.filter_wave_pool_uniform(pool, 800) if use_filtered else pool)
            for cfg in ({"chunk_size": 400, "max_workers": 3}, {"chunk_size": 1000, "required_halo_ns": 30_000_000, "parallel": False}):
                outs = list(plugin.compute(ctx2, "run", streaming_config=cfg))
                assert len(outs) >= 3 and {o.metadata["segment_id"] for o in outs} == {0, 1}
                G.assert_struct_equal(np.concatenate([c.data for c in outs]), want2, float_rtol=FLOAT_RTOL,
                                      what=f"stream generator {cfg}")
    finally:
        dp.close()


def test_rccl_single_rank_gather(sess):
    """RCCL leg of the C ABI with a 1-rank communicator: counts all-gather + send/recv to self,
    device-resident hit rows (rows=None) and host rows."""
    case = G.load_case("v1725_default")
    sess.upload_pool(case["wave_pool"])
    sess.upload_records(case["records"], 10.0)
    hits = sess.threshold_hits(_lib.SRC_RAW)
    sess.rccl_init(0, 1, DeviceSession.rccl_unique_id())
    counts, rows = sess.rccl_gather_rows(None, len(hits), hits.dtype, root=0)
    assert counts.tolist() == [len(hits)]
    G.assert_struct_equal(rows, hits, what="gather of resident rows")
    counts, rows = sess.rccl_gather_rows(hits[:7], 7, hits.dtype, root=0)
    G.assert_struct_equal(rows, hits[:7], what="gather of host rows")


def test_full_size_properties():
    """BASELINE config sizes (10^8 samples, 16-ch V1725): properties that do not need the oracle at scale.

    * determinism: two runs give identical rows;
    * order: rows sorted by (record, window start), windows inside the record, position inside the window;
    * two independent code paths agree: fused integer pass (mask/scan/runs/rows) vs materialised
      float32 pool (k_savgol_span) + literal float64 hit kernel on it: integer fields bit-exact;
    * checksum of checksums: per-record run counts sum to the number of rows;
    * oracle spot check on 2000 records drawn across the whole run.
    """
    n_rec = 125_000  # x 800 = 1e8 samples
    rec, pool = synth.make_run(n_rec, "v1725", cfg=1)
    with DeviceSession(0) as s:
        s.upload_pool(pool)
        blank = rec.copy()
        blank["baseline"] = np.nan
        s.upload_records(blank, 10.0)
        s.set_sg_plan(11, 2)
        fused = s.fused_baseline_filter_hits((0, 40))
        again = s.fused_baseline_filter_hits((0, 40))
        np.testing.assert_array_equal(fused.view(np.uint8), again.view(np.uint8))
        np.testing.assert_array_equal(s.baseline_mean(0, 40), rec["baseline"])

        rid = fused["record_id"]
        assert np.all(np.diff(rid) >= 0)
        same = np.diff(rid) == 0
        assert np.all(np.diff(fused["edge_start"].astype(np.int64))[same] > 0)
        assert np.all((fused["edge_start"] >= 0) & (fused["edge_end"] <= 800) & (fused["edge_start"] < fused["edge_end"]))
        assert np.all((fused["position"] >= fused["edge_start"]) & (fused["position"] < fused["edge_end"]))
        assert np.all(fused["width"] == (fused["edge_end"] - fused["edge_start"]).astype(np.float32))
        assert np.all(fused["integral"] >= 0) and np.all(np.isfinite(fused["height"]))
        counts = np.bincount(rid, minlength=n_rec)
        assert counts.sum() == len(fused) and len(fused) > n_rec  # ~1.7 hits per record

        s.upload_records(rec, 10.0)
        s.savgol(download=False)
        via_pool = s.threshold_hits(_lib.SRC_F32)
        G.assert_struct_equal(via_pool, fused, float_rtol=FLOAT_RTOL, what="fused vs materialised path")

    pick = np.sort(np.random.default_rng(0).choice(n_rec, 2000, replace=False))
    sub = rec[pick].copy()
    sub_pool = pool.reshape(-1, 800)[pick].reshape(-1)
    sub["wave_offset"] = np.arange(len(pick), dtype=np.int64) * 800
    want = O.threshold_hits_chunked(sub, O.filter_wave_pool_uniform(sub_pool, 800))
    got = fused[np.isin(fused["record_id"], rec["record_id"][pick])]
    G.assert_struct_equal(got, want, float_rtol=FLOAT_RTOL, what="oracle spot check")


def test_butterworth_sosfiltfilt_bit_exact(sess):
    """Butterworth branch of wave_pool_filtered: the lane-per-record kernel runs scipy's sosfiltfilt
    recursion literally -> bit-exact float32 against the reference fixture and the oracle."""
    from waveformanalysis_amd.filter_engine import design_bw

    case = G.load_case("v1725_bw")
    fp = G.filter_params(case)
    sos, zi, padlen = design_bw(fp["lowcut"], fp["highcut"], fp["fs"], fp["filter_order"])
    sess.upload_pool(case["wave_pool"])
    sess.upload_records(case["records"])
    got = sess.sosfiltfilt(sos, zi, padlen)
    np.testing.assert_array_equal(got, case["wave_pool_filtered"])
    # hits on the resident Butterworth-filtered pool == reference hits on its filtered pool
    hp = G.hit_params(case)
    sess.upload_records(case["records"], hp["thresholds"])
    G.assert_struct_equal(sess.threshold_hits(_lib.SRC_F32, hp["left_extension"], hp["right_extension"]),
                          case["hits_filt"], float_rtol=FLOAT_RTOL, what="hits on BW pool")

    # ragged records (several shorter than padlen -> copied), different band
    rag = G.load_case("ragged_mixed")
    sos2, zi2, padlen2 = design_bw(0.02, 0.15, 0.5, 2)
    want = O.filter_wave_pool(rag["records"], rag["wave_pool"], "BW", bw_sos=sos2)
    sess.upload_pool(rag["wave_pool"])
    sess.upload_records(rag["records"])
    np.testing.assert_array_equal(sess.sosfiltfilt(sos2, zi2, padlen2), want)

    # plugin path
    ctx = SimpleContext({"filter_type": "BW", "lowcut": fp["lowcut"], "highcut": fp["highcut"], "fs": fp["fs"]},
                        {"records": case["records"], "wave_pool": case["wave_pool"]},
                        plugins=[HipWavePoolFilteredPlugin()])
    np.testing.assert_array_equal(ctx.get_data("run", "wave_pool_filtered"), case["wave_pool_filtered"])


def test_enqueued_passes_equal_the_waited_ones():
    """wfa_hits_enqueue / wfa_hits_wait: passes queued without a host round trip give the rows of the blocking call,
    also when a pass finds more rows than the speculative launch covered (it is redone exactly inside wait)."""
    rec, pool = synth.make_run(4000, "v1725", cfg=33)
    rec_in = rec.copy()
    rec_in["baseline"] = np.nan
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.set_sg_plan(11, 2)
        sess.upload_records(rec_in, 400.0)                       # few hits: a small row bound for what follows
        few = sess.fused_baseline_filter_hits((0, 40), 2, 2)
        sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, 40), 2, 2)
        assert sess.hits_wait() == len(few)
        sess.upload_records(rec_in, 10.0)                        # many more hits than that bound
        sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, 40), 2, 2)
        n = sess.hits_wait()
        want = sess.fused_baseline_filter_hits((0, 40), 2, 2)
        assert n == len(want) > 4 * len(few)
        for _ in range(3):                                       # now within the bound: nothing waits in between
            sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, 40), 2, 2)
        got = sess._fill_hits(sess.hits_wait())
        assert got.tobytes() == want.tobytes()
        sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, 40), 2, 2)
        assert sess._fill_hits(len(want)).tobytes() == want.tobytes()   # fill waits by itself
