"""Hit-table stages on the GPU (hit merge, event grouping of merged hits) against fixtures produced by the
reference's plugins and against the oracle's literal loops on larger crafted tables."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd.device import DeviceSession
from waveformanalysis_amd.dtypes import THRESHOLD_HIT_DTYPE
from waveformanalysis_amd.event_grouping import group_hit_windows, group_hit_windows_flat
from waveformanalysis_amd.hit_merge import compute_cluster_rows, compute_merged_rows
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import (
    HipHitGroupedPlugin,
    HipHitMergeClustersPlugin,
    HipHitMergedComponentsPlugin,
    HipHitMergePlugin,
)

pytestmark = pytest.mark.gpu
COLS = ("dt", "boards", "channels", "heights", "integrals", "timestamps", "record_ids", "sample_starts", "sample_ends")


@pytest.fixture(scope="module")
def sess():
    s = DeviceSession(0)
    yield s
    s.close()


@pytest.mark.parametrize("name", G.merge_case_names())
def test_merge_plugins_match_reference(name):
    case = G.load_merge(name)
    for k, cfg in enumerate(case["configs"]):
        ctx = SimpleContext(dict(cfg), {"hit_threshold": case["hits"]},
                            plugins=[HipHitMergeClustersPlugin(), HipHitMergePlugin(), HipHitMergedComponentsPlugin()])
        G.assert_struct_equal(ctx.get_data("run", "hit_merge_clusters"), case[f"clusters_{k}"], what=f"{name} clusters {k}")
        G.assert_struct_equal(ctx.get_data("run", "hit_merged"), case[f"merged_{k}"], what=f"{name} merged {k}")
        G.assert_struct_equal(ctx.get_data("run", "hit_merged_components"), case[f"components_{k}"], what=f"{name} comps {k}")


@pytest.mark.parametrize("name", G.merge_case_names())
def test_grouping_of_merged_hits(name, sess):
    """Merged hits that span records take their window from the component hits (event_grouping.py:369-416)."""
    case = G.load_merge(name)
    for k in range(len(case["configs"])):
        for tw in case["windows"]:
            tag = f"g{k}_w{int(tw)}"
            df = group_hit_windows(case[f"merged_{k}"], float(tw), component_rows=case[f"components_{k}"],
                                   component_hits=case["hits"], session=sess)
            np.testing.assert_array_equal(df["t_min"].to_numpy(np.int64), case[f"{tag}_t_min"])
            np.testing.assert_array_equal(df["t_max"].to_numpy(np.int64), case[f"{tag}_t_max"])
            np.testing.assert_array_equal(df["n_hits"].to_numpy(np.int64), case[f"{tag}_n_hits"])
            for col in COLS:
                got = np.concatenate(list(df[col])) if len(df) else np.zeros(0)
                np.testing.assert_array_equal(got, case[f"{tag}_{col}"], err_msg=f"{name} {tag} {col}")
    # the plugin wires hit_merged + hit_merged_components + hit_threshold like the reference
    k, tw = 2, 100
    ctx = SimpleContext({"time_window_ns": float(tw)},
                        {"hit_merged": case[f"merged_{k}"], "hit_merged_components": case[f"components_{k}"],
                         "hit_threshold": case["hits"]}, plugins=[HipHitGroupedPlugin()])
    df = ctx.get_data("run", "hit_grouped")
    np.testing.assert_array_equal(df["t_min"].to_numpy(np.int64), case[f"g{k}_w{tw}_t_min"])
    with pytest.raises(ValueError, match="component_rows"):
        group_hit_windows(case["merged_3"], 100.0, session=sess)


def crafted(seed, n, n_records, n_channels=5):
    rng = np.random.default_rng(seed)
    L = 400
    rec_channel = rng.integers(-2, n_channels, n_records)  # negative ids too: keys sort as signed
    rec_board = rng.integers(0, 3, n_records)
    rec_dt = np.where(rng.random(n_records) < 0.9, 4, 2)
    rec_ts = np.zeros(n_records, dtype=np.int64)
    for key in set(zip(rec_board.tolist(), rec_channel.tolist())):
        idx = np.flatnonzero((rec_board == key[0]) & (rec_channel == key[1]))
        gaps = rng.choice([0, 0, 40, 400, 30000], size=len(idx)) * 1000
        rec_ts[idx] = 2**58 + np.cumsum(L * rec_dt[idx] * 1000 + gaps)
    hits = np.zeros(n, dtype=THRESHOLD_HIT_DTYPE)
    rid = rng.integers(0, n_records, n)
    start = rng.integers(0, L - 20, n)
    width = rng.integers(1, 20, n)
    pos = start + rng.integers(0, width)
    hits["record_id"], hits["edge_start"], hits["edge_end"], hits["position"] = rid, start, start + width, pos
    hits["width"], hits["dt"], hits["board"], hits["channel"] = width, rec_dt[rid], rec_board[rid], rec_channel[rid]
    hits["timestamp"] = rec_ts[rid] + pos * rec_dt[rid] * 1000
    hits["height"] = rng.choice([12.0, 12.0, 30.5, 77.25, 140.0], n)
    hits["integral"] = rng.uniform(5, 900, n).astype(np.float32)
    return hits


@pytest.mark.parametrize("cfg", [dict(), dict(merge_gap_ns=30.0), dict(merge_gap_ns=400.0, max_total_width_ns=900.0),
                                 dict(merge_gap_ns=1e5, max_total_width_ns=1e12)])
def test_merge_against_oracle_large(sess, cfg):
    """40 000 hits, 15 hardware channels, chains across records, clusters of hundreds of hits (pairwise sums)."""
    hits = crafted(21, 40000, 3000)
    clusters = O.hit_merge_clusters(hits, **cfg)
    want_rows = O.hit_merge_cluster_rows(clusters)
    got_rows = compute_cluster_rows(sess, hits, cfg.get("merge_gap_ns", 0.0), cfg.get("max_total_width_ns", 10000.0), None, "t")
    G.assert_struct_equal(got_rows, want_rows)
    G.assert_struct_equal(compute_merged_rows(sess, hits, got_rows, None, "t"), O.hit_merged_rows(hits, clusters))
    if cfg.get("merge_gap_ns", 0) >= 1e5:
        assert max(len(c) for c in clusters) > 128


def test_grouping_against_oracle_large(sess):
    hits = crafted(22, 60000, 4000, n_channels=40)
    for tw in (0.0, 100.0, 3000.0):
        flat = group_hit_windows_flat(hits, tw, session=sess)
        events = O.group_hit_windows_literal(hits, tw)
        assert len(events) == len(flat["event_start"]) - 1
        got_members = np.split(flat["order"], flat["event_start"][1:-1])
        for ev, (t_min, t_max, members) in enumerate(events):
            np.testing.assert_array_equal(got_members[ev], members)
        np.testing.assert_array_equal(flat["t_min"], [e[0] for e in events])
        np.testing.assert_array_equal(flat["t_max"], [e[1] for e in events])


def test_hit_table_edge_cases(sess):
    empty = np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)
    assert len(compute_cluster_rows(sess, empty, 10.0, 100.0, None, "t")) == 0
    flat = group_hit_windows_flat(empty, 10.0, session=sess)
    assert len(flat["order"]) == 0 and list(flat["event_start"]) == [0]
    one = crafted(1, 1, 1)
    rows = compute_cluster_rows(sess, one, 10.0, 100.0, None, "t")
    assert rows.tolist() == [(0, 0)]
    merged = compute_merged_rows(sess, one, rows, None, "t")
    assert merged["component_count"].tolist() == [1] and merged["height"][0] == one["height"][0]
    # a membership table that is not the one this config would produce is honoured as given
    hits = crafted(2, 50, 5)
    rows = O.hit_merge_cluster_rows([[3, 1, 4], [0], [10, 20, 30, 40, 49, 7, 8, 9, 11]])
    G.assert_struct_equal(compute_merged_rows(sess, hits, rows, None, "t"),
                          O.hit_merged_rows(hits, [[3, 1, 4], [0], [10, 20, 30, 40, 49, 7, 8, 9, 11]]))
    bad = rows.copy()
    bad["cluster_index"][0] = 5
    with pytest.raises(ValueError, match="not ordered by cluster_index"):
        compute_merged_rows(sess, hits, bad, None, "t")
    no_dt = np.zeros(3, dtype=[(n, THRESHOLD_HIT_DTYPE.fields[n][0]) for n in THRESHOLD_HIT_DTYPE.names if n != "dt"])
    with pytest.raises(ValueError, match="missing required field 'dt'"):
        compute_cluster_rows(sess, no_dt, 10.0, 100.0, None, "hit_merge_clusters")


def test_resident_rows_feed_merge_and_grouping_without_host_columns():
    """wfa_hit_merge_count / wfa_group_hit_windows_count with all column pointers NULL read the device-resident rows of the
    last hit pass (and, on a 1-rank communicator, of the last RCCL gather): same tables as with uploaded columns."""
    from waveformanalysis_amd import _lib, synth
    from waveformanalysis_amd.dtypes import THRESHOLD_HIT_DTYPE

    rec, pool = synth.make_run(4000, "v1725", cfg=91)
    with DeviceSession(0) as sess:
        sess.upload_pool(pool)
        sess.upload_records(rec, 10.0)
        sess.set_sg_plan(11, 2)
        rows = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
        n = len(rows)
        assert n > 3000
        want_g = sess.group_hit_windows(rows["timestamp"], rows["position"], rows["edge_start"], rows["edge_end"], rows["dt"],
                                        rows["board"], rows["channel"], rows["record_id"], 100.0)
        want_m = sess.hit_merge_clusters(rows["timestamp"], rows["position"], rows["edge_start"], rows["edge_end"], rows["dt"],
                                         rows["board"], rows["channel"], 20.0, 10000.0)
        sess.hit_rows_source("hits")
        got_g = sess.group_hit_windows_resident(n, 100.0)
        got_m = sess.hit_merge_clusters_resident(n, 20.0, 10000.0)
        for k in want_g:
            np.testing.assert_array_equal(got_g[k], want_g[k], err_msg=k)
        np.testing.assert_array_equal(got_m[0], want_m[0])
        np.testing.assert_array_equal(got_m[1], want_m[1])
        # through a gather (1 rank): rows stay on the device, nothing is downloaded
        sess.rccl_init(0, 1, DeviceSession.rccl_unique_id())
        counts, none = sess.rccl_gather_rows(None, n, THRESHOLD_HIT_DTYPE, root=0, download=False)
        assert none is None and int(counts.sum()) == n
        sess.hit_rows_source("gather")
        got_g2 = sess.group_hit_windows_resident(n, 100.0)
        for k in want_g:
            np.testing.assert_array_equal(got_g2[k], want_g[k], err_msg=f"gather {k}")
        with pytest.raises(ValueError, match="resident hit table has"):
            sess.group_hit_windows_resident(n + 1, 100.0)
