"""BASELINE.json config 4 at one-GPU scope: the 256-channel VX2730 run dealt to 8 channel shards, every shard through its
own hit pass, every shard's device-resident rows through the RCCL gather into ONE table on the root's device, then the
hit-table stages from that buffer.

The eight "ranks" run one after the other on the one GPU of the box (a 1-rank communicator, append mode of the gather:
one exchange per shard), so everything but the inter-GPU wire is the code path of `bench.py --gpus 8`:
`sharding.make_shard` -> `DeviceSession` pass -> `wfa_rccl_allgather_counts` + `wfa_rccl_gather_rows(rows = NULL)` ->
`wfa_hit_rows_source(ctx, 2)` -> `wfa_hit_merge_count` / `wfa_group_hit_windows_count` with NULL columns.

Expected values: the tables the REFERENCE produced for this run in one process (tests/golden/c5_replay.npz:
`hit_threshold`, `hit_merged`, `hit_merged_components`, `grouped_*`; consumer event_grouping.py:286-471), and for the
Savitzky-Golay route (padded shadow layout, k_sg_runs32 on 1500-sample records) the pinned oracle on the unsharded run.
"""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from tests.test_replay_cpu import load_fixture
from waveformanalysis_amd import _lib, hit_merge as M, replay, sharding
from waveformanalysis_amd.device import DeviceSession
from waveformanalysis_amd.dtypes import HIT_MERGE_CLUSTERS_DTYPE, THRESHOLD_HIT_DTYPE
from waveformanalysis_amd.event_grouping import group_hit_windows

pytestmark = pytest.mark.gpu

N_SHARDS = 8


def _run_shards(sess, shards, source, profile_names=None):
    """One pass + one exchange per shard; returns (rows of every shard as the root received them, total)."""
    got = []
    for sh in shards:
        sess.upload_pool(sh.wave_pool)
        sess.upload_records(sh.records, 10.0)
        if profile_names is not None:
            sess.profile(True)
        n = sess.threshold_hits(source, 2, 2, max_len=sh.max_len, download=False)
        if profile_names is not None:
            profile_names.update(sess.profile_report())
            sess.profile(False)
        counts, rows = sess.rccl_gather_rows(None, n, THRESHOLD_HIT_DTYPE, root=0, download=True)
        assert int(counts.sum()) == n == len(rows)
        got.append(rows)
    return got, sum(len(r) for r in got)


def _gathered_to_reference_index(got, shards):
    """Row j of the gathered (shard-major) table is row ref[j] of the reference's (record, start)-ordered table."""
    key = np.concatenate([sh.orig_index[np.searchsorted(sh.records["record_id"], rows["record_id"])]
                          for rows, sh in zip(got, shards)])
    perm = np.argsort(key, kind="stable")
    ref = np.empty(len(perm), dtype=np.int64)
    ref[perm] = np.arange(len(perm))
    return ref


def test_c4_eight_channel_shards_gather_merge_and_group_match_the_reference():
    d, rec, pool = load_fixture()
    assert len(np.unique(rec["board"].astype(np.int64) * 65536 + rec["channel"])) == 256
    shards = [sharding.make_shard(rec, pool, N_SHARDS, g) for g in range(N_SHARDS)]
    assert sum(len(s.records) for s in shards) == len(rec) and min(len(s.records) for s in shards) > 500
    assert all(np.all(np.diff(s.records["record_id"]) > 0) for s in shards)
    with DeviceSession(0) as sess:
        sess.rccl_init(0, 1, DeviceSession.rccl_unique_id())
        sess.rccl_gather_append(True)
        got, n = _run_shards(sess, shards, _lib.SRC_RAW)
        # rank order on the wire, reference order after the host merge: the reference's hit_threshold table
        hits = sharding.merge_rows(got, [s.orig_index for s in shards], [s.records for s in shards])
        G.assert_struct_equal(hits, d["hit_threshold"], float_rtol=1e-6, what="C4 hit_threshold (8 shards)")
        ref = _gathered_to_reference_index(got, shards)
        np.testing.assert_array_equal(np.concatenate(got), hits[ref])

        # hit merge straight from the gathered device table (no host columns)
        sess.hit_rows_source("gather")
        order, offset = sess.hit_merge_clusters_resident(n, 20.0, 10000.0)
        comp = d["hit_merged_components"]
        np.testing.assert_array_equal(ref[order], comp["hit_index"])
        np.testing.assert_array_equal(np.repeat(np.arange(len(offset) - 1), np.diff(offset)), comp["merged_index"])
        clusters = np.zeros(n, dtype=HIT_MERGE_CLUSTERS_DTYPE)
        clusters["cluster_index"], clusters["hit_index"] = comp["merged_index"], ref[order]
        merged = M.compute_merged_rows(sess, hits, clusters, None, "hit_merged")
        G.assert_struct_equal(merged, d["hit_merged"], float_rtol=1e-6, what="C4 hit_merged (gathered table)")

        # event grouping of the gathered threshold rows from the device buffer == grouping of the host-ordered table
        sess.hit_rows_source("gather")
        res = sess.group_hit_windows_resident(n, 100.0)
        want = sess.group_hit_windows(hits["timestamp"], hits["position"], hits["edge_start"], hits["edge_end"], hits["dt"],
                                      hits["board"], hits["channel"], hits["record_id"], 100.0)
        np.testing.assert_array_equal(ref[res["order"]], want["order"])
        for k in ("event_start", "t_min", "t_max"):
            np.testing.assert_array_equal(res[k], want[k], err_msg=k)
        assert len(res["t_min"]) > 100

        # the chain's endpoint: hit_grouped of the merged rows, against the reference's DataFrame
        df = group_hit_windows(merged, 100.0, component_rows=M.compute_component_rows(merged, clusters),
                               component_hits=hits, session=sess)
        for key, value in replay.flatten_grouped(df).items():
            if value.dtype.kind == "f":
                np.testing.assert_allclose(value, d[key], rtol=1e-6, err_msg=key)
            else:
                np.testing.assert_array_equal(value, d[key], err_msg=key)

        # a second table: append mode off and on again starts from nothing
        sess.rccl_gather_append(False)
        sess.rccl_gather_append(True)
        got2, n2 = _run_shards(sess, shards[:2], _lib.SRC_RAW)
        sess.hit_rows_source("gather")
        assert n2 == len(got[0]) + len(got[1])
        with pytest.raises(ValueError, match="resident hit table has"):
            sess.group_hit_windows_resident(n, 100.0)
        assert len(sess.group_hit_windows_resident(n2, 100.0)["order"]) == n2


def test_c4_shards_on_the_padded_streaming_route_match_the_unsharded_oracle():
    """Same shards, Savitzky-Golay fused hits: 1500-sample records take the padded shadow layout (stride 1504) and the
    streaming kernel k_sg_runs32; gathered + merged rows == oracle on the whole run."""
    _d, rec, pool = load_fixture()
    shards = [sharding.make_shard(rec, pool, N_SHARDS, g) for g in range(N_SHARDS)]
    want = O.threshold_hits(rec, O.filter_wave_pool(rec, pool))
    assert len(want) > 5000
    with DeviceSession(0) as sess:
        sess.set_sg_plan(11, 2)
        sess.rccl_init(0, 1, DeviceSession.rccl_unique_id())
        sess.rccl_gather_append(True)
        names = {}
        got, n = _run_shards(sess, shards, _lib.SRC_SG_FUSED, profile_names=names)
        assert "k_sg_runs32" in names and "k_pad_rows (once per upload)" in names, sorted(names)
        assert not any(k.startswith("k_sg_mask") for k in names), sorted(names)
        hits = sharding.merge_rows(got, [s.orig_index for s in shards], [s.records for s in shards])
        G.assert_struct_equal(hits, want, float_rtol=1e-6, what="C4 fused hits (8 shards, padded streaming route)")
        ref = _gathered_to_reference_index(got, shards)
        sess.hit_rows_source("gather")
        res = sess.group_hit_windows_resident(n, 100.0)
        host = sess.group_hit_windows(hits["timestamp"], hits["position"], hits["edge_start"], hits["edge_end"], hits["dt"],
                                      hits["board"], hits["channel"], hits["record_id"], 100.0)
        np.testing.assert_array_equal(ref[res["order"]], host["order"])
        np.testing.assert_array_equal(res["event_start"], host["event_start"])
