"""N > 1 host logic on CPU: channel sharding, gloo gather (world_size 2), re-assembly in reference order.
The per-shard compute is the oracle here (no GPU in this container); on GPUs the same flow runs with
DeviceSession + RCCL (tests/test_hip_parity.py::test_sharded_equals_unsharded)."""

import os
import socket

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from tests.dist_util import gather_rows_torch
from waveformanalysis_amd import sharding, synth
from waveformanalysis_amd.streaming import Chunk, records_to_chunks


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_records, out_path):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec, pool = synth.make_run(n_records, "v1725", cfg=5, threads=1)
    shard = sharding.make_shard(rec, pool, world, rank)
    hits = O.threshold_hits_chunked(shard.records, shard.wave_pool)  # stand-in for the HIP pass
    gathered = gather_rows_torch(hits, root=0)
    if rank == 0:
        shards = [sharding.make_shard(rec, pool, world, r) for r in range(world)]
        merged = sharding.merge_rows(gathered, [s.orig_index for s in shards], [s.records for s in shards])
        np.save(out_path, merged)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp

    n_records, world = 96, 2
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_records, out), nprocs=world, join=True)
    merged = np.load(out)
    rec, pool = synth.make_run(n_records, "v1725", cfg=5, threads=1)
    G.assert_struct_equal(merged, O.threshold_hits_chunked(rec, pool), what="2-rank gather")


def test_shards_partition_records_and_samples():
    rec, pool = synth.make_run(200, "v1725", cfg=6, threads=1)
    shards = [sharding.make_shard(rec, pool, 4, r) for r in range(4)]
    all_idx = np.sort(np.concatenate([s.orig_index for s in shards]))
    np.testing.assert_array_equal(all_idx, np.arange(len(rec)))
    for s in shards:
        assert len(np.unique(s.records["channel"])) == 4  # 16 channels dealt to 4 shards
        assert s.max_len == 800 and len(s.wave_pool) == 800 * len(s.records)
        i = len(s.records) // 2
        o = int(rec["wave_offset"][s.orig_index[i]])
        np.testing.assert_array_equal(s.wave_pool[800 * i : 800 * i + 800], pool[o : o + 800])


def test_ragged_shard_and_merge_order():
    case = G.load_case("ragged_mixed")
    rec, pool = case["records"], case["wave_pool"]
    shards = [sharding.make_shard(rec, pool, 3, r) for r in range(3)]
    rows = [O.threshold_hits(s.records, s.wave_pool) for s in shards]
    # padded-width semantics need the run's max length, which a shard alone does not know:
    assert all(s.max_len == int(rec["event_length"].max()) for s in shards)
    merged = sharding.merge_rows(rows, [s.orig_index for s in shards], [s.records for s in shards])
    assert np.all(np.diff(merged["record_id"]) >= 0)
    assert len(merged) == sum(len(r) for r in rows)


def test_chunk_contract():
    rec, _pool = synth.make_run(100, "v1725", cfg=7, threads=1)
    chunks = records_to_chunks(rec, 32, run_id="run")
    assert [len(c) for c in chunks] == [32, 32, 32, 4]
    assert all(c.start <= int(c.data["timestamp"].min()) and int(c.data["timestamp"].max()) < c.end for c in chunks)
    with pytest.raises(ValueError):
        Chunk(rec[:4], int(rec["timestamp"][3]), int(rec["timestamp"][3]) + 1)
    with pytest.raises(ValueError):
        Chunk(rec[:4], 10, 5)
