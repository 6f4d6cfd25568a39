"""Device stand-in for rehearsing bench.py's multi-rank control flow on CPU (WFA_BENCH_STUB=tests.bench_stub).

It computes nothing: hit rows are a fixed pattern, the "RCCL" gather goes over the gloo process group.  What the
rehearsal exercises is bench.py itself -- self-launch of the ranks, rendezvous on 127.0.0.1, barrier + max-over-ranks
timing, the gather watchdog, `gather_ok`, the exit status of every rank, the single JSON line."""

from __future__ import annotations

import os

import numpy as np

from waveformanalysis_amd.dtypes import THRESHOLD_HIT_DTYPE


class Session:
    def __init__(self, device_id: int = 0):
        self.device_id = device_id
        self.n = 0
        self.rank, self.n_ranks = 0, 1
        self._gathered = 0

    def upload_pool(self, pool):
        self.n = int(pool.size)

    def last_h2d_rate(self):
        return 0.0

    def upload_records(self, rec, thr):
        self.n_rec = len(rec)
        self._chan0 = int(rec["channel"][0]) if len(rec) else 0

    def set_sg_plan(self, w, p):
        return None

    def _hits(self):
        return self.n_rec // 2 + self.device_id

    def fused_baseline_filter_hits(self, window, le, re, download=True):
        return np.zeros(self._hits(), dtype=THRESHOLD_HIT_DTYPE) if download else self._hits()

    def hits_enqueue(self, *a, **k):
        return None

    def hits_wait(self):
        return self._hits()

    def sync(self):
        return None

    def profile(self, on=True):
        return None

    def profile_report(self):
        return {"k_sg_runs32<baseline>": (1.0, 1)}

    def _fill_hits(self, n):
        rows = np.zeros(n, dtype=THRESHOLD_HIT_DTYPE)
        rows["position"] = np.arange(n) + 1000 * self.device_id  # something a checksum can tell apart
        rows["record_id"] = np.arange(n)[::-1] + 7 * self.device_id
        rows["channel"] = (np.arange(n) * 2 + getattr(self, "_chan0", 0)) % 32
        return rows

    @staticmethod
    def rccl_unique_id():
        return b"stub" + bytes(124)

    def rccl_init(self, rank, n_ranks, uid):
        self.rank, self.n_ranks = rank, n_ranks

    def rccl_gather_rows(self, rows, n_rows, row_dtype, root=0, download=True):
        import torch
        import torch.distributed as dist

        if os.environ.get("WFA_BENCH_STUB_FAIL_GATHER") and self.rank == self.n_ranks - 1:
            raise RuntimeError("stub: transport failure injected on the last rank")
        t = [torch.zeros(1, dtype=torch.int64) for _ in range(self.n_ranks)]
        dist.all_gather(t, torch.tensor([n_rows], dtype=torch.int64))
        counts = np.array([int(x.item()) for x in t], dtype=np.int64)
        self._gathered = int(counts.sum())
        if not download:
            return counts, None
        mine = self._fill_hits(n_rows)
        if os.environ.get("WFA_BENCH_STUB_CORRUPT_C4") and self.rank == self.n_ranks - 1 and len(mine):
            mine["position"][0] += 1                              # what arrives is not what the rank digested
        payload = [None] * self.n_ranks
        dist.all_gather_object(payload, mine.tobytes())
        table = np.concatenate([np.frombuffer(b, dtype=row_dtype) for b in payload]) if self.rank == root else None
        return counts, table

    def hit_rows_source(self, which):
        return None

    def group_hit_windows_resident(self, n, tw):
        assert n == self._gathered
        return {"event_start": np.zeros(n // 3 + 1, np.int64)}

    def close(self):
        return None
