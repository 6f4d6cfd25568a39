#!/usr/bin/env python3
"""Aggregate pass rate of S sessions (= HIP streams) queuing passes on one GPU at the same time, each on its own resident
chunk: does the rows kernel of one pass hide under the streaming kernel of another?"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from waveformanalysis_amd import _lib, synth  # noqa: E402
from waveformanalysis_amd.device import DeviceSession  # noqa: E402

n = 1_250_000
rec, pool = synth.make_run(n, "v1725", cfg=100)
rec["baseline"] = np.nan
K = 200
for S in (1, 2, 3, 1, 2):
    sessions = [DeviceSession(0) for _ in range(S)]
    for s in sessions:
        s.upload_pool(pool)
        s.upload_records(rec, 10.0)
        s.set_sg_plan(11, 2)

    def run(s, k):
        for _ in range(k):
            s.hits_enqueue(_lib.SRC_SG_FUSED, (0, synth.BASELINE_SAMPLES), 2, 2)

    def all_run(k):
        th = [threading.Thread(target=run, args=(s, k)) for s in sessions]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return [s.hits_wait() for s in sessions]

    all_run(150)
    t0 = time.perf_counter()
    hits = all_run(K)
    dt = time.perf_counter() - t0
    print(f"sessions {S}: {dt / (S * K) * 1e3:.4f} ms per pass aggregate, {S * K * n * 800 / dt / 1e9:.0f} Gsamples/s, hits {hits}", flush=True)
    for s in sessions:
        s.close()
