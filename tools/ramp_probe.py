#!/usr/bin/env python3
"""Does the streaming kernel get faster as a run goes on?  Blocks of K enqueued passes, each block synchronised and its
average kernel time read from the HIP-event profile, for a few block sizes."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from waveformanalysis_amd import _lib, synth  # noqa: E402
from waveformanalysis_amd.device import DeviceSession  # noqa: E402

rec, pool = synth.make_run(1_250_000, "v1725", cfg=100)
rec["baseline"] = np.nan
with DeviceSession(0) as sess:
    sess.upload_pool(pool)
    sess.upload_records(rec, 10.0)
    sess.set_sg_plan(11, 2)
    for K in (20, 20, 100, 20, 400, 20):
        out = []
        for b in range(6 if K <= 100 else 2):
            sess.profile(2)
            sess.sync()
            t0 = time.perf_counter()
            for _ in range(K):
                sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, synth.BASELINE_SAMPLES), 2, 2)
            sess.sync()
            wall = (time.perf_counter() - t0) / K * 1e3
            rep = sess.profile_report()
            k = [v for n, v in rep.items() if n.startswith("k_sg_runs32")][0]
            out.append((round(k[0] / k[1], 4), round(wall, 4)))
        print("K", K, out, flush=True)
        time.sleep(0.5)
