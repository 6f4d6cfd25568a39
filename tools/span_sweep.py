#!/usr/bin/env python3
"""Streaming kernel and step time against session options on the bench chunk, at the sustained clock (150 queued passes in
front of every measurement, alternating settings):  python tools/span_sweep.py [preset] [name=value ...]
default settings: span_records = 64 .. 45; e.g. `no_deposit=1`, `rows_grouped=1`."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from waveformanalysis_amd import _lib, synth  # noqa: E402
from waveformanalysis_amd.device import DeviceSession  # noqa: E402

n = 1_250_000
preset = sys.argv[1] if len(sys.argv) > 1 else "v1725"
settings = [tuple(a.split("=")) for a in sys.argv[2:]] or [("span_records", str(v)) for v in (64, 60, 56, 51, 48)]
rec, pool = synth.make_run(n if preset == "v1725" else n * 800 // 1500, preset, cfg=1)
rec["baseline"] = np.nan
out = {}
with DeviceSession(0) as sess:
    sess.upload_pool(pool)
    sess.set_sg_plan(11, 2)

    def measure():
        sess.upload_records(rec, 10.0)
        for _ in range(150):
            sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, synth.BASELINE_SAMPLES), 2, 2)
        sess.hits_wait()
        sess.profile(2)
        t0 = time.perf_counter()
        for _ in range(40):
            sess.hits_enqueue(_lib.SRC_SG_FUSED, (0, synth.BASELINE_SAMPLES), 2, 2)
        sess.hits_wait()
        wall = (time.perf_counter() - t0) / 40 * 1e3
        rep = sess.profile_report()
        k = [v for name, v in rep.items() if name.startswith("k_sg_runs32")]
        return (round(k[0][0] / k[0][1], 4) if k else None, round(wall, 4))

    for rep_i in range(3):
        out.setdefault("default", []).append(measure())
        for name, val in settings:
            sess.set_option(name, int(val))
            out.setdefault(f"{name}={val}", []).append(measure())
            sess.set_option(name, 0)
print(json.dumps(out, indent=0))
