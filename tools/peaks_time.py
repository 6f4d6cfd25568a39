#!/usr/bin/env python3
"""Kernel times of the find_peaks hit detector on the bench chunk (1.25e6 x 800 samples, filtered float32 pool, reference
default options), per route: default, `no_peak_hot` (plateau machine over every sample)."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from waveformanalysis_amd import _lib, synth  # noqa: E402
from waveformanalysis_amd.device import DeviceSession  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
rec, pool = synth.make_run(n, "v1725", cfg=1)
out = {}
with DeviceSession(0) as sess:
    sess.upload_pool(pool)
    sess.set_sg_plan(11, 2)
    sess.upload_records(rec, 10.0)
    sess.savgol(download=False)
    ref = None
    for route in ("default", "no_peak_hot"):
        sess.set_option("no_peak_hot", route == "no_peak_hot")
        sess.find_peaks(_lib.SRC_F32)
        sess.profile(True)
        for _ in range(3):
            rows = sess.find_peaks(_lib.SRC_F32)
        rep = sess.profile_report()
        out[route] = {k: round(v[0] / v[1], 4) for k, v in rep.items()}
        out[route]["total_ms"] = round(sum(v[0] / v[1] for v in rep.values()), 4)
        out[route]["rows"] = len(rows)
        if ref is None:
            ref = rows.tobytes()
        else:
            out["same_bytes"] = ref == rows.tobytes()
print(json.dumps(out, indent=1))
