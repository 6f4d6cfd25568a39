#!/usr/bin/env python3
"""Streaming kernel time against the records per span (option `span_records`) on the bench chunk."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from waveformanalysis_amd import synth  # noqa: E402
from waveformanalysis_amd.device import DeviceSession  # noqa: E402

n = 1_250_000
preset = sys.argv[1] if len(sys.argv) > 1 else "v1725"
sizes = [int(x) for x in sys.argv[2:]] or [64, 62, 60, 59, 56, 52, 51, 48, 45]
rec, pool = synth.make_run(n if preset == "v1725" else n * 800 // 1500, preset, cfg=1)
rec["baseline"] = np.nan
out = {}
with DeviceSession(0) as sess:
    sess.upload_pool(pool)
    sess.set_sg_plan(11, 2)
    ref = None
    for rs in sizes + sizes[:1]:
        sess.set_option("span_records", rs)
        sess.upload_records(rec, 10.0)
        sess.fused_baseline_filter_hits((0, 40), 2, 2, download=False)
        sess.profile(True)
        for _ in range(8):
            nh = sess.fused_baseline_filter_hits((0, 40), 2, 2, download=False)
        rep = sess.profile_report()
        k = [v for name, v in rep.items() if name.startswith("k_sg_runs32")][0]
        out.setdefault(rs, []).append(round(k[0] / k[1], 4))
        tot = round(sum(v[0] / v[1] for v in rep.values()), 4)
        out[rs].append(tot)
print(json.dumps(out))
