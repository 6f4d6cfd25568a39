"""Host-side profile of HipThresholdHitPlugin.compute on the bench chunk (where the end-to-end time goes)."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from waveformanalysis_amd import synth
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import HipThresholdHitPlugin

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
rec, pool = synth.make_run(n, "v1725")
rec_in = rec.copy()
rec_in["baseline"] = np.nan
cfg = {"wave_source": "records", "use_filtered": True, "fuse_filter": True, "fuse_baseline": (0, 40), "threshold": 10.0}
plugin = HipThresholdHitPlugin()
for k in range(3):
    ctx = SimpleContext(cfg, {"records": rec_in, "wave_pool": pool.copy()})
    ctx.wfa_device_pool = None
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    rows = plugin.compute(ctx, "bench")
    pr.disable()
    print(f"call {k}: {time.perf_counter() - t0:.4f} s, {len(rows)} rows")
    if k == 2:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
