"""Where find_peaks spends its time at the bench size: scan only (no candidates) vs default options."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession

rec, pool = synth.make_run(1250000 // 4, "v1725", cfg=1)
s = DeviceSession(0)
s.upload_pool(pool)
s.upload_records(rec, 10.0)
s.set_sg_plan(11, 2)
s.savgol(download=False)
s.profile(True)
for name, kw in (("scan only (height 1e9)", dict(height=1e9)), ("defaults", dict()),
                 ("plain signal, height 40", dict(use_derivative=False, height=40.0, width=3, prominence=5.0))):
    s.profile_reset() if hasattr(s, "profile_reset") else None
    t0 = time.perf_counter()
    out = s.find_peaks(_lib.SRC_F32, **kw)
    dt = time.perf_counter() - t0
    print(name, len(out), "peaks", round(dt * 1e3, 2), "ms wall", {k: round(v[0] / max(v[1], 1), 3) for k, v in s.profile_report().items() if "peaks" in k})
