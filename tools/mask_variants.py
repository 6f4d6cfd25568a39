"""Mask kernel with and without the in-kernel baseline estimate, same box, same records (baseline precomputed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession

rec, pool = synth.make_run(1250000, "v1725", cfg=1)
s = DeviceSession(0)
s.upload_pool(pool)
s.upload_records(rec, 10.0)
s.set_sg_plan(11, 2)
for rep in range(2):
    s.profile(True)
    for _ in range(10):
        n1 = s.threshold_hits(_lib.SRC_SG_FUSED, 2, 2, download=False) if "download" in s.threshold_hits.__code__.co_varnames else len(s.threshold_hits(_lib.SRC_SG_FUSED, 2, 2))
    r1 = {k: round(v[0] / max(v[1], 1), 4) for k, v in s.profile_report().items()}
    s.profile(True)
    for _ in range(10):
        s.fused_baseline_filter_hits((0, 40), 2, 2, download=False)
    r2 = {k: round(v[0] / max(v[1], 1), 4) for k, v in s.profile_report().items()}
    print("records.baseline :", r1)
    print("fused baseline   :", r2)
