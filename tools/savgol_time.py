import sys, numpy as np
sys.path.insert(0, ".")
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession
rec, pool = synth.make_run(1_250_000, "v1725", cfg=1)
with DeviceSession(0) as s:
    s.upload_pool(pool); s.set_sg_plan(11, 2); s.upload_records(rec, 10.0)
    for _ in range(30): s.savgol(download=False)
    s.profile(True)
    for _ in range(20): s.savgol(download=False)
    print({k: round(v[0]/v[1],4) for k,v in s.profile_report().items()})
