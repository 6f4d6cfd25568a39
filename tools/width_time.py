"""Time k_waveform_width on dense rows (int16 and float32) with the detector's own hits."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveformanalysis_amd import synth, dense, _lib
from waveformanalysis_amd.device import DeviceSession

n, L = 125000, 800
rec, pool = synth.make_run(n, "v1725", cfg=7)
pool = (16383 - pool.astype(np.int32)).clip(0, 16383).astype(np.uint16)   # positive pulses: the width plugin wants them
rng = np.random.default_rng(1)
rows = pool.reshape(n, L)
pos = rows.argmax(axis=1).astype(np.int64)
hits_pos = np.concatenate([pos, rng.integers(0, L, n)])
hits_row = np.concatenate([np.arange(n), np.arange(n)]).astype(np.int64)
with DeviceSession(0) as s:
    for name, p, src in (("int16", pool, _lib.SRC_RAW), ("float32", pool.astype(np.float32), _lib.SRC_F32)):
        s.upload_pool(p)
        s.waveform_width(src, hits_pos, hits_row, n, L)
        s.profile(True)
        out = s.waveform_width(src, hits_pos, hits_row, n, L)
        rep = s.profile_report()
        s.profile(False)
        print(name, {k: round(v[0] / v[1], 3) for k, v in rep.items()}, "hits", len(hits_pos), "valid", int(np.count_nonzero(out[1])) if isinstance(out, tuple) else len(out))
