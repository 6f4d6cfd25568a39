"""Fused feature kernel (wfa_features_both) against the two separate kernels on the bench chunk: identical rows, times."""
import sys
import numpy as np
sys.path.insert(0, ".")
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
rec, pool = synth.make_run(n, "v1725")
with DeviceSession(0) as s:
    s.upload_pool(pool)
    s.upload_records(rec, 10.0)
    for rep in range(2):
        s.profile(True)
        b = s.basic_features(_lib.SRC_RAW, (40, 90), (0, None))
        w = s.width_integral(_lib.SRC_RAW, 0.1, 0.9, 4.0)
        fb, fw = s.features_both((40, 90), (0, None), 0.1, 0.9, 4.0)
        rep_ms = {k: round(v[0] / v[1], 4) for k, v in s.profile_report().items()}
    print(rep_ms)
    print("identical:", all(np.array_equal(b[f], fb[f]) for f in b.dtype.names), all(np.array_equal(w[f], fw[f]) for f in w.dtype.names))
