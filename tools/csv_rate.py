"""Decode rate of the VX2730 CSV kernels on synthetic text (rows of 1500 samples), next to pyarrow on the host."""
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waveformanalysis_amd.device import DeviceSession  # noqa: E402

n_rows, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 1500
rng = np.random.default_rng(0)
w = np.clip(8000 + np.round(rng.normal(0, 3, (n_rows, L))), 0, 16383).astype(np.int64)
ts = np.sort(rng.integers(0, 10**15, n_rows))
t0 = time.time()
text = "".join(f"0;{i % 8};{int(t)};0;0;0x4000;1;" + ";".join(map(str, row)) + "\n" for i, (t, row) in enumerate(zip(ts, w.tolist()))).encode()
print(f"text: {len(text) / 1e6:.1f} MB, {n_rows} rows x {L} samples (built in {time.time() - t0:.1f} s)")
with DeviceSession(0) as sess:
    sess.csv_decode(text, ";", 7, (0, 1, 2))
    sess.profile(True)
    t0 = time.time()
    for _ in range(3):
        d = sess.csv_decode(text, ";", 7, (0, 1, 2))
    wall = (time.time() - t0) / 3
    rep = sess.profile_report()
    for k, (ms, n) in rep.items():
        print(f"  {k}: {ms / n:.3f} ms per call")
    kern = sum(ms / n for ms, n in rep.values())
    print(f"GPU kernels: {kern:.2f} ms -> {len(text) / kern / 1e6:.1f} GB/s of text, {n_rows * L / kern / 1e6:.2f} Gsamples/s; "
          f"call incl. H2D of the text and D2H of the row tables: {wall * 1e3:.1f} ms")
    assert d["n_samples"] == n_rows * L
try:
    import pyarrow as pa
    import pyarrow.csv as pc

    types = {f"f{i}": pa.int64() for i in range(7)}
    types["f5"] = pa.string()
    types.update({f"f{i}": pa.int16() for i in range(7, 7 + L)})
    t0 = time.time()
    tab = pc.read_csv(io.BytesIO(text), read_options=pc.ReadOptions(autogenerate_column_names=True),
                      parse_options=pc.ParseOptions(delimiter=";"), convert_options=pc.ConvertOptions(column_types=types))
    dt = time.time() - t0
    print(f"pyarrow.csv.read_csv (all host cores): {dt * 1e3:.0f} ms -> {len(text) / dt / 1e9:.2f} GB/s, {tab.num_rows} rows")
except Exception as exc:  # noqa: BLE001
    print("pyarrow comparison skipped:", exc)
