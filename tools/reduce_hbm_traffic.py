"""rocprofv3 counter_collection CSVs (FETCH_SIZE, WRITE_SIZE; KB per dispatch) -> HBM bytes per launch per kernel, keyed
the way bench.py looks them up: "<bench kernel label>|<preset>|<records>|<L>".

usage: reduce_hbm_traffic.py FETCH.csv WRITE.csv PRESET RECORDS L COMMIT [existing.json]
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half of wide coalesced reads).  Entries of
other workloads already in `existing.json` are kept; entries of this workload are replaced."""
import csv
import json
import sys
from collections import defaultdict

# substring of the kernel's C++ name -> the label bench.py / the C-ABI profile uses
NAMES = [
    ("k_sg_runs32<11, 40>", "k_sg_runs32<baseline>"), ("k_sg_runs32<11, 0>", "k_sg_runs32"),
    ("k_sg_mask_span16", "k_sg_mask_span16<baseline>"), ("k_runs_to_desc", "k_runs_to_desc"), ("k_hit_runs", "k_hit_runs"),
    ("k_hit_rows_grp", "k_hit_rows_grp"), ("k_hit_rows_flat", "k_hit_rows_flat"), ("k_hit_rows_literal", "k_hit_rows_literal"), ("k_savgol_span", "k_savgol_span"),
    ("k_features_leaf<0", "k_basic_features_leaf"), ("k_features_leaf<1", "k_width_integral_leaf"),
    ("k_features_leaf<2", "k_features_both_leaf"), ("k_width_ties", "k_width_ties"),
    ("k_find_peaks_hot", "k_find_peaks_hot"), ("k_find_peaks_staged", "k_find_peaks_staged"), ("k_find_peaks_slots", "k_find_peaks_slots"), ("k_peak_compact", "k_peak_compact"), ("k_peak_eval", "k_peak_eval"),
    ("k_peak_rows", "k_peak_rows"),
]


def per_kernel(path):
    acc = defaultdict(list)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            for key, name in NAMES:
                if key in row["Kernel_Name"]:
                    acc[name].append(float(row["Counter_Value"]) * 1024.0)
                    break
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest  # noqa: E402  (the digest bench.py compares an entry with)

fetch_csv, write_csv, preset, records, L, commit = sys.argv[1:7]
sha16 = csrc_digest()
table = {}
if len(sys.argv) > 7:
    try:
        table = {k: v for k, v in json.load(open(sys.argv[7])).items() if isinstance(v, dict) and "|" in k}
    except Exception:
        table = {}
fetch, n = per_kernel(fetch_csv)
write, _ = per_kernel(write_csv)
suffix = f"|{preset}|{int(records)}|{int(L)}"
table = {k: v for k, v in table.items() if not k.endswith(suffix)}
for k in sorted(fetch):
    f2, w = int(round(2 * fetch[k])), int(round(write.get(k, 0.0)))
    table[k + suffix] = {"bytes": f2 + w, "fetch_bytes": f2, "fetch_size_raw": int(round(fetch[k])), "write_bytes": w,
                         "launches_averaged": n[k], "commit": commit, "csrc_sha16": sha16,
                         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, python3 bench.py --steps 3 --warmup 1 "
                                   f"--no-cpu-baseline --preset {preset} --records {int(records)}"}
table["_note"] = ("HBM bytes per launch. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB); FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced streaming reads). bench.py prints roofline.traffic "
                  "only for an entry whose key matches its kernel, preset, record count and record length.")
print(json.dumps(table, indent=1))
