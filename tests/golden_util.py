"""Helpers shared by the oracle and HIP parity tests: load golden fixtures, resolve options."""

from __future__ import annotations

import glob
import json
import os

import numpy as np

from waveformanalysis_amd.channel_config import per_record_option, scatter_per_record

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names(prefix_exclude=("grouping_", "peaks_", "dense_", "merge_", "sort_", "legacy_", "chunk_", "v1725bin_", "densehit_", "sigpeaks_", "vx2730csv_", "c5_")):
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    return [n for n in names if not n.startswith(tuple(prefix_exclude))]


def grouping_case_names():
    return [n for n in case_names(prefix_exclude=()) if n.startswith("grouping_")]


def peaks_case_names():
    return [n for n in case_names(prefix_exclude=()) if n.startswith("peaks_")]


def load_peaks(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["configs"] = json.loads(bytes(d.pop("options_json")).decode())
    return d


def dense_case_names():
    return sorted(os.path.splitext(f)[0] for f in os.listdir(GOLDEN) if f.startswith("dense_") and f.endswith(".npz"))


def load_dense(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    opt = json.loads(bytes(d.pop("options_json")).decode())
    for cfg in opt["s1s2"]:
        for k, v in cfg.items():
            if isinstance(v, list):
                cfg[k] = tuple(v)
    d["options"] = opt
    return d


def densehit_case_names():
    return sorted(os.path.splitext(f)[0] for f in os.listdir(GOLDEN) if f.startswith("densehit_") and f.endswith(".npz"))


def load_densehit(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["options"] = json.loads(bytes(d.pop("options_json")).decode())
    return d


DENSEHIT_SOURCES = {  # tag -> (fixture array, plugin options selecting it)
    "st": ("st_waveforms", {}),
    "filt": ("filtered_waveforms", {"use_filtered": True}),
    "pad": ("st_pad", {}),
    "filtpad": ("filtered_waveforms", {"wave_source": "filtered_waveforms"}),
}


def densehit_array(case, tag):
    """The dense input of one source tag (the padded variants carry event_length 500 on every third row from 1)."""
    arr = case[DENSEHIT_SOURCES[tag][0]].copy()
    if tag == "filtpad":
        arr["event_length"][1::3] = 500
    return arr


def densehit_record_lengths(case, tag, arr):
    """records/wave_pool event_length of each dense row, looked up by record_id (hit_finder.py:257-286)."""
    lookup = dict(zip(case[f"recid_{tag}"].tolist(), case[f"reclen_{tag}"].tolist()))
    return np.array([lookup[int(r)] for r in arr["record_id"]], dtype=np.int64)


def sigpeaks_case_names():
    return sorted(os.path.splitext(f)[0] for f in os.listdir(GOLDEN) if f.startswith("sigpeaks_") and f.endswith(".npz"))


def load_sigpeaks(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["configs"] = json.loads(bytes(d.pop("options_json")).decode())
    return d


def load_vx2730csv(name="vx2730csv_files"):
    """-> (per-channel lists of (file name, text bytes), variants, fixture dict)."""
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    opt = json.loads(bytes(d.pop("options_json")).decode())
    key = lambda f: "text_" + f.replace("@", "_at_").replace(".", "_dot_")  # noqa: E731
    groups = [[(f, bytes(d[key(f)])) for f in group] for group in opt["files"]]
    for v in opt["variants"]:
        if isinstance(v.get("baseline_samples"), list):
            v["baseline_samples"] = tuple(v["baseline_samples"])
    return groups, opt["variants"], d


def merge_case_names():
    return sorted(os.path.splitext(f)[0] for f in os.listdir(GOLDEN) if f.startswith("merge_") and f.endswith(".npz"))


def load_merge(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["configs"] = json.loads(bytes(d.pop("options_json")).decode())
    return d


def load_grouping(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    data = {k: z[k] for k in z.files}
    data["options"] = json.loads(bytes(data.pop("options_json")).decode())
    return data


def hit_params(case):
    """Per-record thresholds + extensions from the stored ThresholdHitPlugin options."""
    opt = case["options"]["hit"]
    rec = case["records"]
    thr = float(opt.get("threshold", 10.0))
    cc = _literal(opt.get("channel_config"))
    per = per_record_option(rec["board"], rec["channel"], cc, "run", {"threshold": thr})
    thresholds = scatter_per_record(rec["board"], rec["channel"],
                                    {k: float(v.get("threshold", thr)) for k, v in per.items()}, thr)
    return dict(thresholds=thresholds, left_extension=int(opt.get("left_extension", 2)),
                right_extension=int(opt.get("right_extension", 2)))


def bf_params(case):
    opt = case["options"]["bf"]
    rec = case["records"]
    cc = _literal(opt.get("channel_config"))
    per = per_record_option(rec["board"], rec["channel"], cc, "run", {"fixed_baseline": None})
    fixed = scatter_per_record(
        rec["board"], rec["channel"],
        {k: (np.nan if v.get("fixed_baseline") is None else float(v["fixed_baseline"])) for k, v in per.items()},
        np.nan)
    hr = opt.get("height_range", (40, 90))
    ar = opt.get("area_range", (0, None))
    return dict(height_range=tuple(hr), area_range=tuple(ar), fixed_baseline=fixed)


def wi_params(case):
    opt = case["options"]["wi"]
    return dict(q_low=float(opt.get("q_low", 0.10)), q_high=float(opt.get("q_high", 0.90)),
                dt=opt.get("dt"), sampling_rate=float(opt.get("sampling_rate", 0.5)))


def filter_params(case):
    opt = case["options"]["filter"]
    return dict(filter_type=opt.get("filter_type", "SG"), lowcut=float(opt.get("lowcut", 0.1)),
                highcut=float(opt.get("highcut", 0.5)), fs=float(opt.get("fs", 0.5)),
                filter_order=int(opt.get("filter_order", 4)),
                sg_window_size=int(opt.get("sg_window_size", 11)),
                sg_poly_order=int(opt.get("sg_poly_order", 2)))


def _literal(value):
    """options_json stores dicts natively; json turns tuple keys into strings already."""
    return value


def assert_struct_equal(got, want, float_rtol=0.0, float_atol=0.0, what=""):
    """Field-by-field comparison: integer fields exact, float fields exact unless a tolerance is given."""
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} != {want.dtype}"
    assert len(got) == len(want), f"{what}: {len(got)} rows != {len(want)}"
    for name in want.dtype.names:
        g, w = got[name], want[name]
        if w.dtype.kind == "f" and (float_rtol or float_atol):
            np.testing.assert_allclose(g, w, rtol=float_rtol, atol=float_atol, equal_nan=True,
                                       err_msg=f"{what}: field {name}")
        else:
            np.testing.assert_array_equal(g, w, err_msg=f"{what}: field {name}")
