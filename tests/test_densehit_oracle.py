"""Oracle restatements of the dense (st_waveforms / filtered_waveforms) branches of hit_threshold,
waveform_width_integral and hit vs the reference plugins' outputs (tests/golden/densehit_*.npz)."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd.channel_config import per_record_option, scatter_per_record


def _thresholds(arr, cfg):
    thr = float(cfg.get("threshold", 10.0))
    per = per_record_option(arr["board"], arr["channel"], cfg.get("channel_config"), "run", {"threshold": thr})
    return scatter_per_record(arr["board"], arr["channel"],
                              {k: float(v.get("threshold", thr)) for k, v in per.items()}, thr)


@pytest.mark.parametrize("name", G.densehit_case_names())
@pytest.mark.parametrize("tag", list(G.DENSEHIT_SOURCES))
def test_threshold_hits_dense(name, tag):
    case = G.load_densehit(name)
    arr = G.densehit_array(case, tag)
    for k, cfg in enumerate(case["options"]["hit"]):
        got = O.threshold_hits_dense(arr, G.densehit_record_lengths(case, tag, arr), thresholds=_thresholds(arr, cfg),
                                     left_extension=cfg.get("left_extension", 2),
                                     right_extension=cfg.get("right_extension", 2))
        G.assert_struct_equal(got, case[f"hits_{tag}_{k}"], what=f"{name} hits {tag} {k}")


@pytest.mark.parametrize("name", G.densehit_case_names())
@pytest.mark.parametrize("tag", ["st", "filt"])
def test_width_integral_dense(name, tag):
    case = G.load_densehit(name)
    arr = G.densehit_array(case, tag)
    for k, cfg in enumerate(case["options"]["wi"]):
        got = O.width_integral_dense(arr, q_low=cfg.get("q_low", 0.1), q_high=cfg.get("q_high", 0.9), dt=cfg.get("dt"))
        G.assert_struct_equal(got, case[f"wi_{tag}_{k}"], what=f"{name} wi {tag} {k}")


@pytest.mark.parametrize("name", G.densehit_case_names())
@pytest.mark.parametrize("tag", list(G.DENSEHIT_SOURCES))
def test_find_peaks_dense(name, tag):
    case = G.load_densehit(name)
    arr = G.densehit_array(case, tag)
    n = 0
    for k, cfg in enumerate(case["options"]["peak"]):
        if f"peak_{tag}_{k}" not in case:
            continue
        kw = {a: b for a, b in cfg.items() if a != "use_filtered"}
        got = O.find_peak_hits_dense(arr, **kw)
        G.assert_struct_equal(got, case[f"peak_{tag}_{k}"], what=f"{name} peak {tag} {k}")
        n += len(got)
    assert n > 0
