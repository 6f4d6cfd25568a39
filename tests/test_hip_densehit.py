"""Dense (wave_source auto / st_waveforms / filtered_waveforms) branches of hit_threshold, waveform_width_integral
and hit on the GPU against fixtures produced by the reference's plugins (tests/golden/densehit_*.npz).
Integer fields exact; float fields exact except the threshold-hit height/integral (float64 sums rounded to
float32: 1e-6 relative, the tolerance of the records branch)."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import (
    HipHitFinderPlugin,
    HipThresholdHitPlugin,
    HipWaveformWidthIntegralPlugin,
)

pytestmark = pytest.mark.gpu


def _records_table(case, tag):
    rec = np.zeros(len(case[f"recid_{tag}"]), dtype=[("record_id", "i8"), ("event_length", "i4"), ("wave_offset", "i8")])
    rec["record_id"], rec["event_length"] = case[f"recid_{tag}"], case[f"reclen_{tag}"]
    return rec


def _data(case, tag):
    arr = G.densehit_array(case, tag)
    return {"st_waveforms": arr, "filtered_waveforms": arr, "records": _records_table(case, tag)}


@pytest.mark.parametrize("name", G.densehit_case_names())
@pytest.mark.parametrize("tag", list(G.DENSEHIT_SOURCES))
def test_threshold_hits_dense(name, tag):
    case = G.load_densehit(name)
    base = G.DENSEHIT_SOURCES[tag][1]
    for k, cfg in enumerate(case["options"]["hit"]):
        ctx = SimpleContext({"hit_threshold": {**base, **cfg}}, _data(case, tag), plugins=[HipThresholdHitPlugin()])
        got = ctx.get_data("run", "hit_threshold")
        G.assert_struct_equal(got, case[f"hits_{tag}_{k}"], float_rtol=1e-6, what=f"{name} hits {tag} {k}")


@pytest.mark.parametrize("name", G.densehit_case_names())
@pytest.mark.parametrize("tag", ["st", "filt"])
def test_width_integral_dense(name, tag):
    case = G.load_densehit(name)
    base = G.DENSEHIT_SOURCES[tag][1]
    for k, cfg in enumerate(case["options"]["wi"]):
        ctx = SimpleContext({"waveform_width_integral": {**base, **cfg}}, _data(case, tag),
                            plugins=[HipWaveformWidthIntegralPlugin()])
        G.assert_struct_equal(ctx.get_data("run", "waveform_width_integral"), case[f"wi_{tag}_{k}"],
                              what=f"{name} wi {tag} {k}")


@pytest.mark.parametrize("name", G.densehit_case_names())
@pytest.mark.parametrize("tag", list(G.DENSEHIT_SOURCES))
def test_find_peaks_dense(name, tag):
    case = G.load_densehit(name)
    base = G.DENSEHIT_SOURCES[tag][1]
    for k, cfg in enumerate(case["options"]["peak"]):
        if f"peak_{tag}_{k}" not in case:
            continue
        ctx = SimpleContext({"hit": {"use_filtered": False, **base, **cfg}}, _data(case, tag),
                            plugins=[HipHitFinderPlugin()])
        G.assert_struct_equal(ctx.get_data("run", "hit"), case[f"peak_{tag}_{k}"], what=f"{name} peak {tag} {k}")


def test_dense_rows_larger_than_fixture():
    """Random rows (int16 and float32) against the oracle restatement, incl. rows longer than one pairwise leaf."""
    rng = np.random.default_rng(5)
    n, L = 300, 1200
    from waveformanalysis_amd.dtypes import create_record_dtype

    st = np.zeros(n, dtype=create_record_dtype(L))
    wave = 8000 + np.round(rng.normal(0, 3, (n, L)))
    for i in range(n):
        for _ in range(rng.integers(0, 4)):
            t0 = int(rng.integers(50, L - 300))
            t = np.arange(L - t0)
            wave[i, t0:] -= rng.uniform(30, 800) * (np.exp(-t / rng.uniform(10, 120)) - np.exp(-t / 4.0))
    st["wave"] = np.clip(wave, 0, 16383).astype(np.int16)
    st["baseline"] = st["wave"][:, :40].mean(axis=1)
    st["timestamp"] = np.cumsum(rng.integers(10**6, 10**7, n))
    st["record_id"] = rng.permutation(n)
    st["dt"], st["event_length"], st["channel"] = 4, L, rng.integers(0, 8, n)
    st["polarity"] = "unknown"
    records = np.zeros(n, dtype=[("record_id", "i8"), ("event_length", "i4"), ("wave_offset", "i8")])
    records["record_id"], records["event_length"] = st["record_id"], L
    data = {"st_waveforms": st, "records": records}
    got = SimpleContext({}, data, plugins=[HipThresholdHitPlugin()]).get_data("run", "hit_threshold")
    G.assert_struct_equal(got, O.threshold_hits_dense(st, np.full(n, L)), float_rtol=1e-6)
    got = SimpleContext({}, data, plugins=[HipWaveformWidthIntegralPlugin()]).get_data("run", "waveform_width_integral")
    G.assert_struct_equal(got, O.width_integral_dense(st))
    for cfg in ({"height": 10.0, "width": 2, "distance": 1}, {"use_derivative": False, "height": 25.0, "width": 6,
                                                            "prominence": 4.0, "height_method": "diff", "distance": 1}):
        got = SimpleContext({"hit": {"use_filtered": False, **cfg}}, data, plugins=[HipHitFinderPlugin()]).get_data("run", "hit")
        want = O.find_peak_hits_dense(st, **cfg)
        assert len(want) > 50
        G.assert_struct_equal(got, want)


def test_dense_errors():
    case = G.load_densehit(G.densehit_case_names()[0])
    data = _data(case, "st")
    short = dict(data)
    short["records"] = data["records"][:-3]
    with pytest.raises(RuntimeError, match="could not resolve record_id"):
        SimpleContext({}, short, plugins=[HipThresholdHitPlugin()]).get_data("run", "hit_threshold")
    bad = dict(data)
    bad["records"] = data["records"].copy()
    bad["records"]["event_length"][0] = 12
    with pytest.raises(RuntimeError, match="does not match records/wave_pool length"):
        SimpleContext({}, bad, plugins=[HipThresholdHitPlugin()]).get_data("run", "hit_threshold")
