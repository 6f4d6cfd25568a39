"""CPU-side checks of the boundary: the C-ABI library loads and exports every declared symbol,
host-side plan / option logic.  No compute call is made (there is no GPU here)."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from waveformanalysis_amd import _lib, sg_plan
from waveformanalysis_amd.channel_config import resolve_channel_values
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import hip_default

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "wfa_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void)\s+(wfa_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().wfa_abi_version() == _lib.ABI_VERSION


def test_errors_are_reported_without_gpu():
    lib = _lib.load()
    n = C.c_int(-1)
    rc = lib.wfa_device_count(C.byref(n))
    if rc != 0:  # CPU container
        assert n.value == 0
        assert "hipGetDeviceCount" in _lib.last_error()
        with pytest.raises(_lib.WfaError):
            _lib.check(rc)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libwfa_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_sg_plan_matches_scipy_tables():
    from scipy.signal import savgol_coeffs

    plan = sg_plan.build_plan(11, 2)
    assert (plan.window, plan.polyorder, plan.den, plan.int_ok) == (11, 2, 429, True)
    W, H = 11, 5
    t = (W - 1) // 2
    fw = plan.tab[t * plan.stride : t * plan.stride + W]
    np.testing.assert_array_equal(fw, savgol_coeffs(11, 2)[::-1])
    np.testing.assert_array_equal(plan.itab[:W], [-36, 9, 44, 69, 84, 89, 84, 69, 44, 9, -36])
    # edge rows reproduce polyfit/polyval to rounding
    x = np.arange(W, dtype=np.float64) ** 2 * 3.0 - 7.0 * np.arange(W) + 5.0
    el = plan.tab[t * plan.stride + W : t * plan.stride + W + H * W].reshape(H, W)
    np.testing.assert_allclose(el @ x, x[:H], rtol=1e-12)
    # even window is bumped, short tables exist for every odd w > P
    assert sg_plan.build_plan(12, 2).window == 13
    with pytest.raises(ValueError):
        sg_plan.build_plan(3, 3)


def test_sg_plan_integer_rows_are_exact_rationals():
    from fractions import Fraction

    plan = sg_plan.build_plan(7, 3)
    hat = sg_plan.hat_matrix(7, 3)
    W, H = 7, 3
    nl = plan.itab[W : W + H * W].reshape(H, W)
    for i in range(H):
        for k in range(W):
            assert Fraction(int(nl[i, k]), plan.den_edge) == hat[i][k]


def test_channel_config_layers():
    cfg = {"defaults": {"threshold": 5.0}, "groups": [{"channels": ["0:1", (0, 2)], "config": {"threshold": 7.0}}],
           "channels": {"0:2": {"threshold": 9.0}}}
    assert resolve_channel_values(cfg, "run", 0, 0, {"threshold": 1.0})["threshold"] == 5.0
    assert resolve_channel_values(cfg, "run", 0, 1, {"threshold": 1.0})["threshold"] == 7.0
    assert resolve_channel_values(cfg, "run", 0, 2, {"threshold": 1.0})["threshold"] == 9.0
    assert resolve_channel_values({"run": {"1:1": {"threshold": 3.0}}}, "run", 1, 1, {})["threshold"] == 3.0
    with pytest.raises(ValueError, match="Invalid channel key"):
        resolve_channel_values({"bogus": {"threshold": 1}}, "run", 0, 0, {})


def test_plugin_contract_attributes():
    for p in hip_default():
        assert p.provides and p.save_when == {"filtered_waveforms": "target", "signal_peaks_stream": "never"}.get(p.provides, "always")
        assert p.output_dtype is not None or p.provides in ("hit_grouped", "signal_peaks_stream")  # DataFrame / chunk stream
        assert "wave_source" in p.options or p.provides in (
            "wave_pool_filtered", "hit_grouped", "filtered_waveforms", "waveform_width", "s1_s2", "hit_merge_clusters",
            "hit_merged", "hit_merged_components", "signal_peaks_stream")
    bf = [p for p in hip_default() if p.provides == "basic_features"][0]
    assert bf.resolve_depends_on(SimpleContext({})) == ["st_waveforms"]  # reference default: wave_source="auto"
    assert bf.resolve_depends_on(SimpleContext({"use_filtered": True})) == ["filtered_waveforms"]
    assert bf.resolve_depends_on(SimpleContext({"wave_source": "records", "use_filtered": True})) == ["records", "wave_pool_filtered"]
    ww = [p for p in hip_default() if p.provides == "waveform_width"][0]
    assert ww.resolve_depends_on(SimpleContext({"use_filtered": True})) == ["hit", "filtered_waveforms"]
    hit = [p for p in hip_default() if p.provides == "hit_threshold"][0]
    assert hit.resolve_depends_on(SimpleContext({})) == ["st_waveforms"]  # reference default: wave_source="auto"
    assert hit.resolve_depends_on(SimpleContext({"use_filtered": True})) == ["filtered_waveforms"]
    ctx = SimpleContext({"wave_source": "records", "use_filtered": True})
    assert hit.resolve_depends_on(ctx) == ["records", "wave_pool_filtered"]
    ctx = SimpleContext({"wave_source": "records", "use_filtered": True, "fuse_filter": True})
    assert hit.resolve_depends_on(ctx) == ["records", "wave_pool"]
    assert hit.resolve_depends_on(SimpleContext({"wave_source": "st_waveforms"})) == ["st_waveforms"]
    with pytest.raises(ValueError, match="Invalid wave_source"):
        hit.resolve_depends_on(SimpleContext({"wave_source": "bogus"}))
    for name in ("hit", "waveform_width_integral"):
        p = [q for q in hip_default() if q.provides == name][0]
        assert p.resolve_depends_on(SimpleContext({"use_filtered": False})) == ["st_waveforms"]
        assert p.resolve_depends_on(SimpleContext({"wave_source": "records", "use_filtered": False})) == ["records", "wave_pool"]


def test_empty_inputs_need_no_device():
    from waveformanalysis_amd.dtypes import RECORDS_DTYPE, THRESHOLD_HIT_DTYPE

    hit = [p for p in hip_default() if p.provides == "hit_threshold"][0]
    ctx = SimpleContext({"wave_source": "records"},
                        {"records": np.zeros(0, dtype=RECORDS_DTYPE), "wave_pool": np.zeros(0, dtype=np.uint16)})
    out = hit.compute(ctx, "run")
    assert out.dtype == THRESHOLD_HIT_DTYPE and len(out) == 0
