"""Dense (st_waveforms / filtered_waveforms) plugins, waveform_width and the s1_s2 chain on the GPU against
fixtures produced by the reference's plugins.  All float fields are compared exactly (same float64/float32
expression order as numpy)."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import synth
from waveformanalysis_amd.channel_config import resolve_channel_values
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import (
    HipBasicFeaturesPlugin,
    HipFilteredWaveformsPlugin,
    HipS1S2ClassifierPlugin,
    HipWaveformWidthPlugin,
    HipWavePoolFilteredPlugin,
)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", G.dense_case_names())
def test_filtered_waveforms_plugin(name):
    case = G.load_dense(name)
    data = {"st_waveforms": case["st_waveforms"]}
    got = SimpleContext({}, data, plugins=[HipFilteredWaveformsPlugin()]).get_data("run", "filtered_waveforms")
    assert got.dtype == case["filtered_waveforms"].dtype
    G.assert_struct_equal(got, case["filtered_waveforms"])
    with pytest.warns(UserWarning, match="已调整为奇数"):
        got = SimpleContext({"filtered_waveforms": {"channel_config": case["options"]["filter_cc"]}}, data,
                            plugins=[HipFilteredWaveformsPlugin()]).get_data("run", "filtered_waveforms")
    G.assert_struct_equal(got, case["filtered_cc"])


@pytest.mark.parametrize("name", G.dense_case_names())
def test_basic_features_dense(name):
    case = G.load_dense(name)
    data = {"st_waveforms": case["st_waveforms"], "filtered_waveforms": case["filtered_waveforms"]}
    got = SimpleContext({}, data, plugins=[HipBasicFeaturesPlugin()]).get_data("run", "basic_features")
    G.assert_struct_equal(got, case["bf_st"])
    cfg = {"use_filtered": True, "height_range": (30, 400), "area_range": (10, 700)}
    got = SimpleContext({"basic_features": cfg}, data, plugins=[HipBasicFeaturesPlugin()]).get_data("run", "basic_features")
    G.assert_struct_equal(got, case["bf_filt"])
    # explicit source wins over use_filtered (cpu/_wave_source.py:74-90)
    with pytest.warns(UserWarning, match="Ignoring"):
        got = SimpleContext({"basic_features": {"use_filtered": True, "wave_source": "st_waveforms"}}, data,
                            plugins=[HipBasicFeaturesPlugin()]).get_data("run", "basic_features")
    G.assert_struct_equal(got, case["bf_st"])


@pytest.mark.parametrize("name", G.dense_case_names())
def test_waveform_width_and_s1s2_chain(name):
    case = G.load_dense(name)
    data = {"st_waveforms": case["st_waveforms"], "filtered_waveforms": case["filtered_waveforms"], "hit": case["hit"]}
    for k, cfg in enumerate(case["options"]["width"]):
        got = SimpleContext({"waveform_width": dict(cfg)}, data, plugins=[HipWaveformWidthPlugin()]).get_data("run", "waveform_width")
        G.assert_struct_equal(got, case[f"width_{k}"], what=f"{name} width cfg {k}")
    for k, cfg in enumerate(case["options"]["s1s2"]):
        ctx = SimpleContext({"s1_s2": dict(cfg)}, data,
                            plugins=[HipWaveformWidthPlugin(), HipBasicFeaturesPlugin(), HipS1S2ClassifierPlugin()])
        G.assert_struct_equal(ctx.get_data("run", "s1_s2"), case[f"s1s2_{k}"], what=f"{name} s1s2 cfg {k}")


def test_waveform_width_medium_against_oracle():
    """2000 rows, hits at row maxima / random positions, int16 and float32 rows, all option shapes."""
    rec, pool = synth.make_run(2000, "v1725", cfg=44)
    L = 800
    w = pool.reshape(-1, L).astype(np.int64)
    ped = np.rint(rec["baseline"]).astype(np.int64)[:, None]
    flipped = np.clip(2 * ped - w, 0, 16383)
    from waveformanalysis_amd.dtypes import HIT_DTYPE, create_filtered_waveform_dtype, create_record_dtype
    st = np.zeros(len(rec), dtype=create_record_dtype(L))
    st["wave"] = flipped.astype(np.int16)
    for f in ("timestamp", "board", "channel", "dt", "baseline"):
        st[f] = rec[f]
    st["record_id"] = np.arange(len(rec))[::-1]  # lookup by value, not by position
    filt = np.zeros(len(rec), dtype=create_filtered_waveform_dtype(st.dtype))
    for f in st.dtype.names:
        if f != "wave":
            filt[f] = st[f]
    filt["wave"] = O.filter_wave_pool_uniform(st["wave"].reshape(-1).view(np.uint16), L).reshape(-1, L)
    rng = np.random.default_rng(8)
    hits = np.zeros(3 * len(rec), dtype=HIT_DTYPE)
    hits["record_id"] = np.repeat(st["record_id"], 3)
    hits["position"] = np.stack([np.argmax(st["wave"], axis=1), rng.integers(0, L + 5, len(rec)),
                                 np.argmax(filt["wave"], axis=1)], axis=1).reshape(-1)
    hits["timestamp"] = rng.integers(0, 10**12, len(hits))
    hits["channel"] = np.repeat(st["channel"], 3)
    data = {"st_waveforms": st, "filtered_waveforms": filt, "hit": hits}
    for cfg in ({}, {"use_filtered": True}, {"use_filtered": True, "sampling_rate": 0.3, "interpolation": False},
                {"sampling_rate": 0.7, "rise_low": 0.05, "rise_high": 0.5, "fall_high": 0.5, "fall_low": 0.05},
                {"use_filtered": True, "sampling_rate": 0.3, "fall_high": 0.2, "fall_low": 0.6}):
        got = SimpleContext({"waveform_width": dict(cfg)}, data, plugins=[HipWaveformWidthPlugin()]).get_data("run", "waveform_width")
        c = dict(cfg)
        src = filt if c.pop("use_filtered", False) else st
        want = O.waveform_width(hits, src, **c)
        assert len(want) > 1000
        G.assert_struct_equal(got, want, what=str(cfg))


def test_wave_pool_filtered_per_channel_overrides():
    """records path, per-channel filter settings (filtering.py:339-374) into one output pool."""
    rec, pool = synth.make_run(96, "v1725", cfg=45)
    cc = {"0:2": {"sg_window_size": 5, "sg_poly_order": 2}, "0:7": {"filter_type": "BW", "lowcut": 0.01, "highcut": 0.2},
          "0:11": {"sg_window_size": 31, "sg_poly_order": 3}}
    got = SimpleContext({"wave_pool_filtered": {"channel_config": cc}}, {"records": rec, "wave_pool": pool},
                        plugins=[HipWavePoolFilteredPlugin()]).get_data("run", "wave_pool_filtered")
    base = dict(filter_type="SG", lowcut=0.1, highcut=0.5, fs=0.5, filter_order=4, sg_window_size=11, sg_poly_order=2)

    def cfg(i):
        v = resolve_channel_values(cc, "run", int(rec["board"][i]), int(rec["channel"][i]), base)
        if v["filter_type"] == "BW":
            return dict(filter_type="BW", bw_sos=O.design_bw(v["lowcut"], v["highcut"], v["fs"], v["filter_order"]))
        return dict(filter_type="SG", sg_window_size=v["sg_window_size"], sg_poly_order=v["sg_poly_order"])

    want = O.filter_wave_pool(rec, pool, per_record_cfg=cfg)
    np.testing.assert_array_equal(got, want)
    assert not np.array_equal(want, O.filter_wave_pool(rec, pool))
