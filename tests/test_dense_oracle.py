"""Dense branches (st_waveforms / filtered_waveforms), waveform_width and s1_s2: oracle restatements and the
vectorised host table code against fixtures produced by the reference's plugins."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd.channel_config import resolve_channel_values
from waveformanalysis_amd.plugins.s1_s2 import classify
from waveformanalysis_amd.plugins.waveform_width import first_row_of_record_id

BASE = dict(filter_type="SG", lowcut=0.1, highcut=0.5, fs=0.5, filter_order=4, sg_window_size=11, sg_poly_order=2)


def oracle_cfg(values):
    if values["filter_type"] == "BW":
        return dict(filter_type="BW", bw_sos=O.design_bw(values["lowcut"], values["highcut"], values["fs"],
                                                         values["filter_order"]))
    w = int(values["sg_window_size"])
    return dict(filter_type="SG", sg_window_size=w + (w % 2 == 0), sg_poly_order=int(values["sg_poly_order"]))


@pytest.mark.parametrize("name", G.dense_case_names())
def test_filtered_waveforms_oracle(name):
    case = G.load_dense(name)
    st = case["st_waveforms"]
    got = O.filtered_waveforms_dense(st, lambda b, c: oracle_cfg(BASE))
    np.testing.assert_array_equal(got, case["filtered_waveforms"]["wave"])
    cc = case["options"]["filter_cc"]
    got = O.filtered_waveforms_dense(st, lambda b, c: oracle_cfg(resolve_channel_values(cc, "run", b, c, BASE)))
    np.testing.assert_array_equal(got, case["filtered_cc"]["wave"])
    assert not np.array_equal(case["filtered_cc"]["wave"], case["filtered_waveforms"]["wave"])


@pytest.mark.parametrize("name", G.dense_case_names())
def test_basic_features_dense_oracle(name):
    case = G.load_dense(name)
    G.assert_struct_equal(O.basic_features_dense(case["st_waveforms"]), case["bf_st"])
    G.assert_struct_equal(O.basic_features_dense(case["filtered_waveforms"], height_range=(30, 400),
                                                 area_range=(10, 700)), case["bf_filt"])


@pytest.mark.parametrize("name", G.dense_case_names())
def test_waveform_width_oracle(name):
    case = G.load_dense(name)
    for k, cfg in enumerate(case["options"]["width"]):
        cfg = dict(cfg)
        data = case["filtered_waveforms"] if cfg.pop("use_filtered", False) else case["st_waveforms"]
        got = O.waveform_width(case["hit"], data, **cfg)
        assert len(got) > 0
        G.assert_struct_equal(got, case[f"width_{k}"], what=f"{name} width cfg {k}")


@pytest.mark.parametrize("name", G.dense_case_names())
def test_s1_s2_oracle_and_vectorised(name):
    case = G.load_dense(name)
    for k, cfg in enumerate(case["options"]["s1s2"]):
        want = case[f"s1s2_{k}"]
        G.assert_struct_equal(O.s1_s2_classify(case["width_0"], case["bf_st"], **cfg), want, what=f"oracle cfg {k}")
        G.assert_struct_equal(classify(case["width_0"], case["bf_st"], **cfg), want, what=f"vectorised cfg {k}")


def test_s1_s2_randomised_against_oracle():
    rng = np.random.default_rng(3)
    n = 500
    widths = np.zeros(n, dtype=O.WAVEFORM_WIDTH_DTYPE)
    widths["total_width"] = rng.uniform(0, 200, n)
    widths["total_width"][rng.integers(0, n, 20)] = np.nan
    widths["total_width_samples"] = widths["total_width"] / 2
    widths["record_id"] = rng.integers(-5, 260, n)
    widths["peak_position"] = rng.integers(0, 800, n)
    widths["channel"] = rng.integers(0, 16, n)
    feats = np.zeros(250, dtype=O.BASIC_FEATURES_DTYPE)
    feats["height"] = rng.uniform(0, 500, 250)
    feats["area"] = rng.uniform(-100, 9000, 250)
    feats["area"][::17] = np.nan
    with_rid = np.zeros(250, dtype=feats.dtype.descr + [("record_id", "i8")])
    for f in feats.dtype.names:
        with_rid[f] = feats[f]
    with_rid["record_id"] = rng.integers(0, 200, 250)  # duplicates: the first match wins
    for cfg in (dict(s1_width_range=(None, 60.0), s2_width_range=(50.0, None)),
                dict(s1_width_range=(10.0, 80.0), s1_area_range=(0.0, 4000.0), s2_height_range=(100.0, None),
                     conflict_policy="prefer_s1"),
                dict(width_unit="samples", s2_width_range=(5.0, 60.0), s2_area_range=(None, 7000.0),
                     s1_height_range=(None, 250.0), conflict_policy="prefer_s2"),
                dict()):
        for table in (feats, with_rid):
            G.assert_struct_equal(classify(widths, table, **cfg), O.s1_s2_classify(widths, table, **cfg))
    with pytest.raises(ValueError, match="No S1/S2 criteria"):
        classify(widths, feats, strict=True)
    with pytest.raises(ValueError, match="range must be a tuple"):
        classify(widths, feats, s1_width_range=[0, 1])


def test_first_row_lookup():
    ids = np.array([5, 3, 5, 9, 3, 0])
    got = first_row_of_record_id(ids, np.array([5, 3, 9, 0, 7, -1, 100]))
    np.testing.assert_array_equal(got, [0, 1, 3, 5, -1, -1, -1])
    np.testing.assert_array_equal(first_row_of_record_id(np.zeros(0, dtype=np.int64), np.array([1, 2])), [-1, -1])
