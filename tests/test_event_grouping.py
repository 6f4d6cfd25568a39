"""Event grouping (group_hit_windows): the device sort/scan implementation vs the reference's DataFrames
(golden fixtures) and vs the oracle's literal loop on other inputs; validation errors need no device."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import synth
from waveformanalysis_amd.event_grouping import EVENT_COLUMNS, group_hit_windows, group_hit_windows_flat

COLS = ("dt", "boards", "channels", "heights", "integrals", "timestamps", "record_ids", "sample_starts", "sample_ends")


@pytest.mark.gpu
@pytest.mark.parametrize("name", G.grouping_case_names())
def test_matches_reference_dataframe(name):
    case = G.load_grouping(name)
    hits = case["hits"]
    for tw in case["windows"]:
        tag = f"w{int(tw)}"
        df = group_hit_windows(hits, float(tw))
        assert list(df.columns) == EVENT_COLUMNS
        np.testing.assert_array_equal(df["t_min"].to_numpy(np.int64), case[f"{tag}_t_min"])
        np.testing.assert_array_equal(df["t_max"].to_numpy(np.int64), case[f"{tag}_t_max"])
        np.testing.assert_array_equal(df["n_hits"].to_numpy(np.int64), case[f"{tag}_n_hits"])
        np.testing.assert_array_equal(df["dt/ns"].to_numpy(np.float64), case[f"{tag}_dt_ns"])
        np.testing.assert_array_equal(df["event_id"].to_numpy(np.int64), np.arange(len(df)))
        for col in COLS:
            got = np.concatenate(list(df[col])) if len(df) else np.zeros(0)
            np.testing.assert_array_equal(got, case[f"{tag}_{col}"], err_msg=f"{name} {tag} {col}")
            assert got.dtype == case[f"{tag}_{col}"].dtype


@pytest.mark.gpu
def test_flat_form_equals_literal_loop():
    rec, pool = synth.make_run(300, "vx2730", cfg=12, threads=1)
    hits = O.threshold_hits_chunked(rec, pool, threshold=12.0)
    for tw in (0.0, 50.0, 1e4):
        flat = group_hit_windows_flat(hits, tw)
        events = O.group_hit_windows_literal(hits, tw)
        assert len(events) == len(flat["event_start"]) - 1
        for ev, (t_min, t_max, members) in enumerate(events):
            a, b = flat["event_start"][ev], flat["event_start"][ev + 1]
            np.testing.assert_array_equal(flat["order"][a:b], members)
            assert (flat["t_min"][ev], flat["t_max"][ev]) == (t_min, t_max)


def test_errors_and_empty():
    from waveformanalysis_amd.dtypes import THRESHOLD_HIT_DTYPE

    assert len(group_hit_windows(np.zeros(0, dtype=THRESHOLD_HIT_DTYPE), 100.0)) == 0
    h = np.zeros(2, dtype=THRESHOLD_HIT_DTYPE)
    h["dt"] = 2
    with pytest.raises(ValueError, match="time_window_ns"):
        group_hit_windows(h, -1.0)
    h["dt"][1] = 0
    with pytest.raises(ValueError, match="dt must be positive"):
        group_hit_windows(h, 1.0)
    h["dt"] = 2
    h["edge_start"][0] = -1
    with pytest.raises(ValueError, match="component_rows"):
        group_hit_windows(h, 1.0)
