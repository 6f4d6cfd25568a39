"""Legacy helpers (find_hits, group_multi_channel_hits): oracle and implementation against a reference fixture."""

import os

import numpy as np
import pandas as pd
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd.event_grouping import (
    MULTI_CHANNEL_COLUMNS,
    find_cluster_boundaries,
    group_multi_channel_hits,
)


def load():
    z = np.load(os.path.join(G.GOLDEN, "legacy_helpers.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def frame(case):
    return pd.DataFrame({"timestamp": case["df_timestamp"], "channel": case["df_channel"], "area": case["df_area"],
                         "height": case["df_height"]})


def test_find_hits_oracle():
    case = load()
    G.assert_struct_equal(O.find_hits_legacy(case["waves"], case["baselines"], 12.5), case["hits_i16"])
    G.assert_struct_equal(O.find_hits_legacy(case["waves_f32"], case["baselines_f32"], 4.25), case["hits_f32"])
    assert len(case["hits_i16"]) > 60 and len(case["hits_f32"]) > 60


@pytest.mark.parametrize("tw", [100, 40])
def test_group_multi_channel_hits(tw):
    case = load()
    df = frame(case)
    g = group_multi_channel_hits(df, float(tw))
    assert list(g.columns) == MULTI_CHANNEL_COLUMNS
    tag = f"w{tw}"
    np.testing.assert_array_equal(g["t_min"].to_numpy(np.int64), case[f"{tag}_t_min"])
    np.testing.assert_array_equal(g["t_max"].to_numpy(np.int64), case[f"{tag}_t_max"])
    np.testing.assert_array_equal(g["dt/ns"].to_numpy(np.float64), case[f"{tag}_dt_ns"])
    np.testing.assert_array_equal(g["n_hits"].to_numpy(np.int64), case[f"{tag}_n_hits"])
    for col in ("channels", "areas", "heights", "timestamps"):
        np.testing.assert_array_equal(np.concatenate(list(g[col])), case[f"{tag}_{col}"])
    events = O.group_multi_channel_hits_literal(case["df_timestamp"], case["df_channel"], case["df_area"],
                                                case["df_height"], float(tw))
    assert len(events) == len(g)
    np.testing.assert_array_equal([e[0] for e in events], g["t_min"])


def test_group_multi_channel_hits_ties_and_edges():
    rng = np.random.default_rng(4)
    n = 5000
    ts = rng.integers(0, 2000, n) * 1000          # many equal timestamps
    ch = rng.integers(0, 4, n)                      # many equal channels per event
    df = pd.DataFrame({"timestamp": ts, "channel": ch, "charge": rng.uniform(0, 1, n), "peak": rng.uniform(0, 1, n)})
    g = group_multi_channel_hits(df, 3.0)
    events = O.group_multi_channel_hits_literal(ts, ch, df["charge"].to_numpy(), df["peak"].to_numpy(), 3.0)
    assert len(events) == len(g)
    for (t0, t1, members), (_, row) in zip(events, g.iterrows()):
        np.testing.assert_array_equal(row["timestamps"], ts[members])
        np.testing.assert_array_equal(row["areas"], df["charge"].to_numpy()[members])
        assert (row["t_min"], row["t_max"]) == (t0, t1)
    assert len(group_multi_channel_hits(df.iloc[:0], 3.0)) == 0
    with pytest.raises(KeyError, match="area/height"):
        group_multi_channel_hits(df.drop(columns=["peak"]), 3.0)
    np.testing.assert_array_equal(find_cluster_boundaries(np.array([0, 5, 10, 11, 30]), 10.0), [0, 3, 4, 5])
    np.testing.assert_array_equal(find_cluster_boundaries(np.zeros(0), 10.0), [0])


@pytest.mark.gpu
def test_find_hits_gpu():
    from waveformanalysis_amd import synth
    from waveformanalysis_amd.event_grouping import find_hits

    case = load()
    G.assert_struct_equal(find_hits(case["waves"], case["baselines"], 12.5), case["hits_i16"])
    G.assert_struct_equal(find_hits(case["waves_f32"], case["baselines_f32"], 4.25), case["hits_f32"])
    assert len(find_hits(np.zeros((0, 800), dtype=np.int16), np.zeros(0), 1.0)) == 0
    rec, pool = synth.make_run(5000, "vx2730", cfg=28)
    waves = pool.reshape(5000, 1500)
    base = rec["baseline"].astype(np.float64)
    G.assert_struct_equal(find_hits(waves, base, 9.0), O.find_hits_legacy(waves.astype(np.float64), base, 9.0))
