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


def check_multi_channel_fixture(tw, session):
    case = load()
    df = frame(case)
    g = group_multi_channel_hits(df, float(tw), session=session)
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


@pytest.mark.parametrize("tw", [100, 40])
def test_group_multi_channel_hits(tw):
    check_multi_channel_fixture(tw, session=False)   # the host table code


def check_multi_channel_ties(session):
    rng = np.random.default_rng(4)
    n = 5000
    ts = rng.integers(0, 2000, n) * 1000          # many equal timestamps
    ch = rng.integers(0, 4, n)                      # many equal channels per event
    df = pd.DataFrame({"timestamp": ts, "channel": ch, "charge": rng.uniform(0, 1, n), "peak": rng.uniform(0, 1, n)})
    g = group_multi_channel_hits(df, 3.0, session=session)
    events = O.group_multi_channel_hits_literal(ts, ch, df["charge"].to_numpy(), df["peak"].to_numpy(), 3.0)
    assert len(events) == len(g)
    for (t0, t1, members), (_, row) in zip(events, g.iterrows()):
        np.testing.assert_array_equal(row["timestamps"], ts[members])
        np.testing.assert_array_equal(row["areas"], df["charge"].to_numpy()[members])
        assert (row["t_min"], row["t_max"]) == (t0, t1)
    assert len(group_multi_channel_hits(df.iloc[:0], 3.0, session=session)) == 0
    with pytest.raises(KeyError, match="area/height"):
        group_multi_channel_hits(df.drop(columns=["peak"]), 3.0, session=session)
    with pytest.raises(ValueError, match="time_window_ns"):
        group_multi_channel_hits(df, -1.0, session=session)


def test_group_multi_channel_hits_ties_and_edges():
    check_multi_channel_ties(session=False)
    np.testing.assert_array_equal(find_cluster_boundaries(np.array([0, 5, 10, 11, 30]), 10.0), [0, 3, 4, 5])
    np.testing.assert_array_equal(find_cluster_boundaries(np.zeros(0), 10.0), [0])


@pytest.mark.gpu
def test_group_multi_channel_hits_gpu():
    """The device route (two stable radix sorts + the window chain by pointer jumping) against the reference's fixture, the
    literal oracle, and the host table code on tables that stress the chain: one long cluster, clusters of one hit, a
    window of zero, timestamps beyond 2^53 (float64 comparison), negative channels."""
    from waveformanalysis_amd.device import DeviceSession
    from waveformanalysis_amd.event_grouping import _group_multi_channel_order_host

    with DeviceSession(0) as sess:
        for tw in (100, 40):
            check_multi_channel_fixture(tw, session=sess)
        check_multi_channel_ties(session=sess)
        rng = np.random.default_rng(11)
        for n, spread, w_ps in ((1, 10, 5.0), (2, 10, 0.0), (70_000, 50, 1e9), (70_000, 10**9, 3.0), (300_001, 4000, 2500.0),
                                (300_001, 4000, 0.0), (1000, 3, 1.5)):
            ts = np.sort(rng.integers(0, max(2, n * spread), n)).astype(np.int64)
            if n == 1000:
                ts += (1 << 60)                                   # float64 cannot tell neighbours apart up here
            rng.shuffle(ts)
            ch = rng.integers(-3, 40, n).astype(np.int32)
            order, bounds = sess.group_multi_channel(ts, ch, w_ps)
            want_order, want_bounds = _group_multi_channel_order_host(ts, ch, w_ps)
            np.testing.assert_array_equal(bounds, want_bounds, err_msg=f"n {n} window {w_ps}")
            np.testing.assert_array_equal(order, want_order, err_msg=f"n {n} window {w_ps}")


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
