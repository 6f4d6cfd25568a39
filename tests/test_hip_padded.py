"""Uniform records whose length is not a multiple of 16 samples (VX2730: 1500) take the span16 kernels on a padded
shadow layout built on the device.  Lengths around the supported boundary (L % 16 >= half window), both polarities,
fused and given baselines, and odd pool offsets, against the oracle and against the per-record kernels
(WFA_DISABLE_PAD) -- integer fields exact, floats within the threshold-hit tolerance."""

import os

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession

pytestmark = pytest.mark.gpu
FLOAT_RTOL = 1e-6


def _run(rec, pool, fused_baseline, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with DeviceSession(0) as sess:
            sess.upload_pool(pool)
            sess.upload_records(rec, 10.0)
            sess.set_sg_plan(11, 2)
            sess.profile(True)
            if fused_baseline:
                rows = sess.fused_baseline_filter_hits((0, synth.BASELINE_SAMPLES), 2, 2)
            else:
                rows = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)
            again = sess.threshold_hits(_lib.SRC_SG_FUSED, 2, 2)      # shadow re-used, baselines now on the device
            return rows, again, set(sess.profile_report())
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


@pytest.mark.parametrize("L", [1500, 1000, 37, 44, 47, 36, 100])
@pytest.mark.parametrize("polarity", ["unknown", "positive"])
def test_padded_layout_matches_oracle(L, polarity):
    rec, pool = synth.make_run(700 if L > 200 else 3000, "vx2730", cfg=60 + L % 50, L=L, polarity="unknown")
    # dips anywhere in the record, the edge samples included (the generator keeps its pulses away from the ends)
    rng = np.random.default_rng(L)
    w = pool.reshape(len(rec), L).astype(np.int32)
    for i in range(len(rec)):
        for _ in range(int(rng.integers(1, 3))):
            a = int(rng.integers(-4, L))
            w[i, max(a, 0) : a + int(rng.integers(3, 11))] -= int(rng.integers(15, 600))
    pool = w.clip(0, 16383).astype(np.uint16).reshape(-1)
    rec = rec.copy()
    rec["baseline"] = pool.reshape(len(rec), L)[:, : min(40, L)].mean(axis=1)
    if polarity == "positive":
        pool = (16383 - pool.astype(np.int32)).clip(0, 16383).astype(np.uint16)   # pulses go up
        rec = rec.copy()
        rec["polarity"] = "positive"
        rec["baseline"] = pool.reshape(len(rec), L)[:, : min(40, L)].mean(axis=1)
    # an odd start offset: the packed pool is only 2-byte aligned, the shadow does not care
    pool = np.concatenate([np.zeros(3, np.uint16), pool])
    rec = rec.copy()
    rec["wave_offset"] += 3
    want = O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool))
    assert len(want) > 100
    for fused_baseline in (False, True):
        rec_in = rec.copy()
        if fused_baseline:
            rec_in["baseline"] = np.nan
        rows, again, kernels = _run(rec_in, pool, fused_baseline)
        expect_pad = L % 16 >= 5
        assert any("k_pad_rows" in k for k in kernels) == expect_pad, kernels
        assert any("span16" in k for k in kernels) == expect_pad, kernels
        G.assert_struct_equal(rows, want, float_rtol=FLOAT_RTOL, what=f"L={L} fused_baseline={fused_baseline}")
        G.assert_struct_equal(again, want, float_rtol=FLOAT_RTOL, what="second pass")
        plain, _a, kernels2 = _run(rec_in, pool, fused_baseline, WFA_DISABLE_PAD="1")
        assert not any("k_pad_rows" in k for k in kernels2)
        assert plain.tobytes() == rows.tobytes()                      # the per-record kernels agree byte for byte
