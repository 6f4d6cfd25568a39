"""Uniform records whose length is not a multiple of 16 samples (VX2730: 1500) take the span16 kernels on a padded
shadow layout built on the device.  Lengths around the supported boundary (L % 16 >= half window), both polarities,
fused and given baselines, and odd pool offsets, against the oracle and against the per-record kernels
(option no_pad) -- integer fields exact, floats within the threshold-hit tolerance."""

import os

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd import _lib, synth
from waveformanalysis_amd.device import DeviceSession

pytestmark = pytest.mark.gpu
FLOAT_RTOL = 1e-6


def _run(rec, pool, fused_baseline, **options):
    if True:
        with DeviceSession(0) as sess:
            for name, value in options.items():
                sess.set_option(name, value)
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
        assert any("span16" in k or "runs32" in k for k in kernels) == expect_pad, kernels  # a uniform-record kernel
        G.assert_struct_equal(rows, want, float_rtol=FLOAT_RTOL, what=f"L={L} fused_baseline={fused_baseline}")
        G.assert_struct_equal(again, want, float_rtol=FLOAT_RTOL, what="second pass")
        plain, _a, kernels2 = _run(rec_in, pool, fused_baseline, no_pad=True)
        assert not any("k_pad_rows" in k for k in kernels2)
        assert plain.tobytes() == rows.tobytes()                      # the per-record kernels agree byte for byte


@pytest.mark.parametrize("L", [1500, 1000, 100, 37, 47])
def test_padded_materialised_filter_bit_exact(L):
    """wave_pool_filtered through the span kernel on the shadow layout == scipy's float32 output == the per-record
    kernel (option no_pad), for every window the integer plan covers and for a low pedestal (literal branch)."""
    rec, pool = synth.make_run(300 if L > 200 else 2000, "vx2730", cfg=70 + L % 50, L=L)
    rng = np.random.default_rng(L + 1)
    w = pool.reshape(len(rec), L).astype(np.int32)
    w[::7] -= 7990                       # records near zero: numerators below the integer guard
    w[:, -3:] += rng.integers(-40, 40, (len(rec), 3))   # structure in the right edge samples
    pool = w.clip(0, 16383).astype(np.uint16).reshape(-1)
    pool = np.concatenate([np.zeros(4, np.uint16), pool])
    rec = rec.copy()
    rec["wave_offset"] += 4
    for W, P in ((11, 2), (7, 3), (15, 4), (5, 2)):
        want = O.filter_wave_pool(rec, pool, sg_window_size=W, sg_poly_order=P)
        outs = {}
        for no_pad in (False, True):
            with DeviceSession(0) as sess:
                sess.set_option("no_pad", no_pad)
                sess.upload_pool(pool)
                sess.upload_records(rec, 10.0)
                plan = sess.set_sg_plan(W, P)
                sess.profile(True)
                outs[no_pad] = (sess.savgol(), set(sess.profile_report()))
        got, kernels = outs[False]
        # windows without an integer plan take the literal float64 kernel in every layout
        assert any("k_savgol_span<padded>" in k for k in kernels) == bool(plan.int_ok), (W, P, kernels)
        np.testing.assert_array_equal(got, want, err_msg=f"L={L} W={W}")
        np.testing.assert_array_equal(outs[True][0], want)
        assert not any("padded" in k for k in outs[True][1])
