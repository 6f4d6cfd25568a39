"""The Context hooks on the real device: kernel names and times reach a reference-style Profiler, the
stats table gets rates, cleanup() gives device scratch back and the next compute() still answers correctly."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from tests.test_hooks_cpu import Profiler, Stats
from waveformanalysis_amd import synth
from waveformanalysis_amd.device import DevicePool
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import HipBasicFeaturesPlugin, HipThresholdHitPlugin

pytestmark = pytest.mark.gpu


def test_profiler_keys_rates_and_cleanup_on_the_device():
    rec, pool = synth.make_run(4000, "vx2730", cfg=12)
    ctx = SimpleContext({"wave_source": "records", "hit_threshold": {"use_filtered": True, "fuse_filter": True}},
                        {"records": rec, "wave_pool": pool}, [HipThresholdHitPlugin(), HipBasicFeaturesPlugin()])
    ctx.profiler, ctx.stats_collector = Profiler(), Stats()
    ctx.wfa_device_pool = DevicePool([0])
    try:
        want = O.threshold_hits_chunked(rec, O.filter_wave_pool(rec, pool))
        hits = ctx.get_data("run", "hit_threshold")
        G.assert_struct_equal(hits, want, float_rtol=1e-6, what="instrumented compute")
        prof = ctx.profiler
        keys = [k for k in prof.durations if k.startswith("plugin.hit_threshold.hip.")]
        assert {"plugin.hit_threshold.hip.k_sg_runs32", "plugin.hit_threshold.hip.k_hit_rows_flat"} <= set(keys), keys
        assert all(prof.durations[k] > 0 for k in keys) and prof.counts["plugin.hit_threshold.hip"] == 1
        row = ctx.stats_collector.hip_metrics["hit_threshold"][0]
        assert row["samples"] == len(pool) and row["rows"] == len(want) and row["gsamples_per_s"] > 1.0
        # SimpleContext called cleanup(): the padded shadow pool, the run events ... are gone; nothing is left to free
        sess = ctx.wfa_device_pool.session()
        assert sess.release_scratch() == 0
        feats = ctx.get_data("run", "basic_features")                        # the pool is still resident: no upload
        assert sess.uploads == 1
        G.assert_struct_equal(feats, O.basic_features(rec, pool), what="after cleanup")
        ctx._results.clear()
        again = ctx.get_data("run", "hit_threshold")                         # scratch is rebuilt on demand
        G.assert_struct_equal(again, hits, what="second compute after cleanup")
        assert prof.counts["plugin.hit_threshold.hip.k_pad_rows (once per upload)"] == 2
    finally:
        ctx.wfa_device_pool.close()
