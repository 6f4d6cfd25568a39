"""The Context hooks of the HIP plugins without a GPU (reference core/context_execution.py:140-183,
core/foundation/utils.py:92-207, core/plugins/core/stats.py:103-520, core/plugins/core/base.py:602-613):
kernel times into `context.profiler` under plugin.<name>.hip.<kernel>, rates into the stats collector, cleanup()
releases device scratch after every compute() and drops the session after a failed one."""

import contextlib
import time
from collections import defaultdict

import numpy as np
import pytest

from waveformanalysis_amd.plugin_api import Option, SimpleContext
from waveformanalysis_amd.plugins import _common as K


class Profiler:  # the reference's Profiler API (durations / counts / timeit)
    def __init__(self):
        self.durations, self.counts = defaultdict(float), defaultdict(int)

    @contextlib.contextmanager
    def timeit(self, key):
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.durations[key] += time.perf_counter() - t0
            self.counts[key] += 1


class Stats:
    def __init__(self, on=True):
        self.on = on

    def is_enabled(self):
        return self.on


class Sess:
    def __init__(self):
        self.n_samples, self.n_records = 0, 0
        self.prof_on = False
        self.released = 0
        self.closed = False

    def ensure_pool(self, pool, cacheable=True):
        self.n_samples = pool.size

    def upload_records(self, rec):
        self.n_records = len(rec)

    def profile(self, on=True):
        self.prof_on = on

    def profile_report(self):
        return {"k_sg_runs32": (0.5, 1), "k_hit_rows_flat": (0.25, 1)} if self.prof_on else {}

    def release_scratch(self):
        self.released += 1
        return 1 << 20

    def close(self):
        self.closed = True


class Pool:
    def __init__(self):
        self.s = Sess()
        self.dropped = 0

    def session(self):
        if self.s is None:
            self.s = Sess()
        return self.s

    def peek_session(self):
        return self.s

    def drop_session(self):
        self.dropped += 1
        self.s.close()
        self.s = None
        return True


class Hits(K.HipPlugin):
    provides = "hits_x"
    algorithmic_bytes = (2, 29, 60)
    options = {"fail": Option(default=False, type=bool)}

    def compute(self, context, run_id, **_kw):
        pool = context.get_data(run_id, "wave_pool")
        sess = K.resident_session(context, pool)
        sess.upload_records(context.get_data(run_id, "records"))
        if context.get_config(self, "fail"):
            raise ValueError("device said no")
        return np.zeros(7, dtype=[("a", "i4")])


def make_ctx(**extra):
    ctx = SimpleContext({}, {"wave_pool": np.zeros(8000, np.uint16), "records": np.zeros(10, [("x", "i4")])}, [Hits()])
    ctx.wfa_device_pool = Pool()
    for k, v in extra.items():
        setattr(ctx, k, v)
    return ctx


def test_kernel_times_reach_the_profiler_and_rates_the_stats_collector():
    ctx = make_ctx(profiler=Profiler(), stats_collector=Stats())
    rows = ctx.get_data("run", "hits_x")
    assert len(rows) == 7
    prof = ctx.profiler
    assert prof.counts["plugin.hits_x.hip"] == 1
    assert prof.durations["plugin.hits_x.hip.k_sg_runs32"] == pytest.approx(0.5e-3)
    assert prof.durations["plugin.hits_x.hip.k_hit_rows_flat"] == pytest.approx(0.25e-3)
    assert prof.counts["plugin.hits_x.hip.k_sg_runs32"] == 1
    row = ctx.get_plugin("hits_x").device_stats
    assert row["samples"] == 8000 and row["records"] == 10 and row["rows"] == 7
    assert row["device_s"] == pytest.approx(0.75e-3)
    assert row["gsamples_per_s"] == pytest.approx(8000 / 0.75e-3 / 1e9)
    assert row["hbm_GBps_algorithmic"] == pytest.approx((2 * 8000 + 29 * 10 + 60 * 7) / 0.75e-3 / 1e9)
    assert ctx.stats_collector.hip_metrics["hits_x"] == [row]
    assert not ctx.wfa_device_pool.s.prof_on           # timers are switched off again


def test_no_hooks_no_timers():
    ctx = make_ctx()                                   # no profiler, no collector: compute() is not wrapped in anything
    ctx.get_data("run", "hits_x")
    assert not hasattr(ctx.get_plugin("hits_x"), "device_stats")
    ctx2 = make_ctx(stats_collector=Stats(on=False))
    ctx2.get_data("run", "hits_x")
    assert not hasattr(ctx2.get_plugin("hits_x"), "device_stats")


def test_cleanup_releases_scratch_and_drops_the_session_after_an_error():
    ctx = make_ctx(profiler=Profiler())
    ctx.get_data("run", "hits_x")                      # SimpleContext calls cleanup() like context_execution.py:179
    pool = ctx.wfa_device_pool
    assert pool.s.released == 1 and pool.dropped == 0
    bad = make_ctx(profiler=Profiler())
    bad.config["fail"] = True
    first = bad.wfa_device_pool.s
    with pytest.raises(RuntimeError, match="device said no"):
        bad.get_data("run", "hits_x")
    assert bad.wfa_device_pool.dropped == 1 and first.closed and first.released == 0
    # the hooks take what the error manager hands them (core/foundation/error.py:95 passes a dict as `context`)
    from waveformanalysis_amd.plugins.threshold_hit import HipThresholdHitPlugin

    assert HipThresholdHitPlugin().resolve_depends_on({"plugin": "hit_threshold"}, run_id="run")


def test_real_pool_session_bookkeeping():
    from waveformanalysis_amd import device as D

    made = []

    class S:
        def __init__(self, dev):
            made.append(self)
            self.closed = False

        def close(self):
            self.closed = True

    pool = D.DevicePool(device_ids=[0], session_factory=S)
    assert pool.peek_session() is None and not pool.drop_session()
    s = pool.session()
    assert pool.peek_session() is s and pool.drop_session() and s.closed
    assert pool.peek_session() is None and pool.session() is not s
