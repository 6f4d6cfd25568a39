"""Residency rules of DeviceSession / resident_session without a GPU (an (address, size, dtype) key does
not identify the contents of a buffer; sessions borrowed elsewhere must not leave a stale "resident" tag behind)."""

import numpy as np
import pytest

from waveformanalysis_amd import device as D
from waveformanalysis_amd.plugins import _common as K


class FakeSession(D.DeviceSession):
    """DeviceSession with the C calls replaced: counts uploads, remembers what the 'device' holds."""

    def __init__(self):  # no wfa_ctx
        self._h = None
        self._res_pool = None
        self._res_filtered = None
        self.uploads = 0
        self.n_samples = 0
        self.n_records = 0
        self.dev_pool = None
        self.dev_filtered = None

    def upload_pool(self, wave_pool):
        self.forget_resident()
        self.uploads += 1
        self.dev_pool = np.array(wave_pool, copy=True)
        self.n_samples = wave_pool.size

    def upload_filtered_pool(self, pool_f32):
        self._res_filtered = None
        self.uploads += 1
        self.dev_filtered = np.array(pool_f32, copy=True)

    def pool_gather(self, *a, **k):
        self.forget_resident()
        self.dev_pool = None

    def close(self):
        self.forget_resident()


class FakePool:
    def __init__(self):
        self.s = FakeSession()

    def session(self):
        return self.s


class Ctx:
    def __init__(self):
        self.wfa_device_pool = FakePool()


def test_same_object_uploads_once_and_temporaries_always():
    ctx = Ctx()
    pool = np.arange(1000, dtype=np.uint16)
    s = K.resident_session(ctx, pool)
    assert s.uploads == 1
    K.resident_session(ctx, pool)
    assert s.uploads == 1                       # the very same array object: still resident
    K.resident_session(ctx, pool.copy())
    assert s.uploads == 2                       # equal contents, another object: uploaded
    # the failure the advisor reproduced: same-shaped temporaries that recycle one address
    seen = []
    for k in range(3):
        tmp = np.full(20000 * 8, k, dtype=np.uint16)
        s = K.resident_session(ctx, tmp, cacheable=False)
        seen.append(int(s.dev_pool[0]))
        del tmp
    assert seen == [0, 1, 2] and s.uploads == 5
    K.resident_session(ctx, pool)               # a non-cacheable upload displaced the cached pool
    assert s.uploads == 6 and s.dev_pool[5] == 5


def test_cached_pool_is_kept_alive_so_its_address_cannot_be_reused():
    ctx = Ctx()
    a = np.zeros(4096, dtype=np.uint16)
    s = K.resident_session(ctx, a)
    addr = a.__array_interface__["data"][0]
    del a
    b = np.ones(4096, dtype=np.uint16)          # would often land on `addr` had the session not kept `a` alive
    assert b.__array_interface__["data"][0] != addr
    K.resident_session(ctx, b)
    assert s.uploads == 2 and s.dev_pool[0] == 1


def test_borrowed_session_calls_drop_the_tag():
    ctx = Ctx()
    pool = np.arange(64, dtype=np.uint16)
    s = K.resident_session(ctx, pool)
    s.upload_pool(np.zeros(8, dtype=np.uint16))   # what find_hits / HipThresholdHitStream.compute_chunk do
    K.resident_session(ctx, pool)
    assert s.uploads == 3 and s.dev_pool.size == 64
    s.pool_gather()                                # records_builder packs a new pool on the device
    K.resident_session(ctx, pool)
    assert s.uploads == 4
    s.close()
    K.resident_session(ctx, pool)
    assert s.uploads == 5


def test_filtered_twin_follows_the_same_rules():
    ctx = Ctx()
    pool = np.arange(64, dtype=np.uint16)
    filt = np.arange(64, dtype=np.float32)
    s = K.resident_session(ctx, pool, filt)
    assert s.uploads == 2
    K.resident_session(ctx, pool, filt)
    assert s.uploads == 2
    K.resident_session(ctx, pool, filt.copy())
    assert s.uploads == 3
    other = pool.copy()
    K.resident_session(ctx, other, filt)           # a new wave_pool invalidates its filtered twin as well
    assert s.uploads == 5


def test_real_session_methods_reset_tags():
    """The production class resets the tags in upload_pool / pool_gather / savgol / sosfiltfilt / close (source check:
    these need a GPU to run)."""
    import inspect

    for name in ("upload_pool", "pool_gather", "close"):
        assert "forget_resident" in inspect.getsource(getattr(D.DeviceSession, name)), name
    for name in ("upload_filtered_pool", "savgol", "sosfiltfilt"):
        assert "_drop_f32_tags" in inspect.getsource(getattr(D.DeviceSession, name)), name


def test_float32_pool_tag_is_dropped_when_a_filter_overwrites_the_device_buffer():
    """a float32 array uploaded as the pool and a filter output share one device buffer.  After savgol /
    sosfiltfilt / upload_filtered_pool the array remembered as the resident pool is no longer what the device holds."""
    s = FakeSession()
    f32 = np.arange(64, dtype=np.float32)
    assert s.ensure_pool(f32) and not s.ensure_pool(f32)
    s._drop_f32_tags()                       # what savgol(), sosfiltfilt(), upload_filtered_pool() do first
    assert s._res_pool is None and s.ensure_pool(f32)
    u16 = np.arange(64, dtype=np.uint16)     # a uint16 pool lives in the other device buffer: its tag stays
    assert s.ensure_pool(u16)
    s._drop_f32_tags()
    assert not s.ensure_pool(u16)


def test_borrow_many_is_atomic_and_spreads_over_devices():
    """ADVICE r2 (low / medium): two sessions are taken under one lock (nested borrow() calls of concurrent pipelines
    could each hold one and wait for ever), and a ring of sessions covers every device of the pool."""
    import threading

    class S:
        def __init__(self, dev):
            self.device_id = dev

        def close(self):
            pass

    pool = D.DevicePool(device_ids=[0, 1, 2, 3], session_factory=S)
    with pool.borrow_many(8) as ring:
        assert [s.device_id for s in ring] == [0, 1, 2, 3, 0, 1, 2, 3]
    with pool.borrow_many(4) as ring:       # reuse of free sessions: still one per device
        assert sorted(s.device_id for s in ring) == [0, 1, 2, 3]
    with pytest.raises(ValueError, match="cannot borrow"):
        with pool.borrow_many(pool.max_sessions + 1):
            pass
    small = D.DevicePool(device_ids=[0], max_sessions=2, session_factory=S)
    order = []

    def pipeline(tag):
        with small.borrow_many(2) as pair:
            order.append((tag, "in", len(pair)))
            threading.Event().wait(0.05)
            order.append((tag, "out", len(pair)))

    threads = [threading.Thread(target=pipeline, args=(k,)) for k in range(3)]
    [t.start() for t in threads]
    [t.join(timeout=10) for t in threads]
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert [e[1] for e in order] == ["in", "out"] * 3      # one pipeline at a time holds both sessions
    with pytest.raises(ValueError, match="cannot borrow"):
        with D.DevicePool(device_ids=[0], max_sessions=1, session_factory=S).borrow_many(2):
            pass


def test_hit_stream_pipeline_uses_every_device_of_the_pool():
    """the streaming hit pipeline drives two sessions per device: chunk k on device k mod n."""
    from tests.bench_stub import Session
    from waveformanalysis_amd import synth
    from waveformanalysis_amd.plugin_api import SimpleContext
    from waveformanalysis_amd.streaming import HipThresholdHitStream, records_to_chunks

    staged = []

    class Spy(Session):
        def upload_pool(self, pool):
            staged.append(self.device_id)
            super().upload_pool(pool)

        def _hits(self):  # no rows: the driver's chunk checks have nothing to object to
            return 0

    rec, pool = synth.make_run(1200, "v1725", cfg=3, threads=1)
    ctx = SimpleContext({"wave_source": "records"}, {"records": rec, "wave_pool": pool})
    dp = D.DevicePool(device_ids=[0, 1, 2, 3], session_factory=Spy)
    plugin = HipThresholdHitStream(use_filtered=False, max_len=800, device_pool=dp)
    chunks = records_to_chunks(rec, 100, "run")
    timeline = []
    outs = plugin.run_chunks(chunks, ctx, "run", max_workers=8, timeline=timeline)
    assert len(outs) == len(chunks) == 12 and [t[0] for t in timeline] == list(range(12))
    assert staged == [k % 4 for k in range(12)]
    # chunk k + 1 .. k + 7 are queued before anybody waits for chunk k (ring of 8 sessions)
    for (k, _b, queued, collected), later in zip(timeline, timeline[1:]):
        assert queued <= later[1] <= later[2] <= collected
    # the executor knob caps the ring: two workers = the double buffer on two devices
    staged.clear()
    plugin.run_chunks(chunks, ctx, "run", max_workers=2)
    assert set(staged) == {0, 1}


def test_device_pool_bounds_live_sessions_across_compute_calls():
    """every parallel compute() used a fresh ThreadPoolExecutor and each worker thread left a
    session behind.  borrow() reuses sessions: N runs leave at most max_sessions alive."""
    from concurrent.futures import ThreadPoolExecutor

    made = []

    class S:
        def __init__(self, dev):
            self.dev = dev
            self.closed = False
            made.append(self)

        def close(self):
            self.closed = True

    pool = D.DevicePool(device_ids=[0, 1], max_sessions=3, session_factory=S)

    def work(_k):
        with pool.borrow() as s:
            assert not s.closed
            return s.dev

    for _run in range(6):
        with ThreadPoolExecutor(max_workers=4) as ex:
            devs = list(ex.map(work, range(16)))
        assert set(devs) <= {0, 1}
    assert len(made) <= 3 and pool.live_sessions <= 3
    # thread-bound sessions are only weakly held: they go when their thread does
    import gc
    import threading

    t = threading.Thread(target=pool.session)
    t.start()
    t.join()
    gc.collect()
    assert pool.live_sessions <= 3
    pool.close()
    assert all(s.closed for s in made[:3])
