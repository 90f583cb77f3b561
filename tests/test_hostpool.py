"""The recycled result buffers never hand out memory somebody still refers to."""
import gc
import threading

import numpy as np

import pytest

from lars_image_processing_amd import hostpool as hp


@pytest.fixture(autouse=True)
def pool_on(monkeypatch):
    """The pool is opt-in (LARS_HOST_POOL_MB, default 0): switch it on for these tests."""
    monkeypatch.setattr(hp, "LIMIT_BYTES", 1 << 30)
    hp.clear()
    yield
    hp.clear()


def test_pool_is_off_by_default(monkeypatch):
    """Default contract: fresh arrays that own their data; views of recycled buffers only on request."""
    import importlib
    import os
    monkeypatch.delenv("LARS_HOST_POOL_MB", raising=False)
    fresh = importlib.reload(hp)
    try:
        assert fresh.LIMIT_BYTES == 0
        a = fresh.empty((1024, 1024), np.float32)
        assert a.flags.owndata and fresh.stats()[0] == 0
    finally:
        importlib.reload(hp)


def test_reuse_only_after_every_view_is_gone():
    hp.clear()
    a = hp.empty((512, 1024), np.float32)
    addr = a.ctypes.data
    assert a.flags.c_contiguous and a.shape == (512, 1024) and a.dtype == np.float32
    a[:] = 7
    # a slice, a reshaped view and a view of another dtype: each of them keeps the buffer busy on its own
    holders = {"row": a[100], "flat": a.reshape(-1)[3:], "raw": a.view(np.uint8)}
    del a
    while holders:
        b = hp.empty((512, 1024), np.float32)
        assert b.ctypes.data != addr, sorted(holders)
        b[:] = 1
        for view in holders.values():
            assert (view.view(np.float32) == 7).all()
        del b, view
        holders.popitem()
    gc.collect()
    before = hp.stats()
    c = hp.empty((512, 1024), np.float32)
    after = hp.stats()
    assert after[0] == before[0] and after[1] == before[1]      # an idle buffer was reused, nothing new allocated
    assert after[2] == before[2] - c.nbytes


def test_small_and_oversized_requests_bypass_the_pool():
    hp.clear()
    s = hp.empty((100, 100), np.uint8)
    assert s.base is None
    assert hp.stats()[0] == 0
    assert hp.empty(5, np.float64).shape == (5,)


def test_bound_and_threads(monkeypatch):
    hp.clear()
    monkeypatch.setattr(hp, "LIMIT_BYTES", 8 << 20)
    held = [hp.empty((1 << 20,), np.float32) for _ in range(2)]          # 4 MiB each: the bound is reached
    extra = hp.empty((1 << 20,), np.float32)
    assert extra.base is None and hp.stats()[1] == 8 << 20                  # past the bound: a plain array
    del held, extra
    errors = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        for _ in range(200):
            n = int(rng.integers(1 << 18, 1 << 19))
            x = hp.empty((n,), np.float32)
            tag = float(seed * 1000 + _)
            x[:] = tag
            y = hp.empty((n,), np.float32)
            y[:] = -1
            if not (x == tag).all():
                errors.append(seed)
            del x, y
    ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors
    assert hp.stats()[1] <= 8 << 20
    hp.clear()
    assert hp.stats() == (0, 0, 0)
