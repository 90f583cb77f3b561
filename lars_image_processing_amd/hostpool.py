"""Recycled host result buffers -- OPT-IN (``LARS_HOST_POOL_MB`` > 0; the default is 0 = plain ``np.empty``).

The documented contract of the host API is "outputs are fresh host ndarrays owned by the caller" (SURVEY.md 8(b)).
With the pool on, results of 1 MiB or more are VIEWS (``OWNDATA`` False) of recycled byte buffers, and whether a buffer
may be handed out again is read off its CPython reference count: correct for every NumPy consumer (each view holds a
reference to the owner), wrong for a consumer that keeps only a raw pointer (ctypes, a C extension) and not portable to
interpreters without immediate reference counts.  Hence off unless the operator of a long-running host asks for it.

What the large-output calls of the host API pay for is not PCIe but the first touch of freshly allocated result
arrays (about 11 ms of page faults and zeroing per 192 MiB) -- and a caller that recomputes
(the Streamlit callers do, process-images.py:644 / :826 / :912) drops the previous results first.  ``empty()`` hands
out views of byte buffers that stay mapped; a buffer is handed out again only when nothing refers to it any more
(its reference count says so: every view of it, however derived, holds a reference to the owner), so an array the
caller still holds -- or any slice of it -- is never overwritten.

``LARS_HOST_POOL_MB`` bounds the memory kept (default 0 = pool off: plain ``np.empty``; e.g. 1024 for a Streamlit
host that recomputes 4096 x 4096 images).  Results below 1 MiB never go through the pool.
"""
from __future__ import annotations

import os
import sys
import threading

import numpy as np

LIMIT_BYTES = int(os.environ.get("LARS_HOST_POOL_MB", "0")) << 20
MIN_BYTES = 1 << 20
_lock = threading.Lock()
_buffers = []                       # owners (uint8, one allocation each), idle or handed out


def _refs(buf):
    return sys.getrefcount(buf)


def _calibrate():
    """What ``_refs(buf)`` reports, called from a ``for buf in <list>`` loop, for a buffer nothing else refers to:
    measured with the very call shape the pool uses instead of assumed (the count includes the interpreter's own
    temporaries)."""
    probe = [np.empty(1, dtype=np.uint8)]
    for buf in probe:
        if True:
            return _refs(buf)


_IDLE_REFS = _calibrate()


def empty(shape, dtype):
    """Like ``np.empty(shape, dtype)`` (C order); large results come from the pool."""
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if LIMIT_BYTES <= 0 or nbytes < MIN_BYTES or nbytes > LIMIT_BYTES:
        return np.empty(shape, dtype=dtype)
    with _lock:
        chosen = None
        for buf in _buffers:
            if nbytes <= buf.nbytes <= nbytes + nbytes // 8 and _refs(buf) == _IDLE_REFS:
                chosen = buf
                break
        buf = None
        if chosen is None:
            held = sum(b.nbytes for b in _buffers)
            if held + nbytes > LIMIT_BYTES:                               # make room: idle buffers go, oldest first
                keep = []
                for buf in _buffers:
                    if held + nbytes > LIMIT_BYTES and _refs(buf) == _IDLE_REFS:
                        held -= buf.nbytes
                    else:
                        keep.append(buf)
                buf = None
                _buffers[:] = keep
            if held + nbytes > LIMIT_BYTES:
                return np.empty(shape, dtype=dtype)                       # everything kept is in use: do not grow past the bound
            chosen = np.empty(nbytes, dtype=np.uint8)
            _buffers.append(chosen)
        return chosen[:nbytes].view(dtype).reshape(shape)


def stats():
    """(buffers kept, bytes kept, bytes idle) -- for tests and diagnostics."""
    with _lock:
        idle = 0
        for buf in _buffers:
            if _refs(buf) == _IDLE_REFS:
                idle += buf.nbytes
        buf = None
        return len(_buffers), sum(b.nbytes for b in _buffers), idle


def clear():
    """Forget every idle buffer (arrays still held by callers stay valid: they own a reference)."""
    with _lock:
        keep = []
        for buf in _buffers:
            if _refs(buf) != _IDLE_REFS:
                keep.append(buf)
        buf = None
        _buffers[:] = keep
