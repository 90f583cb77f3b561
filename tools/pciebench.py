#!/usr/bin/env python3
"""Host <-> device copies of the host entry points by size: lars_memcpy_h2d / _d2h (hipMemcpy from / to pageable NumPy memory) --
what a transfer of the sizes the reference's callers use actually gets.

    python tools/pciebench.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi  # noqa: E402


def main():
    lab = None
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
        import lablib as lab
        lab.load()
    except (ImportError, OSError, AttributeError):
        lab = None
    dev = _ffi.DeviceBuffer(256 << 20)
    print(f"{'bytes':>12s} {'h2d ms':>9s} {'GB/s':>7s} {'d2h ms (same array)':>20s} {'GB/s':>7s} {'d2h ms (fresh array)':>21s} {'GB/s':>7s}")
    for kib in (64, 256, 768, 1024, 3072, 4096, 12288, 16384, 49152, 65536, 196608):
        n = kib << 10
        src = np.random.default_rng(1).integers(0, 255, n, dtype=np.uint8)
        dst = np.empty(n, np.uint8)
        reps = 20 if n < (32 << 20) else 6

        def t(fn):
            fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            return float(np.median(ts))
        h2d = t(lambda: _ffi.call("lars_memcpy_h2d", C.c_void_p(dev.ptr), _ffi.ptr(src), n))
        d2h = t(lambda: _ffi.call("lars_memcpy_d2h", _ffi.ptr(dst), C.c_void_p(dev.ptr), n))

        def fresh():
            out = np.empty(n, np.uint8)
            _ffi.call("lars_memcpy_d2h", _ffi.ptr(out), C.c_void_p(dev.ptr), n)
            return out
        d2hf = t(fresh)
        line = f"{n:12d} {h2d * 1e3:9.3f} {n / h2d / 1e9:7.1f} {d2h * 1e3:20.3f} {n / d2h / 1e9:7.1f} {d2hf * 1e3:21.3f} {n / d2hf / 1e9:7.1f}"
        if lab is not None:
            a_h2d = t(lambda: lab.call("lars_lab_copy", 1, 1, _ffi.ptr(src), C.c_void_p(dev.ptr), n))
            a_d2h = t(lambda: lab.call("lars_lab_copy", 1, 0, _ffi.ptr(dst), C.c_void_p(dev.ptr), n))
            line += f"   async + sync (what the host entry points did): h2d {a_h2d * 1e3:7.3f} ms {n / a_h2d / 1e9:5.1f} GB/s  d2h {a_d2h * 1e3:7.3f} ms {n / a_d2h / 1e9:5.1f} GB/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
