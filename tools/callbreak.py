#!/usr/bin/env python3
"""Where the time of one small host call goes: calculate_index on a 1024 x 1024 image, piece by piece (laboratory copies vs the entry point)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402


def med(fn, n=40):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def main():
    for edge in (512, 1024, 2048):
        rng = np.random.default_rng(1)
        img = rng.integers(0, 255, (edge, edge, 3), dtype=np.uint8)
        out = np.empty((edge, edge), np.float32)
        dev = _ffi.DeviceBuffer(img.nbytes + out.nbytes + 4096)
        outs = [out, None, None]
        p3 = _ffi.ptr3(outs)
        t_call = med(lambda: _ffi.call("lars_h_calculate_index", _ffi.ptr(img), edge, edge, 3, _ffi.U8, 1, C.byref(p3), None, 0))
        t_api = med(lambda: lars.calculate_index(img, "NDVI"))
        t_h2d = med(lambda: _ffi.call("lars_memcpy_h2d", C.c_void_p(dev.ptr), _ffi.ptr(img), img.nbytes))
        t_d2h = med(lambda: _ffi.call("lars_memcpy_d2h", _ffi.ptr(out), C.c_void_p(dev.ptr), out.nbytes))
        t_empty = med(lambda: np.empty((edge, edge), np.float32))

        def fresh_d2h():
            o = np.empty((edge, edge), np.float32)
            _ffi.call("lars_memcpy_d2h", _ffi.ptr(o), C.c_void_p(dev.ptr), o.nbytes)
        t_fresh = med(fresh_d2h)
        b = lars.TileBatch.from_host(img[None])
        o = b.make_outputs(indices=("NDVI",), index=True)
        a = b.fused_args(("NDVI",), False, None, False, o)

        def kern():
            b.run_fused(a)
            _ffi.call("lars_synchronize", None)
        t_kern = med(kern)
        print(f"edge {edge}: api {t_api:.3f}  C entry point (reused result array) {t_call:.3f}  =? h2d {t_h2d:.3f} + kernel launch+sync {t_kern:.3f} + d2h {t_d2h:.3f}"
              f"   (np.empty {t_empty:.4f}, d2h into a fresh array {t_fresh:.3f})", flush=True)


if __name__ == "__main__":
    main()
