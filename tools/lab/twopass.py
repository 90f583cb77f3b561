#!/usr/bin/env python3
"""Two passes interleaved tile by tile in one launch (lars_d_probe kinds 60-63, liblars_lab.so): is a tile's second read served by
the Infinity Cache when the first one ran one tile period ahead?  us per 4096 x 4096 tile."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lablib
from lars_image_processing_amd import _ffi


def main():
    ntiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tb = 50331648
    src, dst = _ffi.DeviceBuffer(ntiles * tb), _ffi.DeviceBuffer(64 * 4 * tb)
    lablib.probe(3, 1, 8192, None, src.ptr, ntiles * tb)          # touch the source
    lablib.probe(3, 1, 8192, None, dst.ptr, 64 * 4 * tb)
    a, b = C.c_void_p(), C.c_void_p()
    _ffi.call("lars_event_create", C.byref(a)); _ffi.call("lars_event_create", C.byref(b))
    names = {60: "interleaved, same tiles (second read one period later)", 61: "interleaved, first read elsewhere (second read cold)",
             62: "plane-writing blocks only", 63: "read-only blocks only",
             64: "interleaved, the sweep of tile t right before its own writing blocks"}
    for nh, nf in ((64, 1024), (256, 1024), (256, 4096), (1024, 4096), (64, 512)):
        for kind in (62, 63, 60, 61, 64, 60, 61, 64):
            ts = []
            for _ in range(4):
                _ffi.call("lars_event_record", a, None)
                lablib.probe(kind, nh, nf, src.ptr, dst.ptr, ntiles * tb)
                _ffi.call("lars_event_record", b, None)
                ms = C.c_float(0); _ffi.call("lars_event_elapsed_ms", a, b, C.byref(ms)); ts.append(ms.value)
            t = float(np.median(ts[1:]))
            print(f"read-only blocks {nh:5d}  writing blocks {nf:5d}  {names[kind]:68s} {t:8.3f} ms  {t / ntiles * 1e3:6.2f} us per tile", flush=True)


if __name__ == "__main__":
    main()
