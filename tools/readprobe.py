#!/usr/bin/env python3
"""Read-only streaming shapes: grid-strided 12-byte loads (the histogram pass's shape, with 1, 4 or 8 loads in flight) against
wave runs of 1 / 4 / 8 x 768 contiguous bytes.  One process, interleaved."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars

b = lars.TileBatch.synthetic(256, 4096, 4096, seed=1234, profile="vegetation")
nbytes = 256 * b.tile_bytes
ev = [C.c_void_p(), C.c_void_p()]
for e in ev:
    _ffi.call("lars_event_create", C.byref(e))
kinds = [(1, 1, "grid stride, 1 load in flight"), (1, 4, "grid stride, 4 loads in flight"), (1, 8, "grid stride, 8 loads in flight"),
         (25, 1, "wave run of 1 x 768 B"), (23, 1, "wave run of 4 x 768 B"), (24, 1, "wave run of 8 x 768 B")]
times = {}
for r in range(6):
    for kind, unroll, name in kinds:
        for blocks in (4096, 16384, 65536):
            _ffi.call("lars_event_record", ev[0], None)
            _ffi.call("lars_d_probe", kind, unroll, blocks, C.c_void_p(b.tiles.ptr), C.c_void_p(b.tiles.ptr), nbytes, None)
            _ffi.call("lars_event_record", ev[1], None)
            _ffi.call("lars_synchronize", None)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            times.setdefault((name, blocks), []).append(ms.value)
b.compute_wb_tables()
hist = []
for r in range(6):
    _ffi.call("lars_event_record", ev[0], None)
    _ffi.call("lars_d_channel_hist", C.c_void_p(b.tiles.ptr), b.ntiles, b.npix, 3, _ffi.U8, C.c_void_p(b.hist.ptr), None)
    _ffi.call("lars_event_record", ev[1], None)
    _ffi.call("lars_synchronize", None)
    ms = C.c_float(0)
    _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
    hist.append(ms.value)
for (name, blocks), t in times.items():
    print(f"{name:34s} blocks={blocks:6d}  {nbytes / float(np.median(t[1:])) / 1e6:7.1f} GB/s")
print(f"k_chan_hist_u8c3_v2 (the pre-pass itself)                {nbytes / float(np.median(hist[1:])) / 1e6:7.1f} GB/s")
