#!/usr/bin/env python3
"""Does the pipeline's second read of a tile come out of the Infinity Cache?  The same launch with the fused items reading
(a) their own tile, counted one segment earlier, and (b) a tile half a launch away (cold), interleaved."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars

b = lars.TileBatch.synthetic(64, 4096, 4096, seed=1234, profile="vegetation")
stats = b.new_stats()
outs = b.make_outputs(index=True)
ev = [C.c_void_p(), C.c_void_p()]
for e in ev:
    _ffi.call("lars_event_create", C.byref(e))
for spi, head in ((64, 16), (32, 16), (128, 8)):
    times = {0: [], 1: []}
    for r in range(7):
        for cold in (0, 1):
            _ffi.set_tuning(pipe_steps=spi, pipe_head=head, pipe_cold=cold)
            _ffi.call("lars_event_record", ev[0], None)
            b.run_pipeline(stats, outs)
            _ffi.call("lars_event_record", ev[1], None)
            _ffi.call("lars_synchronize", None)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            times[cold].append(ms.value)
    a, c = float(np.median(times[0][1:])), float(np.median(times[1][1:]))
    print(f"spi={spi} head={head}: own tile {a:.3f} ms, cold tile {c:.3f} ms per 64 tiles ({c / a:.3f}x)")
_ffi.set_tuning(pipe_steps=0, pipe_head=0, pipe_cold=0)
