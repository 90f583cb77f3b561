#!/usr/bin/env python3
"""Does the placement of the INPUT tiles matter the way the placement of the output arena does?  `n` batches of 256 tiles
(12 GiB each, plain hipMalloc), the read-only histogram pass and the statistics-only kernels on each, interleaved.

    python tools/inbench.py [batches=10] [rounds=4]
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tiles = 256
    batches = [lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation") for _ in range(n)]
    for b in batches:
        b.compute_wb_tables()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def timed(fn):
        _ffi.call("lars_event_record", ev[0], None)
        fn()
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        return ms.value

    stats = batches[0].new_stats()
    t = np.zeros((n, 2, rounds + 1))
    for r in range(rounds + 1):
        for i, b in enumerate(batches):
            t[i, 0, r] = timed(lambda: _ffi.call("lars_d_channel_hist", C.c_void_p(b.tiles.ptr), b.ntiles, b.npix, 3, _ffi.U8,
                                                 C.c_void_p(b.hist.ptr), None))
            t[i, 1, r] = timed(lambda: b.run_fused(b.fused_args(("NDVI",), True, stats, False, None, None, 0, tiles)))
    nbytes = tiles * 4096 * 4096 * 3
    for i, b in enumerate(batches):
        h, f = np.median(t[i, 0, 1:]), np.median(t[i, 1, 1:])
        print(f"input batch at {b.tiles.ptr:#x}: histogram pass {h:6.3f} ms = {nbytes / h / 1e6:6.0f} GB/s    NDVI statistics only {f:6.3f} ms = {nbytes / f / 1e6:6.0f} GB/s")


if __name__ == "__main__":
    main()
