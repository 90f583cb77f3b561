#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache serve the fused pass if it follows the histogram pass closely?

Times hist -> table -> fused over the batch in groups of G tiles (same stream), statistics only.
"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars
from lars_image_processing_amd import _ffi

def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    a, e = C.c_void_p(), C.c_void_p()
    _ffi.call("lars_event_create", C.byref(a)); _ffi.call("lars_event_create", C.byref(e))
    npix = tiles * 4096 * 4096
    for indices in (("NDVI",), ("NDVI", "GNDVI", "NDWI")):
        for G in (1, 2, 3, 4, 5, 8, 16, tiles):
            ts = []
            for _ in range(4):
                _ffi.call("lars_event_record", a, None)
                for s in range(0, tiles, G):
                    n = min(G, tiles - s)
                    _ffi.call("lars_d_channel_hist", C.c_void_p(b.tiles.ptr + s * b.tile_bytes), n, b.npix, 3, _ffi.U8,
                              C.c_void_p(b.hist.ptr + s * 768 * 4), None)
                    _ffi.call("lars_d_wb_table", C.c_void_p(b.hist.ptr + s * 768 * 4), n, b.npix, _ffi.U8,
                              C.c_void_p(b.table.ptr + s * 768), C.c_void_p(b.percentiles.ptr + s * 48), 0, None)
                    b.run_fused(b.fused_args(indices, True, stats, False, None, None, s, n))
                _ffi.call("lars_event_record", e, None)
                ms = C.c_float(0); _ffi.call("lars_event_elapsed_ms", a, e, C.byref(ms)); ts.append(ms.value)
            t = float(np.median(ts[1:]))
            print(f"{'+'.join(indices):16s} group={G:5d}  {t:8.3f} ms  {npix/t/1e6:9.1f} Gpix/s  "
                  f"{npix*3/t/1e6:8.1f} GB/s algorithmic (input once)")

if __name__ == "__main__":
    main()
