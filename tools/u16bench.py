#!/usr/bin/env python3
"""BASELINE configs[4] shape on one GPU: uint16 8192x8192 tiles, white balance + float32 NDVI + RdYlGn RGBA."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars
from lars_image_processing_amd import _ffi

def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    edge = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    b = lars.TileBatch(tiles, edge, edge, 3, np.uint16)
    # random 16-bit samples: the byte generator over twice the bytes
    _ffi.call("lars_d_synth_u8", C.c_void_p(b.tiles.ptr), tiles, 0, b.npix * 2, 3, 1234, 0, None)
    _ffi.call("lars_synchronize", None)
    a, e = C.c_void_p(), C.c_void_p()
    _ffi.call("lars_event_create", C.byref(a)); _ffi.call("lars_event_create", C.byref(e))
    def timed(fn, n=4):
        ts = []
        for _ in range(n):
            _ffi.call("lars_event_record", a, None); fn(); _ffi.call("lars_event_record", e, None)
            ms = C.c_float(0); _ffi.call("lars_event_elapsed_ms", a, e, C.byref(ms)); ts.append(ms.value)
        return float(np.median(ts[1:]))
    npix = tiles * edge * edge
    res = {}
    for impl, name in ((5, "wb_prepare (one full pass on VALUE windows: counts below + histograms inside; round 5)"),
                       (1, "wb_prepare (2 radix passes + tables)"),
                       (3, "wb_prepare (windows that miss on purpose: full pass + the two radix passes)"),
                       (50, "wb_prepare (value windows once more)")):
        impl = 5 if impl == 50 else impl
        _ffi.set_tuning(u16_hist_impl=impl)
        t = timed(lambda: b.compute_wb_tables())
        res[name] = {"ms": t, "GBs_input_once": npix * 6 / t / 1e6, "frac_8TBs": npix * 6 / t / 1e6 / 8000}
    _ffi.set_tuning(u16_hist_impl=5)
    stats = b.new_stats()
    outs = b.make_outputs(indices=("NDVI",), index=True, rgba=True)
    for name, kw, bpp in (("ndvi_f32+rgba+stats (configs[4])", dict(indices=("NDVI",), outputs=outs), 14),
                          ("ndvi stats only", dict(indices=("NDVI",), outputs=None), 6),
                          ("3idx stats only", dict(indices=("NDVI", "GNDVI", "NDWI"), outputs=None), 6)):
        t = timed(lambda: b.run_fused(b.fused_args(kw["indices"], True, stats, False, kw["outputs"])))
        res[name] = {"ms": t, "GBs": npix * bpp / t / 1e6, "frac_8TBs": npix * bpp / t / 1e6 / 8000, "Gpix_s": npix / t / 1e6}
    for k, v in res.items():
        print(f"{k:96s} " + "  ".join(f"{kk}={vv:9.3f}" for kk, vv in v.items()))
    print(json.dumps(res))

if __name__ == "__main__":
    main()
