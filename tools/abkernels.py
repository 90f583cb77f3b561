#!/usr/bin/env python3
"""A/B of kernel BUILDS in one process against the same memory: the batch, its tables and the output arena come from the product
library; every variant library (a liblars_hip.so built with other flags) is loaded next to it and its lars_d_fused is launched
with the same argument block.  Output arenas differ by up to 15 % between allocations, so builds can only be compared like this.

    python tools/abkernels.py --case u16 build/variants/liblars_a.so build/variants/liblars_b.so
cases: u8 (three planes + statistics), u8planes (three planes, no statistics), u8ndvi (one plane + statistics), u8hist (three planes +
histograms), u8wb (white-balanced image only), rgba8 (RGBA tiles, three planes + statistics), u16 (uint16 8192 x 8192: NDVI + RGBA + statistics)
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402

CASES = {
    # name: (dtype, channels, edge, tiles, ring, indices, planes, rgba, wb image, statistics, histograms, bytes per pixel)
    "u8": (np.uint8, 3, 4096, 256, 64, ("NDVI", "GNDVI", "NDWI"), True, False, False, True, False, 15),
    "u8planes": (np.uint8, 3, 4096, 256, 64, ("NDVI", "GNDVI", "NDWI"), True, False, False, False, False, 15),
    "u8ndvi": (np.uint8, 3, 4096, 256, 64, ("NDVI",), True, False, False, True, False, 7),
    "u8hist": (np.uint8, 3, 4096, 256, 64, ("NDVI", "GNDVI", "NDWI"), True, False, False, True, True, 15),
    "u8wb": (np.uint8, 3, 4096, 256, 64, (), False, False, True, False, False, 6),
    "rgba8": (np.uint8, 4, 4096, 192, 64, ("NDVI", "GNDVI", "NDWI"), True, False, False, True, False, 16),
    "u16": (np.uint16, 3, 8192, 32, 16, ("NDVI",), True, True, False, True, False, 14),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--case", default="u8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--arena", default="auto", choices=["auto", "plain"], help="plain: the first allocation as it comes (often of the slow kind)")
    args = ap.parse_args()
    dtype, ch, edge, tiles, ring, indices, planes, rgba, wbimg, want_stats, hist, bpp = CASES[args.case]
    b = lars.TileBatch(tiles, edge, edge, ch, dtype)
    _ffi.call("lars_d_synth_u8", C.c_void_p(b.tiles.ptr), tiles, 0, b.npix * np.dtype(dtype).itemsize, ch, 1234, 1 if dtype == np.uint8 else 0, None)   # uint16: random bytes
    b.compute_wb_tables()
    outs = b.make_outputs(indices=indices or ("NDVI",), index=planes, rgba=rgba, wb=wbimg, ring=ring, arena=args.arena)
    stats = b.new_stats() if want_stats else None
    print("arena:", outs.arena_report.get("kind"), outs.arena_report.get("chosen_ms"), [round(x, 2) for x in outs.arena_report.get("candidate_ms", [])], flush=True)
    libs = [("product", _ffi.load())]
    for path in args.libs:
        lib = C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)     # its own copies of every symbol, not the product's
        libs.append((os.path.basename(path), lib))
    for _, lib in libs:
        lib.lars_d_fused.restype = C.c_int
        lib.lars_d_fused.argtypes = [C.c_void_p]
        lib.lars_synchronize.restype = C.c_int
        lib.lars_synchronize.argtypes = [C.c_void_p]
    launches = [b.fused_args(indices, True, stats, hist, outs, None, s, min(outs.slots, tiles - s)) for s in range(0, tiles, outs.slots)]
    times = {name: [] for name, _ in libs}
    for _ in range(args.rounds + 1):
        for name, lib in libs:
            lib.lars_synchronize(None)
            _ffi.call("lars_synchronize", None)
            t0 = time.perf_counter()
            for a in launches:
                rc = lib.lars_d_fused(C.byref(a))
                assert rc == 0, (name, rc)
            lib.lars_synchronize(None)
            times[name].append((time.perf_counter() - t0) * 1e3)
    npix = tiles * edge * edge
    for name, t in times.items():
        med = float(np.median(t[1:]))
        print(f"{args.case:9s} {name:28s} {med:8.3f} ms (min {min(t[1:]):8.3f})  {npix * bpp / med / 1e6:7.1f} GB/s = {npix * bpp / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
