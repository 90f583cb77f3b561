#!/usr/bin/env python3
"""How far apart must the three planes of an arena lie for it to be fast?  (product library, no knob: only pointers and launch sizes)

Tile slots in groups of G; the three planes of group j either planar (plane k at k x 4 GiB + j x G x 64 MiB: the product's layout) or
grouped (group j's planes next to each other: plane k at (3 j + k) x G x 64 MiB, i.e. G x 64 MiB between simultaneously written
regions).  Both as launches of G tiles, so only the layout differs.  Fastest two and slowest of the candidate arenas.

    python tools/lab/groupsweep.py [--candidates 10]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402
from lars_image_processing_amd.batch import BatchOutputs  # noqa: E402

IDX = ("NDVI", "GNDVI", "NDWI")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=10)
    ap.add_argument("--tiles", type=int, default=1024)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    outs = BatchOutputs(b, IDX, True, False, False, 64, allocate=False)
    nbytes = 3 * outs.plane_bytes
    tile_b = b.npix * 4
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def launches(arena, G, grouped):
        outs.adopt_arena(arena)
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(IDX, True, stats, False, None, None, st, min(G, b.ntiles - st), raw=True)
            slot = st % 64
            j = slot // G
            for k in range(3):
                a.out_index[k] = arena.ptr + ((3 * j + k) * G * tile_b if grouped else k * outs.plane_bytes + slot * tile_b)
            ls.append(a)
        return ls

    def level(ls):
        out = []
        for _ in range(2):
            _ffi.call("lars_event_record", ev[0], None)
            for a in ls:
                b.run_fused(a)
            _ffi.call("lars_event_record", ev[1], None)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            out.append(ms.value * 64.0 / b.ntiles)                  # ms per 64 tiles
        return out[1]

    arenas = [_ffi.DeviceBuffer(nbytes) for _ in range(args.candidates)]
    planar = [level(launches(a, 64, False)) for a in arenas]
    print("planar levels, launches of 64 tiles:", " ".join(f"{t:.3f}" for t in planar), flush=True)
    order = np.argsort(planar)
    for j in sorted({int(order[0]), int(order[1]), int(order[-1])}):
        row = []
        for G in (64, 32, 16, 8, 4):
            p = level(launches(arenas[j], G, False))
            g = level(launches(arenas[j], G, True)) if G < 64 else p
            row.append(f"G={G:2d} ({G * 64:4d} MiB apart): planar {p:.3f} grouped {g:.3f}")
        print(f"arena {j} ({planar[j]:.3f}):  " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
