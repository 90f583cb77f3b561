#!/usr/bin/env python3
"""Level of the plane-writing launch against the DISTANCE between its three planes: planes of 32 tile slots (2 GiB each) at
0, D and 2 D inside one 24 GiB allocation, launches of 32 tiles (product library, only pointers change).

    python tools/lab/distsweep.py [--allocations 4]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402

IDX = ("NDVI", "GNDVI", "NDWI")
GIB = 1 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--allocations", type=int, default=4)
    ap.add_argument("--tiles", type=int, default=1024)
    ap.add_argument("--slots", type=int, default=32)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    tile_b = b.npix * 4
    G = args.slots
    plane = G * tile_b
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def level(base, dist, start=0):
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(IDX, True, stats, False, None, None, st, min(G, b.ntiles - st), raw=True)
            for k in range(3):
                a.out_index[k] = base + start + k * dist
            ls.append(a)
        out = []
        for _ in range(2):
            _ffi.call("lars_event_record", ev[0], None)
            for a in ls:
                b.run_fused(a)
            _ffi.call("lars_event_record", ev[1], None)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            out.append(ms.value * 64.0 / b.ntiles)
        return out[1]

    total = 24 * GIB
    dists = [2.0, 2.25, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0]
    print(f"# planes of {G} slots ({plane / GIB:.0f} GiB) at 0, D, 2 D of a 24 GiB allocation; ms per 64 tiles, launches of {G} tiles")
    for n in range(args.allocations):
        big = _ffi.DeviceBuffer(total)
        row = [f"{d:g}: {level(big.ptr, int(d * GIB)):.3f}" for d in dists]
        # the same three distances again from another start inside the allocation
        row2 = [f"{d:g}@+1GiB: {level(big.ptr, int(d * GIB), GIB):.3f}" for d in (2.0, 4.0, 8.0)]
        print(f"allocation {n}:  " + "  ".join(row) + "   |  " + "  ".join(row2), flush=True)
        big.free()
        _ffi.call("lars_synchronize", None)


if __name__ == "__main__":
    main()
