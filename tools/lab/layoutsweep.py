#!/usr/bin/env python3
"""Plane placements inside ONE allocation: planes of 64 tile slots (4 GiB) at chosen offsets (GiB) of a 24 GiB allocation, launches of
64 tiles as in the product (only pointers change).  Is a placement with two planes 16 GiB apart fast in every allocation?

    python tools/lab/layoutsweep.py [--allocations 6] [--gib 24]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402

IDX = ("NDVI", "GNDVI", "NDWI")
GIB = 1 << 30
LAYOUTS = [(0, 4, 8), (0, 4, 16), (0, 8, 16), (0, 4, 20), (0, 10, 20), (4, 8, 20), (0, 12, 16), (2, 6, 18), (0, 4, 12), (0, 6, 12), (0, 4, 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--allocations", type=int, default=6)
    ap.add_argument("--gib", type=int, default=24)
    ap.add_argument("--tiles", type=int, default=1024)
    ap.add_argument("--hold", action="store_true", help="keep every allocation (so that the next one comes from other memory)")
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    G = 64
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def level(base, offs):
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(IDX, True, stats, False, None, None, st, min(G, b.ntiles - st), raw=True)
            for k in range(3):
                a.out_index[k] = base + offs[k] * GIB
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

    print(f"# {args.gib} GiB allocations, planes of 4 GiB at (o0, o1, o2) GiB, ms per 64-tile launch")
    held = []
    free_b, total_b = C.c_size_t(), C.c_size_t()
    for n in range(args.allocations):
        _ffi.call("lars_mem_info", C.byref(free_b), C.byref(total_b))
        if free_b.value < args.gib * GIB + (8 << 30):
            break
        big = _ffi.DeviceBuffer(args.gib * GIB)
        row = [f"{o}: {level(big.ptr, o):.3f}" for o in LAYOUTS if max(o) + 4 <= args.gib]
        print(f"allocation {n} @ {big.ptr:#x}:  " + "  ".join(row), flush=True)
        if args.hold:
            held.append(big)
        else:
            big.free()
            _ffi.call("lars_synchronize", None)


if __name__ == "__main__":
    main()
