#!/usr/bin/env python3
"""Planes of one plane-writing launch in DIFFERENT allocations: when every placement inside single allocations is of the slow class (seen in the
first process on a fresh box: profiles/r05_arena_first_process.txt), does mixing allocations split the write streams between the two kinds?

    python tools/lab/crossalloc.py [--allocations 6] [--gib 12]

Holds all allocations; prints the packed level of each, then the level of (two planes in allocation i, the third in allocation j) for all i != j.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--allocations", type=int, default=6)
    ap.add_argument("--gib", type=int, default=12)
    ap.add_argument("--tiles", type=int, default=1024)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    G = 64
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def level(ptrs):
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(IDX, True, stats, False, None, None, st, min(G, b.ntiles - st), raw=True)
            for k in range(3):
                a.out_index[k] = ptrs[k]
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

    print(f"# batch at {b.tiles.ptr:#x}; {args.gib} GiB allocations, planes of 4 GiB; ms per 64-tile launch")
    allocs = []
    for n in range(args.allocations):
        allocs.append(_ffi.DeviceBuffer(args.gib * GIB))
    for n, a in enumerate(allocs):
        print(f"allocation {n} @ {a.ptr:#x}: packed (0, 4, 8): {level([a.ptr, a.ptr + 4 * GIB, a.ptr + 8 * GIB]):.3f}", flush=True)
    print("# two planes at (0, 4) GiB of allocation i, the third at 0 GiB of allocation j")
    for i, a in enumerate(allocs):
        row = []
        for j, c in enumerate(allocs):
            row.append("  --  " if i == j else f"{level([a.ptr, a.ptr + 4 * GIB, c.ptr]):.3f}")
        print(f"i = {i}: " + "  ".join(row), flush=True)
    print("# one plane each at 0 GiB of allocations (i, i + 1, i + 2)")
    for i in range(len(allocs) - 2):
        print(f"i = {i}: {level([allocs[i].ptr, allocs[i + 1].ptr, allocs[i + 2].ptr]):.3f}", flush=True)


if __name__ == "__main__":
    main()
