#!/usr/bin/env python3
"""Which parts of an allocation disturb each other under the plane-writing launch?  Planes of 32 tile slots (2 GiB): plane 0 at the
start, plane 2 at the end, plane 1 walks through the allocation in steps of 1 GiB; launches of 32 tiles (product library; planes of
16 slots and launches of 16 tiles show nothing: 2.79-2.88 ms everywhere).

    python tools/lab/regionmap.py [--gib 24] [--allocations 4]
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
    ap.add_argument("--gib", type=int, default=24)
    ap.add_argument("--allocations", type=int, default=4)
    ap.add_argument("--tiles", type=int, default=512)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    G = 32
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def level(base, segs):
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(IDX, True, stats, False, None, None, st, min(G, b.ntiles - st), raw=True)
            for k in range(3):
                a.out_index[k] = base + segs[k] * GIB
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

    n = args.gib
    print(f"# {n} GiB allocations; planes of 2 GiB: plane 0 at 0, plane 2 at {n - 2} GiB, plane 1 at s GiB; ms per 64 tiles (launches of 32)")
    for a in range(args.allocations):
        big = _ffi.DeviceBuffer(n * GIB)
        row = [f"{level(big.ptr, (0, s, n - 2)):.2f}" for s in range(2, n - 3)]
        print(f"allocation {a} @ {big.ptr:#x}: plane 1 at s=2.. " + " ".join(row), flush=True)
        mid = n // 2
        row = [f"{level(big.ptr, (s, mid, n - 2)):.2f}" if abs(s - mid) >= 2 else " -- " for s in range(0, n - 3)]
        print(f"             plane 1 at {mid}, plane 0 at s=0.. " + " ".join(row), flush=True)
        big.free()
        _ffi.call("lars_synchronize", None)


if __name__ == "__main__":
    main()
