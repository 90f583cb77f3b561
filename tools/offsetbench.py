#!/usr/bin/env python3
"""Does the speed of the plane-writing kernel depend on the OFFSETS between its three output planes inside one
allocation?  One arena, plane k at k * (plane + d): sweep d.  (tools/allocbench.py showed separate allocations falling
into a fast and a slow class; tools/membench.py that no single allocation is slow on its own.)

    python tools/offsetbench.py [tiles=256] [rounds=3] [arenas=2]
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars


class View:
    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def free(self):
        pass


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    arenas = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    idx = ("NDVI", "GNDVI", "NDWI")
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    slots = 64
    plane = slots * b.npix * 4
    K, M, G = 1 << 10, 1 << 20, 1 << 30
    deltas = [0, 4 * K, 8 * K, 16 * K, 32 * K, 64 * K, 128 * K, 256 * K, 512 * K, 1 * M, 2 * M, 3 * M, 4 * M, 8 * M, 16 * M, 36 * M,
              64 * M, 100 * M, 128 * M, 256 * M, 341 * M, 512 * M, 1 * G]
    outs = b.make_outputs(index=False, ring=slots)
    _ffi.set_tuning(traverse=1)
    res = {}
    for a in range(arenas):
        arena = _ffi.DeviceBuffer(3 * plane + 2 * max(deltas) + 4096)
        times = {d: [] for d in deltas}
        for r in range(rounds + 1):
            for d in deltas:
                outs.index = [View(arena.ptr + k * (plane + d), plane) for k in range(3)]
                _ffi.call("lars_event_record", ev[0], None)
                for start in range(0, b.ntiles, slots):
                    b.run_fused(b.fused_args(idx, True, stats, False, outs, None, start, slots))
                _ffi.call("lars_event_record", ev[1], None)
                _ffi.call("lars_synchronize", None)
                ms = C.c_float(0)
                _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
                times[d].append(ms.value)
        print(f"arena {a} at {arena.ptr:#x} ({arena.nbytes >> 20} MiB); plane k at k * (4 GiB + d)")
        for d in deltas:
            gbs = tiles * b.npix * 15 / float(np.median(times[d][1:])) / 1e6
            res[f"arena{a} d={d}"] = gbs
            print(f"  d = {d:>11d} B ({d / M:9.3f} MiB)   {gbs:7.1f} GB/s  {gbs / 8000:.3f}")
        outs.index = [None] * 3
        keep = arena           # keep the arena allocated while the next one is made, so the next lands elsewhere
    _ffi.set_tuning(traverse=-1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
