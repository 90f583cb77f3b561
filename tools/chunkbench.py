#!/usr/bin/env python3
"""Is an arena's speed the blend of its parts?  Arenas built from 64 MiB physical chunks (LARS_MALLOC_KIND=3) and from plain
hipMalloc; the plane-writing kernel on groups of 8 tiles (24 chunks of 64 MiB: one per tile and plane), every group of
every arena several times, interleaved.  Prints GB/s per group: stable differences between the groups of one arena would
mean that single chunks are fast or slow.

    python tools/chunkbench.py [arenas_per_kind=3] [rounds=5]
"""
import ctypes as C, os, sys
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
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    group = int(os.environ.get("GROUP", "8"))
    idx = ("NDVI", "GNDVI", "NDWI")
    os.environ.pop("LARS_MALLOC_KIND", None)
    slots = 64
    b = lars.TileBatch.synthetic(slots, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    plane = slots * b.npix * 4
    out = b.make_outputs(index=False, ring=slots)
    arenas = []
    nmalloc = int(os.environ.get("NMALLOC", str(per)))
    for name, kind in (("vmm 64 MiB", "3"), ("hipMalloc", None)):
        for k in range(per if kind else nmalloc):
            if kind:
                os.environ["LARS_MALLOC_KIND"] = kind
                os.environ["LARS_VMM_CHUNK_MB"] = "64"
            else:
                os.environ.pop("LARS_MALLOC_KIND", None)
            arenas.append((name, _ffi.DeviceBuffer(3 * plane)))
    os.environ.pop("LARS_MALLOC_KIND", None)

    def run(arena, start, count):
        out.index = [View(arena.ptr + k * plane, plane) for k in range(3)]
        _ffi.call("lars_event_record", ev[0], None)
        for _ in range(4):
            b.run_fused(b.fused_args(idx, True, stats, False, out, None, start, count))
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        out.index = [None] * 3
        return 4 * count * b.npix * 15 / ms.value / 1e6

    ngroups = slots // group
    t = np.zeros((len(arenas), ngroups + 1, rounds + 1))
    for r in range(rounds + 1):
        for i, (name, arena) in enumerate(arenas):
            for g in range(ngroups):
                t[i, g, r] = run(arena, g * group, group)
            t[i, ngroups, r] = run(arena, 0, slots)
    for i, (name, arena) in enumerate(arenas):
        med = np.median(t[i, :, 1:], axis=1)
        spread = (t[i, :ngroups, 1:].max(axis=1) - t[i, :ngroups, 1:].min(axis=1)).max()
        print(f"{name:11s} {arena.ptr:#x}: whole arena {med[ngroups]:6.0f} GB/s; groups of {group} tiles " +
              " ".join(f"{v:5.0f}" for v in med[:ngroups]) + f"   (largest spread of one group over the rounds {spread:.0f})")


if __name__ == "__main__":
    main()
