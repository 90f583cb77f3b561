#!/usr/bin/env python3
"""Output arenas from plain hipMalloc against arenas built with the virtual-memory-management API (LARS_MALLOC_KIND=3:
one address range, physical memory created in chunks of LARS_VMM_CHUNK_MB and mapped back to back): does the way the
physical memory is obtained decide an arena's speed class?  One input batch (plain hipMalloc), `n`
arenas per variant, all allocated up front, the plane-writing kernel on every arena, interleaved.

    python tools/vmmbench.py [tiles=256] [rounds=3] [arenas_per_variant=4]
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
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    idx = ("NDVI", "GNDVI", "NDWI")
    os.environ.pop("LARS_MALLOC_KIND", None)
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    slots = 64
    plane = slots * b.npix * 4
    out = b.make_outputs(index=False, ring=slots)
    spec = os.environ.get("VMM_VARIANTS", "malloc,0,1024,64,2")
    # "malloc", "contig", or "<chunk MiB>" / "<chunk MiB>a<alignment MiB of the address range>" / "<chunk MiB>s" (shuffled)
    variants = [("hipMalloc", None, None) if v == "malloc" else ("contiguous", "4", "0") if v == "contig" else
                (("vmm, one handle" if v == "0" else f"vmm, {v} MiB chunks"), "3", v) for v in spec.split(",")]
    arenas = []
    for name, kind, chunk in variants:
        for k in range(per):
            if kind is None:
                os.environ.pop("LARS_MALLOC_KIND", None)
            else:
                os.environ["LARS_MALLOC_KIND"] = kind
                os.environ["LARS_VMM_CHUNK_MB"] = chunk.split("a")[0].rstrip("s")
                os.environ["LARS_VMM_SHUFFLE"] = "1" if chunk.endswith("s") else "0"
                os.environ["LARS_VMM_ALIGN_MB"] = chunk.split("a")[1] if "a" in chunk else "0"
            try:
                arenas.append((name, _ffi.DeviceBuffer(3 * plane)))
            except Exception as exc:                      # noqa: BLE001
                print(f"# {name}: allocation failed: {exc}")
    os.environ.pop("LARS_MALLOC_KIND", None)

    def run(arena):
        out.index = [View(arena.ptr + k * plane, plane) for k in range(3)]
        _ffi.call("lars_event_record", ev[0], None)
        for start in range(0, b.ntiles, slots):
            b.run_fused(b.fused_args(idx, True, stats, False, out, None, start, slots))
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        out.index = [None] * 3
        return tiles * b.npix * 15 / ms.value / 1e6

    knob = os.environ.get("VMM_KNOB", "nt_stores")                 # the tuning key switched on for the second column
    t = {(i, nt): [] for i in range(len(arenas)) for nt in (0, 1)}
    for r in range(rounds + 1):
        for i, (name, arena) in enumerate(arenas):
            for nt in (0, 1):
                _ffi.set_tuning(**{knob: nt})
                t[(i, nt)].append(run(arena))
    _ffi.set_tuning(**{knob: 0})
    for i, (name, arena) in enumerate(arenas):
        g, gn = float(np.median(t[(i, 0)][1:])), float(np.median(t[(i, 1)][1:]))
        print(f"{name:22s} arena at {arena.ptr:#x}: {g:7.1f} GB/s ({g / 8000:.3f})   with {knob}=1 {gn:7.1f} GB/s ({gn / 8000:.3f})")


if __name__ == "__main__":
    main()
