#!/usr/bin/env python3
"""Is an output arena fast or slow on its own, or only relative to the input it is paired with?  Two input batches and
several output arenas, all allocated up front; the plane-writing kernel for every (input, arena) pair, interleaved;
then the slowest arena is freed and allocated again.

    python tools/pairbench.py [tiles=256] [rounds=3] [arenas=8]
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
    narena = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    idx = ("NDVI", "GNDVI", "NDWI")
    batches = [lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234 + k, profile="vegetation") for k in range(2)]
    for b in batches:
        b.compute_wb_tables()
    b0 = batches[0]
    stats = b0.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    slots = 64
    plane = slots * b0.npix * 4
    arenas = [_ffi.DeviceBuffer(3 * plane) for _ in range(narena)]
    outs = [b.make_outputs(index=False, ring=slots) for b in batches]
    _ffi.set_tuning(traverse=1)

    def run(bi, arena):
        b, o = batches[bi], outs[bi]
        o.index = [View(arena.ptr + k * plane, plane) for k in range(3)]
        _ffi.call("lars_event_record", ev[0], None)
        for start in range(0, b.ntiles, slots):
            b.run_fused(b.fused_args(idx, True, stats, False, o, None, start, slots))
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        o.index = [None] * 3
        return tiles * b.npix * 15 / ms.value / 1e6

    def sweep(label):
        t = {(bi, ai): [] for bi in range(2) for ai in range(len(arenas))}
        for r in range(rounds + 1):
            for ai, arena in enumerate(arenas):
                for bi in range(2):
                    t[(bi, ai)].append(run(bi, arena))
        print(f"# {label}: GB/s of the plane-writing kernel (3 planes + statistics), input batch 0 at {batches[0].tiles.ptr:#x}, 1 at {batches[1].tiles.ptr:#x}")
        out = {}
        for ai, arena in enumerate(arenas):
            a, b = float(np.median(t[(0, ai)][1:])), float(np.median(t[(1, ai)][1:]))
            out[ai] = (a, b)
            print(f"arena {ai} at {arena.ptr:#x}: input 0 -> {a:7.1f} ({a / 8000:.3f})   input 1 -> {b:7.1f} ({b / 8000:.3f})")
        return out

    free0, total = C.c_size_t(), C.c_size_t()
    _ffi.call("lars_mem_info", C.byref(free0), C.byref(total))
    print(f"free device memory with everything allocated: {free0.value >> 30} of {total.value >> 30} GiB")
    first = sweep("all arenas allocated up front")
    slow = min(first, key=lambda k: first[k][0])
    print(f"# freeing arena {slow} (the slowest) and allocating it again")
    arenas[slow].free()
    arenas[slow] = _ffi.DeviceBuffer(3 * plane)
    second = sweep("after re-allocating the slowest arena")
    _ffi.set_tuning(traverse=-1)
    print(json.dumps({"first": first, "second": second}))


if __name__ == "__main__":
    main()
