#!/usr/bin/env python3
"""Which kind of memory successive allocations of a process come from: N blocks of 4 GiB, all held; the plane-writing launch with its first two
planes in blocks 0 and 1 and the third in block k is of the fast class exactly when block k is of another kind than blocks 0 and 1 (or they two
differ).  Meant to run as the FIRST process on a fresh box and again right after, to see whether a fresh device hands out one kind for long.

    python tools/lab/kindmap.py [--blocks 48] [--tiles 1024]
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
    ap.add_argument("--blocks", type=int, default=48)
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

    def level(ptrs, passes=2):
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(IDX, True, stats, False, None, None, st, min(G, b.ntiles - st), raw=True)
            for k in range(3):
                a.out_index[k] = ptrs[k]
            ls.append(a)
        out = []
        for _ in range(passes):
            _ffi.call("lars_event_record", ev[0], None)
            for a in ls:
                b.run_fused(a)
            _ffi.call("lars_event_record", ev[1], None)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            out.append(ms.value * 64.0 / b.ntiles)
        return out[-1]

    free_b, total_b = C.c_size_t(), C.c_size_t()
    blocks = []
    for n in range(args.blocks):
        _ffi.call("lars_mem_info", C.byref(free_b), C.byref(total_b))
        if free_b.value < (12 << 30):
            break
        blocks.append(_ffi.DeviceBuffer(4 * GIB))
    print(f"# batch at {b.tiles.ptr:#x} ({b.tiles.nbytes / GIB:.1f} GiB); {len(blocks)} blocks of 4 GiB held; ms per 64-tile launch with planes in blocks (0, 1, k)")
    level([blocks[0].ptr, blocks[1].ptr, blocks[2].ptr], passes=3)                 # warm-up
    row = []
    for k in range(2, len(blocks)):
        row.append((k, blocks[k].ptr, level([blocks[0].ptr, blocks[1].ptr, blocks[k].ptr])))
    for k, ptr, ms in row:
        print(f"k = {k:2d} @ {ptr:#x} (+{(blocks[0].ptr - ptr) / GIB:7.1f} GiB below block 0): {ms:.3f} {'fast' if ms < 2.75 else 'SLOW'}", flush=True)
    print("# planes in blocks (k, k + 1, k + 2)")
    print("  ".join(f"{k}: {level([blocks[k].ptr, blocks[k + 1].ptr, blocks[k + 2].ptr]):.3f}" for k in range(0, len(blocks) - 2, 3)))


if __name__ == "__main__":
    main()
