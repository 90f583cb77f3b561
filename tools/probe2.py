#!/usr/bin/env python3
"""Round 2: shapes of the plane-writing kernel's traffic mix (12 B read : 48 B written per 4 pixels) next to the
kernel itself, interleaved in one process (allocation placement moves everything by a few per cent, so only
numbers of one process compare).  Several independent destination allocations are tried for the best shapes.

    python tools/probe2.py [tiles=256] [rounds=5]
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars

KINDS = [
    (5, "mix 12B load / 3x16B store, grid stride (round-1 probe)"),
    (6, "  same, non-temporal stores"),
    (8, "(i) lane owns 16 px: 48B contiguous load, 4x16B per plane"),
    (9, "(i') wave owns 1024 px: 3 coalesced 16B loads, stores plane-interleaved"),
    (10, "(iii) wave 1024 px, 16B loads, one plane at a time (4 KiB bursts)"),
    (11, "(iii) wave 1024 px, 12B loads, one plane at a time (4 KiB bursts)"),
    (12, "     wave 1024 px, 12B loads, plane-interleaved"),
    (13, "(ii+iii) workgroup-contiguous slabs, 1024-px steps, 4 KiB bursts"),
    (14, "(ii) workgroup-contiguous slabs, 12B load / 3x16B store"),
    (15, "(iv) 4 loads, then 12 stores (step order)"),
    (16, "(iv) 8 loads, then 24 stores (step order)"),
    (17, "(iv) 4 loads, then 12 stores (plane order)"),
    (18, "(iv) 8 loads, then 24 stores (plane order)"),
    (19, "write side alone: 3 planes, 48B per lane and step (48/60 of the bytes)"),
]


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    nbytes = tiles * b.tile_bytes * 5                      # the mix moves 5x the bytes it reads
    nbytes -= nbytes % (60 * 256 * 4)
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def timed(fn):
        _ffi.call("lars_event_record", ev[0], None)
        fn()
        _ffi.call("lars_event_record", ev[1], None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        return ms.value

    res = {}
    for trial in range(3):                                 # three independent sets of output allocations
        dst = _ffi.DeviceBuffer(nbytes // 5 * 4 + 4096)
        outs = b.make_outputs(index=True, ring=64)
        grids = (16384, 65536, 262144)
        variants = [(k, n, g) for k, n in KINDS for g in grids]
        times = {v: [] for v in variants}
        kern = []
        for _ in range(rounds + 1):
            for v in variants:
                times[v].append(timed(lambda: _ffi.call("lars_d_probe", v[0], 1, v[2], C.c_void_p(b.tiles.ptr),
                                                        C.c_void_p(dst.ptr), nbytes, None)))

            def run():
                for start in range(0, b.ntiles, outs.slots):
                    b.run_fused(b.fused_args(("NDVI", "GNDVI", "NDWI"), True, stats, False, outs, None, start, outs.slots))
            kern.append(timed(run))
        k_gbs = tiles * b.npix * 15 / float(np.median(kern[1:])) / 1e6
        print(f"--- allocation set {trial}: k_fused_u8c3 (3 planes + statistics, 64-tile launches) {k_gbs:8.1f} GB/s = {k_gbs / 8000:.3f} of 8 TB/s")
        res[f"set{trial} kernel"] = k_gbs
        for k, n in KINDS:
            best = max(((nbytes * (0.8 if k == 19 else 1.0)) / float(np.median(times[(k, n, g)][1:])) / 1e6, g) for g in grids)
            per = "  ".join(f"{g}: {(nbytes * (0.8 if k == 19 else 1.0)) / float(np.median(times[(k, n, g)][1:])) / 1e6:7.1f}" for g in grids)
            print(f"kind {k:2d} {n:75s} {per}   best {best[0]:7.1f} GB/s ({best[0] / k_gbs:5.3f} x kernel)")
            res[f"set{trial} kind{k}"] = best[0]
        dst.free()
        outs.free()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
