#!/usr/bin/env python3
"""Does the LAYOUT of the three planes inside an output arena decide its speed class?  (laboratory build of the library with the
plane-stride knob: make -C lars_image_processing_amd/csrc lablayout, then LARS_HIP_LIB=build/lablayout/liblars_layout.so python tools/lab/interleave.py)

The product writes plane k of tile slot s at  arena + k * (slots * 64 MiB) + s * 64 MiB  ("planar": three streams 4 GiB apart).
Interleaved:  arena + s * (3 * 64 MiB) + k * 64 MiB  (the three planes of a slot next to each other: 192 MiB per slot).
Same allocation, same bytes, same kernel; per candidate arena two bursts of 16 launches (the whole batch) per layout, mean of
launches 2..16 of the second burst.

    python tools/lab/interleave.py [--candidates 8]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402
from lars_image_processing_amd.batch import BatchOutputs  # noqa: E402

IDX = ("NDVI", "GNDVI", "NDWI")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=8)
    ap.add_argument("--tiles", type=int, default=1024)
    ap.add_argument("--ring", type=int, default=64)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    outs = BatchOutputs(b, IDX, True, False, False, args.ring, allocate=False)
    nbytes = 3 * outs.plane_bytes
    ev = []
    for _ in range(18):
        e = C.c_void_p(); _ffi.call("lars_event_create", C.byref(e)); ev.append(e)

    def burst(launches):
        _ffi.call("lars_event_record", ev[0], None)
        for i, a in enumerate(launches):
            b.run_fused(a)
            _ffi.call("lars_event_record", ev[i + 1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        out = []
        for i in range(len(launches)):
            _ffi.call("lars_event_elapsed_ms", ev[i], ev[i + 1], C.byref(ms)); out.append(ms.value)
        return out

    def launches_for(arena, interleaved):
        outs.adopt_arena(arena)
        ls = []
        for st in range(0, b.ntiles, outs.slots):
            a = b.fused_args(IDX, True, stats, False, outs, None, st, min(outs.slots, b.ntiles - st), raw=True)
            if interleaved:
                for k in range(3):
                    a.out_index[k] = arena.ptr + k * b.npix * 4          # slot 0 of every launch (the ring is rewritten each time)
            ls.append(a)
        return ls

    def level(ls):
        burst(ls)
        return float(np.mean(burst(ls)[1:]))

    print(f"# {args.candidates} candidate arenas of {nbytes / 2**30:.0f} GiB; ms per 64-tile launch (mean of launches 2..16 of the second burst)")
    arenas = [_ffi.DeviceBuffer(nbytes) for _ in range(args.candidates)]
    for rnd in range(2):
        for j, arena in enumerate(arenas):
            row = []
            for inter in (0, 1, 0, 1):
                _ffi.set_tuning(out_stride_planes=3 if inter else 0)
                ls = launches_for(arena, inter)
                burst(ls)
                t = burst(ls)
                row.append(float(np.mean(t[1:])))
            _ffi.set_tuning(out_stride_planes=0)
            print(f"round {rnd} arena {j}: planar {row[0]:.3f} interleaved {row[1]:.3f} planar {row[2]:.3f} interleaved {row[3]:.3f}", flush=True)
    # correctness of the interleaved form on one tile group: plane k of slot s at (3 s + k) * npix
    _ffi.set_tuning(out_stride_planes=3)
    ls = launches_for(arenas[0], 1)
    b.run_fused(ls[-1])
    _ffi.call("lars_synchronize", None)
    got = arenas[0].download(np.float32, (3, b.npix), (3 * 5) * b.npix * 4)          # slot 5 of the last chunk
    _ffi.set_tuning(out_stride_planes=0)
    ls = launches_for(arenas[1], 0)
    b.run_fused(ls[-1])
    _ffi.call("lars_synchronize", None)
    want = [outs.host_index(t, 5, 1).reshape(-1) for t in IDX]
    print("interleaved planes == planar planes:", all(np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)) for k in range(3)))


if __name__ == "__main__":
    main()
