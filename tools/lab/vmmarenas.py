#!/usr/bin/env python3
"""Output arenas by the way their physical memory is obtained, levels read with warm step-length bursts (laboratory):
plain hipMalloc against virtual-memory-management blocks built from chunks of C MiB (in allocation order or shuffled).

    python tools/lab/vmmarenas.py [--each 5] [--chunks 64,256,16]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lablib  # noqa: E402
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402
from lars_image_processing_amd.batch import BatchOutputs  # noqa: E402

IDX = ("NDVI", "GNDVI", "NDWI")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--each", type=int, default=5)
    ap.add_argument("--chunks", default="64,256,16")
    ap.add_argument("--tiles", type=int, default=1024)
    args = ap.parse_args()
    lablib.load()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    outs = BatchOutputs(b, IDX, True, False, False, 64, allocate=False)
    nbytes = 3 * outs.plane_bytes
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def level(arena):
        outs.adopt_arena(arena)
        ls = [b.fused_args(IDX, True, stats, False, outs, None, st, min(outs.slots, b.ntiles - st), raw=True) for st in range(0, b.ntiles, outs.slots)]
        out = []
        for _ in range(2):
            _ffi.call("lars_event_record", ev[0], None)
            for a in ls:
                b.run_fused(a)
            _ffi.call("lars_event_record", ev[1], None)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            out.append(ms.value / len(ls))
        return out[1]

    kinds = [("plain hipMalloc", dict(kind=0))]
    for c in [int(x) for x in args.chunks.split(",")]:
        kinds.append((f"vmm {c} MiB chunks", dict(kind=3, chunk_mb=c, shuffle=0)))
        kinds.append((f"vmm {c} MiB chunks, shuffled", dict(kind=3, chunk_mb=c, shuffle=1)))
    kinds.append(("plain hipMalloc (again)", dict(kind=0)))
    free_b, total_b = C.c_size_t(), C.c_size_t()
    for name, kw in kinds:
        row, held = [], []
        for _ in range(args.each):
            _ffi.call("lars_mem_info", C.byref(free_b), C.byref(total_b))
            if free_b.value < nbytes + (8 << 30):
                break
            t0 = time.perf_counter()
            a = lablib.LabBuffer(nbytes, **kw)
            ms_alloc = (time.perf_counter() - t0) * 1e3
            held.append(a)
            row.append((level(a), ms_alloc))
        print(f"{name:32s} " + "  ".join(f"{t:.3f}" for t, _ in row) + "    alloc ms " + " ".join(f"{m:.0f}" for _, m in row), flush=True)
        for a in held:
            a.free()
        _ffi.call("lars_synchronize", None)


if __name__ == "__main__":
    main()
