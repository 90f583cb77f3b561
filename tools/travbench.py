#!/usr/bin/env python3
"""A/B of the plane-writing kernel's pixel -> wave mappings (lars_set_tuning("traverse", k)) in ONE process, interleaved,
over several sets of output allocations; also checks that records and planes do not depend on the mapping.

    python tools/travbench.py [tiles=256] [rounds=5] [sets=3]
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars

NAMES = {0: "wave runs of 1024 px, one quad per trip", 1: "wave runs of 1024 px, 4 quads unrolled", 2: "round-1 grid stride"}


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    sets = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    idx = ("NDVI", "GNDVI", "NDWI") if os.environ.get("TRAV_INDICES", "3") == "3" else ("NDVI",)
    bpp = 3 + 4 * len(idx)
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    res = {}
    for trial in range(sets):
        outs = b.make_outputs(indices=idx, index=True, ring=64)
        times = {k: [] for k in NAMES}
        recs, planes = {}, {}
        for r in range(rounds + 1):
            for k in NAMES:
                _ffi.set_tuning(traverse=k)
                _ffi.call("lars_event_record", ev[0], None)
                for start in range(0, b.ntiles, outs.slots):
                    b.run_fused(b.fused_args(idx, True, stats, False, outs, None, start, outs.slots))
                _ffi.call("lars_event_record", ev[1], None)
                _ffi.call("lars_synchronize", None)
                ms = C.c_float(0)
                _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
                times[k].append(ms.value)
                if r == 0 and trial == 0:
                    recs[k] = stats.download(_ffi.STATS_DTYPE, (b.ntiles, 3)).tobytes()
                    planes[k] = [outs.host_index(t, 3, 1).tobytes() for t in idx]
        if trial == 0:
            assert recs[0] == recs[1] == recs[2], "statistics depend on the mapping"
            assert planes[0] == planes[1] == planes[2], "planes depend on the mapping"
            print("records and planes identical across mappings")
        for k, name in NAMES.items():
            med = float(np.median(times[k][1:]))
            gbs = tiles * b.npix * bpp / med / 1e6
            res[f"set{trial} traverse{k}"] = gbs
            print(f"set {trial}  traverse={k} {name:42s} {med:8.3f} ms  {gbs:7.1f} GB/s  {gbs / 8000:.3f} of 8 TB/s")
        outs.free()
    _ffi.set_tuning(traverse=-1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
