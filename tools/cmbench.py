#!/usr/bin/env python3
"""A/B of the coverage counters of the statistics-only kernels (lars_set_tuning("count_mode", k)) in one process:
-1 scalar counters (v_cmp -> s_bcnt1), 3 per-lane float counters (saturating packed fma + packed add).  Records must be
identical.

    python tools/cmbench.py [tiles=256] [rounds=6]
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    res = {}
    for profile in ("vegetation", "uniform"):
        b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile=profile)
        b.compute_wb_tables()
        stats = b.new_stats()
        for name, idx in (("3 indices", ("NDVI", "GNDVI", "NDWI")), ("NDVI", ("NDVI",))):
            times, recs = {-1: [], 3: []}, {}
            for r in range(rounds + 1):
                for cm in (-1, 3):
                    _ffi.set_tuning(count_mode=cm)
                    _ffi.call("lars_event_record", ev[0], None)
                    b.run_fused(b.fused_args(idx, True, stats, False, None))
                    _ffi.call("lars_event_record", ev[1], None)
                    _ffi.call("lars_synchronize", None)
                    ms = C.c_float(0)
                    _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
                    times[cm].append(ms.value)
                    if r == 0:
                        recs[cm] = stats.download(_ffi.STATS_DTYPE, (b.ntiles, 3)).tobytes()
            assert recs[-1] == recs[3], "records depend on the counter flavour"
            for cm in (-1, 3):
                med = float(np.median(times[cm][1:]))
                gbs = tiles * b.npix * 3 / med / 1e6
                res[f"{profile} {name} count_mode={cm}"] = gbs
                print(f"{profile:10s} {name:10s} count_mode={cm:2d}  {med:7.3f} ms  {gbs:7.1f} GB/s  {gbs / 8000:.3f} of 8 TB/s   (records identical)")
        stats.free(); b.free()
    _ffi.set_tuning(count_mode=-1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
