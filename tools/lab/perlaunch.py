#!/usr/bin/env python3
"""The sixteen launches of a headline pass one by one, for several placements of the three planes inside the first 24 GiB allocation of a
fresh process (product library): is there, for every input chunk, a placement at the level of the best launches?

    python tools/lab/perlaunch.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402

IDX = ("NDVI", "GNDVI", "NDWI")
GIB = 1 << 30


def main():
    b = lars.TileBatch.synthetic(1024, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    G = 64
    ev = []
    for _ in range(18):
        e = C.c_void_p(); _ffi.call("lars_event_create", C.byref(e)); ev.append(e)
    big = _ffi.DeviceBuffer(24 * GIB)

    def burst(offs):
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(IDX, True, stats, False, None, None, st, min(G, b.ntiles - st), raw=True)
            for k in range(3):
                a.out_index[k] = big.ptr + offs[k] * GIB
            ls.append(a)
        out = None
        for _ in range(3):
            _ffi.call("lars_event_record", ev[0], None)
            for i, a in enumerate(ls):
                b.run_fused(a)
                _ffi.call("lars_event_record", ev[i + 1], None)
            _ffi.call("lars_synchronize", None)
            ms = C.c_float(0)
            out = []
            for i in range(len(ls)):
                _ffi.call("lars_event_elapsed_ms", ev[i], ev[i + 1], C.byref(ms)); out.append(ms.value)
        return out

    rows = {}
    for offs in ((0, 4, 8), (0, 4, 16), (0, 4, 20), (0, 16, 20), (4, 16, 20), (0, 12, 16), (0, 8, 16), (8, 12, 16), (8, 16, 20), (12, 16, 20)):
        t = burst(offs)
        rows[offs] = t
        print(f"planes at {str(offs):12s}: mean {np.mean(t):.3f}  [" + " ".join(f"{x:.3f}" for x in t) + "]", flush=True)
    fast = {k: v for k, v in rows.items() if np.mean(v) < 2.8}
    best = np.min(np.array(list(fast.values())), axis=0)
    print(f"best placement per launch     : mean {best.mean():.3f}  [" + " ".join(f"{x:.3f}" for x in best) + "]")


if __name__ == "__main__":
    main()
