#!/usr/bin/env python3
"""One written plane (NDVI, no statistics: 3 B read + 4 B written per pixel) at several places of one 24 GiB allocation: the sixteen
launches of a pass one by one -- does it matter whether the plane lies in the kind of memory the input chunk lies in?

    python tools/lab/oneplane.py [--allocations 2]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402

GIB = 1 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--allocations", type=int, default=2)
    ap.add_argument("--tiles", type=int, default=1024)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    G = 64
    ev = []
    for _ in range(18):
        e = C.c_void_p(); _ffi.call("lars_event_create", C.byref(e)); ev.append(e)

    def burst(base, off):
        ls = []
        for st in range(0, b.ntiles, G):
            a = b.fused_args(("NDVI",), True, None, False, None, None, st, min(G, b.ntiles - st))
            a.out_index[0] = base + off * GIB
            ls.append(a)
        out = None
        for _ in range(2):
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

    held = []
    for n in range(args.allocations):
        big = _ffi.DeviceBuffer(24 * GIB)
        held.append(big)
        rows = {}
        for off in (0, 4, 8, 12, 16, 20):
            t = burst(big.ptr, off)
            rows[off] = t
            print(f"allocation {n} plane at {off:2d} GiB: mean {np.mean(t):.3f}  [" + " ".join(f"{x:.3f}" for x in t) + "]", flush=True)
        best = np.min(np.array([rows[o] for o in rows]), axis=0)
        print(f"allocation {n} best placement per launch: mean {best.mean():.3f}  [" + " ".join(f"{x:.3f}" for x in best) + "]", flush=True)


if __name__ == "__main__":
    main()
