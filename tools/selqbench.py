#!/usr/bin/env python3
"""Statistics + exact medians of a batch (no planes): the grid of the classic select passes over the tiles whose predicted
window missed, A/B in one process.

    python tools/selqbench.py --tiles 256 --wgs 0,64,128,256,512,1024,2048
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lars_image_processing_amd import _ffi  # noqa: E402
import lars_image_processing_amd as lars  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--wgs", default="0,64,128,256,512,1024,2048")
    ap.add_argument("--profile", default="vegetation")
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, args.tile, args.tile, seed=1234, profile=args.profile)
    b.compute_wb_tables()
    variants = [("window", w) for w in map(int, args.wgs.split(","))] + [("two-pass", 0)]
    times = {v: [] for v in variants}
    ref = None
    for _ in range(args.rounds + 1):
        for v in variants:
            _ffi.set_tuning(selq_window=1 if v[0] == "window" else 0, selq_list_wgs=v[1])
            _ffi.call("lars_synchronize", None)
            t0 = time.perf_counter()
            rec, med = b.process(medians=True, recompute_tables=False)
            times[v].append((time.perf_counter() - t0) * 1e3)
            med = np.asarray(med)
            if ref is None:
                ref = med.copy()
            assert np.array_equal(ref, med, equal_nan=True), v
    _ffi.set_tuning(selq_window=1, selq_list_wgs=0)
    npix = args.tiles * args.tile * args.tile
    for v, t in times.items():
        m = float(np.median(t[1:]))
        print(f"{v[0]:9s} list_wgs={v[1]:5d}  {m:7.3f} ms (min {min(t[1:]):7.3f}) per {args.tiles} tiles  {npix / m / 1e6:7.1f} Gpix/s  "
              f"{npix * 3 / m / 1e6 / 8000:.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
