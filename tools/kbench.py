#!/usr/bin/env python3
"""Kernel micro-benchmark: A/B kernel variants in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24), HIP-event timing on the launch stream.

    python tools/kbench.py --tiles 256 --rounds 5 --what hist,fused
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lars_image_processing_amd import _ffi  # noqa: E402
import lars_image_processing_amd as lars  # noqa: E402

MODES = {
    "stats_ndvi": (("NDVI",), False, False, 3),
    "stats_3idx": (("NDVI", "GNDVI", "NDWI"), False, False, 3),
    "out_ndvi": (("NDVI",), True, False, 7),
    "out_3idx": (("NDVI", "GNDVI", "NDWI"), True, False, 15),
    "out_3idx_hist": (("NDVI", "GNDVI", "NDWI"), True, True, 15),
    "stats_3idx_hist": (("NDVI", "GNDVI", "NDWI"), False, True, 3),
}


class Timer:
    def __init__(self):
        self.a, self.b = C.c_void_p(), C.c_void_p()
        _ffi.call("lars_event_create", C.byref(self.a))
        _ffi.call("lars_event_create", C.byref(self.b))

    def time(self, fn):
        _ffi.call("lars_event_record", self.a, None)
        fn()
        _ffi.call("lars_event_record", self.b, None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", self.a, self.b, C.byref(ms))
        return ms.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--ring", type=int, default=64)
    ap.add_argument("--what", default="hist,fused")
    ap.add_argument("--modes", default=",".join(MODES))
    ap.add_argument("--impls", default="1,2")
    ap.add_argument("--nt", default="0,1")
    ap.add_argument("--bpt", default="0")
    ap.add_argument("--wb", default="1")
    ap.add_argument("--profile", default="vegetation")
    args = ap.parse_args()

    if args.profile in ("uniform", "vegetation"):
        b = lars.TileBatch.synthetic(args.tiles, args.tile, args.tile, seed=1234, profile=args.profile)
    else:                                           # the hand-made contents of tools/jointbench.py (natural, smooth, flat, steps ...)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import jointbench
        b = jointbench.make_batch(args.profile, args.tiles, args.tile)
    b.compute_wb_tables()
    stats = b.new_stats()
    timer = Timer()
    npix = args.tiles * args.tile * args.tile
    results = {}

    if "hist" in args.what:
        variants = [(i, bp) for i in map(int, args.impls.split(",")) for bp in map(int, args.bpt.split(","))]
        times = {v: [] for v in variants}
        for _ in range(args.rounds + 1):
            for v in variants:
                _ffi.set_tuning(hist_impl=v[0], blocks_per_tile=v[1])
                times[v].append(timer.time(lambda: _ffi.call(
                    "lars_d_channel_hist", C.c_void_p(b.tiles.ptr), b.ntiles, b.npix, 3, _ffi.U8,
                    C.c_void_p(b.hist.ptr), None)))
        for v, t in times.items():
            med = float(np.median(t[1:]))
            results[f"hist impl={v[0]} bpt={v[1]}"] = {"ms": med, "min_ms": float(min(t[1:])), "GBs": npix * 3 / med / 1e6}
        _ffi.set_tuning(blocks_per_tile=0)

    if "fused" in args.what:
        modes = [m for m in args.modes.split(",") if m]
        outs_cache = {}
        variants = []
        for m in modes:
            for impl in map(int, args.impls.split(",")):
                for nt in map(int, args.nt.split(",")):
                    if nt and not MODES[m][1]:
                        continue
                    for bp in map(int, args.bpt.split(",")):
                        for wb in map(int, args.wb.split(",")):
                            variants.append((m, impl, nt, bp, wb))
        times = {v: [] for v in variants}
        for _ in range(args.rounds + 1):
            for v in variants:
                m, impl, nt, bp, wb = v
                indices, write, hist, bpp = MODES[m]
                outs = None
                if write:
                    if indices not in outs_cache:
                        outs_cache[indices] = b.make_outputs(indices=indices, index=True, ring=args.ring)
                    outs = outs_cache[indices]
                _ffi.set_tuning(fused_impl=impl, nt_stores=nt, blocks_per_tile=bp)

                def run():
                    if outs is None:
                        b.run_fused(b.fused_args(indices, bool(wb), stats, hist, None))
                    else:
                        for start in range(0, b.ntiles, outs.slots):
                            cnt = min(outs.slots, b.ntiles - start)
                            b.run_fused(b.fused_args(indices, bool(wb), stats, hist, outs, None, start, cnt))
                times[v].append(timer.time(run))
        for v, t in times.items():
            m, impl, nt, bp, wb = v
            med = float(np.median(t[1:]))
            bpp = MODES[m][3]
            results[f"fused {m} impl={impl} nt={nt} bpt={bp} wb={wb}"] = {
                "ms": med, "min_ms": float(min(t[1:])), "GBs": npix * bpp / med / 1e6,
                "frac_8TBs": npix * bpp / med / 1e6 / 8000.0, "Gpix_s": npix / med / 1e6}
    if "medians" in args.what:
        # time-series table of the whole batch (statistics + median per tile): without planes, and with the planes written as well
        import time
        outs = b.make_outputs(index=True, ring=min(args.ring, 16))
        for name, kw in (("medians recompute+select (no planes)", {}), ("medians NDVI only (no planes)", {"indices": ("NDVI",)}),
                         ("medians + planes written (ring)", {"outputs": outs})):
            ts = []
            for _ in range(3):
                _ffi.call("lars_synchronize", None)
                t0 = time.perf_counter()
                b.process(medians=True, recompute_tables=False, white_balance=args.wb != "0", **kw)
                ts.append((time.perf_counter() - t0) * 1e3)
            results[name] = {"ms": float(np.median(ts[1:])), "min_ms": float(min(ts[1:])), "Gpix_s": npix / float(np.median(ts[1:])) / 1e6}
        outs.free()
    for k, v in results.items():
        print(f"{k:58s} " + "  ".join(f"{kk}={vv:9.3f}" for kk, vv in v.items()))
    print(json.dumps(results))


if __name__ == "__main__":
    main()
