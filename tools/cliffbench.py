#!/usr/bin/env python3
"""The launches that used to fall onto k_fused_generic (one pixel per lane) next to their fast neighbours, one process:
RGBA uint8 tiles against RGB tiles, two-index masks against all three indices; planes written (ring) and statistics only.

    python tools/cliffbench.py --tiles 128
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lars_image_processing_amd import _ffi  # noqa: E402
import lars_image_processing_amd as lars  # noqa: E402

KERNELS = {1: "k_fused_u8c3", 2: "k_fused_v2", 3: "k_fused_u8c3<uint16>", 4: "k_fused_generic", 5: "k_fused_u8c3<RGBA>"}


def timed(fn, rounds):
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    out = []
    for _ in range(rounds + 1):
        _ffi.call("lars_event_record", ev[0], None)
        fn()
        _ffi.call("lars_event_record", ev[1], None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        out.append(ms.value)
    return float(np.median(out[1:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=128)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--ring", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    npix = args.tiles * args.tile * args.tile
    for channels in (3, 4):
        b = lars.TileBatch.synthetic(args.tiles, args.tile, args.tile, seed=1234, profile="vegetation", channels=channels)
        stats = b.new_stats()
        for indices in (("NDVI", "GNDVI", "NDWI"), ("NDVI", "NDWI"), ("NDVI", "GNDVI"), ("GNDVI", "NDWI"), ("NDVI",)):
            outs = b.make_outputs(indices=indices, index=True, ring=args.ring, arena="plain")
            t_hist = timed(lambda: b.compute_wb_tables(), args.rounds)
            t_out = timed(lambda: b.run_fused_chunks(indices, True, stats, False, outs), args.rounds)
            k_out = KERNELS[_ffi.get_tuning("last_fused_kernel")]
            t_st = timed(lambda: b.run_fused(b.fused_args(indices, True, stats)), args.rounds)
            k_st = KERNELS[_ffi.get_tuning("last_fused_kernel")]
            t_joint = timed(lambda: b.run_joint(indices, True, stats), args.rounds)
            bpp_out = channels + 4 * len(indices)
            print(f"{channels} channels  {'+'.join(indices):16s} histogram pass {t_hist:7.3f} ms ({npix * channels / t_hist / 1e6:6.0f} GB/s)   "
                  f"planes + statistics {t_out:8.3f} ms ({npix * bpp_out / t_out / 1e6:6.0f} GB/s, {k_out})   "
                  f"statistics only {t_st:7.3f} ms ({npix * channels / t_st / 1e6:6.0f} GB/s, {k_st})   "
                  f"one-read statistics {t_joint:7.3f} ms ({npix * channels / t_joint / 1e6:6.0f} GB/s)", flush=True)
            outs.free()
        stats.free(); b.free()


if __name__ == "__main__":
    main()
