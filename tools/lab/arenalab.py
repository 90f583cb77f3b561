#!/usr/bin/env python3
"""Output arenas by the way they are built, in ONE process: the plane-writing kernel (64 tiles of 4096^2 per launch, three
float32 planes) into each, interleaved rounds.

    python tools/lab/arenalab.py --variants plain,asm64,asm2,asm2s,asm64a1024 --per 3

Round 3's finding (profiles/r03_arena_assembled.txt): the candidate groups of physical memory all probe alike, and an arena
assembled from them is no faster than a plain allocation as it comes -- the product therefore chooses among whole plain
allocations (TileBatch.make_outputs).
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lablib  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402
import lars_image_processing_amd as lars  # noqa: E402

VARIANTS = {
    "plain": dict(arena="plain"),
    "asm64": dict(arena="assembled", arena_chunk_mb=64),
    "asm2": dict(arena="assembled", arena_chunk_mb=2),
    "asm2s": dict(arena="assembled", arena_chunk_mb=2, arena_shuffle=1),
    "asm16s": dict(arena="assembled", arena_chunk_mb=16, arena_shuffle=1),
    "asm64s": dict(arena="assembled", arena_chunk_mb=64, arena_shuffle=1),
    "asm1024": dict(arena="assembled", arena_chunk_mb=1024),
    "asm64a1024": dict(arena="assembled", arena_chunk_mb=64, arena_align_mb=1024),
    "asm2a1024": dict(arena="assembled", arena_chunk_mb=2, arena_align_mb=1024),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--ring", type=int, default=64)
    ap.add_argument("--variants", default="plain,asm64,asm2,asm2s")
    ap.add_argument("--per", type=int, default=3, help="arenas per variant")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--max-groups", type=int, default=0)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    arenas = []
    for name in args.variants.split(","):
        v = dict(VARIANTS[name])
        mode = v.pop("arena")
        for _ in range(args.per):
            lablib.set_tuning(arena_chunk_mb=v.get("arena_chunk_mb", 64), arena_align_mb=v.get("arena_align_mb", 0),
                              arena_shuffle=v.get("arena_shuffle", 0))
            outs = (b.make_outputs(index=True, ring=args.ring, arena="plain", placement_trials=0) if mode == "plain" else
                    lablib.assembled_outputs(b, ring=args.ring, max_groups=args.max_groups or None))
            arenas.append((name, outs, []))
    lablib.set_tuning(arena_chunk_mb=64, arena_align_mb=0, arena_shuffle=0)
    for _ in range(args.rounds):
        for name, outs, times in arenas:
            times.append(b._time_outputs(outs, ("NDVI", "GNDVI", "NDWI")))
    nbytes = args.ring * 4096 * 4096 * 15
    for name, outs, times in arenas:
        ms = float(np.median(times))
        rep = outs.arena_report
        extra = "" if "group_ms" not in rep else f"  search {rep['search_ms']:.0f} ms, groups " + " ".join(f"{x:.3f}" for x in rep["group_ms"])
        print(f"{name:12s} arena at {outs.arena.ptr:#x}: {ms:.3f} ms  {nbytes / ms / 1e6:7.1f} GB/s  ({nbytes / ms / 1e6 / 8000:.3f}){extra}", flush=True)


if __name__ == "__main__":
    main()
