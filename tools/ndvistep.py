#!/usr/bin/env python3
"""The NDVI-plane step (bench mode wb_ndvi_out_stats: 3 B read, 4 B written per pixel + statistics), three ways, one process, one arena:

  A  what the mode does: channel-histogram pass over the batch + tables, then the fused NDVI kernel with statistics, chunk by chunk
  B  the one-read statistics pass for NDVI (one stream: statistics AND the tables of red and NIR from one read), then the
     plane-writing kernel WITHOUT statistics
  C  A with the histogram pass + tables of chunk c + 1 on a second stream under the fused launch of chunk c

    python tools/ndvistep.py [--tiles 1024] [--ring 64] [--rounds 5]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402

IDX = ("NDVI",)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=1024)
    ap.add_argument("--ring", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    outs = b.make_outputs(indices=IDX, index=True, ring=args.ring)
    print("arena:", {k: outs.arena_report.get(k) for k in ("kind", "chosen_ms", "post_free_ms", "candidate_ms")}, flush=True)
    stats = b.new_stats()
    side = C.c_void_p()
    _ffi.call("lars_stream_create", C.byref(side))
    nchunks = -(-b.ntiles // outs.slots)
    ev_hist = []
    for _ in range(nchunks):
        e = C.c_void_p(); _ffi.call("lars_event_create", C.byref(e)); ev_hist.append(e)
    ev = [C.c_void_p() for _ in range(3)]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    npix = b.ntiles * b.npix

    def hist_chunk(start, count, stream):
        _ffi.call("lars_d_channel_hist", C.c_void_p(b.tiles.ptr + start * b.tile_bytes), count, b.npix, b.channels, b.code,
                  C.c_void_p(b.hist.ptr + start * 3 * 256 * 4), stream)
        _ffi.call("lars_d_wb_table", C.c_void_p(b.hist.ptr + start * 3 * 256 * 4), count, b.npix, b.code,
                  C.c_void_p(b.table.ptr + start * b.table_bytes), C.c_void_p(b.percentiles.ptr + start * 48), 0, stream)

    def way_a():
        b.compute_wb_tables()
        _ffi.call("lars_event_record", ev[1], None)
        b.run_fused_chunks(IDX, True, stats, False, outs)

    def way_b():
        b.run_joint(IDX, True, stats)
        _ffi.call("lars_event_record", ev[1], None)
        b.run_fused_chunks(IDX, True, None, False, outs)

    def way_c():
        _ffi.call("lars_event_record", ev[1], None)                      # no separate first pass
        _ffi.call("lars_event_record", ev_hist[0], None)                 # orders the side stream behind what the main stream has queued
        _ffi.call("lars_stream_wait_event", side, ev_hist[0])
        hist_chunk(0, min(outs.slots, b.ntiles), side)
        _ffi.call("lars_event_record", ev_hist[0], side)
        _ffi.call("lars_d_stats_begin", C.c_void_p(stats.ptr), b.ntiles, 1, None)
        for c in range(nchunks):
            start = c * outs.slots
            count = min(outs.slots, b.ntiles - start)
            if c + 1 < nchunks:
                s1 = (c + 1) * outs.slots
                hist_chunk(s1, min(outs.slots, b.ntiles - s1), side)
                _ffi.call("lars_event_record", ev_hist[c + 1], side)
            _ffi.call("lars_stream_wait_event", None, ev_hist[c])
            b.run_fused(b.fused_args(IDX, True, stats, False, outs, None, start, count, raw=True))
        _ffi.call("lars_d_stats_end", C.c_void_p(stats.ptr), b.ntiles, 1, b.npix, None)

    results = {}
    ref = None
    for name, fn in (("A hist pass, then NDVI + statistics", way_a), ("B one-read statistics + tables, then NDVI plane only", way_b),
                     ("C A with the next chunk's hist pass on a second stream", way_c), ("A again", way_a)):
        ts, firsts = [], []
        for _ in range(args.rounds + 1):
            _ffi.call("lars_synchronize", None)
            _ffi.call("lars_synchronize", side)
            _ffi.call("lars_event_record", ev[0], None)
            fn()
            _ffi.call("lars_event_record", ev[2], None)
            _ffi.call("lars_synchronize", side)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[2], C.byref(ms)); ts.append(ms.value)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms)); firsts.append(ms.value)
        b.check_joint()
        rec = stats.download(_ffi.STATS_DTYPE, (b.ntiles, 3))[:, 0]
        plane = outs.host_index("NDVI", (b.ntiles - 1) % outs.slots, 1)
        if ref is None:
            ref = (rec.tobytes(), plane.tobytes())
        same = (rec.tobytes() == ref[0], plane.tobytes() == ref[1])
        t = float(np.median(ts[1:]))
        print(f"{name:58s} {t:7.2f} ms per {b.ntiles} tiles (first pass {float(np.median(firsts[1:])):6.2f})  whole step {npix * 7 / t / 1e6 / 8000:.3f} of 8 TB/s  "
              f"records {'==' if same[0] else '!='} A, last plane {'==' if same[1] else '!='} A", flush=True)
        results[name] = t


if __name__ == "__main__":
    main()
