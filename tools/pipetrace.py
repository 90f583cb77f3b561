#!/usr/bin/env python3
"""Where the persistent pipeline's time goes: per-item timestamps (lars_set_tuning("pipe_trace", 1)) of one launch over
64 tiles of 4096 x 4096: per phase the mean wait / stream / finish times, the gaps between a workgroup's items, and the
per-tile schedule (when a tile's H items, its table and its F items happened).

    python tools/pipetrace.py [spi=64] [head=8]
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars

TRACE_ITEMS, TICK_US = 96, 0.01


def main():
    spi = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    head = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tiles = 64
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    stats = b.new_stats()
    outs = b.make_outputs(index=True)
    _ffi.set_tuning(pipe_steps=spi, pipe_head=head, pipe_trace=1)
    for _ in range(2):
        b.run_pipeline(stats, outs)
        _ffi.call("lars_synchronize", None)
    _ffi.set_tuning(pipe_steps=0, pipe_head=0, pipe_trace=0)
    nsteps = (b.npix // 4 + 255) // 256
    items = (nsteps + spi - 1) // spi
    off = tiles * items * 768 * 4 + (2 + 2 * tiles) * 4
    off = (b._pipe_scratch.ptr + off + 255) // 256 * 256 - b._pipe_scratch.ptr
    tr = b._pipe_scratch.download(np.uint64, (4096, TRACE_ITEMS, 6), off).astype(np.int64)
    used = tr[:, :, 0] != 0
    wgs = int(used.any(axis=1).sum())
    t_base = tr[:, :, 0][used].min()
    t_end = tr[:, :, 4][used].max()
    print(f"spi={spi} head={head}: {items} items per tile and phase, {wgs} workgroups, launch {(t_end - t_base) * TICK_US:.1f} us for {tiles} tiles "
          f"= {(t_end - t_base) * TICK_US / tiles:.2f} us per tile (traced part: first {TRACE_ITEMS} items per workgroup)")
    phase = tr[:, :, 5] & 0xFF
    tile = (tr[:, :, 5] >> 8) & 0xFFFFFF
    for ph, name in ((0, "H"), (1, "F")):
        m = used & (phase == ph) & (tile < tiles)
        a = (tr[:, :, 1] - tr[:, :, 0])[m] * TICK_US
        s = (tr[:, :, 2] - tr[:, :, 1])[m] * TICK_US
        f = (tr[:, :, 4] - tr[:, :, 2])[m] * TICK_US
        what = ("histogram sweep", "-", "count + (last: table)") if ph == 0 else ("wait + table", "stream", "flush")
        print(f"{name} items: {int(m.sum()):6d}   {what[0]}: mean {a.mean():7.2f} p50 {np.median(a):7.2f} p95 {np.percentile(a, 95):7.2f} us   "
              f"{what[1]}: mean {s.mean():7.2f} p50 {np.median(s):7.2f} us   {what[2]}: mean {f.mean():7.2f} p95 {np.percentile(f, 95):7.2f} us")
    gaps = []
    for w in range(4096):
        n = int(used[w].sum())
        if n > 1:
            gaps.append((tr[w, 1:n, 0] - tr[w, :n - 1, 4]) * TICK_US)
    gaps = np.concatenate(gaps)
    print(f"gap between a workgroup's items (barrier + queue): mean {gaps.mean():.2f} p50 {np.median(gaps):.2f} p95 {np.percentile(gaps, 95):.2f} us")
    print("# tile: H first start .. last end | table ready (end of the last H item's workgroup) | F first start .. last end   [us from launch start]")
    for t in range(0, min(tiles, 12)):
        mh = used & (phase == 0) & (tile == t)
        mf = used & (phase == 1) & (tile == t)
        if not mh.any() or not mf.any():
            continue
        h0, h1 = tr[:, :, 0][mh].min(), tr[:, :, 3][mh].max()
        tab = tr[:, :, 4][mh].max()
        f0, f1 = tr[:, :, 0][mf].min(), tr[:, :, 4][mf].max()
        fs = tr[:, :, 1][mf].min()
        print(f"tile {t:2d}: H {(h0 - t_base) * TICK_US:8.1f} .. {(h1 - t_base) * TICK_US:8.1f} | table {(tab - t_base) * TICK_US:8.1f} | "
              f"F taken {(f0 - t_base) * TICK_US:8.1f}, streaming from {(fs - t_base) * TICK_US:8.1f} .. {(f1 - t_base) * TICK_US:8.1f}")


if __name__ == "__main__":
    main()
