#!/usr/bin/env python3
"""Statistics-only passes over a resident batch, the two routes side by side in one process (HIP events on the library
stream; whole step = everything from the raw tiles to final records, white balance included):

    classic   channel-histogram pass + tables, then the per-pixel statistics kernel (+ its median passes)
    joint     ONE read: joint byte-pair histograms in LDS, then the per-tile finish kernel (csrc/joint.hip)

    python tools/jointbench.py --tiles 256 [--content vegetation,uniform,flat,steps,smooth]

Content: the two synthetic profiles of the bench, and three hand-made tiles replicated over the batch that show what the
counting kernel's LDS atomics do on imagery whose neighbouring pixels share cells (flat: one colour; steps: runs of 64
equal pixels; smooth: a slow gradient plus two levels of noise; natural / natural_narrow: 1 / f noise with correlated channels and sensor
noise, stretched over most resp. a third of the 8-bit range).
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

MODES = {"ndvi": ("NDVI",), "3idx": ("NDVI", "GNDVI", "NDWI")}


def handmade(kind, edge):
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:edge, 0:edge]
    if kind == "flat":
        t = np.empty((edge, edge, 3), np.uint8)
        t[...] = (40, 90, 180)
        return t
    if kind == "steps":
        base = ((xx // 64) * 37 + (yy // 8) * 11) % 200
        return np.stack([(base + 10 * c) % 256 for c in range(3)], axis=-1).astype(np.uint8)
    if kind in ("smooth", "smooth_mid"):
        out = []
        for c in range(3):
            if kind == "smooth":
                g = 60 + 50 * c + 40 * np.sin(xx / 900.0 + c) + 30 * np.cos(yy / 700.0) + rng.integers(-2, 3, (edge, edge))
            else:                                     # the same over 160-170 values per channel: safe windows on red and green do not fit
                g = 100 + 15 * c + 50 * np.sin(xx / 900.0 + c) + 35 * np.cos(yy / 700.0) + rng.integers(-2, 3, (edge, edge))
            out.append(np.clip(g, 0, 255))
        return np.stack(out, axis=-1).astype(np.uint8)
    if kind.startswith("natural"):
        hot_sel = None
        # image-like in the statistical sense: 1 / f^2 power spectrum (clouds of structure at every scale), channels that share most of their
        # structure, sensor noise on top; "natural" stretched over most of the 8-bit range (a processed JPEG), "natural_narrow" over a third
        # of it (a raw capture before any stretch -- what a percentile white balance is for)
        fy, fx = np.meshgrid(np.fft.fftfreq(edge), np.fft.rfftfreq(edge), indexing="ij")
        amp = 1.0 / np.maximum(np.hypot(fy, fx), 1.0 / edge) ** 1.0
        common = np.fft.irfft2(amp * np.exp(2j * np.pi * rng.random(amp.shape)), s=(edge, edge))
        out = []
        for c in range(3):
            own = np.fft.irfft2(amp * np.exp(2j * np.pi * rng.random(amp.shape)), s=(edge, edge))
            f = 0.8 * common + 0.6 * own
            f = (f - f.mean()) / f.std()
            # natural[_narrow|_soft|_mid][_clean|_n<sigma>][_sat|_nirsat]: _soft = as wide as it gets without clipping at 0 / 255 (p2 .. p98 spans
            # 130 values), _mid = p2 .. p98 spans 165 values; _clean = no
            # sensor noise (a denoised JPEG), _n0.7 = noise of sigma 0.7 levels (default 1.5); _sat = the brightest 1 % of the scene
            # overexposed in every channel, _nirsat = NIR alone saturated over the brightest 3.6 % (red and green keep moving)
            parts = kind.split("_")[1:]
            scale, centre = ((14.0, 70.0 + 30 * c) if "narrow" in parts else (32.0, 120.0 + 8 * c) if "soft" in parts else
                             (40.0, 120.0 + 8 * c) if "mid" in parts else (55.0, 120.0 + 15 * c))
            sigma = 0.0 if "clean" in parts else next((float(q[1:]) for q in parts if q[0] == "n" and q[1:2].isdigit()), 1.5)
            noise = rng.normal(0, sigma, (edge, edge)) if sigma else 0.0
            v = np.clip(centre + scale * f + noise, 0, 255)
            zc = (common - common.mean()) / common.std()
            if "sat" in parts:
                v[zc > 2.33] = 255
            if "nirsat" in parts and c == 2:
                v[zc > 1.8] = 255
            if "hotmid" in parts:                     # the same 1 % of pixels on one cell in the middle of the value range
                v[zc > 2.33] = 100
            if "hotiid" in parts:                     # 1 % of the pixels, drawn independently, on one cell
                if c == 0:
                    hot_sel = rng.random((edge, edge)) < 0.0113
                v[hot_sel] = 100
            if "hotrows" in parts:                    # whole image rows on one cell: 1 % of the rows, in bands of 8
                v[(yy // 8) % 100 == 7] = 100
            if "wide" in parts:                       # the same 1 % of pixels spread over the top of the value range: no hot cell
                sel = zc > 2.33
                v[sel] = rng.integers(200, 256, int(sel.sum()))
            out.append(v)
        return np.stack(out, axis=-1).astype(np.uint8)
    raise ValueError(kind)


def make_batch(content, ntiles, edge, roll=True):
    if content in ("uniform", "vegetation"):
        return lars.TileBatch.synthetic(ntiles, edge, edge, seed=1234, profile=content)
    b = lars.TileBatch(ntiles, edge, edge, 3, np.uint8)
    b.tiles.upload(handmade(content, edge)[None])
    # Tile i = tile 0 rolled by a pseudo-random number of rows (two device copies).  Plain replicas would put the same image region into
    # the same chunk of every tile, and chunk c of every tile runs on the same XCD (workgroups go round the XCDs): one slow region -- a
    # saturated blob -- would then load ONE XCD in every tile, which no batch of different images does.
    row = edge * 3
    for i in range(1, ntiles):
        r = (i * 997) % edge if roll else 0
        dst = b.tiles.ptr + i * b.tile_bytes
        _ffi.call("lars_memcpy_d2d", C.c_void_p(dst), C.c_void_p(b.tiles.ptr + r * row), b.tile_bytes - r * row, None)
        if r:
            _ffi.call("lars_memcpy_d2d", C.c_void_p(dst + b.tile_bytes - r * row), C.c_void_p(b.tiles.ptr), r * row, None)
    _ffi.call("lars_synchronize", None)
    return b


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
    for e in ev:
        _ffi.call("lars_event_destroy", e)
    return float(np.median(out[1:])), float(min(out[1:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--content", default="vegetation,uniform")
    ap.add_argument("--depths", default="6,4")
    ap.add_argument("--blocks", default="0", help="chunks per tile of the counting kernel (blocks_per_tile), 0 = automatic")
    ap.add_argument("--windows", default="1,0", help="joint_window settings to time: 1 windowed tables where they fit, 0 never, 2 windows that miss")
    ap.add_argument("--skip-classic-medians", action="store_true")
    ap.add_argument("--no-roll", action="store_true", help="hand-made contents: plain replicas of one tile instead of row-rolled ones")
    args = ap.parse_args()
    lib = _ffi.load()
    for content in args.content.split(","):
        b = make_batch(content, args.tiles, args.tile, roll=not args.no_roll)
        npix = args.tiles * args.tile * args.tile
        stats = b.new_stats()
        stats.zero()
        pairs = _ffi.DeviceBuffer(b.ntiles * 4 * 4)
        med_scratch = _ffi.DeviceBuffer(int(lib.lars_quotient_median_scratch_bytes(b.ntiles)))
        print(f"== {content}: {args.tiles} tiles of {args.tile}^2, 3 B/pixel algorithmic")

        def report(name, ms, mn):
            print(f"  {name:44s} {ms:8.3f} ms (min {mn:8.3f})  {npix / ms / 1e6:8.1f} Gpix/s  whole step {npix * 3 / ms / 1e6 / 8000:.3f} of 8 TB/s",
                  flush=True)

        ref = {}
        for mname, indices in MODES.items():
            def classic():
                b.compute_wb_tables()
                b.run_fused(b.fused_args(indices, True, stats))
            ms, mn = timed(classic, args.rounds)
            report(f"classic {mname} statistics", ms, mn)
            _ffi.call("lars_synchronize", None)
            ref[mname] = stats.download(_ffi.STATS_DTYPE, (b.ntiles, 3)).copy()
            ref["tables"], ref["pcts"] = b.host_tables().copy(), b.host_percentiles().copy()

            def classic_med():
                b.compute_wb_tables()
                a = b.fused_args(indices, True, stats)
                _ffi.call("lars_d_stats_medians", C.byref(a), C.c_void_p(pairs.ptr), C.c_void_p(med_scratch.ptr))
            if not args.skip_classic_medians:
                ms, mn = timed(classic_med, args.rounds)
                report(f"classic {mname} statistics + medians", ms, mn)
                _ffi.call("lars_synchronize", None)
                ref[mname + "_med"] = pairs.download(np.float32, (b.ntiles, 2, 2)).copy()
        for depth in map(int, args.depths.split(",")):
          for blocks in map(int, args.blocks.split(",")):
            for window in map(int, args.windows.split(",")):
                _ffi.set_tuning(joint_depth=depth if depth in (4, 6, 8, 12) else 6, joint_win_depth=depth if depth in (4, 5, 6, 12, 15) else 15, blocks_per_tile=blocks, joint_window=window)
                for mname, indices in MODES.items():
                    if window != 1 and len(indices) == 1:
                        continue                                    # one stream: never windowed
                    for med in (False, True):
                        ms, mn = timed(lambda: b.run_joint(indices, True, stats, pairs=pairs if med else None), args.rounds)
                        b.check_joint()
                        nw, nrec = b.joint_window_report()
                        n3 = b.joint_window_modes()[2]
                        report(f"joint   {mname} statistics{' + medians' if med else ''} depth {depth} blocks {blocks} window {window} "
                               f"[{nw} windowed{f' ({n3} on three windows)' if n3 else ''}, {nrec} recounted]", ms, mn)
                        rec = stats.download(_ffi.STATS_DTYPE, (b.ntiles, 3))
                        ids = [_ffi.INDEX_IDS[t] for t in indices]
                        assert rec[:, ids].tobytes() == ref[mname][:, ids].tobytes(), "records differ between the routes"
                        chans = sorted(lars.batch.channels_of(indices))
                        assert np.array_equal(b.host_tables(partial=True)[:, chans], ref["tables"][:, chans]), "tables differ between the routes"
                        assert b.host_percentiles(partial=True)[:, chans].tobytes() == ref["pcts"][:, chans].tobytes(), "percentiles differ"
                        if med and mname + "_med" in ref:
                            got = pairs.download(np.float32, (b.ntiles, 2, 2))
                            assert np.array_equal(got, ref[mname + "_med"], equal_nan=True), "medians differ between the routes"
        _ffi.set_tuning(joint_depth=6, joint_win_depth=15, blocks_per_tile=0, joint_window=1)
        stats.free(); pairs.free(); med_scratch.free(); b.free()


if __name__ == "__main__":
    main()
