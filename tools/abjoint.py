#!/usr/bin/env python3
"""A/B of library BUILDS on the one-read statistics call (lars_d_stats_joint) in one process, on the same batch: box-to-box differences
(+-3 %) are larger than most kernel changes, so two builds can only be compared side by side.

    git archive HEAD lars_image_processing_amd/csrc include | tar -x -C build/head_src
    make -C build/head_src/lars_image_processing_amd/csrc OUT=$PWD/build/variants/liblars_head.so OBJDIR=$PWD/build/obj_head
    python tools/abjoint.py --content vegetation,smooth build/variants/liblars_head.so

Every library runs the call on its own stream, with its own scratch; host clock around call + synchronize (median of the rounds).
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from lars_image_processing_amd import _ffi  # noqa: E402
import jointbench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--content", default="vegetation,uniform")
    ap.add_argument("--window", type=int, default=1, help="lars_set_tuning(\"joint_window\", ..) in every library (0: full tables, two readers)")
    ap.add_argument("--no-roll", action="store_true", help="hand-made contents: plain replicas of one tile (jointbench.make_batch)")
    args = ap.parse_args()
    libs = [("product", _ffi.load())]
    for path in args.libs:
        libs.append((os.path.basename(path), C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)))
    for _, lib in libs[1:]:
        lib.lars_d_stats_joint.restype = C.c_int
        lib.lars_d_stats_joint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.lars_synchronize.restype = C.c_int
        lib.lars_synchronize.argtypes = [C.c_void_p]
        lib.lars_joint_scratch_bytes.restype = C.c_size_t
        lib.lars_joint_scratch_bytes.argtypes = [C.c_int64, C.c_int64, C.c_uint32]
    for _, lib in libs:
        lib.lars_set_tuning.restype = C.c_int
        lib.lars_set_tuning.argtypes = [C.c_char_p, C.c_int]
        assert lib.lars_set_tuning(b"joint_window", args.window) == 0
    for content in args.content.split(","):
        b = jointbench.make_batch(content, args.tiles, args.tile, roll=not args.no_roll)
        npix = args.tiles * args.tile * args.tile
        print(f"== {content}: {args.tiles} tiles of {args.tile}^2")
        for mname, indices in jointbench.MODES.items():
            mask = 0
            for t in indices:
                mask |= 1 << _ffi.INDEX_IDS[t]
            need = max(int(lib.lars_joint_scratch_bytes(b.ntiles, b.npix, mask)) for _, lib in libs)
            stats = {name: b.new_stats() for name, _ in libs}
            scratch = {name: _ffi.DeviceBuffer(need) for name, _ in libs}
            table = _ffi.DeviceBuffer(b.ntiles * b.table_bytes)
            pcts = _ffi.DeviceBuffer(b.ntiles * 3 * 2 * 8)
            table.zero(); pcts.zero()
            times = {name: [] for name, _ in libs}
            blocks = {}
            for name, _ in libs:
                stats[name].zero()
                a = _ffi.FusedArgs()
                a.tiles = b.tiles.ptr
                a.ntiles, a.npix, a.channels, a.dtype = b.ntiles, b.npix, b.channels, b.code
                a.wb_table = table.ptr
                a.index_mask = mask
                a.flags = _ffi.F_STATS
                a.stats = stats[name].ptr
                a.stream = None
                blocks[name] = a
            _ffi.call("lars_synchronize", None)
            for _ in range(args.rounds + 1):
                for name, lib in libs:
                    t0 = time.perf_counter()
                    rc = lib.lars_d_stats_joint(C.byref(blocks[name]), 1, 0, C.c_void_p(pcts.ptr), None, None, C.c_void_p(scratch[name].ptr), need)
                    lib.lars_synchronize(None)
                    times[name].append((time.perf_counter() - t0) * 1e3)
                    assert rc == 0, rc
            ref = stats["product"].download(_ffi.STATS_DTYPE, (b.ntiles, 3))
            ids = [_ffi.INDEX_IDS[t] for t in indices]
            for name, _ in libs:
                ms = float(np.median(times[name][1:]))
                same = stats[name].download(_ffi.STATS_DTYPE, (b.ntiles, 3))[:, ids].tobytes() == ref[:, ids].tobytes()
                print(f"  {mname:5s} {name:24s} {ms:8.3f} ms (min {min(times[name][1:]):8.3f})  whole call {npix * 3 / ms / 1e6 / 8000:.3f} of 8 TB/s"
                      f"  records {'identical' if same else 'DIFFER'}", flush=True)
            for name, _ in libs:
                stats[name].free(); scratch[name].free()
            table.free(); pcts.free()
        b.free()


if __name__ == "__main__":
    main()
