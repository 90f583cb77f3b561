#!/usr/bin/env python3
"""Infinity Cache probe: bandwidth of re-reading a range of S bytes again and again inside one launch, alone and while 4 x
as many bytes are written elsewhere (the plane-writing kernel's mix), by S.  What a histogram pass -> fused pass
sequence over the same tile could get back from the 256 MiB cache.

    python tools/mallbench.py
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi


def main():
    _ffi.call("lars_set_device", 0)
    MiB = 1 << 20
    sizes = [12, 24, 48, 96, 144, 192, 240, 288, 384, 768, 3072]
    src = _ffi.DeviceBuffer(max(sizes) * MiB)
    dst = _ffi.DeviceBuffer(4 * 8 * 768 * MiB + 4096)         # write probes up to S = 768 MiB
    _ffi.call("lars_d_probe", 3, 1, 8192, None, C.c_void_p(src.ptr), src.nbytes, None)
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def timed(kind, nbytes, reps, blocks):
        _ffi.call("lars_event_record", ev[0], None)
        _ffi.call("lars_d_probe", kind, reps, blocks, C.c_void_p(src.ptr), C.c_void_p(dst.ptr), nbytes, None)
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        return ms.value

    res = {}
    print("# S MiB | re-read only: GB/s read | + plain stores of 4S per sweep: GB/s read, GB/s total | + non-temporal stores: read, total")
    for s_mib in sizes:
        nbytes = s_mib * MiB
        nbytes -= nbytes % (12 * 256)
        reps = max(4, min(64, int(6144 / s_mib)))
        row = [s_mib]
        for kind in (20, 21, 22):
            if kind != 20 and s_mib > 768:
                row += [float("nan"), float("nan")]
                continue
            best = None
            for blocks in (2048, 8192):
                t = [timed(kind, nbytes, reps, blocks) for _ in range(3)]
                ms = float(np.median(t[1:]))
                best = ms if best is None or ms < best else best
            rd = nbytes * reps / best / 1e6
            row += [rd] if kind == 20 else [rd, rd * 5]
        res[s_mib] = row[1:]
        print(f"{s_mib:5d}  | {row[1]:8.0f} | {row[2]:8.0f} {row[3]:8.0f} | {row[4]:8.0f} {row[5]:8.0f}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
