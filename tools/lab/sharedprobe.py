#!/usr/bin/env python3
"""What a CU can take in when several workgroups read the same bytes at the same moment (lars_d_probe kinds 31..34 / 41..44):
R workgroups of 1024 threads per chunk, 8 apart in dispatch order (one XCD), one workgroup per CU, six loads in flight.

    python tools/lab/sharedprobe.py [GiB=12] [loads in flight per lane, e.g. 6,8,12,16]
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lablib
from lars_image_processing_amd import _ffi


def main():
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    nbytes = gib << 30
    src = _ffi.DeviceBuffer(nbytes)
    lablib.probe(3, 1, 8192, None, src.ptr, nbytes - nbytes % 960)
    a, b = C.c_void_p(), C.c_void_p()
    _ffi.call("lars_event_create", C.byref(a)); _ffi.call("lars_event_create", C.byref(b))
    depths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6]
    for wide, depth in [(w, d) for w in (0, 1) for d in depths if not (w and d == 16)]:
        for readers in (1, 2, 3, 4):
            for chunks in (512, 1024):
                ts = []
                for _ in range(4):
                    _ffi.call("lars_event_record", a, None)
                    lablib.probe((40 if wide else 30) + readers, depth, chunks, src.ptr, None, nbytes)
                    _ffi.call("lars_event_record", b, None)
                    ms = C.c_float(0); _ffi.call("lars_event_elapsed_ms", a, b, C.byref(ms)); ts.append(ms.value)
                t = float(np.median(ts[1:]))
                print(f"{16 if wide else 12} B per lane  {depth:2d} loads in flight  readers {readers}  chunks {chunks:5d}: {t:7.3f} ms   bytes once {nbytes / t / 1e6:7.0f} GB/s   "
                      f"into the CUs {readers * nbytes / t / 1e6:7.0f} GB/s = {readers * nbytes / t / 1e6 / 256 / 2.1:5.1f} B per clock and CU (at 2.1 GHz)", flush=True)


if __name__ == "__main__":
    main()
