#!/usr/bin/env python3
"""What the device sustains for plain streaming with the hot path's access shapes (lars_d_probe, liblars_lab.so)."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lablib
from lars_image_processing_amd import _ffi

def main():
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    nbytes = gib << 30
    nbytes -= nbytes % (48 * 1024)
    src, dst = _ffi.DeviceBuffer(nbytes), _ffi.DeviceBuffer(nbytes)
    lablib.probe(3, 1, 8192, None, src.ptr, nbytes)
    a, b = C.c_void_p(), C.c_void_p()
    _ffi.call("lars_event_create", C.byref(a)); _ffi.call("lars_event_create", C.byref(b))
    res = {}
    for kind, name, mult in ((0, "read16", 1), (1, "read12", 1), (2, "copy16", 2), (3, "write16", 1), (4, "write16nt", 1),
                             (5, "mix12r48w", 1), (6, "mix12r48w_nt", 1), (7, "mix12r16w", 1)):
        for unroll in ((1, 4) if kind < 2 else (1,)):
            for blocks in (2048, 8192, 16384, 65536, 262144):
                ts = []
                for _ in range(4):
                    _ffi.call("lars_event_record", a, None)
                    lablib.probe(kind, unroll, blocks, src.ptr, dst.ptr, nbytes)
                    _ffi.call("lars_event_record", b, None)
                    ms = C.c_float(0); _ffi.call("lars_event_elapsed_ms", a, b, C.byref(ms)); ts.append(ms.value)
                t = float(np.median(ts[1:]))
                res[f"{name} unroll={unroll} blocks={blocks}"] = nbytes * mult / t / 1e6
    for k, v in res.items():
        print(f"{k:40s} {v:9.1f} GB/s")
    print(json.dumps(res))

if __name__ == "__main__":
    main()
