#!/usr/bin/env python3
"""Does it matter WHICH XCD writes to a piece of memory?  Slices of 1 GiB of several arenas (plain hipMalloc and physically
contiguous), written (and read) by the workgroups of one XCD at a time and by all eight.  GB/s per (slice, XCD): a
pattern over the XCDs would mean that physical memory has a near and a far side.

    python tools/xcdbench.py [arenas_per_kind=2]
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
from tools.kbench import Timer


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    gib = 1 << 30
    arenas = []
    for name, kind in (("hipMalloc", None), ("contiguous", "4")):
        for k in range(per):
            if kind:
                os.environ["LARS_MALLOC_KIND"] = kind
            else:
                os.environ.pop("LARS_MALLOC_KIND", None)
            arenas.append((name, _ffi.DeviceBuffer(12 * gib)))
    os.environ.pop("LARS_MALLOC_KIND", None)
    timer = Timer()
    blocks = 8 * 1024
    for kind, what in ((26, "write"), (27, "read")):
        for name, arena in arenas:
            for sl in (0, 5, 11):
                ptr = arena.ptr + sl * gib
                row = []
                for xcd in list(range(8)) + [8]:
                    ts = [timer.time(lambda: _ffi.call("lars_d_probe", kind, xcd, blocks, None, C.c_void_p(ptr), gib, None)) for _ in range(4)]
                    row.append(gib / float(np.median(ts[1:])) / 1e6)
                print(f"{what:5s} {name:10s} {arena.ptr:#x} + {sl:2d} GiB: per XCD " + " ".join(f"{v:5.0f}" for v in row[:8]) + f"   all XCDs {row[8]:5.0f} GB/s")


if __name__ == "__main__":
    main()
