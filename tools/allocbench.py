#!/usr/bin/env python3
"""Where the output planes live vs how fast the plane-writing kernel runs (one process, interleaved): separate
allocations, one block split three ways, odd paddings between planes, uncached / fine-grained device memory.

    python tools/allocbench.py [tiles=256] [rounds=4]
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars


class View:
    """Non-owning device range with DeviceBuffer's interface."""
    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def free(self):
        pass


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    idx = ("NDVI", "GNDVI", "NDWI")
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    slots = 64
    plane = slots * b.npix * 4
    MiB = 1 << 20
    keep = []
    configs = {}

    def separate(kind=None):
        if kind:
            os.environ["LARS_MALLOC_KIND"] = kind
        bufs = [_ffi.DeviceBuffer(plane) for _ in range(3)]
        os.environ.pop("LARS_MALLOC_KIND", None)
        keep.extend(bufs)
        return bufs

    def block(pad):
        blk = _ffi.DeviceBuffer(3 * plane + 3 * pad + 4096)
        keep.append(blk)
        return [View(blk.ptr + k * (plane + pad), plane) for k in range(3)]

    configs["separate hipMalloc #1"] = separate()
    configs["separate hipMalloc #2"] = separate()
    configs["separate hipMalloc #3"] = separate()
    configs["one block, planes back to back"] = block(0)
    configs["one block, +2 MiB between planes"] = block(2 * MiB)
    configs["one block, +36 MiB between planes"] = block(36 * MiB)
    configs["one block, +341 MiB between planes"] = block(341 * MiB)
    configs["one block, +4 KiB between planes"] = block(4096)
    for kind, name in (("1", "uncached (hipDeviceMallocUncached)"), ("2", "fine-grained (hipDeviceMallocFinegrained)")):
        try:
            configs[name] = separate(kind)
        except _ffi.LarsError as exc:
            print(f"{name}: {exc}")
    outs = b.make_outputs(index=False, ring=slots)
    times = {(n, t): [] for n in configs for t in (1, 2)}
    for r in range(rounds + 1):
        for name, bufs in configs.items():
            outs.index = list(bufs)
            for trav in (1, 2):
                _ffi.set_tuning(traverse=trav)
                _ffi.call("lars_event_record", ev[0], None)
                for start in range(0, b.ntiles, slots):
                    b.run_fused(b.fused_args(idx, True, stats, False, outs, None, start, slots))
                _ffi.call("lars_event_record", ev[1], None)
                _ffi.call("lars_synchronize", None)
                ms = C.c_float(0)
                _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
                times[(name, trav)].append(ms.value)
    res = {}
    for name, bufs in configs.items():
        row = []
        for trav in (1, 2):
            gbs = tiles * b.npix * 15 / float(np.median(times[(name, trav)][1:])) / 1e6
            res[f"{name} traverse={trav}"] = gbs
            row.append(f"traverse={trav}: {gbs:7.1f} GB/s ({gbs / 8000:.3f})")
        print(f"{name:45s} bases {[hex(x.ptr) for x in bufs]}  " + "   ".join(row))
    outs.index = [None] * 3
    _ffi.set_tuning(traverse=-1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
