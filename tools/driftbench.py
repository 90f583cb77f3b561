#!/usr/bin/env python3
"""Time series in one process: plain write / read / mix probes and the plane-writing kernel on two fixed arenas, with
clocks, power and temperatures from rocm-smi every few seconds.  Separates "this allocation is slow" from "the device is
in a slow state right now".

    python tools/driftbench.py [seconds=40] [tiles=128]
"""
import ctypes as C, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars


class View:
    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def free(self):
        pass


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(out.stdout)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "mclk", "fclk", "socclk", "power", "junction", "memory)"))}
        return keep
    except Exception as exc:
        return {"error": str(exc)[:100]}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40
    tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    idx = ("NDVI", "GNDVI", "NDWI")
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    slots = 64
    plane = slots * b.npix * 4
    arenas = [_ffi.DeviceBuffer(3 * plane) for _ in range(3)]
    outs = b.make_outputs(index=False, ring=slots)
    nbytes = tiles * b.tile_bytes * 5
    nbytes -= nbytes % (60 * 256 * 4)
    dst = _ffi.DeviceBuffer(nbytes // 5 * 4 + 4096)
    _ffi.set_tuning(traverse=1)

    def timed(fn):
        _ffi.call("lars_event_record", ev[0], None)
        fn()
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        return ms.value

    def kernel(arena):
        outs.index = [View(arena.ptr + k * plane, plane) for k in range(3)]
        def go():
            for start in range(0, b.ntiles, slots):
                b.run_fused(b.fused_args(idx, True, stats, False, outs, None, start, slots))
        ms = timed(go)
        outs.index = [None] * 3
        return tiles * b.npix * 15 / ms / 1e6

    def probe(kind, n, blocks=65536):
        ms = timed(lambda: _ffi.call("lars_d_probe", kind, 1, blocks, C.c_void_p(b.tiles.ptr), C.c_void_p(dst.ptr), n, None))
        return n * (0.8 if kind == 19 else 1.0) / ms / 1e6

    t0 = time.time()
    last_smi = -1e9
    print("# t_s  kernel@arena0 kernel@arena1 kernel@arena2 | mix(kind 11) write3(kind 19) write16(kind 3) read12(kind 1)   [GB/s]")
    rows = []
    while time.time() - t0 < seconds:
        t = time.time() - t0
        row = [t] + [kernel(a) for a in arenas] + [probe(11, nbytes), probe(19, nbytes), probe(3, nbytes // 5 * 4, 16384), probe(1, nbytes // 5)]
        rows.append(row)
        print(f"{t:6.2f}  " + " ".join(f"{v:7.0f}" for v in row[1:4]) + "  | " + " ".join(f"{v:7.0f}" for v in row[4:]))
        if t - last_smi > 6:
            print("# smi", json.dumps(smi()))
            last_smi = t
        # idle gaps now and then: does the state recover?
        if int(t) % 10 == 9:
            time.sleep(1.0)
    _ffi.set_tuning(traverse=-1)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
