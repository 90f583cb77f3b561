#!/usr/bin/env python3
"""The whole step (histograms -> tables -> three planes + statistics) as separate launches vs the persistent pipeline
(csrc/lab/pipeline.hip, liblars_lab.so), interleaved in one process, same output arena; sweeps the pipeline's item size and
H-item lead.

    python tools/lab/pipebench.py [tiles=256] [rounds=4] [arenas=3]
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lablib
from lars_image_processing_amd import _ffi
import lars_image_processing_amd as lars

IDX = ("NDVI", "GNDVI", "NDWI")


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    narena = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    b = lars.TileBatch.synthetic(tiles, 4096, 4096, seed=1234, profile="vegetation")
    stats = b.new_stats()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    slots = 64

    def timed(fn):
        _ffi.call("lars_event_record", ev[0], None)
        fn()
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        return ms.value

    res = {}
    variants = [("two-pass", None)] + [(f"pipeline spi={spi} head={head}", (spi, head))
                                       for spi, head in ((16, 8), (32, 4), (32, 8), (32, 16), (64, 4), (64, 8), (64, 16), (128, 8), (256, 8))]
    for a in range(narena):
        outs = b.make_outputs(index=True, ring=slots, arena="plain")
        times = {name: [] for name, _ in variants}
        recs = {}
        for r in range(rounds + 1):
            for name, cfg in variants:
                if cfg is None:
                    def go():
                        b.compute_wb_tables()
                        for start in range(0, b.ntiles, slots):
                            b.run_fused(b.fused_args(IDX, True, stats, False, outs, None, start, slots))
                else:
                    lablib.set_tuning(pipe_steps=cfg[0], pipe_head=cfg[1])
                    def go():
                        for start in range(0, b.ntiles, slots):
                            lablib.run_pipeline(b, stats, outs, None, start, slots)
                times[name].append(timed(go))
                if r == 0 and a == 0:
                    recs[name] = (stats.download(_ffi.STATS_DTYPE, (b.ntiles, 3)).tobytes(), outs.host_index("NDWI", 5, 1).tobytes(),
                                  b.host_percentiles().tobytes())
        if a == 0:
            for name, _ in variants[1:]:
                assert recs[name] == recs["two-pass"], f"{name}: results differ from the two-pass path"
            print("records, planes and percentiles identical across variants")
        base = float(np.median(times["two-pass"][1:]))
        for name, _ in variants:
            med = float(np.median(times[name][1:]))
            gbs = tiles * b.npix * 15 / med / 1e6
            res[f"arena{a} {name}"] = med
            print(f"arena {a}  {name:26s} {med:8.3f} ms per {tiles} tiles  {tiles * b.npix / med / 1e6:8.1f} Gpix/s  whole step {gbs:7.1f} GB/s algorithmic = {gbs / 8000:.3f} of 8 TB/s   x{base / med:.3f}")
        outs.free()
    lablib.set_tuning(pipe_steps=0, pipe_head=0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
