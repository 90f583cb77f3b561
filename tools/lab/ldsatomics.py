#!/usr/bin/env python3
"""LDS atomics of the one-read statistics route in isolation (lars_d_probe kinds 70..73): what a returning add costs next to the
plain one, and what the period scan (two barriers + a sweep of the 128 KiB table every 12 steps) costs.

    python tools/lab/ldsatomics.py [steps=4096]
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lablib
from lars_image_processing_amd import _ffi


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    a, b = C.c_void_p(), C.c_void_p()
    _ffi.call("lars_event_create", C.byref(a)); _ffi.call("lars_event_create", C.byref(b))
    names = {70: "ds_add_u32 (no return)", 71: "ds_add_rtn_u32 + threshold check one step later", 72: "ds_add_u32 + scan every 12 steps",
             73: "ds_add_rtn_u32, waited for, unused"}
    for blocks in (256, 1024):
        for kind in (70, 71, 72, 73):
            ts = []
            for _ in range(5):
                _ffi.call("lars_event_record", a, None)
                lablib.probe(kind, steps, blocks, None, None, 1)
                _ffi.call("lars_event_record", b, None)
                ms = C.c_float(0); _ffi.call("lars_event_elapsed_ms", a, b, C.byref(ms)); ts.append(ms.value)
            t = float(np.median(ts[1:]))
            rounds = blocks / 256.0
            # per CU: steps * 16 waves * 4 wave-instructions per round of workgroups
            per_instr_ns = t * 1e6 / (steps * 16 * 4 * rounds)
            print(f"blocks {blocks:5d}  {names[kind]:48s} {t:8.3f} ms   {per_instr_ns:6.2f} ns per wave-instruction and CU = {per_instr_ns * 2.4:5.2f} cycles at 2.4 GHz", flush=True)


if __name__ == "__main__":
    main()
