#!/usr/bin/env python3
"""Instruction-issue probe: cycles per wave64 instruction per SIMD for the instructions the hot loops are made of.

    python tools/issuebench.py [--iters 4096]

Each probe kernel runs `iters` trips of 16 independent instances of one instruction in every
lane; blocks of 4 waves, `waves_per_simd` x 256 blocks (one block per CU and step).  Cycles are
derived against the shader clock reported by the runtime.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lars_image_processing_amd import _ffi  # noqa: E402
from tools.kbench import Timer  # noqa: E402

OPS = {0: "v_add_f32", 14: "v_fma_f32", 16: "v_max_f32", 7: "v_min3_f32", 12: "v_pk_add_f32", 13: "v_pk_mul_f32",
       1: "v_pk_fma_f32", 2: "v_rcp_f32", 3: "v_add_f64", 15: "v_fma_f64", 4: "v_cvt_f64_f32", 8: "v_cvt_f32_ubyte0",
       5: "v_perm_b32", 9: "v_fma_mix_f32", 11: "v_cmp_lt_f32 (vcc)", 6: "v_cmp_lt_f32 + s_bcnt1", 17: "v_addc_co_u32",
       18: "v_dot4_u32_u8", 19: "v_add_u32", 10: "ds_read_b32 (16 in flight)",
       20: "v_add_f32 a,a,b", 21: "v_max_f32 a,a,const", 22: "v_mul_f32 a,a,const", 23: "v_cvt_f32_ubyte0 in place",
       24: "v_mov_b32", 25: "v_add_f32 a,b,c", 26: "v_fma_f32 a,a,b,c", 27: "v_pk_add_f32 a,a,const", 28: "v_add_f64 a,a,const",
       29: "v_perm_b32 a,a,a,s", 30: "v_and_or_b32", 31: "v_lshl_or_b32", 32: "v_bfe_u32", 33: "v_lshrrev_b32", 34: "v_and_b32",
       35: "v_lshlrev_b32_sdwa BYTE_1", 36: "v_lshl_add_u32", 37: "v_pk_fma_f32 clamp", 38: "v_fma_f32 clamp", 39: "v_or_b32",
       40: "v_mfma_f64_4x4x4_4b_f64", 41: "v_cvt_f64_f32 + v_mfma_f64_4x4x4 (2 instr)", 42: "v_cvt_f64_f32 + v_add_f64 (2 instr)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4096)
    ap.add_argument("--mhz", type=float, default=2400.0)
    ap.add_argument("--wps", type=lambda t: [int(x) for x in t.split(",")], default=[1, 2, 4, 8], help="waves per SIMD")
    ap.add_argument("--ops", type=lambda t: [int(x) for x in t.split(",")], default=None)
    args = ap.parse_args()
    dst = _ffi.DeviceBuffer(4096)
    timer = Timer()
    out = {}
    clk = _ffi.DeviceBuffer(64)
    _ffi.call("lars_d_probe", 99, 4000000, 1024, C.c_void_p(clk.ptr), C.c_void_p(clk.ptr), 64, None)
    _ffi.call("lars_synchronize", None)
    host = clk.download(np.uint64, (8,))
    mhz = float(host[0]) / float(host[1]) * 100.0
    print(f"shader clock under a VALU loop: {mhz:.0f} MHz ({int(host[0])} cycles / {int(host[1])} ticks of 100 MHz)")
    args.mhz = mhz
    out["mhz"] = mhz
    for wps in args.wps:
        blocks = 256 * wps
        for op, name in OPS.items():
            if args.ops and op not in args.ops:
                continue
            ts = []
            for _ in range(4):
                ts.append(timer.time(lambda: _ffi.call("lars_d_probe", 100 + op, args.iters, blocks, C.c_void_p(dst.ptr),
                                                       C.c_void_p(dst.ptr), 4096, None)))
            ms = float(np.median(ts[1:]))
            cyc = ms * 1e-3 * args.mhz * 1e6 / (args.iters * 16 * wps)
            out[f"{name} wps={wps}"] = cyc
            print(f"{name:28s} waves/SIMD={wps}  {ms:8.3f} ms  {cyc:6.2f} cycles per instruction per SIMD")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
