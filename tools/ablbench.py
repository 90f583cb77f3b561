#!/usr/bin/env python3
"""What the three-index statistics kernel spends its time on: the same kernel built with one ingredient left out at a time
(LARS_ABLATE, fused_v2.hip; results of those builds are wrong), each library in its own process, same tiles.

    make -C lars_image_processing_amd/csrc abl      # builds build/abl/liblars_abl<mask>.so
    python tools/ablbench.py --masks 0,64,65,66,68,72,71
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="0,64,65,66,68,72,79")
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--modes", default="stats_3idx,stats_ndvi")
    args = ap.parse_args()
    names = {1: "min/max", 2: "float64 sums", 4: "coverage counters", 8: "quotient correction", 64: "(xor sink)"}
    for mask in map(int, args.masks.split(",")):
        lib = os.path.join(ROOT, "build", "abl", f"liblars_abl{mask}.so")
        env = dict(os.environ, LARS_HIP_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--tiles", str(args.tiles), "--rounds", "5", "--what",
                              "fused", "--modes", args.modes, "--impls", "2", "--nt", "0"], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode:
            print(f"mask {mask}: failed\n{out.stderr[-800:]}")
            continue
        res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        left_out = " + ".join(n for b, n in names.items() if mask & b) or "nothing"
        print(f"mask {mask:3d} without {left_out:60s} " + "  ".join(f"{k.split()[1]} {v['ms']:.3f} ms" for k, v in res.items()))


if __name__ == "__main__":
    main()
