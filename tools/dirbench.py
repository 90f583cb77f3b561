#!/usr/bin/env python3
"""Directory driver end to end: files in -> white balance + three indices -> files out.

    python tools/dirbench.py [--files 16] [--edge 4096] [--workers 8]
Writes synthetic uncompressed 8-bit TIFFs to a temporary directory, then times driver.batch_process for the output
flavours (RGBA PNG colormaps at zlib level 1, palette PNGs, uncompressed TIFF colormaps) and worker counts.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import driver, tiffio  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=16)
    ap.add_argument("--edge", type=int, default=4096)
    ap.add_argument("--workers", default="1,8")
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix="lars_dirbench_")
    try:
        src = os.path.join(root, "in")
        os.mkdir(src)
        rng = np.random.default_rng(0)
        base = rng.integers(0, 256, (args.edge, args.edge, 3), dtype=np.uint8)
        for i in range(args.files):
            tiffio.write_tiff(os.path.join(src, f"scene_{i:03d}.tif"), np.roll(base, i * 17, axis=1), rows_per_strip=64)
        npix = args.files * args.edge * args.edge
        results = {}
        for fmt in ("tiff", "png8", "png"):
            for w in map(int, args.workers.split(",")):
                if fmt != "tiff" and w == 1:
                    continue                                  # minutes of zlib on one thread: nothing to learn
                dst = os.path.join(root, f"out_{fmt}_{w}")
                t0 = time.perf_counter()
                res = driver.batch_process(src, dst, process_wb=True, process_ndvi=True, process_gndvi=True, process_ndwi=True,
                                           lut_format=fmt, workers=w, verbose=False)
                dt = time.perf_counter() - t0
                bad = [k for k, v in res.items() if isinstance(v, Exception)]
                assert not bad, (bad, res[bad[0]])
                results[f"{fmt} outputs, {w} workers"] = {"s": dt, "files_per_s": args.files / dt, "Mpix_per_s": npix / dt / 1e6}
                shutil.rmtree(dst)
        for k, v in results.items():
            print(f"{k:28s} " + "  ".join(f"{kk}={vv:9.2f}" for kk, vv in v.items()), flush=True)
        print(json.dumps(results))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
