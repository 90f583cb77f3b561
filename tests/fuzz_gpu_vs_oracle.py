#!/usr/bin/env python3
"""Differential fuzz of the host entry points on the GPU against the oracle (the same random images
tests/fuzz_oracle_vs_reference.py sends through the reference itself in the build container, so that reference == oracle
there and oracle == GPU here): fix_white_balance, calculate_index on the original and on the corrected image, analyze_index,
and process_image (one upload: white balance + indices + statistics).  Arrays bit for bit, statistics exact except the mean
(1e-6 of max(|mean|, mean|x|)).

    python tests/fuzz_gpu_vs_oracle.py [--cases 400] [--seed 0] [--max-edge 48]
"""
import argparse
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars  # noqa: E402
from oracle import index_oracle as orc  # noqa: E402

DTYPES = [np.uint8, np.uint8, np.uint8, np.uint16, np.int16, np.int32, np.float32, np.float64, np.bool_]
TYPES = ("NDVI", "GNDVI", "NDWI")


def random_image(rng, max_edge):
    h, w = int(rng.integers(1, max_edge)), int(rng.integers(1, max_edge))
    c = int(rng.choice([3, 3, 3, 4]))
    dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
    kind = int(rng.integers(0, 6))
    if dt == np.bool_:
        img = rng.integers(0, 2, (h, w, c)).astype(np.bool_)
    elif np.issubdtype(dt, np.floating):
        img = (rng.normal(100, 60, (h, w, c))).astype(dt)
    else:
        info = np.iinfo(dt)
        img = rng.integers(max(info.min, -2000), min(info.max, 70000) + 1, (h, w, c)).astype(dt)
    if kind == 1:
        img[..., int(rng.integers(0, 3))] = img.flat[0]
    elif kind == 2 and dt != np.bool_:
        img[..., int(rng.integers(0, 3))] = 0
    elif kind == 3 and dt == np.uint8:
        img = (rng.integers(0, 2, (h, w, c)) * 255).astype(np.uint8)
    elif kind == 4 and dt == np.uint8:
        img = np.clip(rng.normal(120, 8, (h, w, c)), 0, 255).astype(np.uint8)
    return img


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def stats_ok(got, want, samples):
    if list(got) != list(want):
        return False
    scale = float(np.mean(np.abs(np.asarray(samples, dtype=np.float64))))
    for k, v in want.items():
        if k.startswith("Mean"):
            if abs(got[k] - v) > 1e-6 * max(abs(v), scale):
                return False
        elif not (got[k] == v or (got[k] != got[k] and v != v)):
            return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-edge", type=int, default=48)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    counts = {"wb": 0, "index": 0, "stats": 0, "process_image": 0}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for case in range(args.cases):
            img = random_image(rng, args.max_edge)
            tag = (case, img.dtype.name, img.shape)
            want_wb = orc.wb_app(img)
            got_wb = lars.fix_white_balance(img)
            assert same(got_wb, want_wb), ("fix_white_balance", tag, int((np.asarray(got_wb) != want_wb).sum()))
            counts["wb"] += 1
            for src in (img, want_wb):
                for t in TYPES:
                    want = orc.index_app(src, t)
                    assert same(lars.calculate_index(src, t), want), ("calculate_index", tag, t, src.dtype.name)
                    counts["index"] += 1
                    assert stats_ok(lars.analyze_index(want, t), orc.stats_app(want, t), want), ("analyze_index", tag, t)
                    counts["stats"] += 1
            if img.dtype in (np.uint8, np.uint16):
                res = lars.process_image(img, indices=list(TYPES), white_balance=True)
                assert same(res["corrected"], want_wb), ("process_image corrected", tag)
                for t in TYPES:
                    want = orc.index_app(want_wb, t)
                    assert same(res["indices"][t]["index"], want), ("process_image index", tag, t)
                    assert stats_ok(res["indices"][t]["stats"], orc.stats_app(want, t), want), ("process_image stats", tag, t)
                counts["process_image"] += 1
            if case % 100 == 99:
                print(f"{case + 1} cases ok", flush=True)
    print(f"{args.cases} random images (seed {args.seed}): GPU == oracle for " + ", ".join(f"{v} x {k}" for k, v in counts.items()))


if __name__ == "__main__":
    main()
