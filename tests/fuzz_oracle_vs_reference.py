#!/usr/bin/env python3
"""Differential fuzz of the oracle against the reference ITSELF (build container only: /root/reference does not exist on
the GPU box; pytest does not collect this file and nothing imports it).  Random small images -- shapes, sample types, degenerate channels --
go through the reference's fix_white_balance / calculate_index / analyze_index (loaded as tools/gen_golden.py loads them)
and through oracle/index_oracle.py's statement functions and closed forms; every array must be bit-identical, every
dictionary equal.  The committed goldens pin a dozen fixed cases; this widens the net.

    MPLBACKEND=Agg python tests/fuzz_oracle_vs_reference.py --cases 400 --seed 0
"""
from __future__ import annotations

import argparse
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

import gen_golden  # noqa: E402
from oracle import index_oracle as orc  # noqa: E402

DTYPES = [np.uint8, np.uint8, np.uint8, np.uint16, np.int16, np.int32, np.float32, np.float64, np.bool_]


def random_image(rng):
    h, w = int(rng.integers(1, 48)), int(rng.integers(1, 48))
    c = int(rng.choice([3, 3, 3, 4]))
    dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
    kind = int(rng.integers(0, 6))
    if dt == np.bool_:
        img = rng.integers(0, 2, (h, w, c)).astype(np.bool_)
    elif np.issubdtype(dt, np.floating):
        img = (rng.normal(100, 60, (h, w, c))).astype(dt)
    else:
        info = np.iinfo(dt)
        lo, hi = max(info.min, -2000), min(info.max, 70000)
        img = rng.integers(lo, hi + 1, (h, w, c)).astype(dt)
    if kind == 1:
        img[..., int(rng.integers(0, 3))] = img.flat[0]              # a constant channel (p98 == p2)
    elif kind == 2 and dt != np.bool_:
        img[..., int(rng.integers(0, 3))] = 0                        # a zero band
    elif kind == 3 and dt == np.uint8:
        img = (rng.integers(0, 2, (h, w, c)) * 255).astype(np.uint8)  # two levels only
    elif kind == 4 and dt == np.uint8:
        img = np.clip(rng.normal(120, 8, (h, w, c)), 0, 255).astype(np.uint8)   # narrow range: fractional percentiles
    return img


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    gen_golden._stub_absent_modules()
    app = gen_golden._load(args.reference, "process-images.py", "ref_process_images")
    rng = np.random.default_rng(args.seed)
    counts = {"wb": 0, "wb_closed_form": 0, "index": 0, "index_closed_form": 0, "stats": 0}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for case in range(args.cases):
            img = random_image(rng)
            want_wb = app.fix_white_balance(img)
            assert same(orc.wb_app(img), want_wb), ("wb_app", case, img.dtype, img.shape)
            counts["wb"] += 1
            if img.dtype in (np.uint8, np.uint16):
                assert same(orc.wb_closed_form(img), want_wb), ("wb_closed_form", case, img.dtype, img.shape)
            else:
                assert same(orc.wb_float_closed_form(img)[0], want_wb), ("wb_float_closed_form", case, img.dtype, img.shape)
            counts["wb_closed_form"] += 1
            for src in (img, want_wb):
                for t in ("NDVI", "GNDVI", "NDWI"):
                    want = app.calculate_index(src, t)
                    got = orc.index_app(src, t)
                    assert same(got, want), ("index_app", case, t, src.dtype, src.shape)
                    counts["index"] += 1
                    if src.dtype in (np.uint8, np.uint16):
                        hi, lo = orc._band_pair(t)
                        closed = orc.index_closed_form(src[..., hi], src[..., lo])
                        if t == "NDWI":                      # the kernels' route: -GNDVI + 0.0
                            closed_neg = (np.float32(0) - orc.index_closed_form(src[..., 2], src[..., 1])).astype(np.float32)
                            assert same(closed_neg, want), ("index via -GNDVI", case, t)
                        assert same(closed, want), ("index_closed_form", case, t)
                        counts["index_closed_form"] += 1
                    ws, gs = app.analyze_index(want, t), orc.stats_app(want, t)
                    assert list(ws.keys()) == list(gs.keys()), ("stats keys", case, t)
                    for k in ws:
                        assert (ws[k] == gs[k]) or (ws[k] != ws[k] and gs[k] != gs[k]), ("stats", case, t, k, ws[k], gs[k])
                    counts["stats"] += 1
    print(f"{args.cases} random images (seed {args.seed}): oracle == reference for " +
          ", ".join(f"{v} x {k}" for k, v in counts.items()) + f"; numpy {np.__version__}")


if __name__ == "__main__":
    main()
