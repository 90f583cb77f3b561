#!/usr/bin/env python3
"""Latency of the host entry points (NumPy in, NumPy out, PCIe included) at the sizes the reference really calls them with,
next to the NumPy oracle (== the reference's expressions) on the same box.  Never the bench `value`.

The Streamlit app caps every upload at a 1024-pixel long edge before anything else touches it
(process-images.py:398-422, :1444, :1130), so its calls see 0.07 - 1 Mpix; the directory driver (backend-process.py:49-97)
sees whole files.

    python tests/latency_hostcalls.py [--sizes 256,512,1024,2048,4096] [--calls 30] [--json out.json]

Per size and entry point: p50 and p95 of `calls` calls in milliseconds, fresh result arrays (the default), GPU library and
oracle alike; "x" = oracle p50 / library p50.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lars_image_processing_amd as lars  # noqa: E402
from oracle import index_oracle as orc  # noqa: E402  (the checker, timed here as the CPU column of the table: this script lives under tests/ for that reason)


def pcts(fn, calls, warm=1):
    for _ in range(warm):                                    # warm-up: context, workspace, page faults (and, for the device, its clocks)
        fn()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.percentile(ts, 50)), float(np.percentile(ts, 95))


def after_idle(fn, idle_s, calls=8):
    """Median latency of a call that follows ``idle_s`` seconds of doing nothing (what a UI callback sees)."""
    ts = []
    for _ in range(calls):
        time.sleep(idle_s)
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def oracle_process(img):
    wb = orc.wb_app(img)
    out = {}
    for t in ("NDVI", "GNDVI", "NDWI"):
        plane = orc.index_app(wb, t)
        out[t] = (plane, orc.stats_app(plane, t))
    return wb, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256,512,1024,2048")
    ap.add_argument("--calls", type=int, default=30)
    ap.add_argument("--json", default=None)
    ap.add_argument("--warm", type=int, default=40, help="warm-up calls before the back-to-back measurement")
    ap.add_argument("--idle", type=float, default=0.3, help="seconds of idling before each call of the 'idle' column")
    args = ap.parse_args()
    rng = np.random.default_rng(1234)
    rows = []
    warnings.simplefilter("ignore")
    print(f"# device {lars._ffi.device_name()}; NumPy {np.__version__}; {args.calls} calls per cell; ms per call.  'library': p50 / p95 of calls back to back after "
          f"{args.warm} warm-up calls; 'idle': median of calls that each follow {args.idle} s of idling (device clocks and the link have dropped: what a UI callback sees)")
    print(f"{'edge':>5s} {'entry point':44s} {'library p50':>11s} {'p95':>8s} {'idle':>8s} {'NumPy p50':>10s} {'p95':>8s} {'x':>6s}")
    for edge in [int(x) for x in args.sizes.split(",")]:
        img = np.clip(rng.normal((70, 90, 150), (25, 25, 40), (edge, edge, 3)), 0, 255).astype(np.uint8)   # vegetation-like R, G, NIR
        wb = lars.fix_white_balance(img)
        idx = lars.calculate_index(wb, "NDVI")
        calls = max(5, args.calls if edge <= 2048 else args.calls // 3)
        cases = [
            ("fix_white_balance(img)", lambda: lars.fix_white_balance(img), lambda: orc.wb_app(img)),
            ("calculate_index(wb, 'NDVI')", lambda: lars.calculate_index(wb, "NDVI"), lambda: orc.index_app(wb, "NDVI")),
            ("analyze_index(ndvi, 'NDVI')", lambda: lars.analyze_index(idx, "NDVI"), lambda: orc.stats_app(idx, "NDVI")),
            ("process_image(img): wb + 3 planes + 3 dicts", lambda: lars.process_image(img), lambda: oracle_process(img)),
            ("process_image(img, want_arrays=False)", lambda: lars.process_image(img, want_arrays=False), None),
        ]
        for name, gpu_fn, cpu_fn in cases:
            g50, g95 = pcts(gpu_fn, calls, warm=args.warm)
            gidle = after_idle(gpu_fn, args.idle)
            c50, c95 = pcts(cpu_fn, max(3, calls // 3)) if cpu_fn else (float("nan"), float("nan"))
            rows.append({"edge": edge, "entry": name, "library_ms_p50": g50, "library_ms_p95": g95, "library_ms_after_idle": gidle,
                         "numpy_ms_p50": c50, "numpy_ms_p95": c95})
            print(f"{edge:5d} {name:44s} {g50:11.3f} {g95:8.3f} {gidle:8.3f} {c50:10.3f} {c95:8.3f} {c50 / g50:6.1f}", flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
