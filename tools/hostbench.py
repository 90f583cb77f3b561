#!/usr/bin/env python3
"""PCIe-inclusive timing of the host entry points (NumPy in, NumPy out) -- never the bench `value`."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars

def best(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)

def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rng = np.random.default_rng(1234)
    img = np.clip(rng.normal((70, 90, 150), (25, 25, 40), (edge, edge, 3)), 0, 255).astype(np.uint8)   # vegetation-like R, G, NIR
    mp = edge * edge / 1e6
    lars.fix_white_balance(img)            # warm up: context, workspace
    wb = lars.fix_white_balance(img)
    idx = lars.calculate_index(wb, "NDVI")
    res = {
        "fix_white_balance": mp / best(lambda: lars.fix_white_balance(img)),
        "calculate_index": mp / best(lambda: lars.calculate_index(wb, "NDVI")),
        "analyze_index": mp / best(lambda: lars.analyze_index(idx, "NDVI")),
        "process_image(wb+3idx arrays+stats+median)": mp / best(lambda: lars.process_image(img)),
        "process_image(stats only, no arrays)": mp / best(lambda: lars.process_image(img, want_arrays=False)),
    }
    print(json.dumps({"edge": edge, "Mpix_per_s": res}, indent=1))

if __name__ == "__main__":
    main()
