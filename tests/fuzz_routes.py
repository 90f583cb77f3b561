#!/usr/bin/env python3
"""Differential fuzz of the two statistics routes on the GPU: random batches (shapes incl. ragged ones, RGB / RGBA, contents from
iid noise to constant channels, saturated and two-valued tiles), random index subsets and flags -- the one-read route
(csrc/joint.hip) must give the records, medians, tables, percentiles and channel histograms of the per-pixel route, bit for bit.

    python tests/fuzz_routes.py [--cases 300] [--seed 0] [--max-edge 200]
"""
import argparse
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi as _lars_ffi  # noqa: E402,F401

TYPES = ("NDVI", "GNDVI", "NDWI")
SUBSETS = [c for r in (1, 2, 3) for c in itertools.combinations(TYPES, r)]


def content(rng, ntiles, h, w, ch):
    kind = rng.choice(["iid", "narrow", "const_channel", "two_valued", "saturated", "gradient", "runs", "flat", "mixed_tiles", "rare_outlier"])
    t = np.empty((ntiles, h, w, ch), dtype=np.uint8)
    for i in range(ntiles):
        k = kind if kind != "mixed_tiles" else rng.choice(["iid", "flat", "const_channel", "gradient", "two_valued"])
        if k == "iid":
            t[i] = rng.integers(0, 256, (h, w, ch))
        elif k == "narrow":
            lo = rng.integers(0, 250)
            t[i] = rng.integers(lo, lo + rng.integers(1, 6), (h, w, ch))
        elif k == "const_channel":
            t[i] = rng.integers(0, 256, (h, w, ch))
            t[i, :, :, rng.integers(0, 3)] = rng.integers(0, 256)
        elif k == "two_valued":
            t[i] = np.where(rng.random((h, w, ch)) < rng.uniform(0.005, 0.995), 0, 255)
        elif k == "saturated":
            t[i] = np.clip(rng.normal(240, 30, (h, w, ch)), 0, 255)
        elif k == "gradient":
            g = np.linspace(rng.integers(0, 100), rng.integers(100, 256), w)[None, :, None] + rng.normal(0, rng.uniform(0, 3), (h, w, ch))
            t[i] = np.clip(g, 0, 255)
        elif k == "runs":
            base = rng.integers(0, 256, (h, (w + 63) // 64, ch))
            t[i] = np.repeat(base, 64, axis=1)[:, :w]
        elif k == "flat":
            t[i] = rng.integers(0, 256, (1, 1, ch))
        elif k == "rare_outlier":
            t[i] = rng.integers(100, 104, (h, w, ch))
            for _ in range(int(rng.integers(1, 4))):
                t[i, rng.integers(0, h), rng.integers(0, w), rng.integers(0, 3)] = rng.choice([0, 255])
    return kind, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-edge", type=int, default=200)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    seen, windowed, recounted = {}, {}, {}
    for case in range(args.cases):
        ntiles = int(rng.integers(1, 6))
        h, w = int(rng.integers(1, args.max_edge)), int(rng.integers(1, args.max_edge))
        if ntiles > 1 and (h * w) % 4:
            w += 4 - w % 4                                      # ragged tiles come one per batch (the library's rule)
        ch = int(rng.choice([3, 3, 4]))
        kind, tiles = content(rng, ntiles, h, w, ch)
        indices = SUBSETS[int(rng.integers(0, len(SUBSETS)))]
        wb, hist, sumsq, med = (bool(rng.integers(0, 2)) for _ in range(4))
        # the one-read route's tables: 0 full, 3 windowed where they fit (tiles of any size), 2 windows that miss on purpose (recount),
        # 4 three windows (NIR as well) wherever they fit, 5 the same with NIR windows that miss
        window = int(rng.choice([0, 3, 3, 2, 4, 4, 5]))
        _lars_ffi.set_tuning(joint_window=window)
        b = lars.TileBatch.from_host(tiles)
        what = f"case {case}: {kind} {ntiles}x{h}x{w}x{ch} {indices} wb={wb} hist={hist} sumsq={sumsq} medians={med} window={window}"
        try:
            if not b.can_joint():
                raise AssertionError("can_joint() is False")
            rc = b.process(indices=indices, white_balance=wb, hist=hist, sumsq=sumsq, medians=med, route="classic")
            if wb:
                tab_c, pct_c, hist_c = b.host_tables(), b.host_percentiles(), b.host_hist()
                b.table.zero(); b.percentiles.zero(); b.hist.zero()
            rj = b.process(indices=indices, white_balance=wb, hist=hist, sumsq=sumsq, medians=med, route="joint", channel_hist=window == 0)
            nwin, nrec = b.joint_window_report()
            windowed[window] = windowed.get(window, 0) + nwin
            recounted[window] = recounted.get(window, 0) + nrec
            assert window != 0 or nwin == 0
            assert window not in (2, 5) or nrec == nwin
            (rec_c, med_c), (rec_j, med_j) = (rc if med else (rc, None)), (rj if med else (rj, None))
            a, c = rec_c.copy(), rec_j.copy()
            np.testing.assert_allclose(a["sumsq"], c["sumsq"], rtol=1e-12, atol=2.0 ** -26)   # rounded to 2^-32 per workgroup / per tile
            a["sumsq"] = c["sumsq"] = 0
            assert a.tobytes() == c.tobytes(), "records differ"
            if med:
                assert np.array_equal(med_c, med_j, equal_nan=True), ("medians differ", med_c, med_j)
            if wb:
                chans = sorted({2} | ({0} if "NDVI" in indices else set()) | ({1} if set(indices) & {"GNDVI", "NDWI"} else set()))
                tab_j, pct_j = b.host_tables(partial=True), b.host_percentiles(partial=True)
                hist_j = b.host_hist(partial=True) if window == 0 else None
                for k in chans:
                    assert hist_j is None or np.array_equal(hist_j[:, k], hist_c[:, k]), f"channel histogram {k}"
                    assert pct_j[:, k].tobytes() == pct_c[:, k].tobytes(), f"percentiles {k}"
                    assert np.array_equal(tab_j[:, k], tab_c[:, k]), f"table {k}"
        except Exception as e:
            print("FAILED", what, "--", repr(e)[:400], flush=True)
            np.save(os.path.join("gpurun_out", f"fuzz_fail_{case}.npy"), tiles) if os.path.isdir("gpurun_out") else None
            return 1
        finally:
            b.free()
        seen[kind] = seen.get(kind, 0) + 1
        if case % 50 == 49:
            print(f"{case + 1} cases ok", flush=True)
    _lars_ffi.set_tuning(joint_window=1)
    print(f"{args.cases} random batches (seed {args.seed}): one-read route == per-pixel route; contents {dict((str(k), v) for k, v in seen.items())}; "
          f"tiles counted on windowed tables by joint_window setting {windowed}, recounted {recounted}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
