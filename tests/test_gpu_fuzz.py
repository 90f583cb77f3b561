"""Seeded random images through the three swapped functions on the GPU against the oracle (which
tests/fuzz_oracle_vs_reference.py holds against the reference itself on the same kind of images): every accepted sample
type, 3 and 4 channels, 1 x 1 up to 47 x 47, constant channels, zero bands, two-level and narrow-range images.
White balance and indices bit-exact, statistics as in test_gpu_parity.py."""
import warnings

import numpy as np
import pytest

from oracle import index_oracle as orc
from test_gpu_parity import assert_stats_close, bits

pytestmark = pytest.mark.gpu

DTYPES = [np.uint8, np.uint8, np.uint8, np.uint16, np.int16, np.int32, np.float32, np.float64, np.bool_]
TYPES = ("NDVI", "GNDVI", "NDWI")


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
        img = rng.integers(max(info.min, -2000), min(info.max, 70000) + 1, (h, w, c)).astype(dt)
    if kind == 1:
        img[..., int(rng.integers(0, 3))] = img.flat[0]              # a constant channel (p98 == p2)
    elif kind == 2 and dt != np.bool_:
        img[..., int(rng.integers(0, 3))] = 0                        # a zero band
    elif kind == 3 and dt == np.uint8:
        img = (rng.integers(0, 2, (h, w, c)) * 255).astype(np.uint8)  # two levels only
    elif kind == 4 and dt == np.uint8:
        img = np.clip(rng.normal(120, 8, (h, w, c)), 0, 255).astype(np.uint8)   # narrow range: fractional percentiles
    return img


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_images_match_the_oracle(seed):
    import lars_image_processing_amd as lars
    rng = np.random.default_rng(1000 + seed)
    for case in range(60):
        img = random_image(rng)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_wb = orc.wb_app(img)
        got_wb = lars.fix_white_balance(img)
        assert got_wb.dtype == np.uint8 and got_wb.shape == want_wb.shape, (seed, case)
        np.testing.assert_array_equal(got_wb, want_wb, err_msg=f"white balance, seed {seed} case {case} {img.dtype} {img.shape}")
        for src in (img, want_wb):
            for t in TYPES:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    want = orc.index_app(src, t)
                got = lars.calculate_index(src, t)
                assert got.dtype == np.float32 and got.shape == want.shape
                np.testing.assert_array_equal(bits(got), bits(want), err_msg=f"index {t}, seed {seed} case {case} {src.dtype}")
                assert_stats_close(lars.analyze_index(got, t), orc.stats_app(want, t), samples=want)
