"""Property tests through the C ABI (hypothesis drives shapes and contents; needs a MI355X).
LARS_TEST_DEEP=k multiplies the number of examples for a one-off deep run."""
import os
import warnings

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as hnp

from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu
TYPES = ("NDVI", "GNDVI", "NDWI")
DEEP = max(1, int(os.environ.get("LARS_TEST_DEEP", "1")))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


images = st.one_of(
    hnp.arrays(np.uint8, st.tuples(st.integers(1, 40), st.integers(1, 40), st.sampled_from([3, 4]))),
    hnp.arrays(np.uint16, st.tuples(st.integers(1, 24), st.integers(1, 24), st.just(3))),
    # few distinct values: interpolated percentiles, constant channels, steep tables
    hnp.arrays(np.uint8, st.tuples(st.integers(1, 40), st.integers(1, 40), st.just(3)), elements=st.sampled_from([0, 1, 2, 127, 254, 255])),
)


@settings(max_examples=120 * DEEP, deadline=None)
@given(images)
def test_process_image_equals_oracle(img):
    import lars_image_processing_amd as lars
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_wb = orc.wb_app(img)
    res = lars.process_image(img, want_hist=True)
    np.testing.assert_array_equal(res["corrected"], want_wb)
    for t in TYPES:
        want = orc.index_app(want_wb, t)
        r = res["indices"][t]
        np.testing.assert_array_equal(bits(r["index"]), bits(want))
        ws = orc.stats_app(want, t)
        for key, val in ws.items():
            if key.startswith("Mean"):
                assert abs(r["stats"][key] - val) <= 1e-6 * max(abs(val), float(np.mean(np.abs(want))))
            else:
                assert r["stats"][key] == val, key
        np.testing.assert_array_equal(r["hist"], orc.hist50(want))


@settings(max_examples=60 * DEEP, deadline=None)
@given(hnp.arrays(np.float32, st.integers(1, 3000), elements=st.floats(-1, 1, width=32)), st.sampled_from(TYPES))
def test_analyze_index_on_arbitrary_float32(x, t):
    import lars_image_processing_amd as lars
    got, want = lars.analyze_index(x, t), orc.stats_app(x, t)
    for key, val in want.items():
        if key.startswith("Mean"):
            # + the float32 subnormal range: hypothesis feeds denormals, whose float32 mean has no relative accuracy
            assert abs(got[key] - val) <= 1e-6 * max(abs(val), float(np.mean(np.abs(x)))) + 1e-36
        else:
            assert got[key] == val, key
    np.testing.assert_array_equal(lars.index_histogram(x), orc.hist50(x))


rgn_tiles = st.one_of(
    hnp.arrays(np.uint8, st.tuples(st.integers(1, 3), st.sampled_from([2, 4, 6, 8, 10]), st.sampled_from([2, 4, 6, 14]), st.just(3))),
    # few distinct values: medians on bin / bucket boundaries, exactly 0, exactly +-1, ranks that straddle two values
    hnp.arrays(np.uint8, st.tuples(st.integers(1, 3), st.sampled_from([2, 4, 8]), st.sampled_from([2, 6, 10]), st.just(3)),
               elements=st.sampled_from([0, 1, 2, 3, 127, 128, 254, 255])),
    hnp.arrays(np.uint8, st.tuples(st.just(1), st.integers(1, 9), st.integers(1, 9), st.just(3))),      # odd pixel counts (one tile)
)


@settings(max_examples=80 * DEEP, deadline=None)
@given(rgn_tiles, st.booleans())
def test_recompute_and_select_medians_equal_numpy(tiles, white_balance):
    """Statistics + exact medians without planes (per tile and over the batch) against np.median of the oracle's planes."""
    import lars_image_processing_amd as lars
    b = lars.TileBatch.from_host(tiles)
    rec, med = b.process(medians=True, white_balance=white_balance)
    glob = b.global_medians(white_balance=white_balance)
    planes = {t: [] for t in TYPES}
    for i, img in enumerate(tiles):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            src = orc.wb_app(img) if white_balance else img
        for k, t in enumerate(TYPES):
            plane = orc.index_app(src, t)
            planes[t].append(plane.ravel())
            assert med[i, k] == float(np.median(plane)), (i, t)
            assert float(rec[i, k]["min"]) == float(plane.min()) and float(rec[i, k]["max"]) == float(plane.max())
    for t in TYPES:
        assert glob[t] == float(np.median(np.concatenate(planes[t]))), t
    b.free()
