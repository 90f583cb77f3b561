"""Property tests of the oracle's closed forms (CPU; hypothesis)."""
import warnings

import numpy as np
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as hnp

from oracle import index_oracle as orc

small_images = hnp.arrays(np.uint8, st.tuples(st.integers(1, 12), st.integers(1, 12), st.just(3)))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@settings(max_examples=60, deadline=None)
@given(small_images)
def test_white_balance_closed_form_equals_statement(img):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(orc.wb_closed_form(img), orc.wb_app(img))


@settings(max_examples=60, deadline=None)
@given(small_images)
def test_index_properties(img):
    g = orc.index_app(img, "GNDVI")
    w = orc.index_app(img, "NDWI")
    v = orc.index_app(img, "NDVI")
    assert (np.abs(v) <= 1).all() and (np.abs(g) <= 1).all()
    np.testing.assert_array_equal(bits((-g) + np.float32(0)), bits(w))
    np.testing.assert_array_equal(bits(orc.index_closed_form(img[:, :, 2], img[:, :, 0])), bits(v))
    np.testing.assert_array_equal(orc.hist50_closed_form(v), orc.hist50(v))
    assert orc.hist50(v).sum() == v.size


@settings(max_examples=60, deadline=None)
@given(hnp.arrays(np.int64, 256, elements=st.integers(0, 50)), st.sampled_from([0, 2, 50, 98, 100]))
def test_percentile_from_histogram(hist, q):
    if hist.sum() == 0:
        hist[7] = 1
    samples = np.repeat(np.arange(256), hist).astype(np.float32)
    assert orc.percentile_from_hist(hist, q) == np.percentile(samples, (q,))[0]


@settings(max_examples=30, deadline=None)
@given(st.lists(small_images, min_size=2, max_size=5), st.sampled_from(["NDVI", "GNDVI", "NDWI"]))
def test_merge_of_partials_is_statistics_of_the_union(tiles, t):
    idx = [orc.index_app(x, t) for x in tiles]
    merged = orc.merge_partials(orc.tile_partials(i, t) for i in idx)
    union = np.concatenate([i.ravel() for i in idx])
    assert merged["count"] == union.size and merged["min"] == float(union.min()) and merged["max"] == float(union.max())
    _, thr = orc.coverage_rule(t)
    assert merged["coverage"] == np.count_nonzero(union > np.float32(thr)) / union.size * 100.0
    np.testing.assert_array_equal(merged["hist"], orc.hist50(union))


def test_histogram_bin_from_the_quotient_position():
    """The statistics kernels bin an index value of a uint8 tile as floor(fma(x, 25, 25.5001) + 2^23) - 1
    instead of searching numpy's edges (csrc/fused_v2.hip): exhaustive over the 65536 byte pairs, both signs."""
    edges = orc.hist50_edges(np.float32)
    for i in range(51):                                    # an exact hit compares >= its float32 edge
        assert np.float32(i - 25) / np.float32(25) >= edges[i]
    a, b = np.meshgrid(np.arange(256, dtype=np.float32), np.arange(256, dtype=np.float32), indexing="ij")
    q = orc.index_closed_form(a.ravel(), b.ravel())
    c = np.float32(25.5001)
    for sign in (1.0, -1.0):
        x = q if sign > 0 else (np.float32(0) - q).astype(np.float32)
        want = np.searchsorted(edges, x, side="right") - 1
        want[x == edges[-1]] = 49
        assert np.array_equal(np.bincount(want, minlength=50), np.histogram(x, bins=50, range=(-1, 1))[0])
        t = (q.astype(np.float64) * (25.0 * sign) + float(c)).astype(np.float32)          # one rounding, like fma
        u = (t + np.float32(8388608.0)).astype(np.float32)
        got = np.minimum((u.view(np.uint32) & 0x7FFFFF).astype(np.int64) - 1, 49)
        np.testing.assert_array_equal(got, want)


def _quotients(a, b):
    s = a + b
    return np.where(s == 0, np.float32(0), (a - b) / np.where(s == 0, np.float32(1), s)).astype(np.float32)


def test_select_positions_separate_every_quotient_of_bytes():
    """The claim the two-level select rests on: every distinct (a - b) / (a + b) of bytes has its own (bucket, slot), the
    position is monotone in the value, and batch.select_value finds the value back from the position."""
    from lars_image_processing_amd import batch
    a, b = np.meshgrid(np.arange(256, dtype=np.float32), np.arange(256, dtype=np.float32), indexing="ij")
    vals = np.unique(_quotients(a.ravel(), b.ravel()))                  # sorted, ~40 k values
    bucket, slot = batch.select_position(vals)
    assert bucket.min() == 0 and bucket.max() == batch.SELECT_BINS - 1 and slot.max() < batch.SELECT_SLOTS
    code = bucket * batch.SELECT_SLOTS + slot
    assert (np.diff(code) > 0).all()                                    # strictly increasing: distinct and monotone
    gap = np.diff((vals.astype(np.float64) * 1023.5) * 4096)            # in units of the fraction
    assert gap.min() > 15.5
    rng = np.random.default_rng(0)
    probe = np.concatenate([np.arange(0, vals.size, 97), rng.integers(0, vals.size, 300), [0, vals.size - 1, int(np.searchsorted(vals, 0))]])
    for i in probe:
        got = batch.select_value(int(bucket[i]), int(slot[i]))
        assert got.tobytes() == vals[i].tobytes(), (vals[i], got)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 3000), st.integers(0, 2**32 - 1), st.sampled_from(["bytes", "few", "zeros", "dark"]))
def test_select_host_logic_equals_numpy_median(n, seed, kind):
    """batch.select_order_statistics (bucket pick, slot pick, value look-up, shared / split ranks) with a NumPy stand-in
    for the kernel pass: quotients of uniform bytes, of few distinct values, mostly zeros, and of small sums."""
    from _select_stub import select_pass_on_planes
    from lars_image_processing_amd import batch
    rng = np.random.default_rng(seed)
    pool = {"bytes": np.arange(256), "few": np.array([0, 1, 2, 127, 254, 255]), "zeros": np.array([0, 0, 0, 0, 7]),
            "dark": np.arange(4)}[kind]
    a, r, g = (rng.choice(pool, n).astype(np.float32) for _ in range(3))
    planes = [_quotients(a, r), _quotients(a, g)]
    want = {"NDVI": float(np.median(planes[0])), "GNDVI": float(np.median(planes[1])), "NDWI": float(np.median(np.float32(0) - planes[1]))}
    # predicted window (one full pass), a window predicted from the wrong sample (falls back to the two passes), two passes
    wrong = [np.full(16, -0.9, dtype=np.float32), np.full(16, 0.9, dtype=np.float32)]
    calls = []
    for pass_fn, windowed in ((select_pass_on_planes(planes), True), (select_pass_on_planes(planes, wrong), True),
                              (select_pass_on_planes(planes), False)):
        def counting(first, buckets, fn=pass_fn):
            calls.append(int(first))
            return fn(first, buckets)
        assert batch.medians_from_pairs(batch.select_order_statistics(counting, n, windowed=windowed)) == want
        calls.append(-1)
    assert calls[-3:] == [1, 0, -1]                                     # windowed=False: the two classic passes only
    assert calls[:2] == [3, 2]                                          # windowed: the sample, then the window pass
