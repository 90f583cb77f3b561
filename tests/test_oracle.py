"""The oracle (oracle/index_oracle.py) against what the reference itself returned.

These pin the oracle: every statement function must reproduce the reference's
outputs bit for bit (arrays) / exactly (dict floats) on the committed goldens,
and every closed form must agree with its statement function.
"""
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_case_names, golden_series
from oracle import index_oracle as orc

CASES = golden_case_names()
U8_RGB = [c for c in CASES if c.startswith("u8_") and "rgba" not in c]
TYPES = ("NDVI", "GNDVI", "NDWI")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


@pytest.mark.parametrize("case", CASES)
def test_wb_statement_matches_reference(golden, case):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = orc.wb_app(golden[f"{case}/input"])
    assert got.dtype == np.uint8
    np.testing.assert_array_equal(got, golden[f"{case}/wb"])


@pytest.mark.parametrize("case", CASES)
def test_wb_closed_form_matches_reference(golden, case):
    got = orc.wb_closed_form(golden[f"{case}/input"])
    np.testing.assert_array_equal(got, golden[f"{case}/wb"])


@pytest.mark.parametrize("case", CASES)
def test_percentile_from_hist_is_numpy_percentile(golden, case):
    img = golden[f"{case}/input"]
    nval = 256 if img.dtype == np.uint8 else 65536
    for c in range(3):
        hist = np.bincount(img[:, :, c].ravel(), minlength=nval)
        for j, q in enumerate((2, 98)):
            want = golden[f"{case}/percentiles"][c, j]
            got = orc.percentile_from_hist(hist, q)
            assert got == want, (case, c, q, got, want)


def test_percentile_from_hist_random_trials():
    rng = np.random.default_rng(123)
    for trial in range(400):
        n = int(rng.integers(1, 3000))
        hi = int(rng.integers(1, 257))
        x = rng.integers(0, hi, n).astype(np.uint8)
        hist = np.bincount(x, minlength=256)
        # q must be a sequence, as at process-images.py:437: a scalar q keeps the
        # whole computation in float32, a tuple makes it float64
        qs = (2, 98, 50, 0, 100)
        want = np.percentile(x.astype(np.float32), qs)
        assert want.dtype == np.float64
        for q, w in zip(qs, want):
            assert orc.percentile_from_hist(hist, q) == w


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", ["raw", "wb"])
@pytest.mark.parametrize("t", TYPES)
def test_index_statement_and_closed_form(golden, case, kind, t):
    src = golden[f"{case}/input"] if kind == "raw" else golden[f"{case}/wb"]
    want = golden[f"{case}/index_{kind}_{t}"]
    got = orc.index_app(src, t)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(bits(got), bits(want))
    hi, lo = orc._band_pair(t)
    cf = orc.index_closed_form(src[:, :, hi], src[:, :, lo])
    np.testing.assert_array_equal(bits(cf), bits(want))


@pytest.mark.parametrize("case", CASES)
def test_ndwi_is_negated_gndvi_plus_zero(golden, case):
    g = golden[f"{case}/index_wb_GNDVI"]
    w = golden[f"{case}/index_wb_NDWI"]
    np.testing.assert_array_equal(bits((-g) + np.float32(0)), bits(w))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", ["raw", "wb"])
@pytest.mark.parametrize("t", TYPES)
def test_stats_and_hist(golden, golden_dicts, case, kind, t):
    idx = golden[f"{case}/index_{kind}_{t}"]
    want = golden_dicts["dicts"][f"{case}/stats_{kind}_{t}"]
    got = orc.stats_app(idx, t)
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])
    np.testing.assert_array_equal(orc.hist50(idx), golden[f"{case}/hist50_{kind}_{t}"])
    np.testing.assert_array_equal(orc.hist50_closed_form(idx), golden[f"{case}/hist50_{kind}_{t}"])


@pytest.mark.parametrize("case", U8_RGB)
def test_script_variants(golden, golden_dicts, case):
    img = golden[f"{case}/input"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(orc.wb_app(img), golden[f"{case}/backend_wb"])
        np.testing.assert_array_equal(orc.wb_rgn(img), golden[f"{case}/rgn_wb"])
    f = golden[f"{case}/backend_wb"].astype(np.float32)
    for t in TYPES:
        got = orc.index_bands(f[:, :, 0].copy(), f[:, :, 1].copy(), f[:, :, 2].copy(), t)
        np.testing.assert_array_equal(bits(got), bits(golden[f"{case}/backend_index_{t}"]))
    nd = orc.ndvi_f64(img)
    np.testing.assert_array_equal(bits(nd), bits(golden[f"{case}/ndvi_f64"]))
    want = golden_dicts["dicts"][f"{case}/ndvi_stats"]
    got = orc.stats_ndvi(nd)
    assert got == want
    np.testing.assert_array_equal(orc.hist50(nd), golden[f"{case}/ndvi_f64_hist50"])
    np.testing.assert_array_equal(orc.hist50_closed_form(nd), golden[f"{case}/ndvi_f64_hist50"])


def test_contract_edge_cases(golden_dicts):
    d = golden_dicts["dicts"]
    assert d["contract/wb_none"] == "None" and orc.wb_app(None) is None
    assert d["contract/index_none"] == "None" and orc.index_app(None, "NDVI") is None
    assert d["contract/stats_none"] == "{}" and orc.stats_app(None, "NDVI") == {}
    assert d["contract/wb_empty"] == "None" and orc.wb_app(np.zeros((0, 0, 3), np.uint8)) is None
    assert d["contract/index_unknown"] == "ValueError: Unknown index type: EVI"
    with pytest.raises(ValueError, match="Unknown index type: EVI"):
        orc.index_app(np.zeros((2, 2, 3), np.uint8), "EVI")
    assert d["contract/backend_index_unknown"] == "UnboundLocalError"
    with pytest.raises(UnboundLocalError):
        orc.index_bands(np.ones(1), np.ones(1), np.ones(1), "EVI")
    assert d["contract/index_2d"] == "IndexError"


@pytest.mark.parametrize("name", ["RdYlGn", "RdYlBu", "bwr"])
def test_colormap_closed_form(golden, name):
    lut = golden[f"colormap/{name}_lut"]
    assert lut.shape == (256, 4) and lut.dtype == np.uint8
    got = orc.colormap_closed_form(golden["colormap/probe"], lut)
    np.testing.assert_array_equal(got, golden[f"colormap/{name}_probe_rgba"])


def test_merge_partials_equals_union():
    rng = np.random.default_rng(5)
    tiles = [rng.integers(0, 256, (32, 48, 3), dtype=np.uint8) for _ in range(5)]
    for t in TYPES:
        idx = [orc.index_app(x, t) for x in tiles]
        merged = orc.merge_partials(orc.tile_partials(i, t) for i in idx)
        union = np.concatenate([i.ravel() for i in idx])
        whole = orc.merge_partials([orc.tile_partials(union, t)])
        assert merged["count"] == union.size
        assert merged["min"] == whole["min"] and merged["max"] == whole["max"]
        assert merged["coverage"] == whole["coverage"]
        np.testing.assert_array_equal(merged["hist"], whole["hist"])
        assert abs(merged["mean"] - float(np.mean(union.astype(np.float64)))) < 1e-12


def test_synth_tile_is_deterministic_and_profiled():
    a = orc.synth_tile_u8(1234, 3, 16, 20)
    b = orc.synth_tile_u8(1234, 3, 16, 20)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (16, 20, 3) and a.dtype == np.uint8
    assert not np.array_equal(a, orc.synth_tile_u8(1234, 4, 16, 20))
    v = orc.synth_tile_u8(1234, 3, 16, 20, profile="vegetation")
    assert v[:, :, 0].max() <= 20 + 255 * 3 // 8 and v[:, :, 2].min() >= 60


# ---------------------------------------------------------------- preprocess_large_image ---
def _resize_cases():
    import os
    from conftest import GOLDEN_DIR
    with np.load(os.path.join(GOLDEN_DIR, "resize_outputs.npz"), allow_pickle=False) as z:
        return sorted({k.split("/")[0] for k in z.files})


@pytest.fixture(scope="session")
def resize_golden():
    import os
    from conftest import GOLDEN_DIR
    with np.load(os.path.join(GOLDEN_DIR, "resize_outputs.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("case", _resize_cases())
def test_lanczos_restatement_matches_reference(resize_golden, golden_dicts, case):
    from oracle import resize_oracle as ro
    img = resize_golden[f"{case}/input"]
    md = int(resize_golden[f"{case}/max_dimension"])
    got = ro.preprocess_large_image(img, md)
    want = resize_golden[f"{case}/output"]
    assert got.dtype == np.uint8 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    assert (got is img) == bool(resize_golden[f"{case}/same_object"])
    assert golden_dicts["dicts"]["contract/resize_none"] == "None" and ro.preprocess_large_image(None) is None


# ---- white balance of samples that are neither uint8 nor uint16 (tests/golden/wb_dtypes.npz, reference-produced) ----
def _dtype_golden():
    with np.load(os.path.join(GOLDEN_DIR, "wb_dtypes.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


DTYPE_CASES = sorted({k.split("/")[0] for k in _dtype_golden()})


@pytest.mark.parametrize("case", DTYPE_CASES)
def test_wb_of_other_sample_types(case):
    g = _dtype_golden()
    img, want = g[f"{case}/input"], g[f"{case}/wb"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(orc.wb_app(img), want)                   # the statement
        got, pcts = orc.wb_float_closed_form(img)                             # the kernels' spelling
    np.testing.assert_array_equal(got, want)
    ref_p = g[f"{case}/percentiles"]
    same = (pcts == ref_p) | (np.isnan(pcts) & np.isnan(ref_p))
    assert same.all(), (pcts, ref_p)
    assert want.dtype == np.uint8 and want.shape == img.shape
    if img.shape[2] > 3:
        assert (want[:, :, 3:] == 0).all()


@pytest.mark.parametrize("t", ["NDVI", "GNDVI", "NDWI"])
def test_timeframe_table_and_plotted_points(golden_dicts, t):
    """process-images.py:619-667 and :814-832 on a series of image_data dicts -- no 'corrected_array' key, the key holding None, a
    cached array that is NOT the white balance of its 'array', an RGBA image, an empty image (no row) -- against the DataFrame the
    reference itself returned and the lists its create_time_series_plot handed to errorbar (tools/gen_golden.py)."""
    import warnings
    series = golden_series()
    want = golden_dicts["dicts"][f"timeframe/table_{t}"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rows = orc.timeframe_rows(series, t)
        dates, means, maxs, mins = orc.timeseries_points(series[:4], t)
    assert [list(r.keys()) for r in rows] == [want["columns"]] * len(want["rows"]) and len(rows) == 4      # the empty image leaves no row
    for got, ref in zip(rows, want["rows"]):
        assert [got["Date"].isoformat()] + [got[c] for c in want["columns"][1:]] == ref                      # exact: the same NumPy calls
    pts = golden_dicts["dicts"][f"timeframe/points_{t}"]
    assert [d.isoformat() for d in dates] == pts["dates"] and means == pts["mean"] and maxs == pts["max"] and mins == pts["min"]
    assert golden_dicts["dicts"]["contract/timeseries_plot_short"] == "None"
