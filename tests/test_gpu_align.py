"""Registration, change detection and time-series callers on the GPU.

Pinned: the shift application against scipy.ndimage.shift itself, the bwr change map against the
matplotlib fixture, indices / differences bit-exact against the oracle.  PARITY UNPINNED: the shift
*estimate* (scikit-image is not installed; oracle/align_oracle.py restates the published algorithm and
the tests anchor it on pairs with a known displacement).
"""
import datetime

import numpy as np
import pytest

from oracle import align_oracle as ao
from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lars():
    import lars_image_processing_amd as mod
    from lars_image_processing_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a gfx950 device"
    return mod


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def smooth_scene(h, w, seed):
    """Band-limited random RGNir scene (blurred noise): a unique correlation peak, like a field photo."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    img = rng.uniform(0, 255, (h, w, 3))
    img = ndimage.gaussian_filter(img, (3, 3, 0))
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.astype(np.uint8)


# few distinct image sizes on purpose: rocFFT builds (run-time compiles) a plan per size, seconds each on a fresh box
@pytest.mark.parametrize("shape,true", [((128, 160), (7, -11)), ((128, 160), (-20, 5)), ((97, 301), (0, 0)),
                                        ((256, 256), (100, 90)), ((97, 301), (3, -140)), ((256, 256), (-1, 1))])
def test_align_recovers_known_displacement(lars, shape, true):
    base = smooth_scene(shape[0], shape[1], shape[0] + shape[1])
    moving = np.roll(base, true, axis=(0, 1))
    aligned, shift = lars.align_images(base, moving)
    assert shift.dtype == np.float64 and shift.shape == (3,)
    np.testing.assert_array_equal(shift, [-true[0], -true[1], 0])
    want_aligned, want_shift = ao.align_images(base, moving)
    np.testing.assert_array_equal(shift, want_shift)
    np.testing.assert_array_equal(aligned, want_aligned)


def test_align_matches_oracle_on_noisy_cropped_pairs(lars):
    """Not a circular shift: two crops of one scene plus sensor noise, gray and colour."""
    scene = smooth_scene(300, 340, 5)
    rng = np.random.default_rng(6)
    for dy, dx in ((4, 9), (-13, 2), (0, -7)):
        a = scene[20:276, 20:276].copy()
        b = scene[20 + dy:276 + dy, 20 + dx:276 + dx].astype(np.int16) + rng.integers(-6, 7, (256, 256, 3))
        b = np.clip(b, 0, 255).astype(np.uint8)
        aligned, shift = lars.align_images(a, b)
        want_aligned, want_shift = ao.align_images(a, b)
        np.testing.assert_array_equal(shift, want_shift)
        assert np.max(np.abs(shift[:2] - [dy, dx])) <= 2      # whitened spectra + noise: the peak is a few pixels wide
        np.testing.assert_array_equal(aligned, want_aligned)
        ga, gs = lars.align_images(a[:, :, 0].copy(), b[:, :, 0].copy())     # 2-D input: no rgb2gray, shift of length 2
        wa, ws = ao.align_images(a[:, :, 0], b[:, :, 0])
        assert gs.shape == (2,)
        np.testing.assert_array_equal(gs, ws)
        np.testing.assert_array_equal(ga, wa)


def test_shift_application_is_scipy_ndimage_shift(lars):
    from scipy import ndimage
    from lars_image_processing_amd import _ffi
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    dimg, dout, dshift = _ffi.DeviceBuffer(img.nbytes), _ffi.DeviceBuffer(img.nbytes), _ffi.DeviceBuffer(16)
    dimg.upload(img)
    for sh in ((0, 0), (3, -5), (-36, 52), (40, -60), (100, 7), (-75, -107)):
        dshift.upload(np.array(sh, dtype=np.int64))
        _ffi.call("lars_d_shift_reflect_u8", dimg.ptr, 37, 53, 3, dshift.ptr, dout.ptr, None)
        _ffi.call("lars_synchronize", None)
        got = dout.download(np.uint8, img.shape)
        np.testing.assert_array_equal(got, ndimage.shift(img, (sh[0], sh[1], 0), order=1, mode="reflect"))


def test_align_contract(lars):
    img = smooth_scene(128, 160, 1)
    out, shift = lars.align_images(None, img)
    assert out is img and np.array_equal(shift, [0, 0])
    out, shift = lars.align_images(img, None)
    assert out is None
    with pytest.raises(ValueError, match="same shape"):
        lars.align_images(img, img[:100])
    with pytest.raises(ValueError):
        lars.align_images(np.zeros((8, 8, 4), np.uint8), np.zeros((8, 8, 4), np.uint8))
    keep = img.copy()
    lars.align_images(img, np.roll(img, 3, axis=0))
    np.testing.assert_array_equal(img, keep)


def test_align_downscales_large_inputs_like_upstream(lars):
    base = smooth_scene(1100, 700, 3)
    moving = np.roll(base, (22, -33), axis=(0, 1))
    aligned, shift = lars.align_images(base, moving)
    want_aligned, want_shift = ao.align_images(base, moving)
    assert aligned.shape == want_aligned.shape == (1024, 651, 3)
    np.testing.assert_array_equal(shift, want_shift)
    np.testing.assert_array_equal(aligned, want_aligned)


def test_bwr_change_map_matches_matplotlib(lars, golden):
    got = lars.colorize_difference(golden["colormap/diff_probe"])
    np.testing.assert_array_equal(got, golden["colormap/bwr_diff_probe_rgba"])
    # the general Normalize also reproduces the (-1, 1) fixtures of the index maps
    for name in ("RdYlGn", "RdYlBu", "bwr"):
        got = lars.colorize_difference(golden["colormap/probe"], vmin=-1, vmax=1, cmap=name)
        np.testing.assert_array_equal(got, golden[f"colormap/{name}_probe_rgba"])


@pytest.mark.parametrize("index_type", ["NDVI", "GNDVI", "NDWI"])
@pytest.mark.parametrize("cached", [False, True])
def test_change_detection_matches_oracle(lars, index_type, cached):
    early = smooth_scene(256, 256, 21)
    late = np.clip(np.roll(early, (5, -8), axis=(0, 1)).astype(np.int16)
                   + np.random.default_rng(22).integers(-20, 21, early.shape), 0, 255).astype(np.uint8)
    e_c, l_c = orc.wb_app(early), orc.wb_app(late)
    we, wl, wd, wal, wshift = ao.change_detection(e_c, l_c, index_type)
    if cached:
        res = lars.change_detection(None, None, index_type, early_corrected=e_c, late_corrected=l_c, want_rgba=True)
    else:
        res = lars.change_detection(early, late, index_type, want_rgba=True)
    np.testing.assert_array_equal(res["shift"], wshift)
    np.testing.assert_array_equal(res["aligned_late"], wal)
    np.testing.assert_array_equal(bits(res["early_index"]), bits(we))
    np.testing.assert_array_equal(bits(res["late_index"]), bits(wl))
    np.testing.assert_array_equal(bits(res["diff"]), bits(wd))
    np.testing.assert_array_equal(res["diff_rgba"], ao.colormap_norm_closed_form(wd, lars.colormap_lut("bwr"), -0.5, 0.5))
    # without registration: plain difference of the two indices
    res = lars.change_detection(None, None, index_type, early_corrected=e_c, late_corrected=l_c, align=False)
    np.testing.assert_array_equal(bits(res["diff"]), bits(orc.index_app(l_c, index_type) - orc.index_app(e_c, index_type)))


def test_change_detection_unfused_route_and_errors(lars):
    early = np.random.default_rng(1).integers(0, 256, (40, 50, 4), dtype=np.uint8)       # RGBA: no fused registration
    late = np.random.default_rng(2).integers(0, 256, (40, 50, 4), dtype=np.uint8)
    res = lars.change_detection(early, late, "NDVI", align=False)
    e_c, l_c = orc.wb_app(early), orc.wb_app(late)
    np.testing.assert_array_equal(bits(res["diff"]), bits(orc.index_app(l_c, "NDVI") - orc.index_app(e_c, "NDVI")))
    with pytest.raises(ValueError, match="Unknown index type"):
        lars.change_detection(early, late, "EVI")
    assert lars.change_detection(None, late, "NDVI") is None


def _series(n, shape=(96, 128)):
    out = []
    for k in range(n):
        img = orc.synth_tile_u8(77, k, shape[0], shape[1], profile="vegetation")
        out.append({"metadata": {"upload_date": datetime.datetime(2025, 1, 1 + k)}, "array": img, "original": None})
    return out


@pytest.mark.parametrize("index_type", ["NDVI", "NDWI"])
def test_timeframe_table_matches_reference_rows(lars, index_type):
    series = _series(4)
    series[1]["corrected_array"] = orc.wb_app(series[1]["array"])         # cached by the UI (process-images.py:1132)
    series[2]["corrected_array"] = None
    df = lars.calculate_index_statistics_by_timeframe(series, index_type)
    feature = "Water" if index_type == "NDWI" else "Vegetation"
    assert list(df.columns) == ["Date", "Mean", "Median", "Min", "Max", f"{feature} Coverage (%)"]
    assert len(df) == 4
    for k, img_data in enumerate(series):
        idx = orc.index_app(orc.wb_app(img_data["array"]), index_type)
        want = orc.stats_timeseries_row(idx, index_type, img_data["metadata"]["upload_date"])
        row = df.iloc[k]
        assert row["Date"] == want["Date"]
        assert abs(row["Mean"] - want["Mean"]) <= 1e-6 * max(abs(want["Mean"]), float(np.mean(np.abs(idx))))
        for key in ("Median", "Min", "Max", f"{feature} Coverage (%)"):
            assert row[key] == want[key], key
    dates, means, maxs, mins = lars.time_series_points(series, index_type)
    assert dates == [d["metadata"]["upload_date"] for d in series]
    assert maxs == list(df["Max"]) and mins == list(df["Min"]) and means == list(df["Mean"])


@pytest.mark.parametrize("index_type", ["NDVI", "GNDVI", "NDWI"])
def test_timeframe_table_and_points_match_the_reference_itself(lars, golden_dicts, index_type):
    """The DataFrame the reference's calculate_index_statistics_by_timeframe returned for tools/gen_golden.py's series (cached
    corrected_array that is not the white balance of its array, the key missing, the key None, RGBA, an empty image: no row) and
    the lists its create_time_series_plot drew, against api.calculate_index_statistics_by_timeframe / api.time_series_points:
    columns, dates, median, min, max and coverage exact, the mean within 1e-6 (the reference's float32 pairwise sum)."""
    from conftest import golden_series
    series = golden_series()
    want = golden_dicts["dicts"][f"timeframe/table_{index_type}"]
    df = lars.calculate_index_statistics_by_timeframe(series, index_type)
    assert list(df.columns) == want["columns"] and len(df) == len(want["rows"]) == 4
    for k, ref in enumerate(want["rows"]):
        row = df.iloc[k]
        assert row["Date"].isoformat() == ref[0]
        assert abs(row["Mean"] - ref[1]) <= 1e-6 * max(abs(ref[1]), 0.25)          # mean |x| of these images is 0.4 .. 0.6
        assert [row[c] for c in want["columns"][2:]] == ref[2:], (k, list(row), ref)
    pts = golden_dicts["dicts"][f"timeframe/points_{index_type}"]
    dates, means, maxs, mins = lars.time_series_points(series[:4], index_type)
    assert [d.isoformat() for d in dates] == pts["dates"] and maxs == pts["max"] and mins == pts["min"]
    assert all(abs(a - b) <= 1e-6 * max(abs(b), 0.25) for a, b in zip(means, pts["mean"]))
    assert list(df["Mean"]) == means                                               # one computation behind both entry points


def test_no_figure_plumbing_in_the_package(lars):
    """Figure rendering is out of scope: the package hands the reference's own figure
    functions their numbers and imports no matplotlib; the per-pixel colormap of an index stays available."""
    import os
    import re
    pkg = os.path.dirname(lars.__file__)
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(import|from)\s+matplotlib", text, re.M), name
    # the reference's figure functions exist as names that say where to go instead (no silent breakage for a caller that
    # expected a drop-in for them)
    for gone in ("create_time_series_plot", "create_change_detection_visualization", "create_index_visualization",
                 "create_comparison_view"):
        with pytest.raises(NotImplementedError, match="INTEGRATION.md"):
            getattr(lars.api, gone)(None, "NDVI")
    series = _series(2, shape=(64, 80))
    idx = lars.calculate_index(orc.wb_app(series[0]["array"]), "NDWI")
    rgba = lars.colorize_index(idx, "NDWI")
    assert rgba.shape == idx.shape + (4,) and rgba.dtype == np.uint8
    np.testing.assert_array_equal(rgba, orc.colormap_closed_form(idx, lars.colormap_lut("RdYlBu")))
