"""Parity of the HIP path (through the C ABI) against the committed reference
outputs and against the oracle on seeded inputs.  Needs a real MI355X.

Bars: uint8 outputs and index arrays BIT-EXACT; min / max / median / coverage /
histogram exact; mean within 1e-6 of max(|mean|, mean|x|): the reference's float32
pairwise sum carries an error proportional to sum|x| (~2.5e-7 relative to it,
SURVEY.md 8a-3), ours is the exact sum of the float32 samples divided by N.
"""
import warnings

import numpy as np
import pytest

from conftest import golden_case_names
from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu

CASES = golden_case_names()
U8_RGB = [c for c in CASES if c.startswith("u8_") and "rgba" not in c]
TYPES = ("NDVI", "GNDVI", "NDWI")
MEAN_RTOL = 1e-6


@pytest.fixture(scope="module")
def lars():
    import lars_image_processing_amd as mod
    from lars_image_processing_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a gfx950 device"
    return mod


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def assert_stats_close(got, want, samples=None):
    """``samples``: the index array the statistics describe (sets the scale of the mean's tolerance)."""
    assert list(got.keys()) == list(want.keys())
    scale = 1.0 if samples is None else float(np.mean(np.abs(np.asarray(samples, dtype=np.float64))))
    for key in want:
        if key.startswith("Mean") or key in ("Mean", "mean_ndvi"):
            assert abs(got[key] - want[key]) <= MEAN_RTOL * max(abs(want[key]), scale), key
        elif key in ("std_ndvi",):
            assert got[key] == pytest.approx(want[key], rel=1e-9, abs=1e-12), key
        else:
            assert got[key] == want[key], (key, got[key], want[key])


# ---------------------------------------------------------------- goldens ---
@pytest.mark.parametrize("case", CASES)
def test_white_balance_matches_reference(lars, golden, case):
    got = lars.fix_white_balance(golden[f"{case}/input"])
    assert got.dtype == np.uint8 and got.shape == golden[f"{case}/wb"].shape
    np.testing.assert_array_equal(got, golden[f"{case}/wb"])


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", ["raw", "wb"])
@pytest.mark.parametrize("t", TYPES)
def test_index_bit_exact(lars, golden, case, kind, t):
    src = golden[f"{case}/input"] if kind == "raw" else golden[f"{case}/wb"]
    got = lars.calculate_index(src, t)
    want = golden[f"{case}/index_{kind}_{t}"]
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(bits(got), bits(want))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("t", TYPES)
def test_analyze_index_matches_reference(lars, golden, golden_dicts, case, t):
    idx = golden[f"{case}/index_wb_{t}"]
    assert_stats_close(lars.analyze_index(idx, t), golden_dicts["dicts"][f"{case}/stats_wb_{t}"], idx)
    np.testing.assert_array_equal(lars.index_histogram(idx), golden[f"{case}/hist50_wb_{t}"])
    row = lars.timeseries_row(idx, t, "2025-01-01")
    assert row["Date"] == "2025-01-01" and row["Median"] == golden_dicts["dicts"][f"{case}/stats_wb_{t}"][f"Median {t}"]


@pytest.mark.parametrize("case", U8_RGB)
def test_script_variants_match_reference(lars, golden, golden_dicts, case, tmp_path):
    from PIL import Image
    img = golden[f"{case}/input"]
    # backend-process.py: PIL in, PIL out; 4-argument index
    pil = lars.fix_white_balance(Image.fromarray(img))
    np.testing.assert_array_equal(np.array(pil), golden[f"{case}/backend_wb"])
    f = golden[f"{case}/backend_wb"].astype(np.float32)
    for t in TYPES:
        got = lars.calculate_index(f[:, :, 0].copy(), f[:, :, 1].copy(), f[:, :, 2].copy(), t)
        np.testing.assert_array_equal(bits(got), bits(golden[f"{case}/backend_index_{t}"]))
    # process-rgn.py / process-ndvi.py: through files
    p = tmp_path / "in.png"
    Image.fromarray(img).save(p)
    np.testing.assert_array_equal(lars.fix_white_balance_rgnir(str(p)), golden[f"{case}/rgn_wb"])
    out = tmp_path / "out.png"
    assert lars.fix_white_balance_rgnir(str(p), str(out)) is None
    np.testing.assert_array_equal(np.array(Image.open(out)), golden[f"{case}/rgn_wb"])
    nd = lars.calculate_ndvi(str(p), save_path=None, visualize=False)
    assert nd.dtype == np.float64
    np.testing.assert_array_equal(bits(nd), bits(golden[f"{case}/ndvi_f64"]))
    assert_stats_close(lars.analyze_ndvi_statistics(nd), golden_dicts["dicts"][f"{case}/ndvi_stats"], nd)
    np.testing.assert_array_equal(lars.index_histogram(nd), golden[f"{case}/ndvi_f64_hist50"])


@pytest.mark.parametrize("name,t", [("RdYlGn", "NDVI"), ("RdYlBu", "NDWI")])
def test_colormap_matches_matplotlib(lars, golden, name, t):
    got = lars.colorize_index(golden["colormap/probe"], t)
    np.testing.assert_array_equal(got, golden[f"colormap/{name}_probe_rgba"])


@pytest.mark.parametrize("case", [c for c in CASES if "rgba" not in c])
def test_process_image_one_upload(lars, golden, golden_dicts, case):
    img = golden[f"{case}/input"]
    res = lars.process_image(img, want_hist=True, want_rgba=True)
    np.testing.assert_array_equal(res["corrected"], golden[f"{case}/wb"])
    for t in TYPES:
        r = res["indices"][t]
        np.testing.assert_array_equal(bits(r["index"]), bits(golden[f"{case}/index_wb_{t}"]))
        assert_stats_close(r["stats"], golden_dicts["dicts"][f"{case}/stats_wb_{t}"], golden[f"{case}/index_wb_{t}"])
        np.testing.assert_array_equal(r["hist"], golden[f"{case}/hist50_wb_{t}"])
        lut = lars.colormap_lut("RdYlBu" if t == "NDWI" else "RdYlGn")
        np.testing.assert_array_equal(r["rgba"], orc.colormap_closed_form(golden[f"{case}/index_wb_{t}"], lut))


# ------------------------------------------------- oracle on seeded inputs ---
@pytest.mark.parametrize("shape", [(1, 1), (1, 2), (3, 5), (7, 9), (64, 64), (255, 257), (512, 512), (1000, 1003)])
@pytest.mark.parametrize("profile", ["uniform", "vegetation"])
def test_odd_shapes_against_oracle(lars, shape, profile):
    img = orc.synth_tile_u8(99, shape[0] * 1000 + shape[1], shape[0], shape[1], profile=profile)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_wb = orc.wb_app(img)
    res = lars.process_image(img, want_hist=True)
    np.testing.assert_array_equal(res["corrected"], want_wb)
    for t in TYPES:
        want = orc.index_app(want_wb, t)
        np.testing.assert_array_equal(bits(res["indices"][t]["index"]), bits(want))
        assert_stats_close(res["indices"][t]["stats"], orc.stats_app(want, t), want)
        np.testing.assert_array_equal(res["indices"][t]["hist"], orc.hist50(want))
    # statistics only: the planes never exist on the device either (medians by recompute-and-select)
    lean = lars.process_image(img, want_arrays=False, want_hist=True)
    assert lean["corrected"] is not None
    for t in TYPES:
        assert lean["indices"][t]["index"] is None
        assert lean["indices"][t]["stats"] == res["indices"][t]["stats"], t
        np.testing.assert_array_equal(lean["indices"][t]["hist"], res["indices"][t]["hist"])
    solo = lars.process_image(img, indices=("GNDVI",), white_balance=False, want_arrays=False)
    assert solo["indices"]["GNDVI"]["stats"] == orc.stats_app(orc.index_app(img, "GNDVI"), "GNDVI") or \
        solo["indices"]["GNDVI"]["stats"]["Median GNDVI"] == float(np.median(orc.index_app(img, "GNDVI")))


def test_large_odd_shaped_image(lars):
    """35 Mpix, neither dimension a multiple of anything: many blocks per image, ragged rows, the npix % 4 tail."""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (5001, 7001, 3), dtype=np.uint8)
    img[:, :, 2] = np.clip(img[:, :, 2].astype(np.int32) // 2 + 100, 0, 255).astype(np.uint8)
    res = lars.process_image(img, want_hist=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wb = orc.wb_closed_form(img)                       # == wb_app (test_oracle.py), 10x faster at this size
    np.testing.assert_array_equal(res["corrected"], wb)
    for t in TYPES:
        want = orc.index_app(wb, t)
        np.testing.assert_array_equal(bits(res["indices"][t]["index"]), bits(want))
        assert_stats_close(res["indices"][t]["stats"], orc.stats_app(want, t), want)
        np.testing.assert_array_equal(res["indices"][t]["hist"], orc.hist50(want))


def test_rgba_and_uint16_inputs(lars):
    rng = np.random.default_rng(17)
    rgba = rng.integers(0, 256, (37, 41, 4), dtype=np.uint8)
    got = lars.fix_white_balance(rgba)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(got, orc.wb_app(rgba))
    assert (got[:, :, 3] == 0).all()                       # zeros_like + range(3), process-images.py:432-435
    u16 = rng.integers(0, 65536, (50, 60, 3), dtype=np.uint16)
    np.testing.assert_array_equal(lars.fix_white_balance(u16), orc.wb_app(u16))
    for t in TYPES:
        np.testing.assert_array_equal(bits(lars.calculate_index(u16, t)), bits(orc.index_app(u16, t)))
        np.testing.assert_array_equal(bits(lars.calculate_index(rgba, t)), bits(orc.index_app(rgba, t)))
    f32img = rng.uniform(0, 1, (20, 30, 3)).astype(np.float32)
    for t in TYPES:
        np.testing.assert_array_equal(bits(lars.calculate_index(f32img, t)), bits(orc.index_app(f32img, t)))


def _dtype_golden():
    import os
    from conftest import GOLDEN_DIR
    with np.load(os.path.join(GOLDEN_DIR, "wb_dtypes.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("case", sorted({k.split("/")[0] for k in _dtype_golden()}))
def test_white_balance_of_other_sample_types(lars, case):
    """process-images.py:431 accepts whatever astype(np.float32) accepts: float, signed / wide integer and bool images
    against the reference's own outputs (tests/golden/wb_dtypes.npz), bit for bit, percentiles included.
    (NaN samples: np.percentile returns NaN and the cast of NaN to uint8 is platform-defined -- parity unpinned,
    not exercised.)"""
    from lars_image_processing_amd import api
    g = _dtype_golden()
    img, want = g[f"{case}/input"], g[f"{case}/wb"]
    before = img.copy()
    got, pcts = api._wb_array(np.ascontiguousarray(img), want_percentiles=True)
    np.testing.assert_array_equal(got, want)
    assert got.dtype == np.uint8 and got.shape == img.shape
    ref_p = g[f"{case}/percentiles"]
    assert ((pcts == ref_p) | (np.isnan(pcts) & np.isnan(ref_p))).all(), (pcts, ref_p)
    np.testing.assert_array_equal(lars.fix_white_balance(img), want)         # the public name, same answer
    np.testing.assert_array_equal(img, before)                                # the input is never modified
    # a larger image than the goldens hold, against the oracle (itself pinned by those goldens)
    big = np.random.default_rng(7).normal(0.4, 0.2, (301, 517, 3)).astype(img.dtype if img.dtype.kind == "f" else np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(lars.fix_white_balance(big), orc.wb_app(big))


def test_classification_masks_are_bit_exact(lars, golden):
    """index > threshold as a uint8 mask (north star: bit-exact classification masks), float32 compare at 0.2f / 0."""
    rng = np.random.default_rng(4)
    probe = np.concatenate([rng.uniform(-1, 1, 100003).astype(np.float32),
                            np.array([0.2, np.nextafter(np.float32(0.2), np.float32(1)), np.nextafter(np.float32(0.2), np.float32(0)),
                                      0.0, -0.0, 1.0, -1.0, 0.20000000298], dtype=np.float32)])
    for t, thr in (("NDVI", 0.2), ("GNDVI", 0.2), ("NDWI", 0.0)):
        got = lars.classification_mask(probe, t)
        assert got.dtype == np.uint8
        np.testing.assert_array_equal(got, (probe > thr).astype(np.uint8))
        got2 = lars.classification_mask(probe[:-3].reshape(-1, 2), t)          # 2-D, length not a multiple of 4
        np.testing.assert_array_equal(got2, (probe[:-3].reshape(-1, 2) > thr).astype(np.uint8))
        assert float(got.mean() * 100) == lars.analyze_index(probe, t)[f"{'Water' if t == 'NDWI' else 'Vegetation'} Coverage (%)"]
    case = [c for c in CASES if "rgba" not in c][0]
    idx = golden[f"{case}/index_wb_NDVI"]
    np.testing.assert_array_equal(lars.classification_mask(idx, "NDVI"), (idx > 0.2).astype(np.uint8))
    assert lars.classification_mask(None, "NDVI") is None


def test_division_exhaustive_uint8_pairs(lars):
    """Every (a, b) byte pair: the kernel's quotient is the IEEE float32 quotient."""
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    img = np.stack([b, b, a], axis=-1)                     # red = green = b, nir = a
    for t in TYPES:
        np.testing.assert_array_equal(bits(lars.calculate_index(img, t)), bits(orc.index_app(img, t)))


def test_median_and_stats_on_arbitrary_arrays(lars):
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 10, 1001, 4096, 100000):
        x = rng.uniform(-1, 1, n).astype(np.float32)
        x[rng.integers(0, n, max(1, n // 10))] = np.float32(0.25)       # ties
        assert_stats_close(lars.analyze_index(x, "NDVI"), orc.stats_app(x, "NDVI"), x)
        assert_stats_close(lars.analyze_index(x, "NDWI"), orc.stats_app(x, "NDWI"), x)
        np.testing.assert_array_equal(lars.index_histogram(x), orc.hist50(x))
        x64 = x.astype(np.float64) * 0.999
        assert_stats_close(lars.analyze_ndvi_statistics(x64), orc.stats_ndvi(x64), x64)
    z = np.zeros((5, 5), np.float32)
    z[0, 0] = -0.0
    assert lars.analyze_index(z, "NDVI")["Median NDVI"] == 0.0


def test_inputs_are_not_modified_and_outputs_are_fresh(lars):
    img = orc.synth_tile_u8(1, 1, 32, 32)
    keep = img.copy()
    a = lars.fix_white_balance(img)
    b = lars.fix_white_balance(img)
    np.testing.assert_array_equal(img, keep)
    assert a is not b and a.flags["OWNDATA"] and a.flags["WRITEABLE"]
    view = img[::2, ::2, :]                                 # non-contiguous input
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(lars.fix_white_balance(view), orc.wb_app(view))


# ------------------------------------------------- special values, odd arrays ---
def _same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def test_special_values_and_odd_arrays_follow_numpy(lars):
    """NaN, +-inf, huge, denormal and zero samples, float16 / float64 / int64 / uint32 / bool bands, Fortran order, negative
    strides, 4 and 5 channels: ``calculate_index`` is bit-identical to the reference's expression (oracle) on all of them, and
    ``analyze_index`` reports what NumPy reports (NaN statistics for a NaN sample, infinite mean / extremum for an infinite
    one).  White balance of NaN samples is the documented exception (parity unpinned: ``test_other_sample_types``)."""
    rng = np.random.default_rng(1)

    def bands(dtype, special):
        a = rng.uniform(-300, 300, (19, 23, 3)).astype(dtype)
        if not special:
            return a
        with np.errstate(over="ignore"):                            # float16 takes 1e38 as inf: one more special value
            a[0, 0, :] = np.nan; a[1, 1, 0] = np.inf; a[2, 2, 2] = -np.inf; a[3, 3] = 0; a[4, 4] = 1e38
            a[5, 5] = [-1e-10, 0, 0]; a[6, 6] = [3e38, 0, 3e38]; a[7, 7] = [1e-45, 0, 1e-45]
        return a

    u8 = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    images = {"f32": bands(np.float32, True), "f64": bands(np.float64, True), "f16": bands(np.float16, True),
              "i8": bands(np.int8, False), "i64": (bands(np.float64, False) * 1e15).astype(np.int64),
              "u32": rng.integers(0, 2 ** 32, (9, 11, 3), dtype=np.uint32), "bool": u8 > 127,
              "u8x4": rng.integers(0, 256, (9, 11, 4), dtype=np.uint8), "u8x5": rng.integers(0, 256, (9, 11, 5), dtype=np.uint8),
              "fortran": np.asfortranarray(u8), "reversed": u8[::-1, ::-1], "1x1": u8[:1, :1]}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, img in images.items():
            for t in TYPES:
                want, got = orc.index_app(img, t), lars.calculate_index(img, t)
                assert got.dtype == want.dtype and got.shape == want.shape and got.tobytes() == want.tobytes(), (name, t)
        x = rng.uniform(-1, 1, (37, 53)).astype(np.float32)
        arrays = {"plain": x, "f64": x.astype(np.float64), "i16": (x * 100).astype(np.int16), "bool": x > 0, "1-D": x.reshape(-1),
                  "strided": x[::2, ::3], "zeros": np.array([[-0.0, 0.0, -0.0, 0.0]], dtype=np.float32), "one": x[:1, :1]}
        for key, (i, j, v) in {"nan": (3, 4, np.nan), "+inf": (0, 0, np.inf), "-inf": (1, 1, -np.inf)}.items():
            arrays[key] = x.copy()
            arrays[key][i, j] = v
        arrays["all nan"] = np.full((5, 7), np.nan, dtype=np.float32)
        for name, arr in arrays.items():
            for t in ("NDVI", "NDWI"):
                want, got = orc.stats_app(arr, t), lars.analyze_index(arr, t)
                assert list(got) == list(want), (name, t)
                scale = float(np.mean(np.abs(np.nan_to_num(np.asarray(arr, dtype=np.float64), posinf=0, neginf=0)))) or 1.0
                for k, v in want.items():
                    if k.startswith("Mean") and np.isfinite(v):
                        assert abs(got[k] - v) <= MEAN_RTOL * max(abs(v), scale), (name, t, k)
                    else:
                        assert _same(got[k], v), (name, t, k, got[k], v)
