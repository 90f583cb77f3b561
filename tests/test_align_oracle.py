"""CPU checks of oracle/align_oracle.py: what pins it (scipy, the matplotlib fixture) and, for the
scikit-image pieces that cannot be pinned here, known-displacement anchors."""
import numpy as np
import pytest

from oracle import align_oracle as ao
from oracle import index_oracle as orc


def test_shift_reflect_is_scipy_ndimage_shift():
    from scipy import ndimage
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for sh in ((0, 0, 0), (3, -5, 0), (-36, 52, 0), (40, -60, 0), (100, 7, 0), (-75, -107, 0)):
        np.testing.assert_array_equal(ao.shift_reflect(img, sh), ndimage.shift(img, sh, order=1, mode="reflect"))
    gray = rng.integers(0, 256, (20, 31), dtype=np.uint8)
    np.testing.assert_array_equal(ao.shift_reflect(gray, (-4, 9)), ndimage.shift(gray, (-4, 9), order=1, mode="reflect"))


def test_colormap_norm_closed_form_matches_matplotlib_fixtures(golden):
    got = ao.colormap_norm_closed_form(golden["colormap/diff_probe"], golden["colormap/bwr_lut"], -0.5, 0.5)
    np.testing.assert_array_equal(got, golden["colormap/bwr_diff_probe_rgba"])
    for name in ("RdYlGn", "RdYlBu", "bwr"):
        got = ao.colormap_norm_closed_form(golden["colormap/probe"], golden[f"colormap/{name}_lut"], -1, 1)
        np.testing.assert_array_equal(got, golden[f"colormap/{name}_probe_rgba"])
        np.testing.assert_array_equal(got, orc.colormap_closed_form(golden["colormap/probe"], golden[f"colormap/{name}_lut"]))


@pytest.mark.parametrize("true", [(7, -11), (0, 0), (-20, 5), (60, 70)])
def test_phase_correlation_recovers_known_displacement(true):
    from scipy import ndimage
    rng = np.random.default_rng(3)
    base = ndimage.gaussian_filter(rng.uniform(0, 255, (128, 150, 3)), (2, 2, 0)).astype(np.uint8)
    moving = np.roll(base, true, axis=(0, 1))
    aligned, shift = ao.align_images(base, moving)
    np.testing.assert_array_equal(shift, [-true[0], -true[1], 0])
    # a circular shift comes back exactly except for the rows / columns the reflection fills
    dy, dx = true
    inner = (slice(abs(dy), 128 - abs(dy)), slice(abs(dx), 150 - abs(dx)))
    np.testing.assert_array_equal(aligned[inner], base[inner])


def test_rgb2gray_weights_and_scale():
    img = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], dtype=np.uint8)
    np.testing.assert_allclose(ao.rgb2gray(img)[0], [0.2125, 0.7154, 0.0721, 1.0], rtol=1e-15)
    with pytest.raises(ValueError):
        ao.rgb2gray(np.zeros((4, 4, 4), np.uint8))


def test_align_contract():
    img = np.zeros((8, 8, 3), np.uint8)
    out, shift = ao.align_images(None, img)
    assert out is img and np.array_equal(shift, [0, 0])
    with pytest.raises(ValueError, match="same shape"):
        ao.align_images(img, img[:4])
