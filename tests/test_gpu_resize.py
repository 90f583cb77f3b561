"""preprocess_large_image through the C ABI: bit-identical to the reference (Pillow LANCZOS)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import resize_oracle as ro

pytestmark = pytest.mark.gpu


def _golden():
    with np.load(os.path.join(GOLDEN_DIR, "resize_outputs.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


G = _golden()
CASES = sorted({k.split("/")[0] for k in G})


@pytest.mark.parametrize("case", CASES)
def test_resize_matches_reference_outputs(case):
    import lars_image_processing_amd as lars
    img = G[f"{case}/input"]
    md = int(G[f"{case}/max_dimension"])
    got = lars.preprocess_large_image(img, md)
    want = G[f"{case}/output"]
    assert got.dtype == np.uint8 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    assert (got is img) == bool(G[f"{case}/same_object"])


@pytest.mark.parametrize("shape,md", [((2048, 1536, 3), 1024), ((1300, 2048, 3), 1024), ((1025, 1024, 3), 1024),
                                      ((3000, 4000, 3), 1024), ((1500, 1500, 4), 1024), ((2000, 300), 777)])
def test_resize_matches_pillow_at_ui_sizes(shape, md):
    """The sizes the Streamlit path feeds it (uploads <= 2048 px, analysis <= 1024 px), against Pillow itself."""
    from PIL import Image
    import lars_image_processing_amd as lars
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if len(shape) == 3 and shape[2] == 4:
        img[:100, :, 3] = 0
        img[100:300, :, 3] = 255
    got = lars.preprocess_large_image(img, md)
    h, w = shape[:2]
    if h > w:
        nh, nw = md, int(w * (md / h))
    else:
        nw, nh = md, int(h * (md / w))
    want = np.array(Image.fromarray(img).resize((nw, nh), Image.Resampling.LANCZOS))
    np.testing.assert_array_equal(got, want)
    if shape == (2000, 300):
        np.testing.assert_array_equal(got, ro.preprocess_large_image(img, md))


def test_resize_contract():
    import lars_image_processing_amd as lars
    assert lars.preprocess_large_image(None) is None
    assert lars.preprocess_large_image(np.zeros((0, 0, 3), np.uint8)) is None
    small = np.zeros((10, 20, 3), np.uint8)
    assert lars.preprocess_large_image(small) is small
    with pytest.raises(TypeError):
        lars.preprocess_large_image(np.zeros((2000, 10, 3), np.uint16))
