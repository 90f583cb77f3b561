"""Host logic of the directory driver that needs no GPU: which rank owns which files, argument checks."""
import pytest

from lars_image_processing_amd import driver


def test_files_of_rank_partitions_the_sorted_list():
    files = [f"f{i:02d}.tif" for i in range(11)]
    for world in (1, 2, 3, 8, 16):
        parts = [driver.files_of_rank(files, r, world) for r in range(world)]
        assert [f for p in parts for f in p] == files                      # contiguous blocks, in order, nothing twice
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert driver.files_of_rank([], 0, 4) == []
    with pytest.raises(ValueError):
        driver.files_of_rank(files, 2, 2)
    with pytest.raises(ValueError):
        driver.files_of_rank(files, 0, 0)


def test_lut_format_is_checked_before_any_work(tmp_path):
    with pytest.raises(ValueError, match="lut_format"):
        driver.process_image(tmp_path / "missing.png", tmp_path, lut_format="jpeg")


def test_an_empty_directory_is_an_empty_result(tmp_path):
    (tmp_path / "in").mkdir()
    (tmp_path / "in" / "notes.txt").write_text("not an image")
    assert driver.batch_process(tmp_path / "in", tmp_path / "out", verbose=False) == {}
    assert driver.main([str(tmp_path / "in"), str(tmp_path / "out"), "--quiet"]) == 0


def test_palette_png_decodes_to_the_rgba_pixels():
    """lut_format="png8": entry plane + colormap palette == the per-pixel RGBA image (oracle's closed form of matplotlib).  The
    entry plane itself comes from the device in the product (lars_h_process_image; tests/test_gpu_driver.py) -- here the oracle's."""
    import io
    import numpy as np
    from PIL import Image
    from lars_image_processing_amd import api
    from oracle import index_oracle as orc
    rng = np.random.default_rng(0)
    x = np.clip(rng.normal(0, 0.6, (50, 70)), -1, 1).astype(np.float32)
    x[0, :4] = [1.0, -1.0, 0.0, np.float32(0.9999999)]
    for name in ("RdYlGn", "RdYlBu"):
        lut = api.colormap_lut(name)
        im = Image.fromarray(orc.colormap_entry_closed_form(x), "P")
        im.putpalette(lut.tobytes(), rawmode="RGBA")
        buf = io.BytesIO()
        im.save(buf, format="PNG", compress_level=1)
        got = np.array(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA"))
        np.testing.assert_array_equal(got, orc.colormap_closed_form(x, lut))
