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
