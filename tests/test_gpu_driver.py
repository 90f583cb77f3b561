"""Directory driver against the oracle; needs a MI355X."""
import warnings

import numpy as np
import pytest

from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu


def test_batch_process_directory(tmp_path):
    from PIL import Image
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import driver
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    rng = np.random.default_rng(21)
    imgs = {"a.png": rng.integers(0, 256, (70, 90, 3), dtype=np.uint8),
            "b.tif": rng.integers(0, 256, (33, 41, 3), dtype=np.uint8),
            "c.PNG": orc.synth_tile_u8(5, 1, 64, 64, profile="vegetation")}
    for name, arr in imgs.items():
        Image.fromarray(arr).save(src / name)
    Image.fromarray(rng.integers(0, 256, (20, 20), dtype=np.uint8)).save(src / "gray.png")   # fails like upstream: skipped
    (src / "notes.txt").write_text("not an image")
    res = driver.batch_process(src, dst, process_wb=True, process_ndvi=True, process_gndvi=True, process_ndwi=True,
                               workers=3, verbose=False)
    assert set(res) == {"a.png", "b.tif", "c.PNG", "gray.png"} and isinstance(res["gray.png"], Exception)
    for name, arr in imgs.items():
        stem = name.rsplit(".", 1)[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_wb = orc.wb_app(arr)
        np.testing.assert_array_equal(np.array(Image.open(dst / "white_balanced" / f"{stem}_wb.tif")), want_wb)
        for t in ("NDVI", "GNDVI", "NDWI"):
            want = orc.index_app(want_wb, t)
            lut = lars.colormap_lut("RdYlBu" if t == "NDWI" else "RdYlGn")
            got = np.array(Image.open(dst / t / f"{stem}_{t.lower()}.png"))
            np.testing.assert_array_equal(got, orc.colormap_closed_form(want, lut))
            ws = orc.stats_app(want, t)
            for key, val in ws.items():
                if key.startswith("Mean"):
                    assert abs(res[name][t][key] - val) <= 1e-6 * max(abs(val), float(np.mean(np.abs(want))))
                else:
                    assert res[name][t][key] == val
    # uncompressed TIFF colormap images instead of PNGs: same pixels
    driver.batch_process(src, tmp_path / "tif", process_ndvi=True, process_ndwi=False, lut_format="tiff",
                         workers=2, verbose=False)
    for name in imgs:
        stem = name.rsplit(".", 1)[0]
        np.testing.assert_array_equal(lars.read_tiff(tmp_path / "tif" / "NDVI" / f"{stem}_ndvi.tif"),
                                      np.array(Image.open(dst / "NDVI" / f"{stem}_ndvi.png")))
    # palette PNGs (one byte per pixel + the colormap as palette): the same pixels once converted
    driver.batch_process(src, tmp_path / "p8", process_ndvi=True, process_ndwi=True, lut_format="png8", workers=2, verbose=False)
    for name in imgs:
        stem = name.rsplit(".", 1)[0]
        for t in ("NDVI", "NDWI"):
            im = Image.open(tmp_path / "p8" / t / f"{stem}_{t.lower()}.png")
            assert im.mode == "P"
            np.testing.assert_array_equal(np.array(im.convert("RGBA")), np.array(Image.open(dst / t / f"{stem}_{t.lower()}.png")))
    # the reference's defaults (backend-process.py:12-15: NDWI only, no white-balanced copies), serial path
    res2 = driver.batch_process(src, tmp_path / "ser", workers=1, verbose=False)
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "ser" / "NDWI" / "a_ndwi.png")),
                                  np.array(Image.open(dst / "NDWI" / "a_ndwi.png")))
    assert not (tmp_path / "ser" / "white_balanced").exists() and "Water Coverage (%)" in res2["a.png"]["NDWI"]


def test_colormap_entries_come_from_the_device():
    """process_image(want_entries=True): the uint8 colormap entry of every pixel (one byte per pixel over PCIe) -- the palette
    of lut_format="png8" applied to it is the RGBA image of want_rgba=True; lars_d_colormap_entry_f32 on arbitrary float32 data,
    ragged lengths included, equals the closed form of matplotlib's Normalize(-1, 1) + Colormap.__call__."""
    import ctypes as C
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import _ffi
    for shape in ((257, 263, 3), (64, 65, 4), (1, 1, 3)):
        img = np.random.default_rng(shape[0]).integers(0, 256, shape, dtype=np.uint8)
        ent = lars.process_image(img, want_arrays=False, want_entries=True)
        full = lars.process_image(img, want_arrays=True, want_rgba=True)
        for t in ("NDVI", "GNDVI", "NDWI"):
            e, f = ent["indices"][t], full["indices"][t]
            assert e["entry"].dtype == np.uint8 and e["entry"].shape == shape[:2] and e["rgba"] is None and e["index"] is None
            np.testing.assert_array_equal(e["entry"], orc.colormap_entry_closed_form(f["index"]))
            np.testing.assert_array_equal(lars.colormap_lut(lars.api._colormap_for(t))[e["entry"]], f["rgba"])
            assert e["stats"] == f["stats"]
        np.testing.assert_array_equal(ent["corrected"], full["corrected"])
    with pytest.raises(ValueError):
        lars.process_image(img, want_rgba=True, want_entries=True)
    rng = np.random.default_rng(3)
    for n in (1, 3, 4, 1001, 65536 + 2):
        x = np.concatenate([rng.uniform(-1, 1, n).astype(np.float32)[: max(0, n - 6)],
                            np.array([-1.0, 1.0, 0.0, -0.0, np.nextafter(np.float32(1), np.float32(0)), 0.9921875], np.float32)])[:n]
        dx, de = _ffi.DeviceBuffer(x.nbytes + 16), _ffi.DeviceBuffer(n + 8)
        dx.upload(x)
        _ffi.call("lars_d_colormap_entry_f32", C.c_void_p(dx.ptr), n, C.c_void_p(de.ptr), None)
        _ffi.call("lars_synchronize", None)
        np.testing.assert_array_equal(de.download(np.uint8, (n,)), orc.colormap_entry_closed_form(x))
        dx.free(); de.free()


def test_sixteen_bit_tiff_at_reference_depth_and_at_full_depth(tmp_path):
    """A three-sample 16-bit TIFF: by default it is read as the reference reads it (Pillow keeps the high bytes);
    full_depth=True processes the uint16 samples (BASELINE configs[4])."""
    from PIL import Image
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import driver, tiffio
    src = tmp_path / "in"
    src.mkdir()
    rng = np.random.default_rng(8)
    img = rng.integers(0, 65536, (48, 80, 3), dtype=np.uint16)
    tiffio.write_tiff(src / "scene.tif", img, tile=(16, 32), deflate=True, predictor=True)
    for full, arr in ((False, (img >> 8).astype(np.uint8)), (True, img)):
        dst = tmp_path / f"out{int(full)}"
        res = driver.batch_process(src, dst, process_wb=True, process_ndvi=True, process_ndwi=True,
                                   workers=1, verbose=False, full_depth=full)
        assert not isinstance(res["scene.tif"], Exception), res["scene.tif"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_wb = orc.wb_app(arr)
        np.testing.assert_array_equal(np.array(Image.open(dst / "white_balanced" / "scene_wb.tif")), want_wb)
        for t in ("NDVI", "NDWI"):
            want = orc.index_app(want_wb, t)
            lut = lars.colormap_lut("RdYlBu" if t == "NDWI" else "RdYlGn")
            np.testing.assert_array_equal(np.array(Image.open(dst / t / f"scene_{t.lower()}.png")), orc.colormap_closed_form(want, lut))
            for key, val in orc.stats_app(want, t).items():
                if key.startswith("Mean"):
                    assert abs(res["scene.tif"][t][key] - val) <= 1e-6 * max(abs(val), float(np.mean(np.abs(want))))
                else:
                    assert res["scene.tif"][t][key] == val, (full, t, key)


def test_export_zip(tmp_path):
    import io
    import zipfile
    from PIL import Image
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import driver
    img = orc.synth_tile_u8(9, 2, 50, 70, profile="vegetation")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_wb = orc.wb_app(img)
    data = driver.export_zip(img, ["NDVI", "NDWI"])
    with zipfile.ZipFile(io.BytesIO(data)) as zf:
        assert sorted(zf.namelist()) == ["NDVI_visualization.png", "NDWI_visualization.png", "white_balanced.png"]
        np.testing.assert_array_equal(np.array(Image.open(io.BytesIO(zf.read("white_balanced.png")))), want_wb)
        got = np.array(Image.open(io.BytesIO(zf.read("NDWI_visualization.png"))))
        np.testing.assert_array_equal(got, orc.colormap_closed_form(orc.index_app(want_wb, "NDWI"), lars.colormap_lut("RdYlBu")))
    # with the cached white-balanced image, as the UI calls it
    data2 = driver.export_zip(None, ["NDVI"], corrected_array=want_wb)
    with zipfile.ZipFile(io.BytesIO(data2)) as zf:
        got = np.array(Image.open(io.BytesIO(zf.read("NDVI_visualization.png"))))
        np.testing.assert_array_equal(got, orc.colormap_closed_form(orc.index_app(want_wb, "NDVI"), lars.colormap_lut("RdYlGn")))


def test_ndvi_report_and_zip_keep_the_reference_file_layout(tmp_path):
    """process-ndvi.py:75-110 (BASELINE configs[0]: one 512x512 uint8 RGNir PNG, NDVI only) and process-images.py:567-617."""
    import io
    import zipfile
    from PIL import Image
    import lars_image_processing_amd as lars
    from oracle import index_oracle as orc
    img = np.random.default_rng(0).integers(0, 256, (512, 512, 3), dtype=np.uint8)
    src = tmp_path / "in.png"
    Image.fromarray(img).save(src)
    out = tmp_path / "report"
    ndvi, stats = lars.generate_ndvi_report(str(src), str(out))
    want = orc.ndvi_f64(img)
    np.testing.assert_array_equal(ndvi.view(np.uint64), want.view(np.uint64))
    ws = orc.stats_ndvi(want)
    assert list(stats) == list(ws)
    for k, v in ws.items():
        assert stats[k] == pytest.approx(v, rel=1e-9, abs=1e-12), k
    assert sorted(p.name for p in out.iterdir()) == ["ndvi_histogram.csv", "ndvi_statistics.txt", "ndvi_visualization.png"]
    rows = (out / "ndvi_histogram.csv").read_text().splitlines()
    assert rows[0] == "bin_left,bin_right,pixel_count" and len(rows) == 51
    np.testing.assert_array_equal([int(r.split(",")[2]) for r in rows[1:]], orc.hist50(want))
    picture = np.array(Image.open(out / "ndvi_visualization.png"))
    np.testing.assert_array_equal(picture, orc.colormap_closed_form(want.astype(np.float32), lars.colormap_lut("RdYlGn")))
    text = (out / "ndvi_statistics.txt").read_text().splitlines()
    assert text[0] == "NDVI Statistics:" and text[1:] == [f"{k}: {v:.4f}" for k, v in ws.items()]
    corrected = lars.fix_white_balance(img)
    blob = lars.download_processed_images({"array": img}, corrected, ["NDVI", "NDWI"])
    with zipfile.ZipFile(io.BytesIO(blob)) as zf:
        assert zf.namelist() == ["white_balanced.png", "NDVI_visualization.png", "NDWI_visualization.png"]
        np.testing.assert_array_equal(np.array(Image.open(io.BytesIO(zf.read("white_balanced.png")))), corrected)


def test_ranks_share_a_directory(tmp_path):
    """One process per GPU over one directory (the driver's __main__ under a launcher; two ranks on this box's one GPU):
    every file is processed by exactly one rank, and the union of the outputs equals a single-process run."""
    import json
    import os
    import subprocess
    import sys
    from PIL import Image
    from lars_image_processing_amd import driver
    src = tmp_path / "in"
    src.mkdir()
    rng = np.random.default_rng(33)
    names = [f"img_{i}.png" for i in range(5)]
    for n in names:
        Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)).save(src / n)
    one = driver.batch_process(src, tmp_path / "one", process_wb=True, process_ndvi=True, workers=2, verbose=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-m", "lars_image_processing_amd.driver", str(src), str(tmp_path / "two"),
                              "--wb", "--ndvi", "--workers", "2", "--quiet"], env=env, capture_output=True, text=True,
                             timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["rank"] == rank and line["world"] == 2 and not line["failed"]
        seen.append(line["files"])
    assert seen == [3, 2] and set(one) == set(names)
    for n in names:
        stem = n[:-4]
        for rel in (f"white_balanced/{stem}_wb.tif", f"NDVI/{stem}_ndvi.png", f"NDWI/{stem}_ndwi.png"):
            np.testing.assert_array_equal(np.array(Image.open(tmp_path / "two" / rel)), np.array(Image.open(tmp_path / "one" / rel)))
