#!/usr/bin/env python3
"""Generate tests/golden/*.npz + *.json by RUNNING the reference in this container.

Usage (build container only; /root/reference does not exist on the GPU box):

    MPLBACKEND=Agg python tools/gen_golden.py --reference /root/reference

The reference's four scripts are compiled from their source text (hyphenated
file names, so no normal import; bytecode caches are not used) with stand-in
modules for the packages that are absent here and unrelated to the pixel
maths (streamlit, dotenv, skimage).  Only *data* is written: the inputs this
script synthesises and the arrays / dicts the reference functions return for
them, plus the library versions they were produced under.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import types
import warnings
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _load(ref_dir, filename, modname):
    path = os.path.join(ref_dir, filename)
    with open(path, "r", encoding="utf-8") as fh:
        src = fh.read()
    mod = types.ModuleType(modname)
    mod.__file__ = path
    exec(compile(src, path, "exec"), mod.__dict__)   # __name__ != "__main__": mains stay off
    return mod


def _stub_absent_modules():
    st = mock.MagicMock(name="streamlit")
    st.cache_resource = lambda f=None, **kw: (f if f is not None else (lambda g: g))
    st.cache_data = st.cache_resource
    for name, obj in {
        "streamlit": st,
        "dotenv": mock.MagicMock(name="dotenv"),
        "skimage": mock.MagicMock(name="skimage"),
        "skimage.registration": mock.MagicMock(name="skimage.registration"),
        "skimage.color": mock.MagicMock(name="skimage.color"),
    }.items():
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = obj


def make_cases():
    """name -> input array.  Deterministic; small enough to commit."""
    rng = lambda s: np.random.default_rng(s)
    cases = {}
    cases["u8_64x64"] = rng(0).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    cases["u8_97x113"] = rng(1).integers(0, 256, (97, 113, 3), dtype=np.uint8)
    cases["u8_1x1"] = rng(2).integers(0, 256, (1, 1, 3), dtype=np.uint8)
    cases["u8_rgba_5x7"] = rng(3).integers(0, 256, (5, 7, 4), dtype=np.uint8)
    cases["u8_zero_16x16"] = np.zeros((16, 16, 3), dtype=np.uint8)
    const = rng(4).integers(0, 256, (24, 24, 3), dtype=np.uint8)
    const[:, :, 1] = 77
    cases["u8_const_green_24x24"] = const
    nored = rng(5).integers(0, 256, (24, 24, 3), dtype=np.uint8)
    nored[:, :, 0] = 0
    cases["u8_red_zero_24x24"] = nored
    cases["u8_binary_32x32"] = (rng(6).integers(0, 2, (32, 32, 3)) * 255).astype(np.uint8)
    # vegetation-like: NIR high, red low, narrow ranges -> fractional percentiles
    g = rng(7)
    veg = np.stack([
        np.clip(g.normal(70, 25, (48, 80)), 0, 255),
        np.clip(g.normal(90, 25, (48, 80)), 0, 255),
        np.clip(g.normal(150, 40, (48, 80)), 0, 255),
    ], axis=-1).astype(np.uint8)
    cases["u8_veg_48x80"] = veg
    # few distinct values -> interpolated (non-integer) percentiles
    cases["u8_sparse_31x29"] = (rng(8).integers(0, 5, (31, 29, 3)) * 50 + 3).astype(np.uint8)
    cases["u16_64x64"] = rng(9).integers(0, 65536, (64, 64, 3), dtype=np.uint16)
    cases["u16_12bit_40x40"] = rng(10).integers(0, 4096, (40, 40, 3), dtype=np.uint16)
    return cases


def make_dtype_cases():
    """Images whose samples are neither uint8 nor uint16 (the reference casts anything with astype(np.float32),
    process-images.py:431).  name -> input array."""
    rng = lambda s: np.random.default_rng(s)
    cases = {}
    cases["f32_normal_40x52"] = rng(40).normal(100.0, 40.0, (40, 52, 3)).astype(np.float32)
    cases["f32_unit_33x47"] = rng(41).random((33, 47, 3), dtype=np.float32)
    cases["i32_wide_40x40"] = rng(42).integers(-5000, 70000, (40, 40, 3)).astype(np.int32)
    cases["f64_48x32"] = rng(43).normal(0.3, 0.1, (48, 32, 3))
    cases["i64_big_24x24"] = rng(44).integers(0, 1 << 40, (24, 24, 3)).astype(np.int64)
    cases["f16_20x30"] = rng(45).normal(50.0, 20.0, (20, 30, 3)).astype(np.float16)
    cases["f32_rgba_9x11"] = rng(46).normal(10.0, 3.0, (9, 11, 4)).astype(np.float32)
    cases["i16_neg_31x17"] = rng(47).integers(-300, 300, (31, 17, 3)).astype(np.int16)
    cases["bool_16x16"] = rng(48).integers(0, 2, (16, 16, 3)).astype(bool)
    flat = rng(49).normal(5.0, 1.0, (30, 30, 3)).astype(np.float32)
    flat[:, :, 0] = 2.5                               # constant channel: 0/0 -> NaN -> 0
    flat[:, :, 1] = np.where(rng(50).random((30, 30)) < 0.99, 7.0, flat[:, :, 1]).astype(np.float32)   # p2 == p98, not constant: x/0 -> +-inf
    cases["f32_degenerate_30x30"] = flat
    cases["f32_1x1"] = np.array([[[1.5, -2.0, 3.25]]], dtype=np.float32)
    few = (rng(51).integers(0, 4, (29, 31, 3)) * 0.3 + 0.1).astype(np.float32)      # ties: the two neighbours often coincide
    cases["f32_ties_29x31"] = few
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=GOLDEN)
    args = ap.parse_args()
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    _stub_absent_modules()

    import matplotlib
    matplotlib.use("Agg")
    import PIL
    from PIL import Image

    app = _load(args.reference, "process-images.py", "ref_process_images")
    backend = _load(args.reference, "backend-process.py", "ref_backend_process")
    ndvi_mod = _load(args.reference, "process-ndvi.py", "ref_process_ndvi")
    rgn_mod = _load(args.reference, "process-rgn.py", "ref_process_rgn")

    arrays = {}
    dicts = {}
    cases = make_cases()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # constant channels divide by zero upstream
        for name, img in cases.items():
            arrays[f"{name}/input"] = img
            wb = app.fix_white_balance(img)
            arrays[f"{name}/wb"] = wb
            pcts = np.array([np.percentile(img[:, :, c].astype(np.float32), (2, 98)) for c in range(3)])
            arrays[f"{name}/percentiles"] = pcts          # numpy's, for diagnostics
            for kind, src in (("raw", img), ("wb", wb)):
                for t in ("NDVI", "GNDVI", "NDWI"):
                    idx = app.calculate_index(src, t)
                    assert idx.dtype == np.float32
                    arrays[f"{name}/index_{kind}_{t}"] = idx
                    dicts[f"{name}/stats_{kind}_{t}"] = app.analyze_index(idx, t)
                    arrays[f"{name}/hist50_{kind}_{t}"] = np.histogram(idx, bins=50, range=(-1, 1))[0]
            # backend-process.py: PIL in / PIL out white balance, 4-arg index
            if img.dtype == np.uint8 and img.shape[2] == 3:
                pil_wb = backend.fix_white_balance(Image.fromarray(img))
                arrays[f"{name}/backend_wb"] = np.array(pil_wb)
                f = np.array(pil_wb, dtype=np.float32)
                r, gch, n = f[:, :, 0].copy(), f[:, :, 1].copy(), f[:, :, 2].copy()
                for t in ("NDVI", "GNDVI", "NDWI"):
                    arrays[f"{name}/backend_index_{t}"] = backend.calculate_index(r, gch, n, t)
                # process-rgn.py / process-ndvi.py go through files
                with tempfile.TemporaryDirectory() as tmp:
                    p = os.path.join(tmp, "in.png")
                    Image.fromarray(img).save(p)
                    arrays[f"{name}/rgn_wb"] = rgn_mod.fix_white_balance_rgnir(p)
                    nd = ndvi_mod.calculate_ndvi(p, save_path=None, visualize=False)
                    assert nd.dtype == np.float64
                    arrays[f"{name}/ndvi_f64"] = nd
                    dicts[f"{name}/ndvi_stats"] = ndvi_mod.analyze_ndvi_statistics(nd)
                    arrays[f"{name}/ndvi_f64_hist50"] = np.histogram(nd.flatten(), bins=50, range=(-1, 1))[0]

        # contract edge cases
        dicts["contract/wb_none"] = repr(app.fix_white_balance(None))
        dicts["contract/index_none"] = repr(app.calculate_index(None, "NDVI"))
        dicts["contract/stats_none"] = repr(app.analyze_index(None, "NDVI"))
        dicts["contract/wb_empty"] = repr(app.fix_white_balance(np.zeros((0, 0, 3), np.uint8)))
        try:
            app.calculate_index(cases["u8_64x64"], "EVI")
        except ValueError as e:
            dicts["contract/index_unknown"] = f"ValueError: {e}"
        try:
            backend.calculate_index(np.ones((2, 2), np.float32), np.ones((2, 2), np.float32),
                                    np.ones((2, 2), np.float32), "EVI")
        except Exception as e:
            dicts["contract/backend_index_unknown"] = type(e).__name__
        try:
            app.calculate_index(np.zeros((4, 4), np.uint8), "NDVI")
        except Exception as e:
            dicts["contract/index_2d"] = type(e).__name__

    # white balance of other sample types (separate file: reference_outputs.npz stays byte-identical)
    dtypes = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, img in make_dtype_cases().items():
            dtypes[f"{name}/input"] = img
            dtypes[f"{name}/wb"] = app.fix_white_balance(img)
            dtypes[f"{name}/percentiles"] = np.array([np.percentile(img[:, :, c].astype(np.float32), (2, 98)) for c in range(3)])

    # colormaps: the per-pixel mapping imshow(cmap, vmin=-1, vmax=1) applies
    from matplotlib.colors import Normalize
    probe = np.concatenate([
        np.random.default_rng(11).uniform(-1, 1, 4096).astype(np.float32),
        np.linspace(-1, 1, 513, dtype=np.float32),
        np.array([-1.0, -0.0, 0.0, 1.0, np.nextafter(np.float32(1), np.float32(0)),
                  np.nextafter(np.float32(-1), np.float32(0))], dtype=np.float32),
    ])
    arrays["colormap/probe"] = probe
    for cm_name in ("RdYlGn", "RdYlBu", "bwr"):
        cm = matplotlib.colormaps[cm_name]
        arrays[f"colormap/{cm_name}_lut"] = cm(np.arange(256), bytes=True)       # [256,4] uint8
        arrays[f"colormap/{cm_name}_probe_rgba"] = cm(Normalize(-1, 1)(probe), bytes=True)
    # change map: imshow(diff, cmap='bwr', vmin=-0.5, vmax=0.5) (process-images.py:956); diff = late - early in [-2, 2]
    dprobe = np.concatenate([
        np.random.default_rng(12).uniform(-2, 2, 4096).astype(np.float32),
        np.random.default_rng(13).uniform(-0.5, 0.5, 4096).astype(np.float32),
        np.linspace(-0.5, 0.5, 1025, dtype=np.float32),
        np.array([-2.0, -0.5, -0.0, 0.0, 0.5, 2.0, np.nextafter(np.float32(0.5), np.float32(0)),
                  np.nextafter(np.float32(0.5), np.float32(1)), np.nextafter(np.float32(-0.5), np.float32(0)),
                  np.nextafter(np.float32(-0.5), np.float32(-1))], dtype=np.float32),
    ])
    arrays["colormap/diff_probe"] = dprobe
    arrays["colormap/bwr_diff_probe_rgba"] = matplotlib.colormaps["bwr"](Normalize(-0.5, 0.5)(dprobe), bytes=True)

    # ---- preprocess_large_image (process-images.py:398-422): Pillow LANCZOS down-scale before the hot path
    resize = {}
    r = np.random.default_rng(31)
    resize_cases = {
        "rgb_150x110_to64": (r.integers(0, 256, (150, 110, 3), dtype=np.uint8), 64),
        "rgb_110x150_to64": (r.integers(0, 256, (110, 150, 3), dtype=np.uint8), 64),
        "rgb_97x301_to100": (r.integers(0, 256, (97, 301, 3), dtype=np.uint8), 100),
        "rgb_513x40_to50": (r.integers(0, 256, (513, 40, 3), dtype=np.uint8), 50),
        "rgb_smooth_200x160_to96": ((np.add.outer(np.arange(200), np.arange(160))[:, :, None] * np.array([1, 2, 3]) % 256).astype(np.uint8), 96),
        "rgba_120x90_to64": (r.integers(0, 256, (120, 90, 4), dtype=np.uint8), 64),
        "gray_90x140_to60": (r.integers(0, 256, (90, 140), dtype=np.uint8), 60),
        "rgb_small_40x50_to64": (r.integers(0, 256, (40, 50, 3), dtype=np.uint8), 64),      # returned unchanged
        "rgb_65x64_to64": (r.integers(0, 256, (65, 64, 3), dtype=np.uint8), 64),            # scale barely above 1
    }
    rgba_in = resize_cases["rgba_120x90_to64"][0]
    rgba_in[:10, :, 3] = 0
    rgba_in[10:30, :, 3] = 255
    for name, (img, md) in resize_cases.items():
        out = app.preprocess_large_image(img, max_dimension=md)
        resize[f"{name}/input"] = img
        resize[f"{name}/max_dimension"] = np.array(md)
        resize[f"{name}/output"] = out
        resize[f"{name}/same_object"] = np.array(out is img)
    dicts["contract/resize_none"] = repr(app.preprocess_large_image(None))


    # ---- time series (process-images.py:619-667 calculate_index_statistics_by_timeframe, :801-883 create_time_series_plot): the
    # reference's own DataFrame for a series of image_data dicts -- one without the 'corrected_array' key, one with it None, one with
    # a cached array (deliberately NOT the white balance of its 'array': only the cached branch :636-637 gives these numbers), an RGBA
    # image and an empty one (fix_white_balance -> None -> calculate_index -> None: the row is skipped, :649) -- and the three lists
    # create_time_series_plot hands to errorbar (:830-832), read from the function's own frame when it returns.
    import datetime as dt
    g = np.random.default_rng(41)
    def _ts_img(h, w, ch, centre, spread):
        return np.clip(g.normal(centre, spread, (h, w, ch)), 0, 255).astype(np.uint8)
    series = [
        {"metadata": {"upload_date": dt.datetime(2025, 1, 5, 10, 30)}, "original": None, "array": _ts_img(48, 64, 3, (70, 90, 150), (25, 25, 40))},
        {"metadata": {"upload_date": dt.datetime(2025, 2, 9, 11, 0)}, "original": None, "array": _ts_img(57, 33, 3, (120, 100, 90), (50, 40, 30)),
         "corrected_array": None},
        {"metadata": {"upload_date": dt.datetime(2025, 3, 16, 9, 15)}, "original": None, "array": _ts_img(40, 40, 3, (60, 60, 60), (10, 10, 10)),
         "corrected_array": g.integers(0, 256, (40, 40, 3), dtype=np.uint8)},
        {"metadata": {"upload_date": dt.datetime(2025, 4, 20, 14, 45)}, "original": None, "array": _ts_img(31, 45, 4, (90, 140, 60, 255), (30, 30, 20, 0))},
        {"metadata": {"upload_date": dt.datetime(2025, 5, 25, 8, 0)}, "original": None, "array": np.zeros((0, 0, 3), np.uint8)},
    ]
    timeframe = {}
    for i, d in enumerate(series):
        timeframe[f"img{i}/array"] = d["array"]
        if d.get("corrected_array") is not None:
            timeframe[f"img{i}/corrected_array"] = d["corrected_array"]
    dicts["timeframe/dates"] = [d["metadata"]["upload_date"].isoformat() for d in series]
    dicts["timeframe/has_corrected_key"] = ["corrected_array" in d for d in series]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in ("NDVI", "GNDVI", "NDWI"):
            df = app.calculate_index_statistics_by_timeframe(series, t)
            dicts[f"timeframe/table_{t}"] = {
                "columns": [str(c) for c in df.columns],
                "rows": [[v.isoformat() if hasattr(v, "isoformat") else float(v) for v in row] for row in df.itertuples(index=False, name=None)],
            }
            caught = {}
            def _hook(frame, event, arg, caught=caught):
                if event == "return" and frame.f_code.co_name == "create_time_series_plot":
                    for key in ("dates", "mean_values", "max_values", "min_values"):
                        caught[key] = list(frame.f_locals[key])
            sys.setprofile(_hook)
            try:
                picture = app.create_time_series_plot(series[:4], t)        # with the empty image its lists would differ in length (:814 / :829)
            finally:
                sys.setprofile(None)
            assert picture is not None and len(caught["dates"]) == 4
            dicts[f"timeframe/points_{t}"] = {"dates": [v.isoformat() for v in caught["dates"]], "mean": caught["mean_values"],
                                              "max": caught["max_values"], "min": caught["min_values"]}
    dicts["contract/timeseries_plot_short"] = repr(app.create_time_series_plot(series[:1], "NDVI"))     # fewer than two images: None (:803)

    meta = {
        "numpy": np.__version__,
        "matplotlib": matplotlib.__version__,
        "pillow": PIL.__version__,
        "pandas": __import__("pandas").__version__,
        "python": sys.version.split()[0],
        "reference": "lars-uav/lars-image-processing snapshot 2025-03-28 (mounted at /root/reference)",
        "generator": "tools/gen_golden.py",
    }
    os.makedirs(args.out, exist_ok=True)
    np.savez_compressed(os.path.join(args.out, "reference_outputs.npz"), **arrays)
    np.savez_compressed(os.path.join(args.out, "resize_outputs.npz"), **resize)
    np.savez_compressed(os.path.join(args.out, "wb_dtypes.npz"), **dtypes)
    np.savez_compressed(os.path.join(args.out, "timeframe_inputs.npz"), **timeframe)
    with open(os.path.join(args.out, "reference_dicts.json"), "w") as fh:
        json.dump({"meta": meta, "dicts": dicts}, fh, indent=1)   # insertion order kept: key order is contract
    total = sum(a.nbytes for a in arrays.values())
    print(f"wrote {len(arrays)} arrays ({total/1e6:.2f} MB raw), {len(dicts)} dicts -> {args.out}")
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
