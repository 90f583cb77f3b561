"""The reference's function signatures, served by the HIP library.

Same names, positional arguments, return types and error behaviour as
lars-uav/lars-image-processing, so its callers (Streamlit front-end, MongoDB
I/O, batch script) keep working unchanged:

==============================  ============================================
this module                     reference
==============================  ============================================
``fix_white_balance(arr)``      process-images.py:424  (ndarray -> uint8 ndarray)
``fix_white_balance(pil)``      backend-process.py:17  (PIL.Image -> PIL.Image)
``fix_white_balance_rgnir``     process-rgn.py:4       (path in, array / file out)
``calculate_index(arr, t)``     process-images.py:449
``calculate_index(r, g, n, t)`` backend-process.py:28
``calculate_ndvi(path, ...)``   process-ndvi.py:5      (float64)
``analyze_index(idx, t)``       process-images.py:492
``analyze_ndvi_statistics``     process-ndvi.py:50
==============================  ============================================

``correct_white_balance`` and ``analyze_index_statistics`` are aliases (the
spellings BASELINE.json uses).  Inputs are never modified; outputs are fresh
host ndarrays owned by the caller.  All arithmetic runs on the GPU through
``liblars_hip.so``; there is no NumPy fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _ffi
from ._ffi import INDEX_IDS, INDEX_NAMES, Stats

__all__ = [
    "fix_white_balance", "correct_white_balance", "fix_white_balance_rgnir",
    "calculate_index", "calculate_ndvi", "analyze_index", "analyze_index_statistics",
    "analyze_ndvi_statistics", "index_histogram", "colorize_index", "process_image",
    "timeseries_row", "colormap_lut", "preprocess_large_image",
]

_CMAPS = None


def colormap_lut(name):
    """256x4 uint8 table of a matplotlib colormap (RdYlGn, RdYlBu, bwr).

    ``(cmap._lut[:256] * 255).astype(uint8)`` of matplotlib 3.10.8, shipped as
    data so that the compute path does not import matplotlib.
    """
    global _CMAPS
    if _CMAPS is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "colormaps.npz")
        with np.load(path, allow_pickle=False) as z:
            _CMAPS = {k: np.ascontiguousarray(z[k]) for k in z.files}
    return _CMAPS[name]


def _colormap_for(index_type):
    # process-images.py:690-693, backend-process.py:42
    return "RdYlBu" if index_type == "NDWI" else "RdYlGn"


def _is_pil(obj):
    return hasattr(obj, "getbands") and hasattr(obj, "convert")


def _as_image(img_array, what):
    """Validate an [H, W, C>=3] integer image and make it C-contiguous."""
    arr = np.asarray(img_array)
    if arr.ndim != 3:
        # the reference indexes [:, :, i] and fails the same way on 2-D input
        raise IndexError(f"too many indices for array: array is {arr.ndim}-dimensional, but 3 were indexed")
    if arr.shape[2] < 3:
        raise IndexError(f"index 2 is out of bounds for axis 2 with size {arr.shape[2]}")
    return np.ascontiguousarray(arr)


# ---------------------------------------------------------------------------
# down-scale in front of the path
# ---------------------------------------------------------------------------
def preprocess_large_image(img_array, max_dimension=1024):
    """process-images.py:398-422: LANCZOS down-scale to ``max_dimension`` on the long edge.

    ``None``/empty -> ``None``; an image that already fits is returned as is (the same object,
    as upstream); otherwise Pillow's ``resize((new_w, new_h), LANCZOS)``, bit for bit, on the GPU
    (uint8 images with 1, 3 or 4 channels; 4 = RGBA through premultiplied alpha, as Pillow does).
    """
    if img_array is None or np.size(img_array) == 0:
        return None
    h, w = img_array.shape[:2]
    if max(h, w) <= max_dimension:
        return img_array
    if h > w:
        new_h = max_dimension
        new_w = int(w * (max_dimension / h))
    else:
        new_w = max_dimension
        new_h = int(h * (max_dimension / w))
    arr = np.ascontiguousarray(img_array)
    if arr.dtype != np.uint8 or arr.ndim not in (2, 3) or (arr.ndim == 3 and arr.shape[2] not in (1, 3, 4)):
        raise TypeError(f"preprocess_large_image: uint8 images with 1, 3 or 4 channels (got {arr.dtype}, shape {arr.shape})")
    c = 1 if arr.ndim == 2 else arr.shape[2]
    out = np.empty((new_h, new_w) if arr.ndim == 2 else (new_h, new_w, c), dtype=np.uint8)
    _ffi.call("lars_h_resize_lanczos_u8", _ffi.ptr(arr), h, w, c, new_h, new_w, _ffi.ptr(out))
    return out


# ---------------------------------------------------------------------------
# white balance
# ---------------------------------------------------------------------------
def _wb_array(arr, variant=0, want_percentiles=False):
    code = _ffi.dtype_code(arr.dtype)
    if code is None:
        raise TypeError(f"fix_white_balance: unsupported sample type {arr.dtype}; the HIP path takes uint8 or uint16 "
                        "images (what PIL decodes RGNir files to)")
    h, w, c = arr.shape
    out = np.empty((h, w, c), dtype=np.uint8)
    pcts = np.empty((3, 2), dtype=np.float64) if want_percentiles else None
    _ffi.call("lars_h_fix_white_balance", _ffi.ptr(arr), h, w, c, code, variant, _ffi.ptr(out), _ffi.ptr(pcts))
    return (out, pcts) if want_percentiles else out


def fix_white_balance(img_array):
    """Percentile (2, 98) white balance of an RGNir image.

    ndarray in -> uint8 ndarray out, channels >= 3 zeroed, ``None``/empty ->
    ``None`` (process-images.py:424-447).  PIL.Image in -> PIL.Image out
    (backend-process.py:17-26).
    """
    if _is_pil(img_array):
        from PIL import Image
        return Image.fromarray(fix_white_balance(np.array(img_array)))
    if img_array is None or np.size(img_array) == 0:
        return None
    return _wb_array(_as_image(img_array, "fix_white_balance"))


correct_white_balance = fix_white_balance


def fix_white_balance_rgnir(image_path, save_path=None):
    """process-rgn.py:4-49: file in; array out, or file out when ``save_path`` is given."""
    from PIL import Image
    arr = _as_image(np.array(Image.open(image_path)), "fix_white_balance_rgnir")
    corrected = _wb_array(arr, variant=1)
    corrected = np.ascontiguousarray(corrected[:, :, :3])        # np.dstack of three planes (:41)
    if save_path:
        Image.fromarray(corrected).save(save_path)
        return None
    return corrected


# ---------------------------------------------------------------------------
# indices
# ---------------------------------------------------------------------------
def _index_from_planes(red, green, nir, index_type):
    shape = np.shape(nir)
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in (red, green, nir)]
    n = planes[2].size
    out = np.empty(shape, dtype=np.float32)
    _ffi.call("lars_h_calculate_index_planes", _ffi.ptr(planes[0]), _ffi.ptr(planes[1]), _ffi.ptr(planes[2]),
              n, INDEX_IDS[index_type], _ffi.ptr(out))
    return out


def calculate_index(*args):
    """``calculate_index(img_array, index_type)`` (process-images.py:449-490) or
    ``calculate_index(red, green, nir, index_type)`` (backend-process.py:28-38).

    float32 ``(a-b)/(a+b+1e-10)`` clipped to [-1, 1]; ``None``/empty -> ``None``;
    unknown type -> ``ValueError`` (2-arg) / ``UnboundLocalError`` (4-arg), as upstream.
    """
    if len(args) == 4:
        red, green, nir, index_type = args
        if index_type not in INDEX_IDS:
            raise UnboundLocalError("local variable 'index' referenced before assignment")
        return _index_from_planes(red, green, nir, index_type)
    if len(args) != 2:
        raise TypeError("calculate_index(img_array, index_type) or calculate_index(red, green, nir, index_type)")
    img_array, index_type = args
    if img_array is None or np.size(img_array) == 0:
        return None
    arr = _as_image(img_array, "calculate_index")
    if index_type not in INDEX_IDS:
        raise ValueError(f"Unknown index type: {index_type}")
    code = _ffi.dtype_code(arr.dtype)
    if code is None:
        # any other sample type: the reference casts to float32 first (:456)
        f = arr.astype(np.float32)
        return _index_from_planes(f[:, :, 0], f[:, :, 1], f[:, :, 2], index_type)
    h, w, c = arr.shape
    k = INDEX_IDS[index_type]
    out = np.empty((h, w), dtype=np.float32)
    outs = [None, None, None]
    outs[k] = out
    p3 = _ffi.ptr3(outs)
    _ffi.call("lars_h_calculate_index", _ffi.ptr(arr), h, w, c, code, 1 << k, C.byref(p3), None, 0)
    return out


def calculate_ndvi(image_path, save_path=None, visualize=True):
    """process-ndvi.py:5-48: float64 NDVI of an image file.

    The figure (when ``save_path`` / ``visualize``) is matplotlib plumbing exactly
    as upstream; the arithmetic runs on the GPU.
    """
    from PIL import Image
    arr = _as_image(np.array(Image.open(image_path)), "calculate_ndvi")
    code = _ffi.dtype_code(arr.dtype)
    if code is None:
        raise TypeError(f"calculate_ndvi: unsupported sample type {arr.dtype}")
    h, w, c = arr.shape
    ndvi = np.empty((h, w), dtype=np.float64)
    _ffi.call("lars_h_ndvi_f64", _ffi.ptr(arr), h, w, c, code, _ffi.ptr(ndvi))
    if visualize or save_path:
        import matplotlib.pyplot as plt
        plt.figure(figsize=(12, 8))
        plot = plt.imshow(ndvi, cmap="RdYlGn", vmin=-1, vmax=1)
        plt.colorbar(plot, label="NDVI")
        plt.title("NDVI Values")
        if save_path:
            plt.savefig(save_path)
            plt.close()
        if visualize:
            plt.show()
    return ndvi


# ---------------------------------------------------------------------------
# statistics
# ---------------------------------------------------------------------------
def _coverage_rule(index_type):
    # process-images.py:498-504
    return ("Water", 0.0) if index_type == "NDWI" else ("Vegetation", 0.2)


def _analyze_array(index_array, threshold, want_hist=False, want_std=False):
    """-> (Stats, median, sumsqdev) of a float32 / float64 array via the library."""
    arr = np.asarray(index_array)
    if arr.dtype == np.float32:
        flat = np.ascontiguousarray(arr).reshape(-1)
        st, med = Stats(), np.empty(2, dtype=np.float32)
        _ffi.call("lars_h_analyze_f32", _ffi.ptr(flat), flat.size, np.float32(threshold), int(want_hist),
                  C.byref(st), _ffi.ptr(med))
        median = float(np.float32(np.float32(med[0] + med[1]) / 2))   # mean of the two middles in float32
        return st, median, None
    flat = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
    st, med, ssd = Stats(), np.empty(2, dtype=np.float64), C.c_double(0.0)
    _ffi.call("lars_h_analyze_f64", _ffi.ptr(flat), flat.size, float(threshold), int(want_hist), C.byref(st),
              _ffi.ptr(med), C.byref(ssd) if want_std else None)
    return st, float((med[0] + med[1]) / 2), ssd.value


def _summary(st, median):
    """mean / median / min / max / coverage from a Stats record (NumPy NaN rules)."""
    if st.nans:
        nan = float("nan")
        return nan, nan, nan, nan, st.above / st.count * 100
    return st.sum / st.count, median, st.min, st.max, st.above / st.count * 100


def analyze_index(index_array, index_type):
    """process-images.py:492-513: dict with exactly the upstream keys.

    ``Mean`` is the exact sum of the samples divided by their count (the
    reference's float32 pairwise sum differs from it by ~1e-7 relative);
    median / min / max / coverage are exact.
    """
    if index_array is None or np.size(index_array) == 0:
        return {}
    feature_name, threshold = _coverage_rule(index_type)
    st, median, _ = _analyze_array(index_array, threshold)
    mean, median, mn, mx, cover = _summary(st, median)
    return {
        f"Mean {index_type}": float(mean),
        f"Median {index_type}": float(median),
        f"Min {index_type}": float(mn),
        f"Max {index_type}": float(mx),
        f"{feature_name} Coverage (%)": float(cover),
    }


analyze_index_statistics = analyze_index


def timeseries_row(index_array, index_type, date):
    """process-images.py:646-658: the inlined statistics row of the time-series table."""
    feature_name, threshold = _coverage_rule(index_type)
    st, median, _ = _analyze_array(index_array, threshold)
    mean, median, mn, mx, cover = _summary(st, median)
    return {"Date": date, "Mean": float(mean), "Median": float(median), "Min": float(mn), "Max": float(mx),
            f"{feature_name} Coverage (%)": float(cover)}


def analyze_ndvi_statistics(ndvi_array):
    """process-ndvi.py:50-73 (float64 flavour, with population std)."""
    arr = np.asarray(ndvi_array)
    st, median, ssd = _analyze_array(arr, 0.2, want_std=True)
    mean, median, mn, mx, cover = _summary(st, median)
    if arr.dtype == np.float32:
        # np.std of a float32 array: same two-pass definition
        st64, _, ssd = _analyze_array(arr.astype(np.float64), 0.2, want_std=True)
    std = float("nan") if st.nans else math.sqrt(ssd / st.count)
    return {
        "mean_ndvi": float(mean),
        "median_ndvi": float(median),
        "min_ndvi": float(mn),
        "max_ndvi": float(mx),
        "std_ndvi": float(std),
        "vegetation_coverage": float(cover),
    }


def index_histogram(index_array):
    """Counts of ``plt.hist(x.flatten(), bins=50, range=(-1, 1))`` (process-ndvi.py:97)."""
    st, _, _ = _analyze_array(index_array, 0.0, want_hist=True)
    return np.array(list(st.hist), dtype=np.int64)


# ---------------------------------------------------------------------------
# colormap + one-upload pipeline
# ---------------------------------------------------------------------------
def colorize_index(index_array, index_type):
    """Per-pixel RGBA8 of ``imshow(index, cmap, vmin=-1, vmax=1)`` (process-images.py:690-695)."""
    arr = np.ascontiguousarray(index_array, dtype=np.float32)
    lut = colormap_lut(_colormap_for(index_type))
    out = np.empty(arr.shape + (4,), dtype=np.uint8)
    _ffi.call("lars_h_colormap_f32", _ffi.ptr(arr.reshape(-1)), arr.size, _ffi.ptr(lut), _ffi.ptr(out))
    return out


def process_image(img_array, indices=INDEX_NAMES, white_balance=True, want_arrays=True, want_hist=False,
                  want_rgba=False):
    """White balance -> indices -> statistics of one image in ONE upload.

    What the Streamlit comparison path does with three separate calls per index
    (process-images.py:1457, :1522, :1525).  Returns a dict with
    ``corrected`` (uint8), and per index ``index`` (float32), ``stats`` (the
    ``analyze_index`` dict), ``hist`` (50 bins) and ``rgba``.
    """
    arr = _as_image(img_array, "process_image")
    code = _ffi.dtype_code(arr.dtype)
    if code is None:
        raise TypeError(f"process_image: unsupported sample type {arr.dtype}")
    for t in indices:
        if t not in INDEX_IDS:
            raise ValueError(f"Unknown index type: {t}")
    h, w, c = arr.shape
    mask = 0
    for t in indices:
        mask |= 1 << INDEX_IDS[t]
    out_wb = np.empty((h, w, c), dtype=np.uint8) if white_balance else None
    outs, rgbas, luts = [None] * 3, [None] * 3, [None] * 3
    for t in indices:
        k = INDEX_IDS[t]
        if want_arrays:
            outs[k] = np.empty((h, w), dtype=np.float32)
        if want_rgba:
            rgbas[k] = np.empty((h, w, 4), dtype=np.uint8)
            luts[k] = colormap_lut(_colormap_for(t))
    stats = (Stats * 3)()
    med = np.zeros((3, 2), dtype=np.float32)
    p_out, p_rgba, p_lut = _ffi.ptr3(outs), _ffi.ptr3(rgbas), _ffi.ptr3(luts)
    _ffi.call("lars_h_process_image", _ffi.ptr(arr), h, w, c, code, int(bool(white_balance)), mask, int(want_hist),
              _ffi.ptr(out_wb), C.byref(p_out), C.byref(stats), _ffi.ptr(med), C.byref(p_rgba), C.byref(p_lut))
    result = {"corrected": out_wb, "indices": {}}
    for t in indices:
        k = INDEX_IDS[t]
        st = stats[k]
        feature_name, _ = _coverage_rule(t)
        median = float(np.float32(np.float32(med[k, 0] + med[k, 1]) / 2))
        result["indices"][t] = {
            "index": outs[k],
            "rgba": rgbas[k],
            "hist": np.array(list(st.hist), dtype=np.int64) if want_hist else None,
            "stats": {
                f"Mean {t}": st.sum / st.count,
                f"Median {t}": median,
                f"Min {t}": st.min,
                f"Max {t}": st.max,
                f"{feature_name} Coverage (%)": st.above / st.count * 100,
            },
        }
    return result
