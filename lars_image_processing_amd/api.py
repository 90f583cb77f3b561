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
``preprocess_large_image``      process-images.py:398  (Pillow LANCZOS down-scale)
``align_images``                process-images.py:515  (phase correlation + shift)
``calculate_index_statistics_by_timeframe``  process-images.py:619 (pandas table)
``time_series_points``          process-images.py:814-832 (the numbers ``create_time_series_plot`` draws)
``change_detection``            process-images.py:885-923, :956 (the arrays ``create_change_detection_visualization`` draws)
``download_processed_images``   process-images.py:567  (ZIP of the processed images, per-pixel colormaps)
``generate_ndvi_report``        process-ndvi.py:75     (NDVI image + histogram counts + statistics file; BASELINE configs[0])
==============================  ============================================

Figure rendering (matplotlib: ``create_time_series_plot``, ``create_change_detection_visualization``,
``create_index_visualization``, ``create_comparison_view``, the pictures inside ``calculate_ndvi`` /
``generate_ndvi_report``) is NOT part of this package (out of scope).  The
reference's own figure functions keep working on top of the functions here -- INTEGRATION.md section 1 shows the
import swap -- and this module only hands them their numbers.  Nothing under this package imports matplotlib.

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
from .hostpool import empty as _empty        # large result arrays reuse released buffers (no first-touch page faults)

__all__ = [
    "fix_white_balance", "correct_white_balance", "fix_white_balance_rgnir",
    "calculate_index", "calculate_ndvi", "analyze_index", "analyze_index_statistics",
    "analyze_ndvi_statistics", "index_histogram", "classification_mask", "colorize_index", "process_image",
    "timeseries_row", "colormap_lut", "preprocess_large_image", "align_images", "change_detection",
    "colorize_difference", "calculate_index_statistics_by_timeframe", "time_series_points",
    "calculate_ndvi_array", "generate_ndvi_report", "download_processed_images",
    "create_index_visualization", "create_comparison_view", "create_time_series_plot", "create_change_detection_visualization",
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
    h, w, c = arr.shape
    out = _empty((h, w, c), dtype=np.uint8)
    pcts = np.empty((3, 2), dtype=np.float64) if want_percentiles else None
    if code is None:
        # any other sample type (float, wider or signed integers, bool): the reference's first line is
        # `img_array.astype(np.float32)` (process-images.py:431); everything after that cast runs on the device
        if variant != 0:
            raise TypeError(f"fix_white_balance_rgnir: unsupported sample type {arr.dtype} (image files decode to uint8 / uint16)")
        if arr.dtype.kind not in "fiub":
            raise TypeError(f"fix_white_balance: samples of type {arr.dtype} cannot be cast to float32")
        as_f32 = np.ascontiguousarray(arr.astype(np.float32))
        _ffi.call("lars_h_fix_white_balance_f32", _ffi.ptr(as_f32), h, w, c, _ffi.ptr(out), _ffi.ptr(pcts))
    else:
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
    out = _empty(shape, dtype=np.float32)
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
    out = _empty((h, w), dtype=np.float32)
    outs = [None, None, None]
    outs[k] = out
    p3 = _ffi.ptr3(outs)
    _ffi.call("lars_h_calculate_index", _ffi.ptr(arr), h, w, c, code, 1 << k, C.byref(p3), None, 0)
    return out


def calculate_ndvi_array(img_array):
    """The arithmetic of process-ndvi.py:18-31 on an array: float64 ``(nir - red) / (nir + red + 1e-10)`` clipped to
    [-1, 1] of an ``[H, W, C >= 3]`` uint8 / uint16 image (``astype(float)`` is exact for both)."""
    arr = _as_image(img_array, "calculate_ndvi")
    code = _ffi.dtype_code(arr.dtype)
    if code is None:
        raise TypeError(f"calculate_ndvi: unsupported sample type {arr.dtype}")
    h, w, c = arr.shape
    ndvi = _empty((h, w), dtype=np.float64)
    _ffi.call("lars_h_ndvi_f64", _ffi.ptr(arr), h, w, c, code, _ffi.ptr(ndvi))
    return ndvi


def calculate_ndvi(image_path, save_path=None, visualize=True):
    """process-ndvi.py:5-48: float64 NDVI of an image file.

    The reference also draws a matplotlib figure (``:33-46``); figure rendering is out of scope here.  With
    ``save_path`` this function writes the per-pixel RdYlGn image of the NDVI instead (the colormap of ``:38`` applied
    pixel by pixel on the GPU, full resolution, no axes or colorbar); ``visualize`` is accepted and ignored (upstream:
    ``plt.show()``).  A caller that wants the reference's figure keeps the reference's ``calculate_ndvi`` and lets it
    call ``calculate_ndvi_array`` for lines 18-31.
    """
    from PIL import Image
    ndvi = calculate_ndvi_array(np.array(Image.open(image_path)))
    if save_path:
        Image.fromarray(colorize_index(ndvi.astype(np.float32), "NDVI"), "RGBA").save(save_path)
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


def classification_mask(index_array, index_type):
    """uint8 mask of ``index_array > threshold`` (1 = vegetation, or water for NDWI): the array the reference
    averages for its coverage figures (process-images.py:498-511), compared in float32 like NumPy does."""
    if index_array is None or np.size(index_array) == 0:
        return None
    arr = np.ascontiguousarray(index_array, dtype=np.float32)
    _, threshold = _coverage_rule(index_type)
    out = _empty(arr.shape, dtype=np.uint8)
    _ffi.call("lars_h_threshold_mask_f32", _ffi.ptr(arr.reshape(-1)), arr.size, float(threshold), _ffi.ptr(out))
    return out


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
    out = _empty(arr.shape + (4,), dtype=np.uint8)
    _ffi.call("lars_h_colormap_f32", _ffi.ptr(arr.reshape(-1)), arr.size, _ffi.ptr(lut), _ffi.ptr(out))
    return out


def process_image(img_array, indices=INDEX_NAMES, white_balance=True, want_arrays=True, want_hist=False,
                  want_rgba=False, want_entries=False):
    """White balance -> indices -> statistics of one image in ONE upload.

    What the Streamlit comparison path does with three separate calls per index
    (process-images.py:1457, :1522, :1525).  Returns a dict with
    ``corrected`` (uint8), and per index ``index`` (float32), ``stats`` (the
    ``analyze_index`` dict), ``hist`` (50 bins) and ``rgba``.  ``want_entries=True`` (instead of
    ``want_rgba``) returns ``entry``: the colormap entry of every pixel, uint8 ``[h, w]``, computed on the
    device -- ``colormap_lut(name)[entry]`` is the RGBA image, and a palette PNG needs nothing else
    (one byte per pixel crosses PCIe; backend-process.py:40-47 per pixel).
    """
    if want_rgba and want_entries:
        raise ValueError("process_image: want_rgba or want_entries, not both (they share the output slot of the C ABI)")
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
    out_wb = _empty((h, w, c), dtype=np.uint8) if white_balance else None
    outs, rgbas, luts = [None] * 3, [None] * 3, [None] * 3
    for t in indices:
        k = INDEX_IDS[t]
        if want_arrays:
            outs[k] = _empty((h, w), dtype=np.float32)
        if want_rgba:
            rgbas[k] = _empty((h, w, 4), dtype=np.uint8)
            luts[k] = colormap_lut(_colormap_for(t))
        elif want_entries:
            rgbas[k] = _empty((h, w), dtype=np.uint8)          # no table: the entry plane comes back in the RGBA slot
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
            "rgba": rgbas[k] if want_rgba else None,
            "entry": rgbas[k] if want_entries else None,
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


# ---------------------------------------------------------------------------
# registration, change detection, time series
# ---------------------------------------------------------------------------
_ALIGN_MAX_DIM = 1024                                      # process-images.py:530


def align_images(fixed_img, moving_img):
    """process-images.py:515-565: register ``moving_img`` onto ``fixed_img`` by phase correlation.

    Returns ``(aligned_img, shift)`` like upstream: ``shift`` is what
    ``skimage.registration.phase_cross_correlation`` returns (``[dy, dx]``, with a trailing 0
    for the channel axis of a colour image) and ``aligned_img`` is
    ``scipy.ndimage.shift(moving_img, shift, order=1, mode='reflect')``.  Images larger than 1024
    on a side are down-scaled first (both, independently, as upstream -- the aligned image then has
    the down-scaled size).  ``None`` in -> ``(moving_img, array([0, 0]))``.
    """
    if fixed_img is None or moving_img is None:
        return moving_img, np.array([0, 0])
    fixed_img, moving_img = np.asarray(fixed_img), np.asarray(moving_img)
    if fixed_img.shape[0] > _ALIGN_MAX_DIM or fixed_img.shape[1] > _ALIGN_MAX_DIM:
        fixed_img = preprocess_large_image(fixed_img, _ALIGN_MAX_DIM)
    if moving_img.shape[0] > _ALIGN_MAX_DIM or moving_img.shape[1] > _ALIGN_MAX_DIM:
        moving_img = preprocess_large_image(moving_img, _ALIGN_MAX_DIM)
    for name, a in (("fixed_img", fixed_img), ("moving_img", moving_img)):
        if a.dtype != np.uint8:
            raise TypeError(f"align_images: {name} must be uint8 (got {a.dtype})")
        if a.ndim == 3 and a.shape[2] != 3:
            # skimage.color.rgb2gray's own complaint
            raise ValueError(f"the input array must have size 3 along `channel_axis`, got {a.shape}")
        if a.ndim not in (2, 3):
            raise ValueError(f"align_images: {name} must be [H, W] or [H, W, 3] (got shape {a.shape})")
    if fixed_img.shape != moving_img.shape:
        raise ValueError("images must be same shape")      # phase_cross_correlation's own complaint
    fixed_c, moving_c = np.ascontiguousarray(fixed_img), np.ascontiguousarray(moving_img)
    h, w = moving_c.shape[:2]
    channels = 1 if moving_c.ndim == 2 else 3
    aligned = _empty(moving_c.shape, moving_c.dtype)
    shift = np.zeros(2, dtype=np.float64)
    _ffi.call("lars_h_align_images", _ffi.ptr(fixed_c), _ffi.ptr(moving_c), h, w, channels, _ffi.ptr(aligned), _ffi.ptr(shift))
    if moving_c.ndim == 3:
        shift = np.append(shift, 0)                         # process-images.py:552-554
    return aligned, shift


def colorize_difference(diff, vmin=-0.5, vmax=0.5, cmap="bwr"):
    """Per-pixel RGBA8 of ``imshow(diff, cmap='bwr', vmin=-0.5, vmax=0.5)`` (process-images.py:956)."""
    arr = np.ascontiguousarray(diff, dtype=np.float32)
    out = _empty(arr.shape + (4,), dtype=np.uint8)
    _ffi.call("lars_h_colormap_norm_f32", _ffi.ptr(arr.reshape(-1)), arr.size, float(vmin), float(vmax),
              _ffi.ptr(colormap_lut(cmap)), _ffi.ptr(out))
    return out


def change_detection(early, late, index_type, early_corrected=None, late_corrected=None, align=True, want_rgba=False):
    """The arithmetic of ``create_change_detection_visualization`` (process-images.py:885-923, :956).

    ``early`` / ``late`` are the raw uint8 images, ``*_corrected`` the cached white-balanced arrays
    the UI keeps (process-images.py:894-902); whichever is missing is computed.  One upload: white
    balance, registration of the late image, both indices, ``diff = late_index - early_index`` and
    (``want_rgba``) the per-pixel ``bwr`` map of the difference.  Returns a dict with
    ``early_index``, ``late_index``, ``diff``, ``aligned_late``, ``shift`` and ``diff_rgba``.
    """
    if index_type not in INDEX_IDS:
        raise ValueError(f"Unknown index type: {index_type}")
    e_src = early_corrected if early_corrected is not None else early
    l_src = late_corrected if late_corrected is not None else late
    if e_src is None or l_src is None:
        return None
    e_arr, l_arr = _as_image(e_src, "change_detection"), _as_image(l_src, "change_detection")
    wb_e, wb_l = early_corrected is None, late_corrected is None
    fused = (e_arr.dtype == np.uint8 and l_arr.dtype == np.uint8 and e_arr.shape == l_arr.shape
             and (not align or (e_arr.shape[2] == 3 and max(e_arr.shape[:2]) <= _ALIGN_MAX_DIM)))
    if not fused:
        # sizes the one-upload entry point does not cover: the same steps, one call each
        e_c = fix_white_balance(e_arr) if wb_e else e_arr
        l_c = fix_white_balance(l_arr) if wb_l else l_arr
        aligned, shift = align_images(e_c, l_c) if align else (l_c, np.zeros(3))
        e_idx, l_idx = calculate_index(e_c, index_type), calculate_index(aligned, index_type)
        diff = l_idx - e_idx                               # raises like upstream when the shapes differ (:923)
        return {"early_index": e_idx, "late_index": l_idx, "diff": diff, "aligned_late": aligned, "shift": shift,
                "diff_rgba": colorize_difference(diff) if want_rgba else None}
    h, w, c = e_arr.shape
    e_idx, l_idx, diff = (_empty((h, w), dtype=np.float32) for _ in range(3))
    aligned = _empty((h, w, c), dtype=np.uint8)
    rgba = _empty((h, w, 4), dtype=np.uint8) if want_rgba else None
    shift = np.zeros(2, dtype=np.float64)
    _ffi.call("lars_h_change_detection", _ffi.ptr(e_arr), _ffi.ptr(l_arr), h, w, c, int(wb_e), int(wb_l), int(bool(align)),
              INDEX_IDS[index_type], _ffi.ptr(e_idx), _ffi.ptr(l_idx), _ffi.ptr(diff), _ffi.ptr(rgba),
              _ffi.ptr(colormap_lut("bwr")) if want_rgba else None, -0.5, 0.5, _ffi.ptr(aligned), _ffi.ptr(shift))
    return {"early_index": e_idx, "late_index": l_idx, "diff": diff, "aligned_late": aligned,
            "shift": np.append(shift, 0), "diff_rgba": rgba}


def corrected_of(img_data):
    """The cached white-balanced array of an ``image_data`` dict, or ``None`` (process-images.py:636-641)."""
    if "corrected_array" in img_data and img_data["corrected_array"] is not None:
        return img_data["corrected_array"]
    return None


def _timeframe_records(image_data_list, index_type, want_median):
    """[(date, Stats, median)] of every image of a series: one upload per image, nothing but the
    statistics comes back."""
    if index_type not in INDEX_IDS:
        raise ValueError(f"Unknown index type: {index_type}")
    k = INDEX_IDS[index_type]
    _, threshold = _coverage_rule(index_type)
    out = []
    for img_data in image_data_list:
        date = img_data["metadata"]["upload_date"]
        corrected = corrected_of(img_data)
        src = corrected if corrected is not None else img_data["array"]
        if src is None or np.size(src) == 0:
            continue                                        # calculate_index -> None -> the row is skipped (:649)
        arr = _as_image(src, "time series")
        code = _ffi.dtype_code(arr.dtype)
        if code is None:
            raise TypeError(f"time series: unsupported sample type {arr.dtype}")
        h, w, c = arr.shape
        stats = (Stats * 3)()
        med = np.zeros((3, 2), dtype=np.float32)
        _ffi.call("lars_h_process_image", _ffi.ptr(arr), h, w, c, code, int(corrected is None), 1 << k, 0, None, None,
                  C.byref(stats), _ffi.ptr(med) if want_median else None, None, None)
        median = float(np.float32(np.float32(med[k, 0] + med[k, 1]) / 2)) if want_median else None
        out.append((date, stats[k], median))
    return out


def calculate_index_statistics_by_timeframe(image_data_list, index_type):
    """process-images.py:619-667: pandas DataFrame, one row per image, the upstream columns."""
    import pandas as pd
    feature_name, _ = _coverage_rule(index_type)
    results = []
    for date, st, median in _timeframe_records(image_data_list, index_type, want_median=True):
        results.append({"Date": date, "Mean": st.sum / st.count, "Median": median, "Min": st.min, "Max": st.max,
                        f"{feature_name} Coverage (%)": st.above / st.count * 100})
    return pd.DataFrame(results)


def time_series_points(image_data_list, index_type):
    """The numbers ``create_time_series_plot`` draws (process-images.py:814-832): dates, means, maxima, minima."""
    recs = _timeframe_records(image_data_list, index_type, want_median=False)
    return ([d for d, _, _ in recs], [st.sum / st.count for _, st, _ in recs], [st.max for _, st, _ in recs],
            [st.min for _, st, _ in recs])


def download_processed_images(image_data, corrected_array, selected_indices):
    """process-images.py:567-617: ZIP bytes with ``white_balanced.png`` and ``<INDEX>_visualization.png`` per index.
    The index pictures are full-resolution per-pixel colormap images from one GPU pass (``driver.export_zip``), not
    the reference's matplotlib figures."""
    from .driver import export_zip
    return export_zip(None if image_data is None else image_data.get("array"), selected_indices,
                      corrected_array=corrected_array)


def generate_ndvi_report(image_path, output_dir):
    """process-ndvi.py:75-110 without its two matplotlib figures: writes ``ndvi_visualization.png`` (per-pixel RdYlGn
    image of the NDVI), ``ndvi_histogram.csv`` (the 50 counts ``plt.hist(ndvi.flatten(), bins=50, range=(-1, 1))``
    would draw, with their bin edges) and ``ndvi_statistics.txt`` (same text as upstream) into ``output_dir``;
    returns ``(ndvi_array, stats)``.  NDVI, statistics and histogram counts come from the GPU."""
    os.makedirs(output_dir, exist_ok=True)
    ndvi_array = calculate_ndvi(image_path, os.path.join(output_dir, "ndvi_visualization.png"), visualize=False)
    stats = analyze_ndvi_statistics(ndvi_array)
    counts = index_histogram(ndvi_array)
    edges = np.linspace(-1.0, 1.0, 51)
    with open(os.path.join(output_dir, "ndvi_histogram.csv"), "w") as f:
        f.write("bin_left,bin_right,pixel_count\n")
        for k in range(50):
            f.write(f"{edges[k]:.2f},{edges[k + 1]:.2f},{int(counts[k])}\n")
    with open(os.path.join(output_dir, "ndvi_statistics.txt"), "w") as f:
        f.write("NDVI Statistics:\n")
        for key, value in stats.items():
            f.write(f"{key}: {value:.4f}\n")
    return ndvi_array, stats


# ---------------------------------------------------------------------------
# the reference's figure functions: out of scope here, and saying so loudly
# ---------------------------------------------------------------------------
def _figure_function(name, lines, instead):
    def stub(*_args, **_kwargs):
        raise NotImplementedError(
            f"{name} (process-images.py:{lines}) draws a matplotlib figure and stays in the reference: this package replaces the "
            f"hot-path functions it calls, not the drawing.  Keep the reference's {name} and swap its imports as INTEGRATION.md "
            f"section 1 shows; the numbers it draws come from {instead}.")
    stub.__name__ = name
    stub.__doc__ = f"Not provided: {name} (process-images.py:{lines}) is figure rendering; see INTEGRATION.md (differs from the reference)."
    return stub


create_index_visualization = _figure_function("create_index_visualization", "669-716", "calculate_index / colorize_index")
create_comparison_view = _figure_function("create_comparison_view", "718-799", "process_image / colorize_index")
create_time_series_plot = _figure_function("create_time_series_plot", "801-883", "time_series_points")
create_change_detection_visualization = _figure_function("create_change_detection_visualization", "885-989", "change_detection")
