"""NumPy restatement of the reference's per-pixel multispectral index path.

TEST INFRASTRUCTURE ONLY -- the shipped package never imports this module.
It exists to check the HIP path (tests/, __graft_entry__.smoke()) and to be
timed as the CPU baseline (bench.py, ``cpu_baseline.kind == "port"``).

Parity status: PINNED.  Every function below is checked against outputs of the
reference itself (tests/golden/*.npz, produced by tools/gen_golden.py which
imports /root/reference in the build container; NumPy 2.2.6, matplotlib 3.10.8).

Each function cites the reference lines it restates (paths are relative to the
upstream repository lars-uav/lars-image-processing).

Two layers live here:

* "statement" functions (``wb_app``, ``index_app``, ``stats_app`` ...) use the
  same NumPy primitives as the reference, so they are the reference's numerics
  by construction (type promotion, percentile interpolation, pairwise sums).
* "closed form" functions (``percentile_from_hist``, ``wb_lut_from_percentiles``,
  ``index_closed_form``, ``colormap_closed_form``, ``hist50_closed_form``,
  ``tile_partials`` / ``merge_partials``) spell out the arithmetic the way the
  device kernels do it.  tests/test_oracle.py proves both layers agree, so a
  kernel that matches the closed forms matches the reference.
"""
from __future__ import annotations

import numpy as np

INDEX_TYPES = ("NDVI", "GNDVI", "NDWI")
HIST_BINS = 50          # process-ndvi.py:97  plt.hist(bins=50, range=(-1, 1))
EPSILON = 1e-10         # process-images.py:464, backend-process.py:29, process-ndvi.py:25


# --------------------------------------------------------------------------
# a-1  white balance
# --------------------------------------------------------------------------
def wb_app(img_array):
    """process-images.py:424-447 ``fix_white_balance(img_array)``.

    Percentile (2, 98) stretch of channels 0..2 to 0..255; channels >= 3 stay 0
    (``zeros_like`` at :432 + ``range(3)`` at :435); truncating uint8 cast (:441).
    The maths at backend-process.py:17-26 is the same on ``np.array(img, float32)``.
    """
    if img_array is None or img_array.size == 0:          # :427-428
        return None
    as_f32 = img_array.astype(np.float32)                 # :431
    stretched = np.zeros_like(as_f32)                     # :432
    for band in (0, 1, 2):                                # :435
        plane = as_f32[:, :, band]
        lo, hi = np.percentile(plane, (2, 98))            # :437  (float64 scalars)
        # float32 plane (-) float64 scalar promotes to float64 under NumPy 2
        stretched[:, :, band] = np.clip((plane - lo) / (hi - lo) * 255, 0, 255)  # :438
    return stretched.astype(np.uint8)                     # :441


def wb_rgn(img_array):
    """process-rgn.py:18-44, the array part of ``fix_white_balance_rgnir``.

    float64 throughout, with an extra clip of the channel to [p2, p98] before
    normalising (:29), ``np.dstack`` of exactly three channels (:41), uint8 cast (:44).
    """
    as_f64 = img_array.astype(float)                      # :18

    def _stretch(plane):                                  # :25-33
        lo, hi = np.percentile(plane, (2, 98))
        inner = np.clip(plane, lo, hi)
        return np.clip((inner - lo) / (hi - lo) * 255, 0, 255)

    planes = [_stretch(as_f64[:, :, band]) for band in (0, 1, 2)]   # :36-38
    return np.dstack(planes).astype(np.uint8)             # :41-44


# --------------------------------------------------------------------------
# a-2  index
# --------------------------------------------------------------------------
def _band_pair(index_type):
    """(minuend band, subtrahend band) as channel numbers: R=0, G=1, NIR=2."""
    if index_type == "NDVI":      # process-images.py:466-470
        return 2, 0
    if index_type == "GNDVI":     # :472-476
        return 2, 1
    if index_type == "NDWI":      # :478-482
        return 1, 2
    raise ValueError(f"Unknown index type: {index_type}")   # :485


def index_app(img_array, index_type):
    """process-images.py:449-490 ``calculate_index(img_array, index_type)`` (float32)."""
    if img_array is None or img_array.size == 0:          # :452-453
        return None
    as_f32 = img_array.astype(np.float32)                 # :456
    hi_band, lo_band = _band_pair(index_type)
    a = as_f32[:, :, hi_band]
    b = as_f32[:, :, lo_band]
    ratio = (a - b) / (a + b + EPSILON)                   # :468-482
    return np.clip(ratio, -1, 1)                          # :490


def index_bands(red, green, nir, index_type):
    """backend-process.py:28-38 ``calculate_index(red, green, nir, index_type)``.

    No ``else`` branch upstream: an unknown type leaves ``index`` unbound and the
    ``np.clip`` line raises UnboundLocalError (a NameError subclass).
    """
    bands = {0: red, 1: green, 2: nir}
    if index_type not in INDEX_TYPES:
        raise UnboundLocalError("local variable 'index' referenced before assignment")
    hi_band, lo_band = _band_pair(index_type)
    a, b = bands[hi_band], bands[lo_band]
    return np.clip((a - b) / (a + b + EPSILON), -1, 1)    # :31-38


def ndvi_f64(img_array):
    """process-ndvi.py:18-31, the array part of ``calculate_ndvi`` (float64)."""
    as_f64 = img_array.astype(float)                      # :18
    nir = as_f64[:, :, 2]                                 # :21
    red = as_f64[:, :, 0]                                 # :22
    return np.clip((nir - red) / (nir + red + EPSILON), -1, 1)   # :28-31


# --------------------------------------------------------------------------
# a-3 / a-5  statistics
# --------------------------------------------------------------------------
def coverage_rule(index_type):
    """process-images.py:498-504 -> (feature name, threshold)."""
    return ("Water", 0.0) if index_type == "NDWI" else ("Vegetation", 0.2)


def stats_app(index_array, index_type):
    """process-images.py:492-513 ``analyze_index(index_array, index_type)``."""
    if index_array is None or index_array.size == 0:      # :495-496
        return {}
    feature, threshold = coverage_rule(index_type)
    return {
        f"Mean {index_type}": float(np.mean(index_array)),            # :507
        f"Median {index_type}": float(np.median(index_array)),        # :508
        f"Min {index_type}": float(np.min(index_array)),              # :509
        f"Max {index_type}": float(np.max(index_array)),              # :510
        f"{feature} Coverage (%)": float(np.mean(index_array > threshold) * 100),   # :511
    }


def stats_timeseries_row(index_array, index_type, date):
    """process-images.py:646-658: the inlined twin used by the time-series table."""
    feature, threshold = coverage_rule(index_type)
    return {
        "Date": date,
        "Mean": float(np.mean(index_array)),
        "Median": float(np.median(index_array)),
        "Min": float(np.min(index_array)),
        "Max": float(np.max(index_array)),
        f"{feature} Coverage (%)": float(np.mean(index_array > threshold) * 100),
    }


def timeframe_rows(image_data_list, index_type):
    """process-images.py:619-667 ``calculate_index_statistics_by_timeframe`` up to the DataFrame: the list of row dicts.
    The cached ``corrected_array`` is used where the dict holds one (:636-637), the white balance is computed otherwise
    (:639); an image whose index comes back as ``None`` (empty array: :427-428, :452-453) leaves no row (:649)."""
    rows = []
    for img_data in image_data_list:
        date = img_data["metadata"]["upload_date"]                                   # :632
        if "corrected_array" in img_data and img_data["corrected_array"] is not None:
            corrected = img_data["corrected_array"]
        else:
            arr = img_data["array"]
            corrected = None if arr is None or arr.size == 0 else wb_app(arr)
        index_array = None if corrected is None or corrected.size == 0 else index_app(corrected, index_type)   # :644
        if index_array is not None:
            rows.append(stats_timeseries_row(index_array, index_type, date))          # :650-660
    return rows


def timeseries_points(image_data_list, index_type):
    """process-images.py:814-832: the lists ``create_time_series_plot`` draws -- dates, means, maxima, minima."""
    dates, means, maxs, mins = [], [], [], []
    for img_data in image_data_list:
        dates.append(img_data["metadata"]["upload_date"])                            # :815-816
        if "corrected_array" in img_data and img_data["corrected_array"] is not None:
            corrected = img_data["corrected_array"]
        else:
            corrected = wb_app(img_data["array"])
        index_array = index_app(corrected, index_type)                               # :825
        means.append(float(np.mean(index_array)))                                    # :830
        maxs.append(float(np.max(index_array)))                                      # :831
        mins.append(float(np.min(index_array)))                                      # :832
    return dates, means, maxs, mins


def stats_ndvi(ndvi_array):
    """process-ndvi.py:50-73 ``analyze_ndvi_statistics(ndvi_array)``."""
    out = {
        "mean_ndvi": float(np.mean(ndvi_array)),          # :61
        "median_ndvi": float(np.median(ndvi_array)),      # :62
        "min_ndvi": float(np.min(ndvi_array)),            # :63
        "max_ndvi": float(np.max(ndvi_array)),            # :64
        "std_ndvi": float(np.std(ndvi_array)),            # :65
    }
    above = np.sum(ndvi_array > 0.2)                      # :69
    out["vegetation_coverage"] = float(above / ndvi_array.size * 100)   # :70-71
    return out


def hist50(index_array):
    """process-ndvi.py:97: counts of ``plt.hist(x.flatten(), bins=50, range=(-1, 1))``.

    matplotlib delegates to ``numpy.histogram``; only the counts matter here.
    """
    return np.histogram(np.asarray(index_array).ravel(), bins=HIST_BINS, range=(-1, 1))[0]


# --------------------------------------------------------------------------
# a-7  colormap (per-pixel LUT implied by imshow(cmap, vmin=-1, vmax=1))
# --------------------------------------------------------------------------
def colormap_name(index_type):
    """process-images.py:690-693, backend-process.py:42: RdYlBu for NDWI else RdYlGn."""
    return "RdYlBu" if index_type == "NDWI" else "RdYlGn"


def colormap_mpl(index_array, cmap_name):
    """What ``imshow(index, cmap, vmin=-1, vmax=1)`` maps each sample to (RGBA8).

    process-images.py:695, backend-process.py:43, process-ndvi.py:38.  Needs
    matplotlib; used only to pin ``colormap_closed_form`` and the LUT fixtures.
    """
    import matplotlib
    from matplotlib.colors import Normalize
    cmap = matplotlib.colormaps[cmap_name]
    return cmap(Normalize(-1, 1)(index_array), bytes=True)


# ==========================================================================
# Closed forms (the arithmetic the device kernels implement)
# ==========================================================================
def percentile_from_hist(hist, q_percent):
    """np.percentile(x, q) (method 'linear') from a histogram of integer samples.

    ``hist[v]`` = number of samples equal to integer v.  Follows numpy's
    ``_quantile`` + ``_lerp``: virtual index (n-1)*q in float64, neighbours by
    order statistic, ``a + (b-a)*t`` and, for t >= 0.5, ``b - (b-a)*(1-t)``.
    """
    hist = np.asarray(hist, dtype=np.int64)
    n = int(hist.sum())
    cum = np.cumsum(hist)
    q = np.float64(q_percent) / np.float64(100)           # int / float -> float64
    vi = np.float64(n - 1) * q
    lo = int(np.floor(vi))
    hi = min(lo + 1, n - 1)
    t = vi - np.float64(lo)
    a = np.float64(np.searchsorted(cum, lo, side="right"))    # value of order stat lo
    b = np.float64(np.searchsorted(cum, hi, side="right"))
    d = b - a
    r = a + d * t
    if t >= 0.5:
        r = b - d * (np.float64(1) - t)
    return np.float64(r)


def wb_lut_from_percentiles(lo, hi, nvalues=256):
    """uint8 table T with ``wb_app(x)[..., c] == T[x[..., c]]`` for integer input.

    Same expression as process-images.py:438/:441 evaluated on every possible
    sample value: float64 arithmetic, clip, float32 store, truncating uint8 cast.
    A degenerate channel (hi == lo) gives NaN/inf -> cast result 0 (the cast of
    NaN is platform-defined; x86 and the kernels both give 0).
    """
    v = np.arange(nvalues, dtype=np.float32)
    with np.errstate(all="ignore"):
        y = np.clip((v - np.float64(lo)) / (np.float64(hi) - np.float64(lo)) * 255, 0, 255)
        y32 = y.astype(np.float32)
        out = y32.astype(np.uint8)
    out[~np.isfinite(y32)] = 0
    return out


def wb_closed_form(img_array):
    """``wb_app`` for integer images via histogram -> percentiles -> table."""
    if img_array is None or img_array.size == 0:
        return None
    nvalues = 256 if img_array.dtype == np.uint8 else 65536
    out = np.zeros(img_array.shape, dtype=np.uint8)
    for band in (0, 1, 2):
        plane = img_array[:, :, band]
        hist = np.bincount(plane.ravel(), minlength=nvalues)
        lo = percentile_from_hist(hist, 2)
        hi = percentile_from_hist(hist, 98)
        out[:, :, band] = wb_lut_from_percentiles(lo, hi, nvalues)[plane]
    return out


def wb_float_closed_form(img_array):
    """``wb_app`` for any other sample type, spelled the way csrc/wb_generic.hip does it: the reference's own
    ``astype(np.float32)`` (process-images.py:431), then per channel the two order statistics next to each percentile's
    virtual index (n - 1) * q, numpy's ``_lerp`` for float32 data (difference b - a in float32, blend in float64,
    ``b - d * (1 - t)`` where ``t >= 0.5``), the float64 stretch of :438, a float32 store and a truncating cast (:441;
    NaN -> 0).  Returns ``(uint8 image, float64 percentiles[3][2])``."""
    f = img_array.astype(np.float32)
    n = f.shape[0] * f.shape[1]
    out = np.zeros(f.shape, dtype=np.uint8)
    pcts = np.zeros((3, 2), dtype=np.float64)
    for band in (0, 1, 2):
        ordered = np.sort(f[:, :, band].ravel())
        for j, q in enumerate((2.0, 98.0)):
            vi = np.float64(n - 1) * (q / 100.0)
            k0 = int(np.floor(vi))
            k1 = min(k0 + 1, n - 1)
            t = vi - np.floor(vi)
            a, b = ordered[k0], ordered[k1]
            with np.errstate(all="ignore"):
                d = np.float64(np.float32(b - a))               # float32 subtraction, then widened
                r = np.float64(a) + d * t
                if t >= 0.5:
                    r = np.float64(b) - d * (1.0 - t)
            pcts[band, j] = r
        lo, span = pcts[band, 0], pcts[band, 1] - pcts[band, 0]
        with np.errstate(all="ignore"):
            v = (f[:, :, band].astype(np.float64) - lo) / span * 255.0
            nan = np.isnan(v)
            v = np.where(v < 0.0, 0.0, np.where(v > 255.0, 255.0, v))
            u = np.where(nan, 0.0, v).astype(np.float32).astype(np.uint8)
        out[:, :, band] = u
    return out, pcts


def index_closed_form(a, b):
    """(a-b)/(a+b) in IEEE float32 with +0.0 where a+b == 0.

    For non-negative integer-valued float32 a, b < 2**23 this is bit-identical to
    ``clip((a-b)/(a+b+1e-10), -1, 1)`` in float32: the epsilon is absorbed for
    every sum >= 1, and for sum == 0 the numerator is +0.0.
    """
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    s = a + b
    with np.errstate(all="ignore"):
        q = (a - b) / np.where(s == 0, np.float32(1), s)
    return q.astype(np.float32)


def colormap_closed_form(index_array, lut_rgba8):
    """RGBA8 = LUT[min(int((x + 1f) * 128f), 255)] -- float32 arithmetic.

    Equals ``colormap_mpl`` for float32 x in [-1, 1] (Normalize(-1,1) in float32,
    then ``int(norm*256)`` with 256 folded to 255).
    """
    x = np.asarray(index_array, dtype=np.float32)
    scaled = (x + np.float32(1)) * np.float32(128)
    idx = np.minimum(scaled.astype(np.int32), 255)
    return np.asarray(lut_rgba8, dtype=np.uint8)[idx]


def colormap_entry_closed_form(index_array):
    """The colormap entry of every sample: ``min(int((x + 1f) * 128f), 255)`` in float32 -- ``lut[entry]`` is
    ``colormap_closed_form`` (matplotlib's Normalize(-1, 1) + Colormap.__call__; process-images.py:690-695)."""
    x = np.asarray(index_array, dtype=np.float32)
    scaled = (x + np.float32(1)) * np.float32(128)
    return np.clip(scaled.astype(np.int32), 0, 255).astype(np.uint8)


def hist50_edges(dtype=np.float32):
    """The 51 bin edges numpy.histogram builds for bins=50, range=(-1, 1).

    For a float32 sample array the edges are float32 (weak Python scalars), for
    float64 they are float64.
    """
    return np.linspace(-1, 1, HIST_BINS + 1, endpoint=True, dtype=dtype)


def hist50_closed_form(index_array):
    """Bin by the edges: e[i] <= x < e[i+1], last bin closed (numpy's corrected result)."""
    x = np.asarray(index_array).ravel()
    edges = hist50_edges(x.dtype if x.dtype in (np.float32, np.float64) else np.float64)
    guess = ((x - edges[0]) / (edges[-1] - edges[0]) * HIST_BINS).astype(np.int64)
    guess = np.clip(guess, 0, HIST_BINS - 1)
    guess -= (x < edges[guess])
    guess += (x >= edges[guess + 1]) & (guess != HIST_BINS - 1)
    return np.bincount(guess, minlength=HIST_BINS).astype(np.int64)


# --------------------------------------------------------------------------
# Batched / sharded statistics (new entry point; semantics SURVEY.md 8(e))
# --------------------------------------------------------------------------
def tile_partials(index_array, index_type):
    """Order-independent partial record of one tile for one index."""
    x = np.asarray(index_array, dtype=np.float32).ravel()
    _, threshold = coverage_rule(index_type)
    x64 = x.astype(np.float64)
    return {
        "count": int(x.size),
        "sum": float(np.sum(x64)),
        "sumsq": float(np.sum(x64 * x64)),
        "above": int(np.count_nonzero(x > np.float32(threshold))),
        "min": float(x.min()),
        "max": float(x.max()),
        "hist": hist50(x).astype(np.int64),
    }


def merge_partials(records):
    """Fold per-tile (or per-rank) records into global statistics."""
    records = list(records)
    count = sum(r["count"] for r in records)
    total = float(np.sum(np.array([r["sum"] for r in records], dtype=np.float64)))
    totsq = float(np.sum(np.array([r["sumsq"] for r in records], dtype=np.float64)))
    above = sum(r["above"] for r in records)
    mean = total / count
    var = max(totsq / count - mean * mean, 0.0)
    return {
        "count": count,
        "mean": mean,
        "std": float(np.sqrt(var)),
        "min": min(r["min"] for r in records),
        "max": max(r["max"] for r in records),
        "coverage": above / count * 100.0,
        "hist": np.sum([r["hist"] for r in records], axis=0).astype(np.int64),
    }


# --------------------------------------------------------------------------
# Synthetic RGNir tiles (same counter hash as the device generator)
# --------------------------------------------------------------------------
def _mix32(x):
    """32-bit finalizer (lowbias32); uint32 array in, uint32 array out."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def synth_tile_u8(seed, tile, h, w, channels=3, profile="uniform"):
    """Counter-hash synthetic tile; must equal csrc ``lars_synth_u8``.

    One hash per 4-byte word of the interleaved buffer: word k of tile t draws
    ``mix32(mix32(k + seed) ^ (t * 0x9E3779B9))`` and its four bytes are the
    four samples (little endian).  ``profile='vegetation'`` remaps each byte by
    a per-channel affine squeeze so percentiles are non-trivial:
    R -> 20 + v*3/8, G -> 40 + v/2, NIR -> 60 + v*3/4 (integer arithmetic).
    """
    nbytes = h * w * channels
    nwords = (nbytes + 3) // 4
    k = np.arange(nwords, dtype=np.uint64)
    with np.errstate(over="ignore"):
        a = _mix32(((k + np.uint64(seed)) & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        salt = np.uint32((int(tile) * 0x9E3779B9) & 0xFFFFFFFF)
        words = _mix32(a ^ salt)
    raw = words.view(np.uint8)[:nbytes].copy()
    if profile == "vegetation":
        ch = np.arange(nbytes, dtype=np.int64) % channels
        v = raw.astype(np.int64)
        r = 20 + (v * 3) // 8
        g = 40 + v // 2
        n = 60 + (v * 3) // 4
        raw = np.where(ch == 0, r, np.where(ch == 1, g, np.where(ch == 2, n, v))).astype(np.uint8)
    elif profile != "uniform":
        raise ValueError(profile)
    return raw.reshape(h, w, channels)
