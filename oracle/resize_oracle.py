"""NumPy restatement of the LANCZOS down-scale the reference applies before the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/index_oracle.py).

Reference: ``preprocess_large_image(img_array, max_dimension=1024)``, process-images.py:398-422:
``PIL.Image.fromarray(img).resize((new_w, new_h), Image.Resampling.LANCZOS)``.  The arithmetic lives
in Pillow (third-party, unpinned in requirements.txt; 12.2.0 in the build container):
``src/libImaging/Resample.c`` -- ``precompute_coeffs`` (float64 weights of the truncated sinc,
normalised per output sample), ``normalize_coeffs_8bpc`` (round to 22-bit fixed point) and the two
integer passes ``ImagingResampleHorizontal_8bpc`` / ``ImagingResampleVertical_8bpc``
(accumulator starts at 2^21, result = clip8(acc >> 22)), horizontal first, vertical second on the
uint8 intermediate.  This file restates that published algorithm.

Parity status: PINNED -- tests/golden/resize_outputs.npz holds outputs of the reference function
itself (tools/gen_golden.py, Pillow 12.2.0) and tests/test_oracle.py checks this restatement
against them bit for bit.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0


def _lanczos(x):
    def sinc(v):
        if v == 0.0:
            return 1.0
        v = v * math.pi
        return math.sin(v) / v
    if -3.0 <= x < 3.0:
        return sinc(x) * sinc(x / 3)
    return 0.0


def precompute_coeffs(in_size, out_size):
    """-> (ksize, bounds[out_size, 2] (xmin, count), coeffs int32 [out_size, ksize]) for the box (0, in_size)."""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = float(in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            w = _lanczos((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS))
    return ksize, bounds, np.trunc(fixed).astype(np.int64).astype(np.int32)


def _pass(img, out_size, axis):
    """One integer resampling pass along ``axis`` of a uint8 array [H, W, C]."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)                  # [n_in, other, C]
    _, bounds, coeffs = precompute_coeffs(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        k = coeffs[xx, :cnt].astype(np.int64)
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k, src[xmin:xmin + cnt], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def premultiply_rgba(arr):
    """Pillow ``rgbA2rgba`` (Convert.c): colour = MULDIV255(colour, alpha)."""
    a = arr[:, :, 3:4].astype(np.int64)
    t = arr[:, :, :3].astype(np.int64) * a + 128
    out = arr.copy()
    out[:, :, :3] = (((t >> 8) + t) >> 8).astype(np.uint8)
    return out


def unpremultiply_rgba(arr):
    """Pillow ``rgba2rgbA`` (Convert.c): alpha 0 or 255 copies, else CLIP8(255 * colour / alpha)."""
    a = arr[:, :, 3:4].astype(np.int64)
    c = arr[:, :, :3].astype(np.int64)
    q = np.minimum((255 * c) // np.maximum(a, 1), 255)
    out = arr.copy()
    out[:, :, :3] = np.where((a == 0) | (a == 255), c, q).astype(np.uint8)
    return out


def resize_lanczos_u8(img, new_h, new_w):
    """Pillow's ``Image.resize((new_w, new_h), LANCZOS)`` for uint8 [H, W] / [H, W, 3] / [H, W, 4] arrays.

    Four channels are RGBA to Pillow: Image.resize premultiplies by alpha, resamples, and divides
    again (Image.py: mode RGBA -> RGBa -> resize -> RGBA)."""
    arr = np.asarray(img)
    squeeze = arr.ndim == 2
    if squeeze:
        arr = arr[:, :, None]
    h, w = arr.shape[:2]
    rgba = arr.shape[2] == 4
    out = premultiply_rgba(arr) if rgba else arr
    if new_w != w:
        out = _pass(out, new_w, 1)          # horizontal pass first (ImagingResampleInner)
    if new_h != h:
        out = _pass(out, new_h, 0)
    if rgba:
        out = unpremultiply_rgba(out)
    out = np.ascontiguousarray(out)
    return out[:, :, 0] if squeeze else out


def preprocess_large_image(img_array, max_dimension=1024):
    """process-images.py:398-422."""
    if img_array is None or img_array.size == 0:          # :401-402
        return None
    h, w = img_array.shape[:2]
    if max(h, w) <= max_dimension:                        # :407-408 (the very same object comes back)
        return img_array
    if h > w:                                             # :411-416
        new_h = max_dimension
        new_w = int(w * (max_dimension / h))
    else:
        new_w = max_dimension
        new_h = int(h * (max_dimension / w))
    return resize_lanczos_u8(img_array, new_h, new_w)     # :419-422
