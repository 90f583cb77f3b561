"""CPU restatement of the registration / change-detection steps around the hot path.

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product path.

Reference: ``align_images`` (process-images.py:515-565) and
``create_change_detection_visualization`` (process-images.py:885-989).

Pinning status
--------------
* ``shift_reflect`` (the ``ndimage.shift(moving, shift, order=1, mode='reflect')`` of :557):
  PINNED -- scipy is installed, tests compare the restatement with scipy itself.
* ``colormap_norm_closed_form`` (``imshow(diff, cmap='bwr', vmin=-0.5, vmax=0.5)``, :956):
  PINNED by tests/golden/reference_outputs.npz (``colormap/bwr_diff_probe_rgba``, produced by
  matplotlib 3.10.8 through tools/gen_golden.py).
* ``rgb2gray`` / ``phase_cross_correlation``: PARITY UNPINNED.  Both live in scikit-image
  (requirements.txt: ``scikit-image``, unpinned), which is not installed here and not vendored
  in the reference; the reference holds no fixture for them.  The functions below restate the
  published algorithms (skimage.color.rgb2gray: ITU-R 709 luma weights 0.2125 / 0.7154 / 0.0721
  on the image scaled to [0, 1]; skimage.registration.phase_cross_correlation with its defaults
  ``upsample_factor=1, space='real', normalization='phase'``: FFT cross-power spectrum divided by
  max(|.|, 100 eps), inverse FFT, argmax of the magnitude, wrap to signed shifts).  The result is an
  integer shift; tests anchor it on synthetic pairs whose true displacement is known.
"""
from __future__ import annotations

import numpy as np

from . import index_oracle as orc

GRAY_WEIGHTS = (0.2125, 0.7154, 0.0721)


def rgb2gray(img):
    """skimage.color.rgb2gray on an [H, W, 3] image (uint8 -> float64 in [0, 1] first)."""
    a = np.asarray(img)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"the input array must have size 3 along `channel_axis`, got {a.shape}")
    if a.dtype.kind == "u":
        f = a.astype(np.float64) * (1.0 / np.iinfo(a.dtype).max)
    else:
        f = a.astype(np.float64)
    return f[..., 0] * GRAY_WEIGHTS[0] + f[..., 1] * GRAY_WEIGHTS[1] + f[..., 2] * GRAY_WEIGHTS[2]


def phase_cross_correlation(reference_image, moving_image):
    """Integer shift that registers ``moving_image`` with ``reference_image`` (skimage defaults)."""
    ref = np.asarray(reference_image)
    mov = np.asarray(moving_image)
    if ref.shape != mov.shape:
        raise ValueError("images must be same shape")
    src = np.fft.fftn(ref.astype(np.float64))
    tgt = np.fft.fftn(mov.astype(np.float64))
    prod = src * tgt.conj()
    eps = np.finfo(np.float64).eps
    prod /= np.maximum(np.abs(prod), 100 * eps)
    cc = np.fft.ifftn(prod)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    shape = np.array(cc.shape)
    midpoint = np.fix(shape / 2)
    shift = np.array(maxima, dtype=np.float64)
    wrap = shift > midpoint
    shift[wrap] -= shape[wrap]
    shift[shape == 1] = 0
    return shift


def shift_reflect(img, shift):
    """``ndimage.shift(img, shift, order=1, mode='reflect')`` for integer-valued shifts: a gather.

    out[i] = in[reflect(i - shift)] per axis, 'reflect' = half-sample symmetric (d c b a | a b c d | d c b a).
    """
    a = np.asarray(img)
    out = a
    for axis, s in enumerate(shift):
        s = int(round(float(s)))
        n = a.shape[axis]
        src = np.arange(n) - s
        period = 2 * n
        src = np.mod(src, period)
        src = np.where(src >= n, period - 1 - src, src)
        out = np.take(out, src, axis=axis)
    return out


def preprocess_large_image(img, max_dimension=1024):
    """process-images.py:398-422 through Pillow itself (Pillow is installed)."""
    from PIL import Image
    if img is None or np.size(img) == 0:
        return None
    h, w = img.shape[:2]
    if max(h, w) <= max_dimension:
        return img
    if h > w:
        new_h, new_w = max_dimension, int(w * (max_dimension / h))
    else:
        new_w, new_h = max_dimension, int(h * (max_dimension / w))
    return np.array(Image.fromarray(img).resize((new_w, new_h), Image.LANCZOS))


def align_images(fixed_img, moving_img):
    """process-images.py:515-565."""
    if fixed_img is None or moving_img is None:
        return moving_img, np.array([0, 0])
    max_dim = 1024
    if fixed_img.shape[0] > max_dim or fixed_img.shape[1] > max_dim:
        fixed_img = preprocess_large_image(fixed_img, max_dim)
    if moving_img.shape[0] > max_dim or moving_img.shape[1] > max_dim:
        moving_img = preprocess_large_image(moving_img, max_dim)
    fixed_gray = rgb2gray(fixed_img) if fixed_img.ndim == 3 else fixed_img
    moving_gray = rgb2gray(moving_img) if moving_img.ndim == 3 else moving_img
    shift = phase_cross_correlation(fixed_gray, moving_gray)
    if moving_img.ndim == 3 and len(shift) == 2:
        shift = np.append(shift, 0)
    return shift_reflect(moving_img, shift), shift


def change_detection(early_corrected, late_corrected, index_type):
    """process-images.py:905-923: -> (early_index, late_index, diff, aligned_late, shift)."""
    aligned_late, shift = align_images(early_corrected, late_corrected)
    early_index = orc.index_app(early_corrected, index_type)
    late_index = orc.index_app(aligned_late, index_type)
    return early_index, late_index, late_index - early_index, aligned_late, shift


def colormap_norm_closed_form(x, lut_rgba8, vmin, vmax):
    """RGBA8 of ``cmap(Normalize(vmin, vmax)(x), bytes=True)`` for float32 x (matplotlib 3.10 colors.py).

    Normalize: ``(x - vmin) / (vmax - vmin)`` in float32; Colormap.__call__: ``xa = norm * 256``;
    ``xa == 256 -> 255``; ``xa < 0 -> under`` (first colour), ``xa >= 256 -> over`` (last colour);
    NaN -> bad (transparent black).
    """
    x = np.asarray(x, dtype=np.float32)
    norm = (x - np.float32(vmin)) / np.float32(np.float32(vmax) - np.float32(vmin))
    xa = norm * np.float32(256)
    with np.errstate(invalid="ignore"):
        idx = np.clip(np.where(np.isnan(xa), 0, xa), -1, 256).astype(np.int32)
    idx = np.where(xa < 0, 0, np.minimum(idx, 255))
    out = np.asarray(lut_rgba8, dtype=np.uint8)[idx]
    out[np.isnan(x)] = 0
    return out
