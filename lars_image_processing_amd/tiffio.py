"""Minimal TIFF reader / writer for multi-sample 16-bit imagery (BASELINE configs[4]: uint16 RGNir tiles).

The reference opens every file with ``PIL.Image.open`` (backend-process.py:52, process-ndvi.py:15), and Pillow
has no mode for three 16-bit samples per pixel: it hands such a file over as 8-bit RGB (the high bytes), so the
reference never sees the full depth.  This module reads them into the ``[H, W, C]`` uint16 array the hot path takes
(``fix_white_balance`` / ``calculate_index`` accept any integer dtype).  Scope: classic (32-bit offset) TIFF, first image of the file, unsigned integer
samples of 8 or 16 bits, chunky or planar layout, strips or tiles, either byte order, uncompressed, LZW (through the
library's host-side decoder) or Deflate (compression 8 / 32946), with or without the horizontal predictor.  Anything else raises ``TiffError`` naming the
feature -- never a silently wrong array.

No oracle exists in the reference for this loader; tests pin it by round trips, against Pillow on the
files both can read (8-bit RGB, 16-bit single band), and against hand-assembled files.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8}
_TYPE_CODES = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i"}

IMAGE_WIDTH, IMAGE_LENGTH, BITS_PER_SAMPLE, COMPRESSION, PHOTOMETRIC = 256, 257, 258, 259, 262
STRIP_OFFSETS, SAMPLES_PER_PIXEL, ROWS_PER_STRIP, STRIP_BYTE_COUNTS = 273, 277, 278, 279
PLANAR_CONFIG, PREDICTOR, TILE_WIDTH, TILE_LENGTH, TILE_OFFSETS, TILE_BYTE_COUNTS = 284, 317, 322, 323, 324, 325
EXTRA_SAMPLES, SAMPLE_FORMAT = 338, 339


class TiffError(ValueError):
    """The file is not a TIFF this reader covers (the message names what is missing)."""


def _read_ifd(buf, endian):
    if len(buf) < 8:
        raise TiffError("file shorter than a TIFF header")
    magic, first = struct.unpack_from(endian + "HI", buf, 2)
    if magic == 43:
        raise TiffError("BigTIFF (64-bit offsets) is not supported")
    if magic != 42:
        raise TiffError(f"bad TIFF magic {magic}")
    if first + 2 > len(buf):
        raise TiffError("first directory lies outside the file")
    (count,) = struct.unpack_from(endian + "H", buf, first)
    tags = {}
    for i in range(count):
        at = first + 2 + 12 * i
        if at + 12 > len(buf):
            raise TiffError("directory entry outside the file")
        tag, typ, n = struct.unpack_from(endian + "HHI", buf, at)
        size = _TYPE_SIZES.get(typ)
        if size is None:
            continue                                     # unknown field type: the TIFF spec says skip it
        nbytes = size * n
        where = at + 8 if nbytes <= 4 else struct.unpack_from(endian + "I", buf, at + 8)[0]
        if where + nbytes > len(buf):
            raise TiffError(f"value of tag {tag} lies outside the file")
        code = _TYPE_CODES.get(typ)
        if code is None:
            continue                                     # rationals, floats, ascii: nothing this reader needs
        tags[tag] = struct.unpack_from(endian + str(n) + code, buf, where)
    return tags


def _one(tags, tag, default=None):
    v = tags.get(tag)
    if v is None or len(v) == 0:
        if default is None:
            raise TiffError(f"required tag {tag} is missing")
        return default
    return int(v[0])


def _lzw_all(buf, offsets, counts, chunk_bytes):
    """Every LZW strip / tile through the library's host-side decoder (csrc/tiff_codec.cpp; threads, no GPU involved):
    (uint8[nchunks][chunk_bytes], bytes produced per chunk)."""
    import os
    from . import _ffi
    n = len(offsets)
    data = np.empty((n, chunk_bytes), dtype=np.uint8)
    produced = np.zeros(n, dtype=np.int64)
    off = np.asarray(offsets, dtype=np.uint64)
    cnt = np.asarray(counts, dtype=np.uint64)
    src = np.frombuffer(buf, dtype=np.uint8)
    try:
        _ffi.call("lars_h_tiff_lzw_decode_chunks", _ffi.ptr(src), src.size, _ffi.ptr(off), _ffi.ptr(cnt), n, _ffi.ptr(data),
                  chunk_bytes, _ffi.ptr(produced), min(8, os.cpu_count() or 1))
    except _ffi.LarsError as exc:
        raise TiffError(str(exc)) from None
    return data, produced


def _chunk(buf, offset, nbytes, compression, want):
    if offset + nbytes > len(buf):
        raise TiffError("strip / tile data outside the file")
    raw = bytes(buf[offset:offset + nbytes])
    if compression != 1:
        # bounded inflate: a small strip must not be able to inflate past what the directory says it holds
        try:
            z = zlib.decompressobj()
            raw = z.decompress(raw, want)
        except zlib.error as exc:
            raise TiffError(f"corrupt Deflate data: {exc}") from None
        if z.unconsumed_tail:
            raise TiffError(f"Deflate strip / tile inflates past its {want} bytes")
    if len(raw) < want:
        raise TiffError(f"strip / tile holds {len(raw)} bytes, {want} expected")
    return raw[:want]


MAX_DECODED_BYTES = 16 << 30      # refuse directories that claim more than this (a corrupt header must not allocate the host away)


def read_tiff(path_or_bytes, max_bytes=MAX_DECODED_BYTES):
    """``[H, W, C]`` (``[H, W]`` for one sample per pixel) uint8 / uint16 array of the first image of a TIFF."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        buf = memoryview(path_or_bytes)
    else:
        buf = memoryview(np.fromfile(path_or_bytes, dtype=np.uint8)).cast("B")     # one read, no second copy for raw strips
    head = bytes(buf[:2])
    if head == b"II":
        endian = "<"
    elif head == b"MM":
        endian = ">"
    else:
        raise TiffError("not a TIFF file (byte-order mark)")
    tags = _read_ifd(buf, endian)
    width, height = _one(tags, IMAGE_WIDTH), _one(tags, IMAGE_LENGTH)
    spp = _one(tags, SAMPLES_PER_PIXEL, 1)
    bits = tags.get(BITS_PER_SAMPLE) or (1,)
    if len(set(bits)) != 1 or len(bits) not in (1, spp):
        raise TiffError(f"samples of different widths {bits} are not supported")
    bits = int(bits[0])
    if bits not in (8, 16):
        raise TiffError(f"{bits}-bit samples are not supported (8 or 16)")
    fmt = tags.get(SAMPLE_FORMAT) or (1,)
    if any(int(f) != 1 for f in fmt):
        raise TiffError(f"sample format {fmt} is not supported (unsigned integer only)")
    compression = _one(tags, COMPRESSION, 1)
    if compression not in (1, 5, 8, 32946):
        names = {2: "CCITT", 6: "old JPEG", 7: "JPEG", 32773: "PackBits"}
        raise TiffError(f"compression {compression} ({names.get(compression, 'unknown')}) is not supported (none, LZW or Deflate)")
    predictor = _one(tags, PREDICTOR, 1)
    if predictor not in (1, 2):
        raise TiffError(f"predictor {predictor} is not supported")
    planar = _one(tags, PLANAR_CONFIG, 1)
    if planar not in (1, 2):
        raise TiffError(f"planar configuration {planar}")
    if width <= 0 or height <= 0 or spp <= 0:
        raise TiffError("empty image")
    if width * height * spp * (bits // 8) > max_bytes:
        raise TiffError(f"image of {width} x {height} x {spp} x {bits} bits exceeds max_bytes={max_bytes}")
    dtype = np.dtype(np.uint8 if bits == 8 else (endian + "u2"))
    planes = spp if planar == 2 else 1                    # separately stored sample planes
    inner = 1 if planar == 2 else spp                     # samples per pixel inside one chunk

    if TILE_WIDTH in tags:
        tw, th = _one(tags, TILE_WIDTH), _one(tags, TILE_LENGTH)
        if tw <= 0 or th <= 0 or tw * th * spp * (bits // 8) > max_bytes:
            raise TiffError(f"bad tile size {tw} x {th}")
        offsets, counts = tags.get(TILE_OFFSETS), tags.get(TILE_BYTE_COUNTS)
        across, down = -(-width // tw), -(-height // th)
        chunk_h, chunk_w = th, tw
    else:
        rps = min(_one(tags, ROWS_PER_STRIP, height), height)
        if rps <= 0:
            raise TiffError("rows per strip must be positive")
        offsets, counts = tags.get(STRIP_OFFSETS), tags.get(STRIP_BYTE_COUNTS)
        across, down = 1, -(-height // rps)
        chunk_h, chunk_w = rps, width
    if offsets is None:
        raise TiffError("no strip / tile offsets")
    if counts is None:
        if compression != 1 or len(offsets) != 1:
            raise TiffError("strip / tile byte counts are missing")
        counts = (width * height * inner * dtype.itemsize,)
    if len(offsets) != across * down * planes or len(counts) != len(offsets):
        raise TiffError(f"{len(offsets)} strips / tiles, {across * down * planes} expected")

    from .hostpool import empty
    out = empty((planes, height, width, inner), dtype.newbyteorder("="))
    full = chunk_h * chunk_w * inner * dtype.itemsize        # bytes of a whole strip / tile
    decoded = _lzw_all(buf, offsets, counts, full) if compression == 5 else None
    k = 0
    for p in range(planes):
        for ty in range(down):
            y0 = ty * chunk_h
            rows_here = min(chunk_h, height - y0)
            # tiles are stored whole (padded); the last strip holds only the rows that exist
            stored_rows = chunk_h if TILE_WIDTH in tags else rows_here
            for tx in range(across):
                x0 = tx * chunk_w
                cols_here = min(chunk_w, width - x0)
                want = stored_rows * chunk_w * inner * dtype.itemsize
                if decoded is not None:
                    data, produced = decoded
                    if produced[k] < want:
                        raise TiffError(f"strip / tile holds {int(produced[k])} bytes, {want} expected")
                    raw = data[k, :want]
                else:
                    raw = _chunk(buf, int(offsets[k]), int(counts[k]), compression, want)
                k += 1
                a = np.frombuffer(raw, dtype=dtype).reshape(stored_rows, chunk_w, inner)
                if predictor == 2:
                    a = np.cumsum(a, axis=1, dtype=a.dtype.newbyteorder("="))         # wraps modulo 2^bits, as the predictor does
                out[p, y0:y0 + rows_here, x0:x0 + cols_here] = a[:rows_here, :cols_here]
    img = out[0] if planar == 1 else np.ascontiguousarray(np.moveaxis(out[..., 0], 0, -1))
    return img[..., 0] if spp == 1 else img


def write_tiff(path, array, rows_per_strip=None, tile=None, byteorder="<", planar=1, deflate=False, predictor=False):
    """Write ``[H, W]`` / ``[H, W, C]`` uint8 or uint16 samples as a classic TIFF (strips, or tiles when
    ``tile=(rows, cols)``).  Photometric: RGB for >= 3 samples, BlackIsZero otherwise; samples past the third are
    declared unspecified extra samples.  Returns the number of bytes written."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[..., None]
    if a.ndim != 3 or a.dtype not in (np.uint8, np.uint16) or a.size == 0:
        raise TiffError("write_tiff takes a non-empty [H, W] or [H, W, C] uint8 / uint16 array")
    if byteorder not in ("<", ">") or planar not in (1, 2):
        raise TiffError("byteorder must be '<' or '>', planar 1 or 2")
    h, w, c = a.shape
    bits = a.dtype.itemsize * 8
    dt = np.dtype(np.uint8 if bits == 8 else byteorder + "u2")
    planes = [a] if planar == 1 else [a[..., k:k + 1] for k in range(c)]
    chunks = []
    if tile:
        th, tw = int(tile[0]), int(tile[1])
        if th % 16 or tw % 16 or th <= 0 or tw <= 0:
            raise TiffError("tile sizes must be positive multiples of 16")
        for pl in planes:
            for y0 in range(0, h, th):
                for x0 in range(0, w, tw):
                    t = np.zeros((th, tw, pl.shape[2]), dtype=a.dtype)
                    part = pl[y0:y0 + th, x0:x0 + tw]
                    t[:part.shape[0], :part.shape[1]] = part
                    chunks.append(t)
    else:
        rps = h if not rows_per_strip else max(1, min(int(rows_per_strip), h))
        for pl in planes:
            for y0 in range(0, h, rps):
                chunks.append(pl[y0:y0 + rps])
    blobs = []
    for t in chunks:
        if predictor:
            t = np.concatenate([t[:, :1], np.diff(t, axis=1)], axis=1)      # unsigned wrap-around = modulo 2^bits
        raw = np.ascontiguousarray(t, dtype=a.dtype).astype(dt, copy=False)
        # uncompressed chunks are written straight from the array's memory (no 64 MiB byte-string copies)
        blobs.append(zlib.compress(raw.tobytes(), 6) if deflate else memoryview(raw.reshape(-1).view(np.uint8)))

    entries = []                                             # (tag, type, values)

    def add(tag, typ, *values):
        entries.append((tag, typ, values))

    add(IMAGE_WIDTH, 4, w)
    add(IMAGE_LENGTH, 4, h)
    add(BITS_PER_SAMPLE, 3, *([bits] * c))
    add(COMPRESSION, 3, 8 if deflate else 1)
    add(PHOTOMETRIC, 3, 2 if c >= 3 else 1)
    add(SAMPLES_PER_PIXEL, 3, c)
    add(PLANAR_CONFIG, 3, planar)
    if predictor:
        add(PREDICTOR, 3, 2)
    extra = c - 3 if c > 3 else (c - 1 if c == 2 else 0)
    if extra:
        add(EXTRA_SAMPLES, 3, *([0] * extra))
    add(SAMPLE_FORMAT, 3, *([1] * c))
    header = 8
    data_at = header
    offsets = []
    for b in blobs:
        offsets.append(data_at)
        data_at += len(b) + (len(b) & 1)                     # word alignment
    if tile:
        add(TILE_WIDTH, 4, int(tile[1]))
        add(TILE_LENGTH, 4, int(tile[0]))
        add(TILE_OFFSETS, 4, *offsets)
        add(TILE_BYTE_COUNTS, 4, *[len(b) for b in blobs])
    else:
        add(ROWS_PER_STRIP, 4, rps)
        add(STRIP_OFFSETS, 4, *offsets)
        add(STRIP_BYTE_COUNTS, 4, *[len(b) for b in blobs])
    entries.sort(key=lambda e: e[0])
    ifd_at = data_at
    if ifd_at + 2 + 12 * len(entries) + 4 + sum(4 * len(e[2]) for e in entries) >= 1 << 32:
        raise TiffError("image too large for a classic TIFF (4 GiB)")
    overflow_at = ifd_at + 2 + 12 * len(entries) + 4
    ifd = struct.pack(byteorder + "H", len(entries))
    overflow = b""
    for tag, typ, values in entries:
        code = _TYPE_CODES[typ]
        payload = struct.pack(byteorder + str(len(values)) + code, *values)
        if len(payload) <= 4:
            field = payload.ljust(4, b"\0")
        else:
            field = struct.pack(byteorder + "I", overflow_at + len(overflow))
            overflow += payload + (b"\0" if len(payload) & 1 else b"")
        ifd += struct.pack(byteorder + "HHI", tag, typ, len(values)) + field
    ifd += struct.pack(byteorder + "I", 0)
    with open(path, "wb") as fh:
        fh.write((b"II" if byteorder == "<" else b"MM") + struct.pack(byteorder + "HI", 42, ifd_at))
        for b in blobs:
            fh.write(b)
            if len(b) & 1:
                fh.write(b"\0")
        fh.write(ifd)
        fh.write(overflow)
    return overflow_at + len(overflow)


def read_image(path, full_depth=False):
    """Any raster the pipeline takes -> ``ndarray``.

    ``full_depth=False``: ``np.array(PIL.Image.open(path))``, what the reference does (backend-process.py:52) --
    Pillow opens a three-sample 16-bit TIFF as 8-bit RGB (the high bytes), so the reference never sees the low bytes.
    ``full_depth=True``: such files come back as the ``[H, W, C]`` uint16 array they hold (this module's reader);
    every other file still goes through Pillow.
    """
    from PIL import Image
    p = str(path)
    if full_depth and p.lower().endswith((".tif", ".tiff")):
        try:
            arr = read_tiff(p)
            if arr.dtype == np.uint16 and arr.ndim == 3:
                return arr
        except TiffError:
            pass                                             # Pillow may still know the flavour (LZW, JPEG, ...)
    return np.array(Image.open(p))
