"""ctypes binding of liblars_lab.so (include/lars_lab.h): the LABORATORY library -- streaming probes, the persistent
one-launch pipeline, allocation kinds.  Not part of the
product; build it with ``make -C lars_image_processing_amd/csrc lab`` (``__graft_entry__.build()`` does)."""
from __future__ import annotations

import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lars_image_processing_amd import _ffi  # noqa: E402
from lars_image_processing_amd._ffi import FusedArgs, STATS_DTYPE, DeviceBuffer  # noqa: E402

LIB_PATH = os.path.join(ROOT, "lars_image_processing_amd", "liblars_lab.so")
_P, _I, _I64, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_size_t


SIGNATURES = {
    "lars_lab_malloc": (_I, [C.POINTER(_P), _SZ, _I, _I, _I, _I]),
    "lars_lab_free": (_I, [_P]),
    "lars_lab_set_tuning": (_I, [C.c_char_p, _I]),
    "lars_pipeline_scratch_bytes": (_SZ, [_I64, _I64]),
    "lars_d_pipeline": (_I, [C.POINTER(FusedArgs), _P, _P, _I, _P]),
    "lars_d_probe": (_I, [_I, _I, _I, _P, _P, _I64, _P]),
}
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def load():
    global _lib
    if _lib is None:
        _ffi.load()                                  # liblars_hip.so first (RTLD_GLOBAL): the laboratory links against it
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def call(name, *args):
    return _ffi.check(getattr(load(), name)(*args))


def set_tuning(**kw):
    for k, v in kw.items():
        call("lars_lab_set_tuning", k.encode(), int(v))


class LabBuffer(_ffi._DeviceRange):
    """Device memory from ``lars_lab_malloc`` (or adopted from the laboratory), freed with ``lars_lab_free``."""

    def __init__(self, nbytes, kind=0, chunk_mb=0, align_mb=0, shuffle=0, adopt=None):
        self.nbytes = int(nbytes)
        if adopt is not None:
            self.ptr = int(adopt)
            return
        p = C.c_void_p()
        call("lars_lab_malloc", C.byref(p), self.nbytes, int(kind), int(chunk_mb), int(align_mb), int(shuffle))
        self.ptr = p.value

    def free(self):
        if getattr(self, "ptr", None):
            try:
                load().lars_lab_free(C.c_void_p(self.ptr))
            finally:
                self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def probe(kind, unroll, blocks, src, dst, nbytes, stream=None):
    call("lars_d_probe", int(kind), int(unroll), int(blocks), C.c_void_p(src) if src else None, C.c_void_p(dst) if dst else None,
         int(nbytes), stream)


# ---- the persistent pipeline (csrc/lab/pipeline.hip) -------------------------------------------------------------------
def can_pipeline(batch, indices, outputs, hist=False, sumsq=False):
    """What ``lars_d_pipeline`` serves: uint8 RGNir tiles, all three planes written, basic statistics."""
    return (batch.code == _ffi.U8 and batch.channels == 3 and batch.npix % 4 == 0 and batch.npix * 3 < (1 << 31)
            and tuple(sorted(indices)) == tuple(sorted(_ffi.INDEX_NAMES)) and outputs is not None and not hist and not sumsq
            and all(outputs.index[k] is not None for k in range(3)) and outputs.wb is None and all(r is None for r in outputs.rgba))


def run_pipeline(batch, stats, outputs, stream=None, tile_start=0, tile_count=None, rgn_variant=0):
    """Channel histograms -> percentile tables -> fused pass of tiles [tile_start, tile_start + tile_count) in ONE persistent
    launch.  Fills ``batch.hist`` / ``batch.table`` / ``batch.percentiles`` like ``compute_wb_tables`` and ``stats`` / the planes
    like ``run_fused``: same bytes."""
    tile_count = batch.ntiles - tile_start if tile_count is None else tile_count
    if batch.table is None:
        batch.table = DeviceBuffer(batch.ntiles * batch.table_bytes)
        batch.percentiles = DeviceBuffer(batch.ntiles * 3 * 2 * 8)
    if batch.hist is None:
        batch.hist = DeviceBuffer(batch.ntiles * 3 * 256 * 4)
    need = int(load().lars_pipeline_scratch_bytes(tile_count, batch.npix))
    if getattr(batch, "_pipe_scratch", None) is None or batch._pipe_scratch.nbytes < need:
        if getattr(batch, "_pipe_scratch", None) is not None:
            _ffi.call("lars_synchronize", stream)
            batch._pipe_scratch.free()
        batch._pipe_scratch = DeviceBuffer(need)
    batch._table_channels = {0, 1, 2}
    a = batch.fused_args(_ffi.INDEX_NAMES, True, stats, False, outputs, stream, tile_start, tile_count)
    call("lars_d_pipeline", C.byref(a), C.c_void_p(batch.percentiles.ptr + tile_start * 48),
         C.c_void_p(batch.hist.ptr + tile_start * 3072), int(rgn_variant), C.c_void_p(batch._pipe_scratch.ptr))
