"""ctypes binding of liblars_lab.so (include/lars_lab.h): the LABORATORY library -- streaming probes and allocation
kinds.  Not part of the product; build it with ``make -C lars_image_processing_amd/csrc lab`` (``__graft_entry__.build()`` does)."""
from __future__ import annotations

import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lars_image_processing_amd import _ffi  # noqa: E402
from lars_image_processing_amd._ffi import DeviceBuffer  # noqa: E402,F401

LIB_PATH = os.path.join(ROOT, "lars_image_processing_amd", "liblars_lab.so")
_P, _I, _I64, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_size_t


SIGNATURES = {
    "lars_lab_malloc": (_I, [C.POINTER(_P), _SZ, _I, _I, _I, _I]),
    "lars_lab_free": (_I, [_P]),
    "lars_d_probe": (_I, [_I, _I, _I, _P, _P, _I64, _P]),
    "lars_d_probe_mix3": (_I, [_P, _P, _P, _P, _I64, _I, _P]),
    "lars_lab_copy": (_I, [_I, _I, _P, _P, _SZ]),
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


def probe_mix3(src, d0, d1, d2, nquads, blocks=65536, stream=None):
    call("lars_d_probe_mix3", C.c_void_p(src), C.c_void_p(d0), C.c_void_p(d1), C.c_void_p(d2), int(nquads), int(blocks), stream)

