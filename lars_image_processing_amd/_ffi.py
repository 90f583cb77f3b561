"""ctypes binding of liblars_hip.so (include/lars_hip.h).

The library is the product: there is no NumPy fallback.  If the shared object
is missing (run ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C lars_image_processing_amd/csrc``) or no gfx950 device is present,
every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LARS_HIP_LIB", os.path.join(_HERE, "liblars_hip.so"))

HIST_BINS = 50
U8, U16 = 1, 2
NDVI, GNDVI, NDWI = 0, 1, 2
INDEX_IDS = {"NDVI": NDVI, "GNDVI": GNDVI, "NDWI": NDWI}
INDEX_NAMES = ("NDVI", "GNDVI", "NDWI")
F_STATS, F_HIST, F_SUMSQ, F_RAW = 1, 2, 4, 8
COMM_ID_BYTES = 128


class LarsError(RuntimeError):
    """A liblars_hip entry point returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"liblars_hip error {code}: {message}")
        self.code = code


class Stats(C.Structure):
    """``lars_stats`` (include/lars_hip.h)."""
    _fields_ = [
        ("sum", C.c_double), ("sumsq", C.c_double),
        ("count", C.c_uint64), ("above", C.c_uint64), ("nans", C.c_uint64),
        ("min", C.c_double), ("max", C.c_double), ("threshold", C.c_double),
        ("index_id", C.c_uint32), ("reserved", C.c_uint32),
        ("hist", C.c_uint64 * HIST_BINS),
    ]


STATS_DTYPE = np.dtype([
    ("sum", "<f8"), ("sumsq", "<f8"), ("count", "<u8"), ("above", "<u8"), ("nans", "<u8"),
    ("min", "<f8"), ("max", "<f8"), ("threshold", "<f8"), ("index_id", "<u4"), ("reserved", "<u4"),
    ("hist", "<u8", (HIST_BINS,)),
])
assert STATS_DTYPE.itemsize == C.sizeof(Stats) == 472


class FusedArgs(C.Structure):
    """``lars_fused_args`` (include/lars_hip.h)."""
    _fields_ = [
        ("tiles", C.c_void_p), ("ntiles", C.c_int64), ("npix", C.c_int64),
        ("channels", C.c_int32), ("dtype", C.c_int32),
        ("wb_table", C.c_void_p), ("index_mask", C.c_uint32), ("flags", C.c_uint32),
        ("out_index", C.c_void_p * 3), ("out_wb", C.c_void_p),
        ("out_rgba", C.c_void_p * 3), ("cmap_lut", C.c_void_p * 3),
        ("stats", C.c_void_p), ("stream", C.c_void_p),
    ]


# name -> (restype, argtypes).  Every symbol include/lars_hip.h declares.
_P, _I, _I64, _U32, _SZ, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_size_t, C.c_float, C.c_double
SIGNATURES = {
    "lars_abi_version": (_I, []),
    "lars_last_error": (C.c_char_p, []),
    "lars_device_count": (_I, [C.POINTER(_I)]),
    "lars_set_device": (_I, [_I]),
    "lars_get_device": (_I, [C.POINTER(_I)]),
    "lars_device_name": (_I, [C.c_char_p, _SZ]),
    "lars_malloc": (_I, [C.POINTER(_P), _SZ]),
    "lars_free": (_I, [_P]),
    "lars_memset": (_I, [_P, _I, _SZ, _P]),
    "lars_memcpy_h2d": (_I, [_P, _P, _SZ]),
    "lars_memcpy_d2h": (_I, [_P, _P, _SZ]),
    "lars_memcpy_d2d": (_I, [_P, _P, _SZ, _P]),
    "lars_stream_create": (_I, [C.POINTER(_P)]),
    "lars_stream_destroy": (_I, [_P]),
    "lars_synchronize": (_I, [_P]),
    "lars_shutdown": (_I, []),
    "lars_mem_info": (_I, [C.POINTER(_SZ), C.POINTER(_SZ)]),
    "lars_event_create": (_I, [C.POINTER(_P)]),
    "lars_event_destroy": (_I, [_P]),
    "lars_event_record": (_I, [_P, _P]),
    "lars_event_elapsed_ms": (_I, [_P, _P, C.POINTER(_F)]),
    "lars_stream_wait_event": (_I, [_P, _P]),
    "lars_d_channel_hist": (_I, [_P, _I64, _I64, _I, _I, _P, _P]),
    "lars_d_wb_table": (_I, [_P, _I64, _I64, _I, _P, _P, _I, _P]),
    "lars_wb_table_bytes": (_SZ, [_I]),
    "lars_d_wb_prepare": (_I, [_P, _I64, _I64, _I, _I, _P, _P, _I, _P]),
    "lars_d_fused": (_I, [C.POINTER(FusedArgs)]),
    "lars_d_stats_begin": (_I, [_P, _I64, _U32, _P]),
    "lars_d_stats_end": (_I, [_P, _I64, _U32, _I64, _P]),
    "lars_d_index_planes_f32": (_I, [_P, _P, _P, _I64, _I, _P, _P]),
    "lars_d_ndvi_f64": (_I, [_P, _I64, _I, _I, _P, _P]),
    "lars_d_array_stats_f32": (_I, [_P, _I64, _F, _I, _P, _P]),
    "lars_d_array_stats_f64": (_I, [_P, _I64, _D, _I, _P, _P, _P]),
    "lars_select_scratch_bytes": (_SZ, []),
    "lars_d_median_pair_f32": (_I, [_P, _I64, _P, _P, _P]),
    "lars_d_median_pair_f64": (_I, [_P, _I64, _P, _P, _P]),
    "lars_d_median_pair_batch_f32": (_I, [_P, _I64, _I64, _I64, _P, _P, _P]),
    "lars_d_colormap_f32": (_I, [_P, _I64, _P, _P, _P]),
    "lars_d_colormap_entry_f32": (_I, [_P, _I64, _P, _P]),
    "lars_quotient_median_scratch_bytes": (_SZ, [_I64]),
    "lars_d_stats_medians": (_I, [C.POINTER(FusedArgs), _P, _P]),
    "lars_d_quotient_median_pairs": (_I, [_P, _I64, _I64, _I, _I, _P, _U32, _P, _P, _P]),
    "lars_joint_scratch_bytes": (_SZ, [_I64, _I64, _U32]),
    "lars_d_stats_joint": (_I, [C.POINTER(FusedArgs), _I, _I, _P, _P, _P, _P, _SZ]),
    "lars_joint_window_report": (_I, [_P, _I64, C.POINTER(_I64), C.POINTER(_I64)]),
    "lars_joint_window_modes": (_I, [_P, _I64, C.POINTER(_I64)]),
    "lars_h_tiff_lzw_decode": (_I, [_P, _I64, _P, _I64, C.POINTER(_I64)]),
    "lars_h_tiff_lzw_decode_chunks": (_I, [_P, _I64, _P, _P, _I64, _P, _I64, _P, _I]),
    "lars_d_quotient_select_hist": (_I, [_P, _I64, _I64, _I, _I, _P, _U32, _I, _P, _P, _P]),
    "lars_d_synth_u8": (_I, [_P, _I64, _I64, _I64, _I, _U32, _I, _P]),
    "lars_stats_merge": (_I, [_P, _I64, _P]),
    "lars_d_stats_fold": (_I, [_P, _I64, _U32, _P, _P]),
    "lars_build_flags": (C.c_uint, []),
    "lars_set_tuning": (_I, [C.c_char_p, _I]),
    "lars_get_tuning": (_I, [C.c_char_p, C.POINTER(_I)]),
    "lars_d_quot_selfcheck": (_I, [_U32, C.POINTER(C.c_uint64), C.POINTER(_U32 * 2)]),
    "lars_h_fix_white_balance": (_I, [_P, _I64, _I64, _I, _I, _I, _P, _P]),
    "lars_h_fix_white_balance_f32": (_I, [_P, _I64, _I64, _I, _P, _P]),
    "lars_h_calculate_index": (_I, [_P, _I64, _I64, _I, _I, _U32, C.POINTER(_P * 3), _P, _I]),
    "lars_h_calculate_index_planes": (_I, [_P, _P, _P, _I64, _I, _P]),
    "lars_h_ndvi_f64": (_I, [_P, _I64, _I64, _I, _I, _P]),
    "lars_h_analyze_f32": (_I, [_P, _I64, _F, _I, _P, _P]),
    "lars_h_analyze_f64": (_I, [_P, _I64, _D, _I, _P, _P, _P]),
    "lars_h_process_image": (_I, [_P, _I64, _I64, _I, _I, _I, _U32, _I, _P, C.POINTER(_P * 3), _P, _P,
                                  C.POINTER(_P * 3), C.POINTER(_P * 3)]),
    "lars_h_colormap_f32": (_I, [_P, _I64, _P, _P]),
    "lars_h_threshold_mask_f32": (_I, [_P, _I64, _F, _P]),
    "lars_d_threshold_mask_f32": (_I, [_P, _I64, _F, _P, _P]),
    "lars_h_colormap_norm_f32": (_I, [_P, _I64, _F, _F, _P, _P]),
    "lars_h_align_images": (_I, [_P, _P, _I64, _I64, _I, _P, _P]),
    "lars_h_change_detection": (_I, [_P, _P, _I64, _I64, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P]),
    "lars_d_gray_c128": (_I, [_P, _I64, _I, _P, _P]),
    "lars_phase_scratch_bytes": (_SZ, []),
    "lars_d_phase_correlation": (_I, [_P, _P, _I64, _I64, _P, _P, _P]),
    "lars_d_shift_reflect_u8": (_I, [_P, _I64, _I64, _I, _P, _P, _P]),
    "lars_d_diff_f32": (_I, [_P, _P, _I64, _P, _P]),
    "lars_d_colormap_norm_f32": (_I, [_P, _I64, _F, _F, _P, _P, _P]),
    "lars_h_resize_lanczos_u8": (_I, [_P, _I64, _I64, _I, _I64, _I64, _P]),
    "lars_comm_available": (_I, []),
    "lars_comm_unique_id": (_I, [_P]),
    "lars_comm_init": (_I, [C.POINTER(_P), _I, _I, _P]),
    "lars_comm_count": (_I, [_P, C.POINTER(_I)]),
    "lars_comm_destroy": (_I, [_P]),
    "lars_comm_allreduce_stats": (_I, [_P, _P, _I64, _I, _P]),
    "lars_comm_allreduce_f64": (_I, [_P, _P, _I64, _I]),
    "lars_comm_barrier": (_I, [_P]),
}

_lib = None
_lock = threading.Lock()


def load():
    """dlopen liblars_hip.so and declare every prototype.  Never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP library first "
                "(`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C lars_image_processing_amd/csrc`).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.lars_abi_version() != 1:
            raise ImportError(f"{LIB_PATH}: ABI version {lib.lars_abi_version()} != 1")
        _lib = lib
    return _lib


def check(status):
    if status != 0:
        msg = load().lars_last_error()
        raise LarsError(status, msg.decode("utf-8", "replace") if msg else "")
    return status


def call(name, *args):
    """Invoke an entry point and raise LarsError on a negative status."""
    return check(getattr(load(), name)(*args))


def ptr(arr):
    """Host pointer of a C-contiguous ndarray (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.c_void_p)


def ptr3(arrs):
    """``void *[3]`` from three optional ndarrays / integer device addresses."""
    out = (C.c_void_p * 3)()
    for k, a in enumerate(arrs):
        if a is None:
            out[k] = None
        elif isinstance(a, np.ndarray):
            out[k] = a.ctypes.data
        else:
            out[k] = int(a)
    return out


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.uint8:
        return U8
    if dt == np.uint16:
        return U16
    return None


def set_tuning(**kw):
    """e.g. set_tuning(fused_impl=1, nt_stores=1); results never depend on tuning."""
    for k, v in kw.items():
        call("lars_set_tuning", k.encode(), int(v))


def get_tuning(key):
    v = C.c_int(0)
    call("lars_get_tuning", key.encode(), C.byref(v))
    return v.value


def device_count():
    n = C.c_int(0)
    status = load().lars_device_count(C.byref(n))
    return n.value if status == 0 else 0


def device_name():
    buf = C.create_string_buffer(256)
    call("lars_device_name", buf, 256)
    return buf.value.decode()


class _DeviceRange:
    """``ptr`` / ``nbytes`` of device memory with the copy helpers."""
    ptr = None
    nbytes = 0

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        call("lars_memcpy_h2d", C.c_void_p(self.ptr + offset), ptr(arr), arr.nbytes)

    def download(self, dtype, shape, offset=0):
        from .hostpool import empty
        out = empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        call("lars_memcpy_d2h", ptr(out), C.c_void_p(self.ptr + offset), out.nbytes)
        return out

    def zero(self, stream=None):
        """Asynchronous memset on ``stream`` (None = this thread's library stream): pass the stream the consuming
        kernels are launched on, or the memset is not ordered against them."""
        call("lars_memset", C.c_void_p(self.ptr), 0, self.nbytes, stream)


class DeviceBuffer(_DeviceRange):
    """A hipMalloc allocation owned by Python (freed on ``free()`` / GC)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        call("lars_malloc", C.byref(p), self.nbytes)
        self.ptr = p.value

    def free(self):
        if getattr(self, "ptr", None):
            try:
                load().lars_free(C.c_void_p(self.ptr))
            finally:
                self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceSlice(_DeviceRange):
    """``nbytes`` at ``offset`` of a DeviceBuffer, which stays the owner (``free()`` here does nothing)."""

    def __init__(self, owner, offset, nbytes):
        assert 0 <= offset and offset + nbytes <= owner.nbytes
        self.owner, self.ptr, self.nbytes = owner, owner.ptr + int(offset), int(nbytes)

    def free(self):
        self.ptr = None
