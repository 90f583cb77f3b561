"""Host-only sanitizer target: `make asan` builds the code that sees untrusted bytes (the TIFF LZW
decoder) and the host-side fold with g++ -fsanitize=address,undefined; this test replays valid, truncated and corrupted
streams through it (every buffer at its exact size on the heap) and requires (1) no sanitizer report and (2) the same
status / output as the shipped library.  Build container only: never run on the GPU box."""
import ctypes as C
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from lars_image_processing_amd import _ffi
from test_tiffio import lzw_encode

CSRC = os.path.join(ROOT, "lars_image_processing_amd", "csrc")
BIN = os.path.join(ROOT, "build", "asan", "lars_host_asan")


def fnv(data):
    h = 1469598103934665603
    for byte in bytes(data):
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def asan_bin():
    if _ffi.device_count() > 0:
        pytest.skip("sanitizer target is for the build container, not the GPU box")
    if not shutil.which("g++"):
        pytest.skip("no g++")
    subprocess.check_call(["make", "-C", CSRC, "asan"], stdout=subprocess.DEVNULL)
    return BIN


def lib_lzw(src, ndst):
    src = np.frombuffer(bytes(src) or b"\0", dtype=np.uint8)[:len(src)]
    dst = np.zeros(max(ndst, 1), dtype=np.uint8)
    n = C.c_int64(-1)
    rc = _ffi.load().lars_h_tiff_lzw_decode(_ffi.ptr(np.ascontiguousarray(src)) if len(src) else _ffi.ptr(dst), len(src),
                                            _ffi.ptr(dst), ndst, C.byref(n))
    return rc, (n.value if rc == 0 else -1), (fnv(dst[:n.value]) if rc == 0 else 0)


def make_cases():
    rng = np.random.default_rng(2024)
    payloads = [rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(), (np.arange(9000) // 5 % 251).astype(np.uint8).tobytes(),
                bytes(6000), b"\x07", b"", bytes(range(256)) * 20]
    cases = []                                              # (kind, a, b, bytes)
    for data in payloads:
        enc = lzw_encode(data)
        for ndst in {len(data), max(0, len(data) - 8), 0, 1, len(data) + 3}:
            cases.append((0, ndst, 0, enc))
        for _ in range(25):                                 # corrupted and truncated streams
            bad = bytearray(enc)
            how = rng.integers(0, 4)
            if how == 0 and bad:
                for _ in range(int(rng.integers(1, 6))):
                    bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
            elif how == 1:
                bad = bad[:int(rng.integers(0, len(bad) + 1))]
            elif how == 2 and bad:
                cut = int(rng.integers(0, len(bad)))
                bad[cut:] = bytes([0xFF]) * (len(bad) - cut)
            else:
                bad += rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            cases.append((0, int(rng.integers(0, len(data) + 10)), 0, bytes(bad)))
    for _ in range(40):                                     # pure noise
        cases.append((0, int(rng.integers(0, 3000)), 0, rng.integers(0, 256, int(rng.integers(0, 400)), dtype=np.uint8).tobytes()))
    # chunked decode: a file of several strips; valid tables, and tables that point outside the file (must be refused)
    strips = [lzw_encode(rng.integers(0, 7, 1500, dtype=np.uint8).tobytes()) for _ in range(5)]
    blob, offs = b"II*\0junk", []
    for sdata in strips:
        offs.append(len(blob))
        blob += sdata
    counts = [len(sdata) for sdata in strips]
    for bad_entry in (None, ("off", 3), ("cnt", 1), ("cnt_huge", 4)):
        o, c = list(offs), list(counts)
        if bad_entry:
            if bad_entry[0] == "off":
                o[bad_entry[1]] = len(blob) + 5
            elif bad_entry[0] == "cnt":
                c[bad_entry[1]] = len(blob)
            else:
                c[bad_entry[1]] = 2 ** 63
        table = struct.pack(f"<{len(o)}Q", *o) + struct.pack(f"<{len(c)}Q", *c)
        for chunk_bytes in (1500, 700):
            cases.append((1, len(o), chunk_bytes, table + blob))
    # host-side fold
    rec = np.zeros(6, dtype=_ffi.STATS_DTYPE)
    rec["sum"] = rng.normal(size=6); rec["count"] = rng.integers(1, 1000, 6); rec["above"] = rng.integers(0, 10, 6)
    rec["min"] = -rng.random(6); rec["max"] = rng.random(6); rec["hist"] = rng.integers(0, 99, (6, 50))
    for n in (6, 1, 0):
        cases.append((2, n, 0, rec.tobytes()))
    return cases


def test_decoder_and_fold_under_address_and_ub_sanitizers(asan_bin, tmp_path):
    cases = make_cases()
    path = tmp_path / "cases.bin"
    with open(path, "wb") as fh:
        for kind, a, b, data in cases:
            fh.write(struct.pack("<4I", kind, a, b, len(data)) + data)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([asan_bin, str(path)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
    lines = out.stdout.strip().splitlines()
    assert lines[-1] == f"done {len(cases)} cases"
    refused = 0
    for i, ((kind, a, b, data), line) in enumerate(zip(cases, lines)):
        words = dict(w.split("=") for w in line.split()[2:])
        if kind == 0:
            rc, n, h = lib_lzw(data, a)
            assert (int(words["rc"]), int(words["n"]), int(words["h"], 16)) == (rc, n, h), (i, line)
            assert n <= a
        elif kind == 1:
            table = a * 16
            offs = np.frombuffer(data[:a * 8], dtype=np.uint64).copy()
            cnts = np.frombuffer(data[a * 8:table], dtype=np.uint64).copy()
            blob = np.frombuffer(data[table:], dtype=np.uint8).copy()
            dst = np.zeros(a * b, dtype=np.uint8)
            produced = np.full(a, -1, dtype=np.int64)
            rc = _ffi.load().lars_h_tiff_lzw_decode_chunks(_ffi.ptr(blob), blob.size, _ffi.ptr(offs), _ffi.ptr(cnts), a,
                                                           _ffi.ptr(dst), b, _ffi.ptr(produced), 3)
            assert int(words["rc"]) == rc, (i, line)
            refused += rc != 0
            if rc == 0:
                assert int(words["h"], 16) == fnv(dst) and int(words["p"], 16) == fnv(produced.tobytes())
        else:
            rec = np.frombuffer(data, dtype=_ffi.STATS_DTYPE).copy()
            out_rec = np.zeros(1, dtype=_ffi.STATS_DTYPE)
            rc = _ffi.load().lars_stats_merge(_ffi.ptr(rec), a, _ffi.ptr(out_rec))
            assert int(words["rc"]) == rc and (rc != 0) == (a == 0)
            if rc == 0:
                assert int(words["h"], 16) == fnv(out_rec.tobytes())
    assert refused == 6                                     # every table that points outside the file
