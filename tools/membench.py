#!/usr/bin/env python3
"""Is streaming bandwidth a property of WHERE an allocation landed?  Allocates many chunks, times a plain write-only and
a plain read-only sweep of each (lars_d_probe kinds 3 and 1), then sweeps sub-ranges of the slowest and the fastest chunk.

    python tools/membench.py [chunk_GiB=4] [chunks=32]
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lars_image_processing_amd import _ffi


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
    nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    nbytes = int(gib * (1 << 30))
    nbytes -= nbytes % (48 * 1024)
    _ffi.call("lars_set_device", 0)
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def timed(kind, ptr, n, reps=1, blocks=16384):
        _ffi.call("lars_event_record", ev[0], None)
        for _ in range(reps):
            _ffi.call("lars_d_probe", kind, 1, blocks, C.c_void_p(ptr), C.c_void_p(ptr), n, None)
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        return n * reps / ms.value / 1e6

    chunks = [_ffi.DeviceBuffer(nbytes) for _ in range(nchunks)]
    for c in chunks:
        timed(3, c.ptr, nbytes)                                  # first touch
    rows = []
    for rnd in range(3):
        for i, c in enumerate(chunks):
            w = timed(3, c.ptr, nbytes)
            r = timed(1, c.ptr, nbytes)
            if rnd == 0:
                rows.append([i, c.ptr, [], []])
            rows[i][2].append(w)
            rows[i][3].append(r)
    print(f"# {nchunks} chunks of {gib} GiB: write-only / read-only GB/s (3 rounds, interleaved)")
    for i, ptr, w, r in rows:
        print(f"chunk {i:2d} va {ptr:#x}  write {np.median(w):7.1f} ({min(w):7.1f}..{max(w):7.1f})   read {np.median(r):7.1f} ({min(r):7.1f}..{max(r):7.1f})")
    wmed = np.array([np.median(w) for _, _, w, _ in rows])
    rmed = np.array([np.median(r) for _, _, _, r in rows])
    print(f"write: min {wmed.min():.0f} median {np.median(wmed):.0f} max {wmed.max():.0f}   read: min {rmed.min():.0f} median {np.median(rmed):.0f} max {rmed.max():.0f}"
          f"   corr(write, read) = {np.corrcoef(wmed, rmed)[0, 1]:.2f}")
    # inside the slowest and the fastest chunk (by write): 16 sub-ranges each, 10 sweeps per timing
    for label, i in (("slowest", int(np.argmin(wmed))), ("fastest", int(np.argmax(wmed)))):
        sub = nbytes // 16
        sub -= sub % (48 * 1024)
        vals = [timed(3, chunks[i].ptr + k * sub, sub, reps=10, blocks=4096) for k in range(16)]
        print(f"{label} chunk {i}: write GB/s of its 16 sub-ranges of {sub >> 20} MiB: " + " ".join(f"{v:6.0f}" for v in vals))
    print(json.dumps({"write": wmed.tolist(), "read": rmed.tolist()}))


if __name__ == "__main__":
    main()
