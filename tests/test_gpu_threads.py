"""Re-entrancy of the host entry points: Streamlit runs sessions as threads of one process.
Every thread gets its own stream and workspace inside liblars_hip.so; ctypes releases the GIL during a call."""
import threading
import warnings

import numpy as np
import pytest

from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu


def test_concurrent_sessions_get_their_own_results():
    import lars_image_processing_amd as lars
    nthreads, rounds = 6, 5
    images = [orc.synth_tile_u8(31, t, 200 + 37 * t, 300 - 23 * t, profile="vegetation" if t % 2 else "uniform")
              for t in range(nthreads)]
    want = []
    for img in images:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wb = orc.wb_app(img)
        want.append((wb, {t: orc.index_app(wb, t) for t in ("NDVI", "GNDVI", "NDWI")}))
    errors = []
    barrier = threading.Barrier(nthreads)

    def session(k):
        try:
            barrier.wait()
            for _ in range(rounds):
                res = lars.process_image(images[k], want_hist=True)
                np.testing.assert_array_equal(res["corrected"], want[k][0])
                for t, plane in want[k][1].items():
                    np.testing.assert_array_equal(res["indices"][t]["index"].view(np.uint32), plane.view(np.uint32))
                    assert res["indices"][t]["stats"][f"Median {t}"] == float(np.median(plane))
                wb = lars.fix_white_balance(images[k])
                np.testing.assert_array_equal(wb, want[k][0])
                st = lars.analyze_index(want[k][1]["NDWI"], "NDWI")
                assert st["Max NDWI"] == float(want[k][1]["NDWI"].max())
                aligned, shift = lars.align_images(want[k][0], np.roll(want[k][0], (2, -3), axis=(0, 1)))
                np.testing.assert_array_equal(shift, [-2, 3, 0])
        except BaseException as e:                           # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=session, args=(k,)) for k in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    assert not any(th.is_alive() for th in threads)


def test_threads_give_their_device_memory_back():
    """A session thread that ends releases its stream, workspace and FFT plan (thread-exit hook in runtime.cpp)."""
    import ctypes as C
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import _ffi

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        _ffi.call("lars_mem_info", C.byref(f), C.byref(t))
        return f.value

    img = orc.synth_tile_u8(3, 0, 2048, 2048, profile="vegetation")       # ~80 MiB of workspace per process_image call
    lars.process_image(img)                                               # main thread's own workspace, stays
    before = free_bytes()
    grown = []

    def session():
        lars.process_image(img)
        grown.append(before - free_bytes())

    for _ in range(8):
        th = threading.Thread(target=session)
        th.start()
        th.join()
    import time
    deadline = time.time() + 5.0                          # join() returns before the OS thread has run its exit hooks
    while before - free_bytes() >= 16 << 20 and time.time() < deadline:
        time.sleep(0.05)
    after = free_bytes()
    assert max(grown) > 32 << 20                          # a live session does hold a workspace
    assert before - after < 16 << 20, (before, after)     # and eight finished ones hold nothing
