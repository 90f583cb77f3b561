"""The laboratory library (liblars_lab.so, include/lars_lab.h): experiments that are not part of the product stay
buildable and correct -- the persistent one-launch pipeline, output arenas assembled from timed groups of physical memory,
the streaming probes (needs a MI355X)."""
import os
import sys
import warnings

import numpy as np
import pytest

from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "lab"))
TYPES = ("NDVI", "GNDVI", "NDWI")


@pytest.fixture(scope="module")
def lars():
    import lars_image_processing_amd as mod
    from lars_image_processing_amd import _ffi
    assert _ffi.device_count() >= 1
    return mod


@pytest.fixture(scope="module")
def lablib():
    import lablib as lab
    assert lab.available(), "build the laboratory library: make -C lars_image_processing_amd/csrc lab"
    lab.load()
    return lab


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("shape,ntiles,steps,head", [((256, 256), 5, 8, 2), ((64, 96), 3, 0, 0), ((130, 62), 4, 1, 1), ((512, 512), 9, 16, 4),
                                                       ((256, 256), 1, 8, 3)])
def test_pipelined_launch_equals_the_two_pass_path(lars, lablib, shape, ntiles, steps, head):
    """csrc/lab/pipeline.hip: histograms -> tables -> fused pass in one persistent launch.  Histograms, percentiles, tables,
    planes and statistics records must be the bytes the separate launches produce (and hence the oracle's)."""
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch.synthetic(ntiles, shape[0], shape[1], seed=31, profile="vegetation")
    want_outs = b.make_outputs(index=True)
    b.compute_wb_tables()
    want_hist, want_tab, want_pct = b.host_hist().copy(), b.host_tables().copy(), b.host_percentiles().copy()
    stats = b.new_stats()
    b.run_fused(b.fused_args(("NDVI", "GNDVI", "NDWI"), True, stats, False, want_outs))
    _ffi.call("lars_synchronize", None)
    want_rec = stats.download(_ffi.STATS_DTYPE, (ntiles, 3)).tobytes()
    want_planes = [want_outs.host_index(t, 0, ntiles).tobytes() for t in ("NDVI", "GNDVI", "NDWI")]
    # forget everything, run the pipeline
    for buf in (b.hist, b.table, b.percentiles, stats):
        buf.zero()
    outs = b.make_outputs(index=True)
    assert lablib.can_pipeline(b, ("NDVI", "GNDVI", "NDWI"), outs)
    lablib.set_tuning(pipe_steps=steps, pipe_head=head)
    try:
        lablib.run_pipeline(b, stats, outs)
        _ffi.call("lars_synchronize", None)
    finally:
        lablib.set_tuning(pipe_steps=0, pipe_head=0)
    np.testing.assert_array_equal(b.host_hist(), want_hist)
    np.testing.assert_array_equal(b.host_percentiles(), want_pct)
    np.testing.assert_array_equal(b.host_tables(), want_tab)
    got = stats.download(_ffi.STATS_DTYPE, (ntiles, 3))
    assert (got["count"] == shape[0] * shape[1]).all()
    assert got.tobytes() == want_rec
    for t, want in zip(("NDVI", "GNDVI", "NDWI"), want_planes):
        assert outs.host_index(t, 0, ntiles).tobytes() == want, t
    # a tile of the result against the oracle as well
    tile = b.host_tiles(ntiles - 1, 1)[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wb = orc.wb_app(tile)
    np.testing.assert_array_equal(bits(outs.host_index("GNDVI", ntiles - 1, 1)[0]), bits(orc.index_app(wb, "GNDVI")))
    # a sub-range of the batch into a ring
    ring = b.make_outputs(index=True, ring=2)
    if ntiles >= 4:
        stats.zero()
        lablib.run_pipeline(b, stats, ring, tile_start=2, tile_count=2)
        _ffi.call("lars_synchronize", None)
        part = stats.download(_ffi.STATS_DTYPE, (ntiles, 3))
        assert part[2:4].tobytes() == got[2:4].tobytes() and not part[:2]["count"].any()
        assert ring.host_index("NDVI", 0, 2).tobytes() == outs.host_index("NDVI", 2, 2).tobytes()
    for o in (want_outs, outs, ring):
        o.free()
    stats.free(); b.free()



def test_assembled_output_arena(lars, lablib):
    """lars_d_output_arena: the planes' memory is put together from candidate groups of physical memory that were timed with
    the batch's own launch; the arena behaves like any other (same records; planes compared on the device), reports what
    the search did and goes back to the driver when freed."""
    import ctypes as C
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch.synthetic(70, 1024, 1024, seed=9, profile="vegetation")
    assert lablib.arena_group_slots(3, b.npix) == 64 and lablib.arena_group_slots(3, 4096 * 4096) == 16
    assert lablib.arena_group_slots(2, 8192 * 8192) == 8
    free0, total = C.c_size_t(), C.c_size_t()
    _ffi.call("lars_mem_info", C.byref(free0), C.byref(total))
    plain = b.make_outputs(index=True, ring=64, arena="plain")
    built = lablib.assembled_outputs(b, ring=64, max_groups=3)
    rep = built.arena_report
    assert rep["kind"].startswith("assembled from 1 of ") and 1 <= len(rep["group_ms"]) <= 3
    assert rep["rejected"] == len(rep["group_ms"]) - 1 and rep["chosen_ms"] == pytest.approx(min(rep["group_ms"])) and rep["search_ms"] > 0
    assert [built.index[k].ptr - built.arena.ptr for k in range(3)] == [0, built.plane_bytes, 2 * built.plane_bytes]
    assert built.plane_bytes == 64 * b.npix * 4
    rec_a = b.process(outputs=plain)
    rec_b = b.process(outputs=built)
    assert rec_a.tobytes() == rec_b.tobytes()
    # the planes, compared by a kernel (late - early over the whole ring): all zeros
    diff = _ffi.DeviceBuffer(64 * b.npix * 4)
    for k in range(3):
        _ffi.call("lars_d_diff_f32", C.c_void_p(plain.index[k].ptr), C.c_void_p(built.index[k].ptr), 64 * b.npix, C.c_void_p(diff.ptr), None)
        _ffi.call("lars_synchronize", None)
        assert not diff.download(np.float32, (64 * b.npix,)).any(), k
    diff.free(); plain.free(); built.free()
    free1 = C.c_size_t()
    _ffi.call("lars_mem_info", C.byref(free1), C.byref(total))
    assert free1.value >= free0.value - (64 << 20)                   # every candidate group went back to the driver
    with pytest.raises(_ffi.LarsError):
        lablib.assembled_outputs(b, ring=32)                         # not whole groups of 64 slots
    b.free()


def test_streaming_probe_and_allocation_kinds(lablib):
    from lars_image_processing_amd import _ffi
    nbytes = 96 << 20
    src = _ffi.DeviceBuffer(nbytes)
    src.zero()
    for kind, chunk in ((0, 0), (3, 32), (3, 0)):
        dst = lablib.LabBuffer(nbytes, kind=kind, chunk_mb=chunk)
        lablib.probe(2, 1, 4096, src.ptr, dst.ptr, nbytes - nbytes % 960)     # copy, 16 bytes per lane
        _ffi.call("lars_synchronize", None)
        dst.free()
    src.free()
