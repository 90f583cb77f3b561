"""The laboratory library (liblars_lab.so, include/lars_lab.h): experiments that are not part of the product stay
buildable and correct -- the persistent one-launch pipeline and the streaming probes (needs a MI355X)."""
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



def test_streaming_probe_and_allocation_kinds(lablib):
    from lars_image_processing_amd import _ffi
    nbytes = 96 << 20
    src = _ffi.DeviceBuffer(nbytes)
    src.zero()
    dst = lablib.LabBuffer(nbytes, kind=0)
    lablib.probe(2, 1, 4096, src.ptr, dst.ptr, nbytes - nbytes % 960)         # copy, 16 bytes per lane
    lablib.probe(32, 1, 64, src.ptr, None, nbytes)                            # two readers per chunk, 12-byte loads
    lablib.probe(43, 1, 64, src.ptr, None, nbytes)                            # three readers, 16-byte loads
    _ffi.call("lars_synchronize", None)
    assert not dst.download(np.uint8, (1 << 20,)).any()                        # the copy of a zeroed buffer
    dst.free(); src.free()
