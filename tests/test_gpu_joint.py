"""The one-read statistics route (joint byte-pair histograms, csrc/joint.hip) against the per-pixel kernels, the oracle
and NumPy itself (needs a MI355X).  Everything it reports must be the same BITS as the classic route's records:
every statistic is a function of the pair counts."""
import itertools
import warnings

import numpy as np
import pytest

from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu
TYPES = ("NDVI", "GNDVI", "NDWI")


@pytest.fixture(scope="module")
def lars():
    import lars_image_processing_amd as mod
    from lars_image_processing_amd import _ffi
    assert _ffi.device_count() >= 1
    return mod


def same_records(a, b):
    """Records of two routes: everything but the sum of squares bit for bit.  That one is a float64 sum (order-dependent) that
    the per-pixel kernels round to 2^-32 once per workgroup and the one-read route once per tile: a few units of 2^-32 apart."""
    a, b = a.copy(), b.copy()
    np.testing.assert_allclose(a["sumsq"], b["sumsq"], rtol=1e-12, atol=2.0 ** -26)
    a["sumsq"] = b["sumsq"] = 0
    assert a.tobytes() == b.tobytes()


def subsets():
    for r in (1, 2, 3):
        yield from itertools.combinations(TYPES, r)


@pytest.fixture
def window(request):
    """lars_set_tuning("joint_window", ...) for the test's duration: 0 full tables only, 3 windowed tables for tiles of any size
    (csrc/joint_win.hip; the default, 1, needs 2^20 pixels per tile), 2 windows that miss on purpose (every tile is recounted), 4 three
    windows (NIR as well) wherever they fit, 5 the same with NIR windows that miss."""
    from lars_image_processing_amd import _ffi
    _ffi.set_tuning(joint_window=request.param)
    yield request.param
    _ffi.set_tuning(joint_window=1)


WINDOWS = pytest.mark.parametrize("window", [0, 3, 2, 4, 5], indirect=True)


@WINDOWS
@pytest.mark.parametrize("profile", ["uniform", "vegetation"])
@pytest.mark.parametrize("shape,ntiles", [((64, 64), 5), ((96, 160), 3), ((33, 35), 1), ((128, 130), 2)])
def test_joint_route_equals_classic_route(lars, profile, shape, ntiles, window):
    b = lars.TileBatch.synthetic(ntiles, shape[0], shape[1], seed=77, profile=profile)
    tiles = b.host_tiles()
    for wb in (True, False):
        for indices in subsets():
            rec_c, med_c = b.process(indices=indices, white_balance=wb, hist=True, sumsq=True, medians=True, route="classic")
            if wb:
                want_tab, want_pct, want_hist = b.host_tables(), b.host_percentiles(), b.host_hist()
                b.table.zero(); b.percentiles.zero(); b.hist.zero()
            # the channel histograms are a by-product only full tables can give: asked for with window 0, not with the others
            rec_j, med_j = b.process(indices=indices, white_balance=wb, hist=True, sumsq=True, medians=True, route="joint",
                                     channel_hist=window == 0)
            nwin, nrec = b.joint_window_report()
            two_streams = "NDVI" in indices and len(indices) > 1
            if window == 0 or not wb or not two_streams:
                assert (nwin, nrec) == (0, 0)
            elif window in (2, 5):
                assert nwin > 0 and nrec == nwin                 # one-row windows at the median: every windowed tile is counted again
            elif profile == "vegetation":
                assert (nwin, nrec) == (ntiles, 0)               # 96 + 128 values: both tables fit one workgroup
                assert b.joint_window_modes() == ((0, ntiles, 0) if window == 3 else (0, 0, ntiles))
            same_records(rec_c, rec_j)
            np.testing.assert_array_equal(med_c, med_j)
            # without the optional parts: sums of squares and bins stay zero
            rec_p = b.process(indices=indices, white_balance=wb, route="joint")
            assert (rec_p["sumsq"] == 0).all() and (rec_p["hist"] == 0).all()
            rec_p["hist"] = rec_j["hist"]; rec_p["sumsq"] = rec_j["sumsq"]
            assert rec_p.tobytes() == rec_j.tobytes()
            if wb:
                # the by-products: channel histograms, percentiles and tables of the channels the indices read
                chans = sorted({2} | ({0} if "NDVI" in indices else set()) | ({1} if set(indices) & {"GNDVI", "NDWI"} else set()))
                got_tab, got_pct = b.host_tables(partial=True), b.host_percentiles(partial=True)
                if window == 0:
                    got_hist = b.host_hist(partial=True)
                for c in chans:
                    if window == 0:
                        np.testing.assert_array_equal(got_hist[:, c], want_hist[:, c])
                    np.testing.assert_array_equal(got_pct[:, c], want_pct[:, c])
                    np.testing.assert_array_equal(got_tab[:, c], want_tab[:, c])
    # and against the oracle / NumPy directly (all three indices, white balance)
    rec, med = b.process(hist=True, medians=True, route="joint")
    for i in range(ntiles):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wbimg = orc.wb_app(tiles[i])
        for k, t in enumerate(TYPES):
            plane = orc.index_app(wbimg, t)
            part = orc.tile_partials(plane, t)
            assert float(rec[i, k]["sum"]) == part["sum"] and int(rec[i, k]["above"]) == part["above"]
            assert float(rec[i, k]["min"]) == float(plane.min()) and float(rec[i, k]["max"]) == float(plane.max())
            assert int(rec[i, k]["count"]) == plane.size
            np.testing.assert_array_equal(rec[i, k]["hist"], orc.hist50(plane))
            assert med[i, k] == float(np.median(plane)), (i, t)
    b.free()


def _tile_from(fn, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([fn(yy, xx, c).astype(np.uint8) for c in range(3)], axis=-1)


@WINDOWS
@pytest.mark.parametrize("blocks", [0, 1, 3])
def test_joint_counters_never_overflow(lars, blocks, window):
    """Tiles that pile hundreds of thousands of pixels onto single cells: the 16-bit counters are emptied onto the
    workgroup's list before they can wrap (constant tiles, two-level tiles, flat areas with a textured rim)."""
    from lars_image_processing_amd import _ffi
    h = w = 1024
    tiles = np.stack([
        np.full((h, w, 3), 0, np.uint8),                                               # nodata
        np.full((h, w, 3), 255, np.uint8),                                             # saturated
        _tile_from(lambda y, x, c: np.full_like(y, (17, 140, 201)[c]), h, w),          # one colour
        _tile_from(lambda y, x, c: np.where((x // 7 + y // 3) % 2 == 0, (10, 30, 250)[c], (200, 90, 40)[c]), h, w),
        _tile_from(lambda y, x, c: np.where(y < 900, (60, 70, 180)[c], (x * 7 + y * 13 + c * 31) % 256), h, w),
        orc.synth_tile_u8(5, 0, h, w, profile="vegetation"),
    ])
    b = lars.TileBatch.from_host(tiles)
    try:
        _ffi.set_tuning(blocks_per_tile=blocks)
        for wb in (False, True):
            rec_j, med_j = b.process(white_balance=wb, hist=True, medians=True, route="joint")
            _ffi.set_tuning(blocks_per_tile=0)
            rec_c, med_c = b.process(white_balance=wb, hist=True, medians=True, route="classic")
            _ffi.set_tuning(blocks_per_tile=blocks)
            assert rec_j.tobytes() == rec_c.tobytes()
            np.testing.assert_array_equal(med_j, med_c)
            for i in (0, 2, 3, 4):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    img = orc.wb_app(tiles[i]) if wb else tiles[i]
                for k, t in enumerate(TYPES):
                    plane = orc.index_app(img, t)
                    assert med_j[i, k] == float(np.median(plane)), (i, t, wb)
                    assert int(rec_j[i, k]["above"]) == orc.tile_partials(plane, t)["above"]
    finally:
        _ffi.set_tuning(blocks_per_tile=0)
        b.free()


def test_joint_full_size_tiles(lars):
    """4096 x 4096 tiles (BASELINE configs[1]/[2] size), one workgroup per tile and stream (2^24 pixels: the most a
    workgroup may count, a constant tile fills its list to the last entry) and the automatic split."""
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch.synthetic(6, 4096, 4096, seed=1234, profile="vegetation")
    const = np.empty((1, 4096, 4096, 3), np.uint8)
    const[...] = (9, 200, 77)
    b.tiles.upload(const, 5 * b.tile_bytes)                                          # tile 5: one colour
    rec_c, med_c = b.process(hist=True, medians=True, route="classic")
    try:
        for blocks in (1, 0, 5):
            for window in (1, 0, 2, 4, 5):
                _ffi.set_tuning(blocks_per_tile=blocks, joint_window=window)
                rec_j, med_j = b.process(hist=True, medians=True, route="joint")
                # the default: all six tiles on windowed tables (96 + 128 values; the one-colour tile: 3 + 3), nothing recounted;
                # 4 / 5: on three windows (NIR as well: sweeps and hand-over lists of that kernel at full size), none / all recounted
                assert b.joint_window_report() == {1: (6, 0), 0: (0, 0), 2: (6, 6), 4: (6, 0), 5: (6, 6)}[window]
                assert b.joint_window_modes() == {1: (0, 6, 0), 0: (6, 0, 0), 2: (0, 6, 0), 4: (0, 0, 6), 5: (0, 0, 6)}[window]
                assert rec_j.tobytes() == rec_c.tobytes(), (blocks, window)
                np.testing.assert_array_equal(med_j, med_c)
    finally:
        _ffi.set_tuning(blocks_per_tile=0, joint_window=1)
    n = 4096 * 4096
    assert (rec_j["count"] == n).all() and (rec_j["hist"].sum(axis=2) == n).all()
    tile = b.host_tiles(3, 1)[0]
    for c in range(3):
        assert b.host_percentiles()[3, c].tolist() == [float(v) for v in np.percentile(tile[:, :, c].astype(np.float32), (2, 98))]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wbimg = orc.wb_app(tile)
    for k, t in enumerate(TYPES):
        plane = orc.index_app(wbimg, t)
        assert med_j[3, k] == float(np.median(plane)) and float(rec_j[3, k]["sum"]) == orc.tile_partials(plane, t)["sum"]
    b.free()


def test_measured_route_choice(lars):
    """route="auto" without medians: both routes are timed once on the first tiles of a large batch and the faster one is
    remembered (flat or smooth imagery makes the LDS atomics of the one-read route queue up); whatever is chosen, the
    records are the same bytes."""
    from lars_image_processing_amd import batch as lb
    yy, xx = np.mgrid[0:1024, 0:1024]
    smooth = np.stack([np.clip(60 + 50 * c + 40 * np.sin(xx / 300.0 + c) + 30 * np.cos(yy / 200.0) + ((xx * 7 + yy * 3 + c) % 3), 0, 255)
                       for c in range(3)], axis=-1).astype(np.uint8)
    tiles = np.stack([np.roll(smooth, 17 * t, axis=1) for t in range(32)])
    for content in (tiles, None):
        b = lars.TileBatch.from_host(content) if content is not None else lars.TileBatch.synthetic(32, 1024, 1024, seed=3, profile="uniform")
        assert lb.get_stats_route() == "auto"
        assert b.pick_stats_route(("NDVI",)) == "joint" and not hasattr(b, "_route_ms")        # 2^25 pixels: too small to measure
        chosen = b.pick_stats_route(TYPES, min_pixels=0)
        assert chosen in ("joint", "classic") and b.pick_stats_route(TYPES) == chosen and set(b._route_ms) == {"joint", "classic"}
        rec = b.process()
        assert rec.tobytes() == b.process(route="joint").tobytes() == b.process(route="classic").tobytes()
        rec_m, med = b.process(medians=True)                        # medians: always the one-read route
        assert rec_m.tobytes() == rec.tobytes()
        b.free()
    small = lars.TileBatch.synthetic(3, 64, 64, seed=1)
    assert small.pick_stats_route(("NDVI",)) == "joint" and not hasattr(small, "_route_ms")
    small.free()


def test_joint_bad_arguments(lars):
    import ctypes as C
    from lars_image_processing_amd import _ffi
    lib = _ffi.load()
    b = lars.TileBatch.synthetic(2, 16, 16, seed=1)
    stats = b.new_stats()
    scratch = _ffi.DeviceBuffer(int(lib.lars_joint_scratch_bytes(2, 256, 7)))
    a = b.fused_args(("NDVI",), False, stats)
    assert lib.lars_d_stats_joint(C.byref(a), 0, 0, None, None, None, None, 0) == -1       # no scratch
    a.index_mask = 0
    assert lib.lars_d_stats_joint(C.byref(a), 0, 0, None, None, None, C.c_void_p(scratch.ptr), scratch.nbytes) == -1
    a.index_mask = 1
    a.dtype = _ffi.U16
    assert lib.lars_d_stats_joint(C.byref(a), 0, 0, None, None, None, C.c_void_p(scratch.ptr), scratch.nbytes) == -1
    a.dtype = _ffi.U8
    a.out_index[0] = scratch.ptr
    assert lib.lars_d_stats_joint(C.byref(a), 0, 0, None, None, None, C.c_void_p(scratch.ptr), scratch.nbytes) == -1
    a.out_index[0] = None
    assert lib.lars_d_stats_joint(C.byref(a), 0, 0, None, None, None, C.c_void_p(scratch.ptr), 1024) == -1         # scratch too small
    assert b"scratch holds 1024 bytes" in lib.lars_last_error()
    assert lib.lars_d_stats_joint(C.byref(a), 0, 0, None, None, None, C.c_void_p(scratch.ptr), scratch.nbytes) == 0
    _ffi.call("lars_synchronize", None)
    assert lib.lars_joint_scratch_bytes(0, 256, 7) == 0 and lib.lars_joint_scratch_bytes(2, 256, 0) == 0
    with pytest.raises(ValueError):
        wide = lars.TileBatch.from_host(np.zeros((1, 8, 8, 3), np.uint16))
        wide.process(route="joint")
    scratch.free(); stats.free(); b.free()


def test_route_timing_leaves_the_tables_alone_and_supplied_tables_are_used(lars):
    """(i) pick_stats_route() times both routes over the first tiles only -- on tables of its own; the batch's
    tables and their validity stay as they were, so a classic pass with recompute_tables=False afterwards is a normal pass;
    (ii) process(recompute_tables=False) uses the tables the caller computed (here process-rgn.py's flavour, rgn_variant=1)
    instead of silently taking the one-read route with flavour 0; (iii) the host getters refuse half-filled tables."""
    b = lars.TileBatch.synthetic(40, 1024, 1024, seed=31, profile="vegetation")
    want = b.process(route="classic")
    want_tab = b.host_tables()
    # (i) fresh batch state, route timing first
    b2 = lars.TileBatch.synthetic(40, 1024, 1024, seed=31, profile="vegetation")
    assert b2.pick_stats_route(("NDVI", "GNDVI", "NDWI"), min_pixels=0) in ("joint", "classic") and set(b2._route_ms) == {"joint", "classic"}
    assert b2.table is None and b2._table_channels == set()
    got = b2.process(recompute_tables=False, route="classic")          # no tables yet: they are computed, over ALL tiles
    assert got.tobytes() == want.tobytes()
    np.testing.assert_array_equal(b2.host_tables(), want_tab)
    b2.compute_wb_tables()
    b2._route_cache.clear()
    b2.pick_stats_route(("NDVI",), min_pixels=0)
    assert b2._table_channels == {0, 1, 2}
    np.testing.assert_array_equal(b2.host_tables(), want_tab)          # untouched by the timing
    # (ii) supplied tables of the other flavour are used, on every route setting that may honour them
    b2.compute_wb_tables(rgn_variant=1)
    tab1 = b2.host_tables()
    rec1 = b2.process(recompute_tables=False)                            # route from the module's setting ("auto")
    assert b2._rgn_variant == 1
    np.testing.assert_array_equal(b2.host_tables(), tab1)
    ref1 = b2.process(recompute_tables=False, route="classic")
    assert rec1.tobytes() == ref1.tobytes()
    rec1j = b2.process(rgn_variant=1, route="joint")                    # the one-read route computes the same flavour itself
    assert rec1j.tobytes() == ref1.tobytes()
    np.testing.assert_array_equal(b2.host_tables(), tab1)
    with pytest.raises(ValueError):
        b2.process(recompute_tables=False, route="joint")
    # (iii) a one-read pass over NDVI alone with another flavour leaves green invalid
    b2.process(indices=("NDVI",), route="joint")                         # flavour 0 again: red and NIR rows only
    assert b2._table_channels == {0, 2}
    with pytest.raises(RuntimeError):
        b2.host_tables()
    assert b2.host_tables(partial=True)[:, (0, 2)].tobytes() == want_tab[:, (0, 2)].tobytes()
    rec_g = b2.process(indices=("GNDVI",), recompute_tables=False)     # green is missing: tables are recomputed, not reused
    assert rec_g[:, 1].tobytes() == want[:, 1].tobytes()
    b.free(); b2.free()


def test_joint_rejects_tiles_beyond_its_counters(lars):
    from lars_image_processing_amd import _ffi
    import ctypes as C
    a = _ffi.FusedArgs()
    a.tiles, a.ntiles, a.npix, a.channels, a.dtype = 256, 1, 1 << 32, 3, _ffi.U8
    a.index_mask, a.flags, a.stats = 1, _ffi.F_STATS, 256
    with pytest.raises(_ffi.LarsError):
        _ffi.call("lars_d_stats_joint", C.byref(a), 1, 0, None, None, None, C.c_void_p(256), 1 << 40)


def _hot_tiles(kind, n, h, w, seed=3):
    """Content whose pixels pile onto few counter pairs: what the 16-bit hand-over has to survive."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    flat = t.reshape(n, h * w, 3)
    if kind == "three_colours_by_quad":                       # lanes differ (no flat wave), three cells take everything
        cols = np.array([(10, 200, 90), (11, 201, 91), (250, 3, 128)], np.uint8)
        flat[:] = cols[(np.arange(h * w) // 4) % 3][None]
    elif kind == "mostly_one_colour":                          # one hot cell under adds from every lane, the rest scattered noise
        mask = rng.random((n, h * w)) < 0.9
        flat[mask] = (77, 140, 31)
    elif kind == "two_colours_by_pixel":                       # every lane holds the same quad: the flat path's counted adds cross the mark
        cols = np.array([(5, 6, 7), (200, 100, 50)], np.uint8)
        flat[:] = cols[np.arange(h * w) % 2][None]
    elif kind == "one_colour":
        flat[:] = (255, 0, 255)
    return t


@WINDOWS
@pytest.mark.parametrize("kind", ["iid", "three_colours_by_quad", "mostly_one_colour", "two_colours_by_pixel", "one_colour"])
def test_hot_cells_survive_the_16_bit_counters(lars, kind, window):
    """Content whose pixels pile onto a few counter pairs moves those pairs onto the hand-over list thousands of times (plain
    adds, flat-wave adds of up to 256 at once, one and two streams, one workgroup per tile and several): records, medians
    and tables stay identical to the per-pixel route."""
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch.from_host(_hot_tiles(kind, 3, 1024, 1024))
    try:
        for indices in (("NDVI",), ("GNDVI", "NDWI"), TYPES):
            want, want_med = b.process(indices=indices, hist=True, medians=True, route="classic")
            want_tab = b.host_tables()
            for blocks, depth in ((0, 6), (1, 6), (0, 12), (3, 8)):
                _ffi.set_tuning(blocks_per_tile=blocks, joint_depth=depth, joint_win_depth={6: 15, 12: 12, 8: 5}[depth] if blocks else 4)
                got, got_med = b.process(indices=indices, hist=True, medians=True, route="joint")
                assert got.tobytes() == want.tobytes(), (kind, indices, blocks, depth)
                np.testing.assert_array_equal(got_med, want_med)
                for c in sorted(lars.batch.channels_of(indices)):
                    np.testing.assert_array_equal(b.host_tables(partial=True)[:, c], want_tab[:, c])
    finally:
        _ffi.set_tuning(blocks_per_tile=0, joint_depth=6, joint_win_depth=15)
        b.free()


def _narrow(rng, h, w, lo, hi, ch=3):
    return rng.integers(lo, hi, (h, w, ch), dtype=np.uint8)


@pytest.mark.parametrize("channels", [3, 4])
def test_windowed_and_full_tiles_in_one_batch(lars, channels):
    """A batch in which some tiles get windows (narrow red and green) and others do not (full-range samples): each tile takes its own
    way through ONE call -- k_joint_count_win for the first kind, k_joint_count for the second -- RGB and RGBA."""
    from lars_image_processing_amd import _ffi
    rng = np.random.default_rng(11)
    h, w = 512, 640
    tiles = np.stack([
        _narrow(rng, h, w, 40, 160, channels),                                          # 120 + 120 rows: fits
        rng.integers(0, 256, (h, w, channels), dtype=np.uint8),                          # 250 + 250: does not
        np.clip(rng.normal(90, 12, (h, w, channels)), 0, 255).astype(np.uint8),          # bell-shaped: fits
        np.concatenate([_narrow(rng, h, w // 2, 0, 256, channels), _narrow(rng, h, w // 2, 100, 110, channels)], axis=1),   # wide half
        _narrow(rng, h, w, 250, 256, channels),                                          # up against the range's end
        _narrow(rng, h, w, 0, 3, channels),                                              # and its start
    ])
    tiles[2, :, :, 2] = rng.integers(0, 256, (h, w), dtype=np.uint8)                     # NIR keeps all 256 values either way
    b = lars.TileBatch.from_host(tiles)
    try:
        want, want_med = b.process(hist=True, medians=True, route="classic")
        want_tab, want_pct = b.host_tables(), b.host_percentiles()
        for window, expect in ((3, (4, 0)), (2, None), (0, (0, 0)), (4, None), (5, None)):
            for blocks in (0, 1, 3):
                _ffi.set_tuning(joint_window=window, blocks_per_tile=blocks)
                got, got_med = b.process(hist=True, medians=True, route="joint")
                report = b.joint_window_report()
                assert expect is None or report == expect, (window, report)
                if window in (2, 5):
                    assert report[0] >= 4 and report[1] == report[0]
                if window == 4:
                    assert report[0] >= 4 and report[1] == 0
                assert got.tobytes() == want.tobytes(), (window, blocks)
                np.testing.assert_array_equal(got_med, want_med)
                np.testing.assert_array_equal(b.host_tables(), want_tab)
                assert b.host_percentiles().tobytes() == want_pct.tobytes()
    finally:
        _ffi.set_tuning(joint_window=1, blocks_per_tile=0)
        b.free()


def test_windowed_ragged_single_tile_and_flavours(lars):
    """One tile whose pixel count is not a multiple of four (the tail pixels take the scalar path of the windowed kernel), the
    process-rgn.py flavour of the tables, the sum of squares and the 50 bins -- windowed against full tables against per pixel."""
    from lars_image_processing_amd import _ffi
    rng = np.random.default_rng(5)
    tile = np.clip(rng.normal(120, 20, (1, 333, 335, 3)), 0, 255).astype(np.uint8)
    tile[0, -1, -3:, :] = (0, 255, 7)                                                     # the tail: values outside every window
    b = lars.TileBatch.from_host(tile)
    try:
        for variant in (0, 1):
            b.compute_wb_tables(rgn_variant=variant)
            want = b.process(hist=True, sumsq=True, recompute_tables=False, route="classic")
            want_tab = b.host_tables()
            for window in (3, 0, 2, 4, 5):
                _ffi.set_tuning(joint_window=window)
                got = b.process(hist=True, sumsq=True, rgn_variant=variant, route="joint")
                assert b.joint_window_report() == {3: (1, 0), 0: (0, 0), 2: (1, 1), 4: (1, 0), 5: (1, 1)}[window]
                assert b.joint_window_modes() == {3: (0, 1, 0), 0: (1, 0, 0), 2: (0, 1, 0), 4: (0, 0, 1), 5: (0, 0, 1)}[window]
                same_records(want, got)
                np.testing.assert_array_equal(b.host_tables(), want_tab)
    finally:
        _ffi.set_tuning(joint_window=1)
        b.free()


def test_three_windows_where_two_do_not_fit(lars):
    """Red and green spanning 170 values each need 344 table rows -- more than one workgroup holds with NIR whole -- but NIR itself spans
    150: rows of 90 dwords instead of 133 let all three windows share one workgroup (JointWin mode 2).  NIR clamps like the other two;
    the finish kernel checks its percentiles against its window too.  Tiles with saturated ends, a ragged single tile, RGBA."""
    from lars_image_processing_amd import _ffi
    rng = np.random.default_rng(21)

    def tiles_of(n, h, w, ch):
        t = np.empty((n, h, w, ch), np.uint8)
        t[..., 0] = rng.integers(20, 190, (n, h, w))
        t[..., 1] = rng.integers(40, 210, (n, h, w))
        t[..., 2] = rng.integers(60, 210, (n, h, w))
        if ch == 4:
            t[..., 3] = 255
        t[:, : max(1, h // 128), :, :3] = 255                                             # 0.6-0.8 % of overexposed rows: above every window, and
                                                                                          # enough to push the safe windows' upper ends to 255
        t[:, -1, : w // 3, :3] = 0
        return t

    for shape, ch in (((3, 256, 512), 3), ((1, 333, 335), 3), ((2, 128, 256), 4)):
        b = lars.TileBatch.from_host(tiles_of(shape[0], shape[1], shape[2], ch))
        try:
            want, want_med = b.process(hist=True, medians=True, route="classic")
            want_tab, want_pct = b.host_tables(), b.host_percentiles()
            for window, modes, recounted in ((3, (0, 0, shape[0]), 0), (5, (0, 0, shape[0]), shape[0]), (0, (shape[0], 0, 0), 0)):
                for blocks in (0, 3):
                    _ffi.set_tuning(joint_window=window, blocks_per_tile=blocks)
                    got, got_med = b.process(hist=True, medians=True, route="joint")
                    assert b.joint_window_modes() == modes and b.joint_window_report()[1] == recounted, (shape, window)
                    assert got.tobytes() == want.tobytes(), (shape, window, blocks)
                    np.testing.assert_array_equal(got_med, want_med)
                    np.testing.assert_array_equal(b.host_tables(), want_tab)
                    assert b.host_percentiles().tobytes() == want_pct.tobytes()
        finally:
            _ffi.set_tuning(joint_window=1, blocks_per_tile=0)
            b.free()


def test_window_report_and_channel_histograms(lars):
    """Asking for the channel histograms keeps the full tables (clamped counts cannot give them); not asking leaves host_hist()
    invalid instead of handing out clamped histograms."""
    b = lars.TileBatch.synthetic(2, 1024, 1024, seed=9, profile="vegetation")
    stats = b.new_stats()
    try:
        b.run_joint(TYPES, True, stats)                                                    # 2^20 pixels: windowed by default
        b.check_joint()
        assert b.joint_window_report() == (2, 0)
        with pytest.raises(RuntimeError):
            b.host_hist()
        rec_w = stats.download(lars.batch.STATS_DTYPE, (2, 3)).copy()
        b.run_joint(TYPES, True, stats, channel_hist=True)
        b.check_joint()
        assert b.joint_window_report() == (0, 0)
        assert stats.download(lars.batch.STATS_DTYPE, (2, 3)).tobytes() == rec_w.tobytes()
        hist = b.host_hist()
        tiles = b.host_tiles()
        for i in range(2):
            for c in range(3):
                np.testing.assert_array_equal(hist[i, c], np.bincount(tiles[i, :, :, c].ravel(), minlength=256))
    finally:
        stats.free()
        b.free()


@pytest.mark.parametrize("channels", [3, 4])
def test_plane_writing_calls_that_take_the_one_read_statistics(lars, channels):
    """process(outputs=...) sends three kinds of plane-writing calls through the one-read statistics pass followed by a fused launch
    WITHOUT statistics (batch.py): planes of one value stream, any planes with medians, any planes with the 50-bin histograms (they
    fall out of the counted cells).  Records, planes and medians against
    route="classic" -- packed outputs and a ring, RGB and RGBA; the optional sum of squares within its documented few units of
    2^-32; and the tables the batch holds afterwards cover only the channels the indices read."""
    rng = np.random.default_rng(21)
    tiles = np.clip(rng.normal((90, 120, 150, 255)[:channels], (30, 25, 40, 0)[:channels], (5, 192, 256, channels)), 0, 255).astype(np.uint8)
    b = lars.TileBatch.from_host(tiles)
    try:
        for ring in (None, 2):
            for indices, kw in ((("NDVI",), dict(hist=True, sumsq=True)), (TYPES, dict(medians=True)), (("GNDVI", "NDWI"), dict(hist=True)),
                                (TYPES, dict(hist=True)), (TYPES, dict(hist=True, medians=True, sumsq=True))):
                oc = b.make_outputs(indices=indices, index=True, ring=ring, arena="plain")
                oj = b.make_outputs(indices=indices, index=True, ring=ring, arena="plain")
                got_c = b.process(indices=indices, outputs=oc, route="classic", **kw)
                got_j = b.process(indices=indices, outputs=oj, **kw)                       # the default route
                rec_c, med_c = got_c if kw.get("medians") else (got_c, None)
                rec_j, med_j = got_j if kw.get("medians") else (got_j, None)
                same_records(rec_c, rec_j)
                if med_c is not None:
                    np.testing.assert_array_equal(med_c, med_j)
                for t in indices:
                    pc, pj = oc.host_index(t, 0, oc.slots), oj.host_index(t, 0, oj.slots)
                    assert pc.view(np.uint32).tobytes() == pj.view(np.uint32).tobytes(), (ring, indices, t)
                assert b.host_tables().shape == (5, 3, 256)            # the classic pass before left all three channels' rows valid
                oc.free(); oj.free()
        # on a batch that has seen nothing else, such a call leaves only the rows of the channels its indices read
        fresh = lars.TileBatch.from_host(tiles)
        of = fresh.make_outputs(indices=("NDVI",), index=True, arena="plain")
        fresh.process(indices=("NDVI",), outputs=of)
        with pytest.raises(RuntimeError):
            fresh.host_tables()
        assert fresh._table_channels == {0, 2} and fresh.host_tables(partial=True).shape == (5, 3, 256)
        of.free(); fresh.free()
    finally:
        b.free()


def test_tuning_rejects_values_the_kernels_do_not_have(lars):
    from lars_image_processing_amd import _ffi
    for key, bad in (("joint_depth", 5), ("joint_win_depth", 8), ("joint_window", 6), ("out_stride_planes", 3)):
        with pytest.raises(_ffi.LarsError):
            _ffi.set_tuning(**{key: bad})
    assert _ffi.get_tuning("joint_depth") == 6 and _ffi.get_tuning("joint_win_depth") == 15 and _ffi.get_tuning("joint_window") == 1


def _sampled_pixels(tile_index, npix):
    """The pixels k_joint_predict samples of tile `tile_index` of a call (csrc/joint_win.hip): 1024 segments of 16 quads, segment i
    somewhere inside its own stretch of nquads / 1024 quads, placed by the counter hash the synthetic tiles use."""
    nquads = npix >> 2
    stretch = nquads // 1024
    span = stretch - 15
    i = np.arange(1024, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = orc._mix32(((i + np.uint64((tile_index * 0x9E3779B9) & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)).astype(np.uint64)
    start = i * np.uint64(stretch) + ((h * np.uint64(span)) >> np.uint64(32))
    quads = (start[:, None] + np.arange(16, dtype=np.uint64)[None, :]).ravel()
    return (quads[:, None] * np.uint64(4) + np.arange(4, dtype=np.uint64)[None, :]).ravel().astype(np.int64)


def test_a_window_that_misses_by_itself_is_recounted(lars):
    """No test hook: a tile whose darkest 3 % of red samples all lie where the prediction's subsample does not look.  The sample sees
    none of them, the red window starts above the tile's 2nd percentile, k_joint_finish finds the percentile's order statistics on
    the window's edge, flags the tile, and the recount on full tables delivers what the per-pixel route delivers.  The neighbouring
    tile (same values, dark pixels spread evenly) keeps its window."""
    h = w = 1024
    npix = h * w
    rng = np.random.default_rng(17)
    tiles = rng.integers(100, 160, (2, npix, 3), dtype=np.uint8)
    dark = rng.integers(5, 30, (2, npix), dtype=np.uint8)
    seen = np.zeros(npix, bool)
    seen[_sampled_pixels(0, npix)] = True
    hidden = np.flatnonzero(~seen)[: int(0.03 * npix)]                   # 3 % of the tile, none of it sampled (the sample is 1 / 16)
    tiles[0, hidden, 0] = dark[0, hidden]
    even = np.arange(0, npix, 33)[: int(0.03 * npix)]
    tiles[1, even, 0] = dark[1, even]
    b = lars.TileBatch.from_host(tiles.reshape(2, h, w, 3))
    try:
        want, want_med = b.process(hist=True, medians=True, route="classic")
        want_tab, want_pct = b.host_tables(), b.host_percentiles()
        assert want_pct[0, 0, 0] < 30 and want_pct[1, 0, 0] < 30            # the 2nd percentile of red lies among the dark pixels
        got, got_med = b.process(hist=True, medians=True, route="joint")
        assert b.joint_window_report() == (2, 1)                             # both tiles windowed, the first one counted again
        assert got.tobytes() == want.tobytes()
        np.testing.assert_array_equal(got_med, want_med)
        np.testing.assert_array_equal(b.host_tables(), want_tab)
        assert b.host_percentiles().tobytes() == want_pct.tobytes()
    finally:
        b.free()
