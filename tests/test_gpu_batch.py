"""The batched, device-resident tile path against the oracle (needs a MI355X)."""
import warnings

import numpy as np
import pytest

from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu
TYPES = ("NDVI", "GNDVI", "NDWI")


@pytest.fixture(scope="module", params=[(2, "auto"), (2, "classic"), (1, "classic")], ids=["impl2-joint", "impl2-classic", "impl1-classic"])
def lars(request):
    """Every batch test runs against both kernel generations and both statistics routes (one read through joint
    byte-pair histograms, or histogram pass + per-pixel kernel): results must not depend on either."""
    import lars_image_processing_amd as mod
    from lars_image_processing_amd import _ffi, batch
    assert _ffi.device_count() >= 1
    impl, route = request.param
    _ffi.set_tuning(fused_impl=impl, hist_impl=impl)
    batch.set_stats_route(route)
    yield mod
    _ffi.set_tuning(fused_impl=0, hist_impl=2)
    batch.set_stats_route("auto")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("profile", ["uniform", "vegetation"])
@pytest.mark.parametrize("shape", [(64, 64), (30, 50), (33, 35)])
def test_device_generator_equals_host_generator(lars, profile, shape):
    b = lars.TileBatch.synthetic(5, shape[0], shape[1], seed=1234, profile=profile, first_tile=7)
    got = b.host_tiles()
    for i in range(5):
        np.testing.assert_array_equal(got[i], orc.synth_tile_u8(1234, 7 + i, shape[0], shape[1], profile=profile))
    b.free()


@pytest.mark.parametrize("profile", ["uniform", "vegetation"])
@pytest.mark.parametrize("shape", [(64, 64), (96, 160), (33, 35)])
def test_batch_matches_oracle_per_tile(lars, profile, shape):
    from lars_image_processing_amd import batch as lb
    ntiles = 6
    b = lars.TileBatch.synthetic(ntiles, shape[0], shape[1], seed=42, profile=profile)
    tiles = b.host_tiles()
    outs = b.make_outputs(index=True, wb=True, rgba=True)
    rec = b.process(hist=True, sumsq=True, outputs=outs)
    tables = b.host_tables()
    pcts = b.host_percentiles()
    wb = outs.host_wb(0, ntiles)
    for i in range(ntiles):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_wb = orc.wb_app(tiles[i])
        np.testing.assert_array_equal(wb[i], want_wb)
        for c in range(3):
            hist = np.bincount(tiles[i][:, :, c].ravel(), minlength=256)
            lo, hi = orc.percentile_from_hist(hist, 2), orc.percentile_from_hist(hist, 98)
            assert pcts[i, c, 0] == lo and pcts[i, c, 1] == hi
            present = hist > 0
            np.testing.assert_array_equal(tables[i, c][present], orc.wb_lut_from_percentiles(lo, hi)[present])
        for k, t in enumerate(TYPES):
            want = orc.index_app(want_wb, t)
            np.testing.assert_array_equal(bits(outs.host_index(t, i, 1)[0]), bits(want))
            part = orc.tile_partials(want, t)
            r = rec[i, k]
            assert int(r["count"]) == part["count"] and int(r["above"]) == part["above"]
            assert float(r["min"]) == part["min"] and float(r["max"]) == part["max"]
            assert float(r["sum"]) == part["sum"]                      # exact fixed-point sum
            assert float(r["sumsq"]) == pytest.approx(part["sumsq"], rel=1e-9)   # filled because sumsq=True
            np.testing.assert_array_equal(np.array(r["hist"], dtype=np.int64), part["hist"])
            lut = lars.colormap_lut("RdYlBu" if t == "NDWI" else "RdYlGn")
            np.testing.assert_array_equal(outs.host_rgba(t, i, 1)[0], orc.colormap_closed_form(want, lut))
    # global statistics == statistics of the union
    for k, t in enumerate(TYPES):
        merged = lb.summarize(lb.merge_records(rec[:, k]))
        parts = []
        for i in range(ntiles):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                parts.append(orc.tile_partials(orc.index_app(orc.wb_app(tiles[i]), t), t))
        want = orc.merge_partials(parts)
        assert merged["count"] == want["count"] and merged["coverage"] == want["coverage"]
        assert merged["min"] == want["min"] and merged["max"] == want["max"]
        assert merged["mean"] == pytest.approx(want["mean"], rel=1e-12, abs=1e-15)
        np.testing.assert_array_equal(merged["hist"], want["hist"])
    outs.free()
    b.free()


def test_one_rank_share_of_the_sharded_batch(lars):
    """BASELINE configs[3]: 16384 tiles over 8 GPUs = 2048 tiles of 4096 x 4096 per rank (96 GiB of samples), statistics only.
    Size-independent identities over all records, equality of a few tiles with their single-tile runs (same global tile
    index, hence same samples), and the fold."""
    from lars_image_processing_amd import batch as lb
    import ctypes as C
    from lars_image_processing_amd import _ffi
    free_b, total_b = C.c_size_t(0), C.c_size_t(0)
    _ffi.call("lars_mem_info", C.byref(free_b), C.byref(total_b))
    # no skip: this is the only test at BASELINE configs[3]'s per-rank size, and the MI355X it is written for has 288 GB
    assert free_b.value >= 120 * 2**30, f"only {free_b.value >> 30} GiB of device memory free: 96 GiB of tiles do not fit"
    n, ntiles, first = 4096 * 4096, 2048, 3 * 2048                       # the share of rank 3
    b = lars.TileBatch.synthetic(ntiles, 4096, 4096, seed=1234, profile="vegetation", first_tile=first)
    rec = b.process(hist=True)
    assert rec.shape == (ntiles, 3)
    assert (rec["count"] == n).all() and (rec["hist"].sum(axis=2) == n).all()
    assert (rec["min"] >= -1).all() and (rec["max"] <= 1).all() and (rec["min"] <= rec["max"]).all()
    assert (rec["sum"][:, 2] == -rec["sum"][:, 1]).all()                    # NDWI == -GNDVI
    assert (rec["min"][:, 2] == -rec["max"][:, 1]).all() and (rec["max"][:, 2] == -rec["min"][:, 1]).all()
    for j in (0, 1023, 2047):
        one = lars.TileBatch.synthetic(1, 4096, 4096, seed=1234, profile="vegetation", first_tile=first + j)
        assert one.process(hist=True)[0].tobytes() == rec[j].tobytes(), j
        one.free()
    glob = lb.local_fold(rec)
    for k in range(3):
        g = lb.summarize(glob[k])
        assert g["count"] == ntiles * n and int(np.sum(g["hist"])) == ntiles * n
        assert g["min"] == float(rec["min"][:, k].min()) and g["max"] == float(rec["max"][:, k].max())
        assert abs(g["mean"] - float(rec["sum"][:, k].sum()) / (ntiles * n)) <= 1e-12
    b.free()


@pytest.mark.parametrize("ntiles", [1, 37, 300])
def test_device_fold_equals_host_fold(lars, ntiles):
    """lars_d_stats_fold: the per-index fold of a batch's per-tile records on the device is lars_stats_merge over the
    tiles in tile order, bit for bit (sums added in the same order); indices outside the mask stay untouched."""
    from lars_image_processing_amd import _ffi, batch as lb
    b = lars.TileBatch.synthetic(ntiles, 32, 48, seed=5, profile="vegetation")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    b.run_fused(b.fused_args(TYPES, True, stats, True, sumsq=True))
    for indices in (TYPES, ("NDVI", "NDWI"), ("GNDVI",)):
        folded = b.fold_stats(stats, indices)
        _ffi.call("lars_synchronize", None)
        got = folded.download(_ffi.STATS_DTYPE, (3,))
        want = lb.local_fold(stats.download(_ffi.STATS_DTYPE, (ntiles, 3)), indices)
        assert got.tobytes() == want.tobytes(), indices
        folded.free()
    stats.free()
    b.free()


def test_stats_only_equals_stats_with_outputs_and_ring(lars):
    b = lars.TileBatch.synthetic(8, 128, 128, seed=7, profile="vegetation")
    rec_a = b.process(hist=True, sumsq=True)
    outs = b.make_outputs(index=True, ring=3)
    rec_b = b.process(hist=True, sumsq=True, outputs=outs)
    assert (rec_a["sumsq"] > 0).all() and (b.process(hist=True)["sumsq"] == 0).all()      # only on request
    np.testing.assert_allclose(rec_a["sumsq"], rec_b["sumsq"], rtol=1e-12, atol=2.0 ** -26)   # rounded to 2^-32 once per workgroup
    rec_a["sumsq"] = rec_b["sumsq"] = 0     # every other field is order-independent, hence identical
    assert rec_a.tobytes() == rec_b.tobytes()
    # the ring holds the last chunk: tiles 6, 7 in slots 0, 1
    tiles = b.host_tiles()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.index_app(orc.wb_app(tiles[7]), "NDVI")
    np.testing.assert_array_equal(bits(outs.host_index("NDVI", 1, 1)[0]), bits(want))
    # no white balance: indices of the raw samples
    rec_raw = b.process(white_balance=False)
    for i in (0, 5):
        part = orc.tile_partials(orc.index_app(tiles[i], "GNDVI"), "GNDVI")
        assert float(rec_raw[i, 1]["sum"]) == part["sum"] and int(rec_raw[i, 1]["above"]) == part["above"]
    outs.free()
    b.free()


def test_single_index_masks_leave_other_records_untouched(lars):
    b = lars.TileBatch.synthetic(3, 64, 64, seed=3)
    tiles = b.host_tiles()
    for k, t in enumerate(TYPES):
        rec = b.process(indices=(t,), white_balance=False)
        for j in range(3):
            if j != k:
                assert rec[:, j].tobytes() == bytes(rec[:, j].nbytes)
        part = orc.tile_partials(orc.index_app(tiles[1], t), t)
        assert float(rec[1, k]["sum"]) == part["sum"] and float(rec[1, k]["min"]) == part["min"]
    rec2 = b.process(indices=("NDVI", "NDWI"), white_balance=False)       # two-index mask -> generic kernel
    assert rec2[:, 1].tobytes() == bytes(rec2[:, 1].nbytes)
    part = orc.tile_partials(orc.index_app(tiles[2], "NDWI"), "NDWI")
    assert float(rec2[2, 2]["sum"]) == part["sum"] and int(rec2[2, 2]["above"]) == part["above"]
    b.free()


def test_full_size_tile_properties(lars):
    """4096x4096 (BASELINE config size): size-independent properties + one oracle spot check."""
    from lars_image_processing_amd import batch as lb
    b = lars.TileBatch.synthetic(2, 4096, 4096, seed=1234, profile="vegetation")
    outs = b.make_outputs(index=True, wb=True)
    rec = b.process(hist=True, outputs=outs)
    n = 4096 * 4096
    for i in range(2):
        for k in range(3):
            r = rec[i, k]
            assert int(r["count"]) == n and int(np.sum(r["hist"])) == n
            assert -1.0 <= float(r["min"]) <= float(r["max"]) <= 1.0
        # NDWI == -GNDVI: sums negate exactly, extrema swap
        assert float(rec[i, 2]["sum"]) == -float(rec[i, 1]["sum"])
        assert float(rec[i, 2]["min"]) == -float(rec[i, 1]["max"]) and float(rec[i, 2]["max"]) == -float(rec[i, 1]["min"])
    ndvi = outs.host_index("NDVI", 1, 1)[0]
    wb = outs.host_wb(1, 1)[0]
    exact = int((ndvi.astype(np.float64) * 2.0 ** 32).astype(np.int64).sum())      # samples are multiples of 2^-32
    assert float(rec[1, 0]["sum"]) == float(exact) / 2.0 ** 32
    assert float(rec[1, 0]["min"]) == float(ndvi.min()) and int(rec[1, 0]["above"]) == int((ndvi > np.float32(0.2)).sum())
    np.testing.assert_array_equal(bits(ndvi), bits(orc.index_closed_form(wb[:, :, 2], wb[:, :, 0])))
    # white balance is idempotent on its table: every output value is a table value of its input
    tiles1 = b.host_tiles(1, 1)[0]
    tab = b.host_tables()[1]
    for c in range(3):
        np.testing.assert_array_equal(wb[:, :, c], tab[c][tiles1[:, :, c]])
    # a 1024-row strip against the reference statement
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_wb = orc.wb_app(tiles1)
    np.testing.assert_array_equal(wb, want_wb)
    assert lb.summarize(lb.merge_records(rec[:, 0]))["count"] == 2 * n
    # exact medians at full size, none of them through a stored plane: per tile and over both tiles
    rec_m, med = b.process(medians=True)
    assert med[1, 0] == float(np.median(ndvi))
    assert float(rec_m[1, 0]["sum"]) == float(rec[1, 0]["sum"]) and int(rec_m[1, 0]["above"]) == int(rec[1, 0]["above"])
    both = np.concatenate([outs.host_index("NDVI", 0, 1)[0].ravel(), ndvi.ravel()])
    assert b.global_medians(("NDVI",))["NDVI"] == float(np.median(both))
    outs.free()
    b.free()


def test_quotient_selfcheck_exhaustive(lars):
    """rcp + mul + 2 fma == IEEE float32 division for EVERY operand pair of the uint8 (den <= 510)
    and uint16 (den <= 131070) domains, checked on the device itself (1.7e10 pairs)."""
    import ctypes as C
    from lars_image_processing_amd import _ffi
    for max_den in (510, 131070):
        bad = C.c_uint64(123)
        first = (C.c_uint32 * 2)()
        _ffi.call("lars_d_quot_selfcheck", max_den, C.byref(bad), C.byref(first))
        assert bad.value == 0, (max_den, bad.value, first[0], C.c_int32(first[1]).value)


def test_tuning_does_not_change_results(lars):
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch.synthetic(4, 200, 300, seed=11, profile="vegetation")
    outs = b.make_outputs(index=True)
    ref = None
    keep = (_ffi.get_tuning("fused_impl"), _ffi.get_tuning("hist_impl"))
    for impl in (1, 2):
        for nt in (0, 1):
            for bpt in (0, 1, 7):
                _ffi.set_tuning(fused_impl=impl, hist_impl=impl, nt_stores=nt, blocks_per_tile=bpt)
                rec = b.process(hist=True, sumsq=True, outputs=outs)
                sumsq = rec["sumsq"].copy()
                rec["sumsq"] = 0            # the only order-dependent field (double sums of squares)
                got = (rec.tobytes(), outs.host_index("NDWI", 0, 4).tobytes(), b.host_tables().tobytes())
                if ref is None:
                    ref, ref_sumsq = got, sumsq
                assert got == ref, (impl, nt, bpt)
                np.testing.assert_allclose(sumsq, ref_sumsq, rtol=1e-12)
    _ffi.set_tuning(fused_impl=keep[0], hist_impl=keep[1], nt_stores=0, blocks_per_tile=0)
    outs.free()
    b.free()


def test_rccl_communicator_single_rank(lars):
    """librccl loads, a 1-rank communicator initialises, and the stats all-gather + fold and the
    f64 all-reduce round-trip through device memory (the N>1 code path with N = 1)."""
    from lars_image_processing_amd import _ffi, dist, batch as lb
    uid = dist._rccl_unique_id()
    assert len(uid) == _ffi.COMM_ID_BYTES and any(uid)
    comm = dist.Comm(0, 1, uid)
    b = lars.TileBatch.synthetic(3, 64, 64, seed=5)
    rec = b.process(hist=True)
    local = lb.local_fold(rec)
    glob = comm.allreduce_stats(local)
    assert glob.tobytes() == local.tobytes()
    assert comm.allreduce_f64([1.5, -2.0], "max").tolist() == [1.5, -2.0]
    assert comm.allreduce_f64([1.5, -2.0], "sum").tolist() == [1.5, -2.0]
    comm.barrier()
    comm.destroy()
    b.free()


@pytest.mark.parametrize("kind", ["full16", "12bit", "narrow", "constant_channel", "odd_shape"])
def test_uint16_batch_against_oracle(lars, kind):
    """uint16 tiles: two-level radix percentiles, threshold-table white balance, fast 24-byte-per-lane kernel."""
    import ctypes as C
    from lars_image_processing_amd import _ffi
    rng = np.random.default_rng({"full16": 1, "12bit": 2, "narrow": 3, "constant_channel": 4, "odd_shape": 5}[kind])
    h, w = (61, 67) if kind == "odd_shape" else (96, 128)
    n = 3
    if kind == "full16":
        tiles = rng.integers(0, 65536, (n, h, w, 3), dtype=np.uint16)
    elif kind == "12bit":
        tiles = rng.integers(0, 4096, (n, h, w, 3), dtype=np.uint16)
    elif kind == "narrow":                      # few distinct values: fractional percentiles, steep staircase
        tiles = (rng.integers(0, 4, (n, h, w, 3)) * 257 + 30000).astype(np.uint16)
    elif kind == "constant_channel":
        tiles = rng.integers(0, 65536, (n, h, w, 3), dtype=np.uint16)
        tiles[..., 1] = 1234
    else:
        tiles = rng.integers(100, 60000, (n, h, w, 3), dtype=np.uint16)
    b = lars.TileBatch.from_host(tiles)
    outs = b.make_outputs(index=True, wb=True, rgba=True)
    rec = b.process(hist=True, outputs=outs)
    pcts = b.host_percentiles()
    tables = b.host_tables()
    wb = outs.host_wb(0, n)
    for i in range(n):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_wb = orc.wb_app(tiles[i])
        np.testing.assert_array_equal(wb[i], want_wb)
        for c in range(3):
            plane = tiles[i][:, :, c]
            lo, hi = np.percentile(plane.astype(np.float32), (2, 98))
            assert pcts[i, c, 0] == lo and pcts[i, c, 1] == hi
            present = np.unique(plane)
            np.testing.assert_array_equal(tables[i, c][present], orc.wb_lut_from_percentiles(lo, hi, 65536)[present])
        for k, t in enumerate(TYPES):
            want = orc.index_app(want_wb, t)
            np.testing.assert_array_equal(bits(outs.host_index(t, i, 1)[0]), bits(want))
            part = orc.tile_partials(want, t)
            r = rec[i, k]
            assert int(r["above"]) == part["above"] and float(r["min"]) == part["min"] and float(r["max"]) == part["max"]
            assert float(r["sum"]) == part["sum"]
            np.testing.assert_array_equal(np.array(r["hist"], dtype=np.int64), part["hist"])
    # raw uint16 indices (no white balance): operands up to 65535
    outs_raw = b.make_outputs(index=True)
    b.process(white_balance=False, outputs=outs_raw)
    for t in TYPES:
        np.testing.assert_array_equal(bits(outs_raw.host_index(t, 1, 1)[0]), bits(orc.index_app(tiles[1], t)))
    outs_raw.free()
    # the full-histogram route (lars_d_channel_hist + lars_d_wb_table) must produce the same table blob
    blob_fast = b.table.download(np.uint8, (n, b.table_bytes))
    hist = _ffi.DeviceBuffer(n * 3 * 65536 * 4)
    _ffi.call("lars_d_channel_hist", C.c_void_p(b.tiles.ptr), n, b.npix, 3, _ffi.U16, C.c_void_p(hist.ptr), None)
    _ffi.call("lars_d_wb_table", C.c_void_p(hist.ptr), n, b.npix, _ffi.U16, C.c_void_p(b.table.ptr),
              C.c_void_p(b.percentiles.ptr), 0, None)
    _ffi.call("lars_synchronize", None)
    blob_full = b.table.download(np.uint8, (n, b.table_bytes))
    used = 196608 + 3 * 260 * 4 + 48
    np.testing.assert_array_equal(blob_fast[:, :used], blob_full[:, :used])
    h16 = hist.download(np.uint32, (n, 3, 65536))
    np.testing.assert_array_equal(h16[0, 2], np.bincount(tiles[0][:, :, 2].ravel(), minlength=65536))
    hist.free(); outs.free(); b.free()


def test_uint16_percentiles_one_pass_and_its_recount(lars):
    """lars_d_wb_prepare for uint16 tiles: ONE full pass on value windows predicted from a subsample (the count of the samples below
    each window and the histogram inside it), the two radix passes, and np.percentile; with windows that miss on purpose every tile
    takes the two passes after the one.  Heavy-tailed, constant and few-level channels included, and marks next to a bin boundary."""
    from lars_image_processing_amd import _ffi
    rng = np.random.default_rng(21)
    h, w = 512, 768
    ramp = (np.arange(h * w * 3, dtype=np.int64).reshape(h, w, 3) * 7919 % 65536).astype(np.uint16)
    tiles = np.stack([
        rng.integers(0, 65536, (h, w, 3), dtype=np.uint16),
        np.clip(rng.normal(30000, 900, (h, w, 3)), 0, 65535).astype(np.uint16),
        np.full((h, w, 3), 4242, np.uint16),
        np.where(rng.random((h, w, 3)) < 0.03, 65535, 7).astype(np.uint16),
        (rng.integers(0, 4, (h, w, 3)) * 21845).astype(np.uint16),
        np.clip(rng.normal(255.6 * 50, 40, (h, w, 3)), 0, 65535).astype(np.uint16),      # the marks sit next to a bin boundary
        ramp,
        (rng.integers(0, 4096, (h, w, 3)) * 16).astype(np.uint16),                         # 12-bit samples in the high bits
        np.clip(rng.normal(800, 300, (h, w, 3)), 0, 4095).astype(np.uint16),                # 12-bit samples in the low bits, clipped at 0
        np.where(rng.random((h, w, 3)) < 0.5, rng.integers(0, 65536, (h, w, 3)), 30000).astype(np.uint16),   # half the samples on one value
    ])
    b = lars.TileBatch.from_host(tiles)
    b.compute_wb_tables()                                            # allocates the blobs (their padding is never written)
    got = {}
    try:
        for impl in (5, 1, 3):
            _ffi.set_tuning(u16_hist_impl=impl)
            b.table.zero(); b.percentiles.zero()
            b.compute_wb_tables()
            _ffi.call("lars_synchronize", None)
            got[impl] = (b.host_percentiles().tobytes(), b.table.download(np.uint8, (b.ntiles, b.table_bytes)).tobytes())
    finally:
        _ffi.set_tuning(u16_hist_impl=5)
    assert got[1] == got[3] == got[5]
    pcts = b.host_percentiles()
    for i in range(len(tiles)):
        for c in range(3):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = [float(v) for v in np.percentile(tiles[i][:, :, c].astype(np.float32), (2, 98))]
            assert pcts[i, c].tolist() == want, (i, c)
    b.free()


def test_uint16_full_size_tile(lars):
    """BASELINE configs[4] at its full size: one 8192 x 8192 uint16 tile (384 MiB), float32 NDVI + RdYlGn RGBA written.
    Percentiles against np.percentile on the whole channel; planes bit-exact on strips (first, middle, last rows) with the
    white balance taken from the reference expression on those strips; statistics through size-independent identities."""
    edge = 8192
    rng = np.random.default_rng(16)
    tile = rng.integers(0, 65536, (edge, edge, 3), dtype=np.uint16)
    tile[:, :, 0] >>= 2                                   # a 14-bit channel next to two 16-bit ones
    b = lars.TileBatch.from_host(tile[None])
    outs = b.make_outputs(indices=("NDVI",), index=True, rgba=True)
    rec = b.process(indices=("NDVI",), hist=True, outputs=outs)
    pcts = b.host_percentiles()[0]
    luts = []
    for c in range(3):
        lo, hi = np.percentile(tile[:, :, c].astype(np.float32), (2, 98))
        assert pcts[c, 0] == lo and pcts[c, 1] == hi, c
        luts.append(orc.wb_lut_from_percentiles(lo, hi, 65536))
    ndvi = outs.host_index("NDVI", 0, 1)[0]
    rgba = outs.rgba[0].download(np.uint8, (edge, edge, 4))
    lut_rgba = lars.colormap_lut("RdYlGn")
    for rows in (slice(0, 48), slice(edge // 2 - 24, edge // 2 + 24), slice(edge - 48, edge)):
        strip = tile[rows]
        wb = np.stack([luts[c][strip[:, :, c]] for c in range(3)], axis=2)
        want = orc.index_app(wb, "NDVI")
        np.testing.assert_array_equal(bits(ndvi[rows]), bits(want))
        np.testing.assert_array_equal(rgba[rows], orc.colormap_closed_form(want, lut_rgba))
    r = rec[0, 0]
    assert int(r["count"]) == edge * edge and int(np.sum(r["hist"])) == edge * edge
    assert float(r["min"]) == float(ndvi.min()) and float(r["max"]) == float(ndvi.max())
    assert int(r["above"]) == int(np.count_nonzero(ndvi > np.float32(0.2)))
    assert float(r["sum"]) == float(np.sum(ndvi, dtype=np.float64))
    np.testing.assert_array_equal(np.array(r["hist"], dtype=np.int64), np.histogram(ndvi, bins=50, range=(-1, 1))[0])
    outs.free(); b.free()


def test_uint16_rgba_channels_take_generic_path(lars):
    rng = np.random.default_rng(9)
    tiles = rng.integers(0, 65536, (2, 33, 47, 4), dtype=np.uint16)
    b = lars.TileBatch.from_host(tiles)
    outs = b.make_outputs(index=True, wb=True)
    b.process(outputs=outs)
    wb = outs.host_wb(0, 2)
    for i in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = orc.wb_app(tiles[i])
        np.testing.assert_array_equal(wb[i], want)
        np.testing.assert_array_equal(bits(outs.host_index("NDVI", i, 1)[0]), bits(orc.index_app(want, "NDVI")))
    outs.free(); b.free()


def test_output_ring_placement_trials(lars):
    """make_outputs(placement_trials=k): k candidate rings are timed, one survives and works like any other."""
    b = lars.TileBatch.synthetic(6, 64, 96, seed=5, profile="vegetation")
    plain = b.make_outputs(index=True, ring=2)
    tuned = b.make_outputs(index=True, ring=2, placement_trials=3)
    assert 1 <= len(tuned.placement_ms["arenas"]) <= 3 and tuned.placement_ms["chosen"] == min(tuned.placement_ms["arenas"]) > 0
    assert not hasattr(plain, "placement_ms")
    # the three index planes are slices of one allocation, in index order
    assert [tuned.index[k].ptr - tuned.arena.ptr for k in range(3)] == [0, tuned.plane_bytes, 2 * tuned.plane_bytes]
    rec_a = b.process(outputs=plain)
    ndvi_a = plain.host_index("NDVI", 1, 1)
    rec_b = b.process(outputs=tuned)
    np.testing.assert_array_equal(bits(tuned.host_index("NDVI", 1, 1)), bits(ndvi_a))
    assert rec_a.tobytes() == rec_b.tobytes()
    # the search's own rules: it ends once both speed classes have been seen (best 7 % under the worst) or at the limit asked for; the
    # survivor is timed once more after the rejected candidates were freed; the diagnostic pick keeps the slowest instead
    wide = b.make_outputs(index=True, ring=2, placement_trials=6)
    rep = wide.arena_report
    assert 2 <= len(rep["candidate_ms"]) <= 6 and rep["rejected"] == len(rep["candidate_ms"]) - 1 == len(rep["malloc_ms"]) - 1
    assert rep["chosen_ms"] == min(rep["candidate_ms"]) and rep["post_free_ms"] > 0
    assert rep["transient_bytes"] == len(rep["candidate_ms"]) * wide.arena.nbytes
    if len(rep["candidate_ms"]) < 6:
        assert min(rep["candidate_ms"]) <= 0.93 * max(rep["candidate_ms"]) and "both classes" in rep["kind"]
    worst = b.make_outputs(index=True, ring=2, placement_trials=3, pick="slowest")
    assert worst.arena_report["chosen_ms"] == max(worst.arena_report["candidate_ms"])
    worst.free()
    # planes may sit anywhere inside the arena (what the search over placements of multi-GiB arenas does): same results
    from lars_image_processing_amd._ffi import DeviceBuffer
    spaced = lars.batch.BatchOutputs(b, ("NDVI", "GNDVI", "NDWI"), True, False, False, 2, allocate=False)
    pb = spaced.plane_bytes
    spaced.adopt_arena(DeviceBuffer(7 * pb + 4096), (2 * pb + 256, 0, 5 * pb + 4096))
    assert [spaced.index[k].ptr - spaced.arena.ptr for k in range(3)] == [2 * pb + 256, 0, 5 * pb + 4096]
    rec_c = b.process(outputs=spaced)
    assert rec_c.tobytes() == rec_a.tobytes()
    np.testing.assert_array_equal(bits(spaced.host_index("NDVI", 1, 1)), bits(ndvi_a))
    with pytest.raises(AssertionError):
        spaced.adopt_arena(spaced.arena, (0, pb // 2 & ~255, 3 * pb))                    # overlapping planes
    spaced.free()
    plain.free(); tuned.free(); wide.free(); b.free()


def test_planes_stay_aligned_inside_the_arena(lars):
    """Odd plane sizes (slots * npix not a multiple of 4): planes start on 256-byte boundaries inside the arena, so the
    quad-per-lane kernel keeps running; multi-GiB arenas are chosen among timed plain allocations by default."""
    from lars_image_processing_amd import _ffi
    odd = lars.TileBatch.synthetic(1, 33, 35, seed=2)
    o = odd.make_outputs(index=True)
    assert o.plane_bytes % 256 == 0 and all(o.index[k].ptr % 256 == 0 for k in range(3))
    assert o.arena_report["kind"] == "plain hipMalloc"
    rec = odd.process(outputs=o)
    assert _ffi.get_tuning("last_fused_kernel") in (1, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.index_app(orc.wb_app(odd.host_tiles()[0]), "GNDVI")
    np.testing.assert_array_equal(bits(o.host_index("GNDVI", 0, 1)[0]), bits(want))
    assert float(rec[0, 1]["sum"]) == orc.tile_partials(want, "GNDVI")["sum"]
    with pytest.raises(ValueError):
        odd.make_outputs(index=True, arena="assembled")
    o.free(); odd.free()


def test_rgba_tiles_and_two_index_masks_take_the_fast_kernels(lars):
    """RGBA uint8 uploads (alpha ignored, zero in the white-balanced image: process-images.py:432-435) and the two-index
    masks the multiselect can hand over (process-images.py:1501-1505) run on the quad-per-lane kernels, not on the
    one-pixel-per-lane fallback -- with the same results as the RGB tiles / the three-index launch."""
    from lars_image_processing_amd import _ffi
    rng = np.random.default_rng(11)
    rgb = np.stack([orc.synth_tile_u8(3, t, 64, 96, profile="vegetation") for t in range(5)])
    rgba = np.concatenate([rgb, rng.integers(0, 256, rgb.shape[:3] + (1,), dtype=np.uint8)], axis=-1)
    b3, b4 = lars.TileBatch.from_host(rgb), lars.TileBatch.from_host(rgba)
    o3, o4 = b3.make_outputs(index=True, wb=True, rgba=True), b4.make_outputs(index=True, wb=True, rgba=True)
    rec3 = b3.process(hist=True, sumsq=True, outputs=o3)
    rec4 = b4.process(hist=True, sumsq=True, outputs=o4)
    assert _ffi.get_tuning("last_fused_kernel") == 5
    np.testing.assert_allclose(rec3["sumsq"], rec4["sumsq"], rtol=1e-12, atol=2.0 ** -26)
    rec3["sumsq"] = rec4["sumsq"] = 0
    assert rec3.tobytes() == rec4.tobytes()
    np.testing.assert_array_equal(b4.host_hist(), b3.host_hist())
    np.testing.assert_array_equal(b4.host_tables(), b3.host_tables())
    wb4 = o4.host_wb(0, 5)
    np.testing.assert_array_equal(wb4[..., :3], o3.host_wb(0, 5))
    assert not wb4[..., 3].any()
    for t in TYPES:
        np.testing.assert_array_equal(bits(o4.host_index(t, 0, 5)), bits(o3.host_index(t, 0, 5)))
        np.testing.assert_array_equal(o4.host_rgba(t, 0, 5), o3.host_rgba(t, 0, 5))
    # statistics only: the one-read route and the per-pixel kernels, medians included
    for route in ("joint", "classic"):
        r3, m3 = b3.process(medians=True, route="classic")
        r4, m4 = b4.process(medians=True, route=route)
        assert r3.tobytes() == r4.tobytes(), route
        np.testing.assert_array_equal(m3, m4)
    # a single RGBA tile whose pixel count is not a multiple of 4, without white balance
    odd = lars.TileBatch.from_host(rgba[:1, :33, :35])
    oo = odd.make_outputs(index=True)
    ro = odd.process(white_balance=False, outputs=oo)
    assert _ffi.get_tuning("last_fused_kernel") == 5
    want = orc.index_app(rgba[0, :33, :35], "GNDVI")
    np.testing.assert_array_equal(bits(oo.host_index("GNDVI", 0, 1)[0]), bits(want))
    assert float(ro[0, 1]["sum"]) == orc.tile_partials(want, "GNDVI")["sum"]
    assert odd.process(white_balance=False, route="joint").tobytes() == ro.tobytes()
    oo.free(); odd.free()
    # two-index masks: three-index kernel, third plane and record untouched
    full = b3.process(hist=True)
    for pair in (("NDVI", "NDWI"), ("NDVI", "GNDVI"), ("GNDVI", "NDWI")):
        op = b3.make_outputs(indices=pair, index=True)
        for route in ("classic", "joint"):
            rp = b3.process(indices=pair, hist=True, outputs=op if route == "classic" else None, route=route)
            if route == "classic":
                assert _ffi.get_tuning("last_fused_kernel") in (1, 2)
            for k, t in enumerate(TYPES):
                if t in pair:
                    assert rp[:, k].tobytes() == full[:, k].tobytes(), (pair, t, route)
                else:
                    assert not rp[:, k].tobytes().strip(b"\0"), (pair, t, route)
        for t in pair:
            np.testing.assert_array_equal(bits(op.host_index(t, 0, 5)), bits(o3.host_index(t, 0, 5)))
        rs = b3.process(indices=pair, route="classic")                 # statistics only through the per-pixel kernel
        assert _ffi.get_tuning("last_fused_kernel") in (1, 2)
        for k, t in enumerate(TYPES):
            if t in pair:
                assert rs[:, k]["sum"].tolist() == full[:, k]["sum"].tolist() and rs[:, k]["above"].tolist() == full[:, k]["above"].tolist()
        op.free()
    o3.free(); o4.free(); b3.free(); b4.free()


def test_batch_medians_are_numpy_medians(lars):
    b = lars.TileBatch.synthetic(21, 64, 96, seed=13, profile="vegetation")        # 21 tiles over a ring of 16: two chunks
    rec, med = b.process(medians=True)
    tiles = b.host_tiles()
    for i in (0, 7, 15, 16, 20):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wb = orc.wb_app(tiles[i])
        for k, t in enumerate(TYPES):
            assert med[i, k] == float(np.median(orc.index_app(wb, t))), (i, t)
    rows = lars.timeseries_rows(rec, med, "NDVI", dates=[f"2025-01-{d + 1:02d}" for d in range(21)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.stats_timeseries_row(orc.index_app(orc.wb_app(tiles[7]), "NDVI"), "NDVI", "2025-01-08")
    assert list(rows[7].keys()) == list(want.keys())
    for key, val in want.items():
        if key == "Mean":
            assert abs(rows[7][key] - val) <= 1e-6 * max(abs(val), 0.1), key
        else:
            assert rows[7][key] == val, key
    # with planes written as well (statistics kernel with outputs, then the two select passes): same medians, and the
    # ring holds the last chunk's planes
    outs = b.make_outputs(index=True, ring=8)
    rec_o, med_o = b.process(medians=True, outputs=outs)
    np.testing.assert_array_equal(med_o, med)
    assert rec_o["count"].tolist() == rec["count"].tolist() and rec_o["min"].tolist() == rec["min"].tolist()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_plane = orc.index_app(orc.wb_app(tiles[20]), "GNDVI")
    np.testing.assert_array_equal(outs.host_index("GNDVI", 20 % 8, 1)[0].view(np.uint32), want_plane.view(np.uint32))
    outs.free()
    # tiles the select does not serve (pixel count not a multiple of 4 in a batch; uint16) take the batched radix select
    # over stored planes
    odd = lars.TileBatch.from_host(np.stack([orc.synth_tile_u8(3, t, 63, 65, profile="vegetation") for t in range(3)]))
    wide = lars.TileBatch.from_host(np.random.default_rng(4).integers(0, 65536, (2, 40, 48, 3), dtype=np.uint16))
    for bb in (odd, wide):
        _, med_p = bb.process(medians=True)
        for i in range(bb.ntiles):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                wb = orc.wb_app(bb.host_tiles(i, 1)[0])
            for k, t in enumerate(TYPES):
                assert med_p[i, k] == float(np.median(orc.index_app(wb, t))), (i, t)
        bb.free()
    big = lars.TileBatch.synthetic(3, 512, 384, seed=3, profile="uniform")         # several workgroups per tile
    _, med_b = big.process(medians=True)
    for i in range(3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wb = orc.wb_app(big.host_tiles(i, 1)[0])
        for k, t in enumerate(TYPES):
            assert med_b[i, k] == float(np.median(orc.index_app(wb, t))), (i, t)
    big.free()
    # one index, two indices (separate statistics pass), and the 50-bin histograms next to the medians
    rec1, med1 = b.process(indices=("NDVI",), medians=True)
    np.testing.assert_array_equal(med1[:, 0], med[:, 0])
    assert np.isnan(med1[:, 1:]).all() and rec1[3, 0]["sum"] == rec[3, 0]["sum"]
    rec5, med5 = b.process(indices=("NDVI", "NDWI"), medians=True)
    np.testing.assert_array_equal(med5[:, [0, 2]], med[:, [0, 2]])
    assert np.isnan(med5[:, 1]).all()
    rech, medh = b.process(medians=True, hist=True)
    np.testing.assert_array_equal(medh, med)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wb7 = orc.wb_app(tiles[7])
    for k, t in enumerate(TYPES):
        np.testing.assert_array_equal(rech[7, k]["hist"], orc.hist50(orc.index_app(wb7, t)))
    rec2, med2 = b.process(indices=("NDWI",), medians=True, white_balance=False)
    assert np.isnan(med2[:, 0]).all() and med2[3, 2] == float(np.median(orc.index_app(tiles[3], "NDWI")))
    odd = lars.TileBatch.synthetic(2, 5, 7, seed=1)                                # odd sample count: single middle element
    _, m = odd.process(medians=True, white_balance=False)
    assert m[1, 0] == float(np.median(orc.index_app(odd.host_tiles()[1], "NDVI")))
    b.free(); odd.free()


def test_bad_arguments_fail_cleanly(lars):
    """Every entry point validates its arguments and reports through lars_last_error (no crash, no launch)."""
    import ctypes as C
    from lars_image_processing_amd import _ffi
    lib = _ffi.load()
    b = lars.TileBatch.synthetic(2, 16, 16, seed=1)
    b.compute_wb_tables()
    a = b.fused_args(("NDVI",), True, None, False, None)
    cases = []
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.tiles = None; cases.append(bad)
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.ntiles = 0; cases.append(bad)
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.channels = 2; cases.append(bad)
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.dtype = 7; cases.append(bad)
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.index_mask = 8; cases.append(bad)
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.flags = _ffi.F_STATS; cases.append(bad)          # stats == NULL
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.index_mask = 0; cases.append(bad)                # nothing to do
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.wb_table = None; bad.out_wb = b.tiles.ptr; cases.append(bad)
    bad = _ffi.FusedArgs.from_buffer_copy(a); bad.out_rgba[0] = b.tiles.ptr; cases.append(bad)    # no LUT
    for bad in cases:
        assert lib.lars_d_fused(C.byref(bad)) == -1 and lib.lars_last_error()
    assert lib.lars_d_channel_hist(None, 1, 16, 3, _ffi.U8, None, None) == -1
    assert lib.lars_d_wb_table(None, 1, 16, _ffi.U8, None, None, 0, None) == -1
    assert lib.lars_d_wb_prepare(C.c_void_p(b.tiles.ptr), 1, 16, 3, 9, C.c_void_p(b.table.ptr), None, 0, None) == -1
    assert lib.lars_h_fix_white_balance(None, 4, 4, 3, _ffi.U8, 0, None, None) == -1
    assert lib.lars_h_analyze_f32(None, 0, 0.2, 0, None, None) == -1
    assert lib.lars_d_median_pair_f32(None, 10, None, None, None) == -1
    # the newer entry points: select passes, statistics + medians, registration, change map, masks
    stats = b.new_stats()
    u4 = (C.c_uint32 * 4)(0, 0, 0, 0)
    big_bucket = (C.c_uint32 * 4)(2048, 0, 0, 0)
    hist64 = _ffi.DeviceBuffer(2 * 2 * 2048 * 8)
    tp = C.c_void_p(b.tiles.ptr)
    assert lib.lars_d_quotient_select_hist(tp, 2, 256, 3, _ffi.U8, None, 3, 0, big_bucket, C.c_void_p(hist64.ptr), None) == -1
    assert lib.lars_d_quotient_select_hist(tp, 2, 256, 4, _ffi.U8, None, 3, 1, u4, C.c_void_p(hist64.ptr), None) == -1
    assert lib.lars_d_quotient_select_hist(tp, 2, 256, 3, _ffi.U16, None, 3, 1, u4, C.c_void_p(hist64.ptr), None) == -1
    assert lib.lars_d_quotient_select_hist(tp, 2, 256, 3, _ffi.U8, None, 0, 1, u4, C.c_void_p(hist64.ptr), None) == -1
    assert lib.lars_d_quotient_median_pairs(tp, 2, 256, 3, _ffi.U8, None, 3, None, None, None) == -1
    sm = b.fused_args(("NDVI", "GNDVI"), True, stats, False, None)                      # two indices: not served
    assert lib.lars_d_stats_medians(C.byref(sm), C.c_void_p(hist64.ptr), C.c_void_p(hist64.ptr)) == -1
    sm = b.fused_args(("NDVI",), True, None, False, None)                               # no statistics records
    assert lib.lars_d_stats_medians(C.byref(sm), C.c_void_p(hist64.ptr), C.c_void_p(hist64.ptr)) == -1
    assert lib.lars_h_align_images(None, None, 8, 8, 3, None, None) == -1
    img8 = np.zeros((8, 8, 4), np.uint8)
    assert lib.lars_h_align_images(_ffi.ptr(img8), _ffi.ptr(img8), 8, 8, 4, _ffi.ptr(img8.copy()), None) == -1
    assert lib.lars_h_change_detection(_ffi.ptr(img8), _ffi.ptr(img8), 8, 8, 4, 0, 0, 0, 7, None, None, None, None, None,
                                       C.c_float(-0.5), C.c_float(0.5), None, None) == -1      # index_id out of range
    x = np.zeros(16, np.float32)
    rg = np.zeros((16, 4), np.uint8)
    lut = np.zeros((256, 4), np.uint8)
    assert lib.lars_h_colormap_norm_f32(_ffi.ptr(x), 16, C.c_float(0.5), C.c_float(0.5), _ffi.ptr(lut), _ffi.ptr(rg)) == -1   # vmax == vmin
    assert lib.lars_h_threshold_mask_f32(None, 16, C.c_float(0.2), None) == -1
    assert lib.lars_d_threshold_mask_f32(C.c_void_p(hist64.ptr + 4), 16, C.c_float(0.2), C.c_void_p(hist64.ptr), None) == -1  # misaligned
    assert lib.lars_d_shift_reflect_u8(tp, 16, 16, 3, C.c_void_p(hist64.ptr), tp, None) == -1                                 # in place
    hist64.free(); stats.free()
    assert lib.lars_set_tuning(b"no_such_knob", 1) == -1
    assert lib.lars_set_device(99) == -1 and b"out of range" in lib.lars_last_error()
    # float images are white-balanced like the reference does (a constant image: 0/0 -> NaN -> 0 everywhere)
    assert not lars.fix_white_balance(np.zeros((4, 4, 3), np.float32)).any()
    with pytest.raises(TypeError):
        lars.fix_white_balance(np.zeros((4, 4, 3), dtype=complex))
    # the library still works afterwards
    rec = b.process(indices=("NDVI",))
    assert int(rec[0, 0]["count"]) == 256
    b.free()


@pytest.mark.parametrize("white_balance", [True, False])
@pytest.mark.parametrize("shape,ntiles", [((64, 96), 5), ((63, 65), 1), ((128, 128), 2)])
def test_global_medians_without_planes(shape, ntiles, white_balance):
    """np.median over all pixels of all tiles, from two select passes that recompute the index values."""
    import warnings
    import lars_image_processing_amd as lars
    from oracle import index_oracle as orc
    tiles = np.stack([orc.synth_tile_u8(5, t, shape[0], shape[1], profile="vegetation" if t % 2 else "uniform")
                      for t in range(ntiles)])
    b = lars.TileBatch.from_host(tiles)
    got = b.global_medians(white_balance=white_balance)
    planes = {t: [] for t in ("NDVI", "GNDVI", "NDWI")}
    for img in tiles:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            src = orc.wb_app(img) if white_balance else img
        for t in planes:
            planes[t].append(orc.index_app(src, t).ravel())
    for t in planes:
        assert got[t] == float(np.median(np.concatenate(planes[t]))), t
    # one index only, and agreement with the per-tile medians of the plane-writing route for a single tile
    if ntiles == 1:
        rec, med = b.process(white_balance=white_balance, medians=True)
        for k, t in enumerate(("NDVI", "GNDVI", "NDWI")):
            assert float(med[0, k]) == got[t]


def test_every_quotient_of_bytes_is_found_back_on_the_device():
    """The value look-up of the two-level select (bucket, slot -> n/d) for all 65536 (red, nir) pairs: 4-pixel tiles
    of one colour each, so a tile's median is its single quotient; GNDVI runs through the pairs in another order."""
    import lars_image_processing_amd as lars
    from oracle import index_oracle as orc
    r, n = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    r, n = r.ravel(), n.ravel()
    g = (r.astype(np.int64) * 7 + 13).astype(np.uint8)
    for lo in (0, 32768):
        px = np.stack([r[lo:lo + 32768], g[lo:lo + 32768], n[lo:lo + 32768]], axis=1)        # [tiles, 3]
        tiles = np.ascontiguousarray(np.repeat(px[:, None, None, :], 4, axis=2))           # [tiles, 1, 4, 3]
        b = lars.TileBatch.from_host(tiles)
        med = b.tile_medians(white_balance=False)
        rec, med2 = b.process(white_balance=False, medians=True)
        for k, t in enumerate(("NDVI", "GNDVI", "NDWI")):
            want = orc.index_app(px[None], t)[0].astype(np.float64)                          # one value per tile
            assert np.array_equal(med[:, k], want), t
            assert np.array_equal(med2[:, k], want), t
            assert not np.signbit(med[want == 0, k]).any() or t == "NDWI"
        b.free()


def test_caller_stream_is_ordered_against_the_zeroing(lars):
    """process(..., stream=s) / select_histogram(..., stream=s): the records, the median scratch and the select
    histograms are zeroed on the SAME stream the kernels run on (a memset on the library stream could land after the
    kernels started accumulating).  Several rounds on a non-blocking stream must equal the library-stream results."""
    import ctypes as C
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch.synthetic(9, 128, 192, seed=77, profile="vegetation")
    b.compute_wb_tables()
    want_rec, want_med = b.process(medians=True, recompute_tables=False)
    want_hist = b.select_histogram(True, [0, 0, 0, 0])
    s = C.c_void_p()
    _ffi.call("lars_stream_create", C.byref(s))
    try:
        for _ in range(6):
            rec, med = b.process(medians=True, recompute_tables=False, stream=s)
            assert rec.tobytes() == want_rec.tobytes()
            np.testing.assert_array_equal(med, want_med)
            np.testing.assert_array_equal(b.select_histogram(True, [0, 0, 0, 0], stream=s), want_hist)
        outs = b.make_outputs(index=True, ring=4)
        rec2, med2 = b.process(medians=True, recompute_tables=True, outputs=outs, stream=s)
        assert rec2.tobytes() == want_rec.tobytes()
        np.testing.assert_array_equal(med2, want_med)
        outs.free()
    finally:
        _ffi.call("lars_synchronize", s)
        _ffi.call("lars_stream_destroy", s)
    b.free()


@pytest.mark.parametrize("profile", ["vegetation", "uniform"])
@pytest.mark.parametrize("indices", [("NDVI", "GNDVI", "NDWI"), ("NDVI",), ("NDWI",)])
def test_one_pass_medians_and_their_fallback(lars, profile, indices):
    """Statistics + exact medians without planes: the statistics kernel counts the slots of a predicted window of buckets
    next to the buckets themselves, so the slot pass is only taken where the prediction missed.  Three routes must give
    np.median exactly: predicted windows (default), windows pointed at the wrong place on purpose (every tile falls back
    to the slot pass) and no windows at all (always two passes); statistics records identical throughout.  Includes
    constant tiles (every value in one slot) and a tile whose two middle values straddle buckets."""
    from lars_image_processing_amd import _ffi
    tiles = [orc.synth_tile_u8(41, t, 128, 256, profile=profile) for t in range(6)]
    tiles.append(np.full((128, 256, 3), 77, dtype=np.uint8))                              # constant: WB -> NaN -> 0, index 0
    half = np.zeros((128, 256, 3), dtype=np.uint8)
    half[:64] = (10, 40, 200)
    half[64:] = (200, 40, 10)                                                             # two values, N/2 each: the middles differ
    tiles.append(half)
    b = lars.TileBatch.from_host(np.stack(tiles))
    want = np.full((b.ntiles, 3), np.nan)
    for i, tile in enumerate(tiles):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wb = orc.wb_app(tile)
        for t in indices:
            want[i, TYPES.index(t)] = float(np.median(orc.index_app(wb, t)))
    results = {}
    try:
        for mode in (1, 2, 0):
            _ffi.set_tuning(selq_window=mode)
            rec, med = b.process(indices=indices, medians=True)
            results[mode] = rec.tobytes()
            np.testing.assert_array_equal(med, want, err_msg=f"selq_window={mode}")
            # the medians on their own (what a launch that writes planes uses): prediction + one window sweep + fallback
            np.testing.assert_array_equal(b.tile_medians(indices), want, err_msg=f"tile_medians, selq_window={mode}")
    finally:
        _ffi.set_tuning(selq_window=1)
    assert results[1] == results[2] == results[0]
    b.free()


def test_one_pass_medians_at_the_ends_of_the_range(lars):
    """Windows clamped at -1 and +1, and a tile whose two middle values are -1 and +1 (no window can hold both: the
    classic passes take over); without white balance so that the bytes are what the index sees."""
    h, w = 128, 256
    top = np.zeros((h, w, 3), dtype=np.uint8); top[..., 2] = 200                      # NDVI = GNDVI = +1
    bottom = np.zeros((h, w, 3), dtype=np.uint8); bottom[..., 0] = 200; bottom[..., 1] = 200      # NDVI = GNDVI = -1
    split = top.copy(); split[: h // 2] = bottom[: h // 2]                            # N/2 values -1, N/2 values +1
    near = top.copy(); near[..., 0] = 1; near[..., 1] = 2                             # 199/201 and 198/202: just below +1
    rng = np.random.default_rng(5)
    mostly = bottom.copy(); mostly[rng.random((h, w)) < 0.3] = (3, 7, 250)            # 70 % at -1, the rest near +1
    tiles = [top, bottom, split, near, mostly]
    b = lars.TileBatch.from_host(np.stack(tiles))
    rec, med = b.process(medians=True, white_balance=False)
    for i, tile in enumerate(tiles):
        for k, t in enumerate(TYPES):
            assert med[i, k] == float(np.median(orc.index_app(tile, t))), (i, t)
    np.testing.assert_array_equal(b.tile_medians(white_balance=False), med)
    # over all five tiles at once (sampled prediction, window pass, fallback where the window misses)
    glob = b.global_medians(white_balance=False)
    for t in TYPES:
        assert glob[t] == float(np.median(np.concatenate([orc.index_app(tile, t).ravel() for tile in tiles]))), t
    for sub in ([top, top], [bottom], [split]):
        bb = lars.TileBatch.from_host(np.stack(sub))
        g = bb.global_medians(white_balance=False)
        for t in TYPES:
            assert g[t] == float(np.median(np.concatenate([orc.index_app(tile, t).ravel() for tile in sub]))), (t, len(sub))
        bb.free()
    b.free()
