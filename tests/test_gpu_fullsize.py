"""The timed configuration of bench.py at its own size (needs a MI355X): 4096 x 4096 uint8 tiles, three float32 planes
into a ring of 64 tile slots, the batch worked off in ring-sized launches with the statistics records opened and closed
ONCE around them (LARS_F_RAW, batch.run_fused_chunks) -- two full launches and a partial one."""
import ctypes as C
import warnings

import numpy as np
import pytest

from oracle import index_oracle as orc

pytestmark = pytest.mark.gpu
TYPES = ("NDVI", "GNDVI", "NDWI")
BANDS = {"NDVI": (2, 0), "GNDVI": (2, 1), "NDWI": (1, 2)}          # (a, b) of (a - b) / (a + b), process-images.py:466-482


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_chunked_ring_launches_at_bench_size():
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import _ffi
    ntiles, ring, edge = 130, 64, 4096
    b = lars.TileBatch.synthetic(ntiles, edge, edge, seed=1234, profile="vegetation")
    outs = b.make_outputs(index=True, ring=ring, arena="plain")
    b.compute_wb_tables()
    stats = b.new_stats()
    stats.zero()
    assert b.run_fused_chunks(TYPES, True, stats, False, outs) == 3                    # 64 + 64 + 2 tiles
    _ffi.call("lars_synchronize", None)
    rec = stats.download(_ffi.STATS_DTYPE, (ntiles, 3))
    n = edge * edge
    assert (rec["count"] == n).all()
    # records of the first / last tile of a launch and of the partial launch against single-tile runs of the library
    one = lars.TileBatch(1, edge, edge, 3, np.uint8)
    one_out = one.make_outputs(index=True)
    for t in (0, 63, 64, 129):
        _ffi.call("lars_memcpy_d2d", C.c_void_p(one.tiles.ptr), C.c_void_p(b.tiles.ptr + t * b.tile_bytes), b.tile_bytes, None)
        r1 = one.process(outputs=one_out, route="classic")
        assert r1[0].tobytes() == rec[t].tobytes(), t
        if t == 129:                                                                   # slot 129 % 64 = 1 of the ring
            for name in TYPES:
                assert one_out.host_index(name, 0, 1).tobytes() == outs.host_index(name, 1, 1).tobytes(), name
    one_out.free(); one.free()
    # what the ring holds afterwards: tiles 128, 129 (last, partial launch) in slots 0, 1; tiles 66..127 (second launch)
    # in slots 2..63 -- bit-equal to the closed form on the white-balanced samples, and each record's sum equal to the exact
    # integer sum of the stored plane (float32 quotients of bytes are multiples of 2^-32)
    tables = b.host_tables()
    for tile, slot in ((129, 1), (127, 63), (66, 2)):
        raw = b.host_tiles(tile, 1)[0]
        wb = np.stack([tables[tile, c][raw[:, :, c]] for c in range(3)], axis=-1)
        if tile == 129:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                np.testing.assert_array_equal(wb, orc.wb_app(raw))                   # the table IS fix_white_balance
        for k, name in enumerate(TYPES):
            plane = outs.host_index(name, slot, 1)[0]
            a, c = BANDS[name]
            np.testing.assert_array_equal(bits(plane), bits(orc.index_closed_form(wb[:, :, a], wb[:, :, c])), err_msg=f"{name} tile {tile}")
            exact = int((plane.astype(np.float64) * 2.0 ** 32).astype(np.int64).sum())
            assert float(rec[tile, k]["sum"]) == float(exact) / 2.0 ** 32, (name, tile)
            assert float(rec[tile, k]["min"]) == float(plane.min()) and float(rec[tile, k]["max"]) == float(plane.max())
            thr = np.float32(0.0 if name == "NDWI" else 0.2)
            assert int(rec[tile, k]["above"]) == int((plane > thr).sum())
    # the one-read route over the same 130 tiles: the same records
    rec_j = b.process(route="joint")
    assert rec_j.tobytes() == rec.tobytes()
    outs.free(); stats.free(); b.free()
