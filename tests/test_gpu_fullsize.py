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


def test_placement_search_of_a_multi_gib_arena():
    """make_outputs(arena="auto") for several planes of 2 GiB and more each: ONE allocation with room to spare, the planes tried in
    several placements inside it (profiles/r04_arena_two_kinds.txt), the fastest kept -- and the planes it writes there are the
    planes a packed arena gets, bit for bit.  Smaller planes (they run alike wherever they lie) and callers that ask for one trial
    at most get one packed allocation: no spare room is held for them."""
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import _ffi
    ntiles, ring, edge = 32, 32, 4096                                  # three planes of 32 slots: 2 GiB each, 6 GiB packed
    b = lars.TileBatch.synthetic(ntiles, edge, edge, seed=77, profile="vegetation")
    plain = b.make_outputs(index=True, ring=ring, arena="plain")
    assert plain.arena_report["kind"] == "plain hipMalloc" and plain.arena.nbytes == 3 * plain.plane_bytes
    for kw in (dict(ring=12), dict(ring=ring, placement_trials=1), dict(ring=ring, placement_trials=0)):
        small = b.make_outputs(index=True, **kw)                         # 0.75 GiB planes; one trial at most: packed, nothing held back
        assert small.arena_report["kind"] == "plain hipMalloc" and small.arena.nbytes == 3 * small.plane_bytes, kw
        small.free()
    auto = b.make_outputs(index=True, ring=ring)
    rep = auto.arena_report
    assert rep["allocations"] == len(rep["malloc_ms"]) and rep["packed_bytes"] == 3 * auto.plane_bytes
    assert len(rep["candidate_ms"]) == len(rep["placements"]) >= 2
    assert rep["chosen_ms"] == min(rep["candidate_ms"]) and rep["post_free_ms"] > 0
    offs = [int(round(o * (1 << 30))) for o in rep["chosen_offsets_gib"]]
    assert all(abs(a - b_) <= (1 << 20) for a, b_ in zip(offs, auto.plane_offsets))
    if auto.arena2 is None:                                              # the usual end: one allocation with room to spare
        assert rep["rejected"] == len(rep["malloc_ms"]) - 1
        assert rep["arena_bytes"] == auto.arena.nbytes > 3 * auto.plane_bytes
        assert [auto.index[k].ptr - auto.arena.ptr for k in range(3)] == list(auto.plane_offsets)
    else:                                                                # every allocation of one kind: the planes split between two
        assert rep["rejected"] == len(rep["malloc_ms"]) - 2 and "two allocations" in rep["kind"]
        assert rep["arena_bytes"] == auto.arena.nbytes + auto.arena2.nbytes
        assert [auto.index[k].ptr - auto.arena.ptr for k in range(2)] == list(auto.plane_offsets[:2])
        assert auto.index[2].ptr - auto.arena2.ptr == auto.plane_offsets[2]
    assert {tuple(p["offsets_gib"]) for p in rep["placements"]} >= {tuple(round(j * auto.plane_bytes / (1 << 30), 3) for j in range(3))}   # packed is among them
    rec_p = b.process(outputs=plain)
    rec_a = b.process(outputs=auto)
    assert rec_p.tobytes() == rec_a.tobytes()
    for name in TYPES:
        assert plain.host_index(name, 0, ring).tobytes() == auto.host_index(name, 0, ring).tobytes(), name
    # one plane: nothing to split, one packed allocation
    single = b.make_outputs(indices=("NDVI",), index=True, ring=ring)
    assert single.arena_report["kind"] == "plain hipMalloc"
    single.free(); auto.free()
    # Every allocation of one kind throughout (forced here: no gap between candidates counts as a class): the planes are split between two
    # allocations, tried pair by pair; whatever wins, the planes written there are the same bits, and both allocations belong to the outputs.
    gap, trials = lars.batch.ARENA_CLASS_GAP, lars.batch.ARENA_TRIALS
    lars.batch.ARENA_CLASS_GAP, lars.batch.ARENA_TRIALS = 0.0, 2
    try:
        cross = b.make_outputs(index=True, ring=ring)
    finally:
        lars.batch.ARENA_CLASS_GAP, lars.batch.ARENA_TRIALS = gap, trials
    rep = cross.arena_report
    split = [p for p in rep["placements"] if isinstance(p["allocation"], list)]
    extra = lars.batch.ARENA_EXTRA_BLOCKS                           # then small allocations for the third plane alone, next to allocation 0's first two
    assert rep["allocations"] == 2 + extra and len(split) == 2 + extra
    assert {tuple(p["allocation"]) for p in split} == {(0, 1), (1, 0)} | {(0, 2 + k) for k in range(extra)}
    assert rep["rejected"] == rep["allocations"] - (2 if cross.arena2 is not None else 1)
    assert rep["chosen_ms"] == min(rep["candidate_ms"])
    assert rep["arena_bytes"] == cross.arena.nbytes + (cross.arena2.nbytes if cross.arena2 is not None else 0)
    rec_c = b.process(outputs=cross)
    assert rec_c.tobytes() == rec_p.tobytes()
    for name in TYPES:
        assert plain.host_index(name, 0, ring).tobytes() == cross.host_index(name, 0, ring).tobytes(), name
    # ... and explicitly: two planes in one buffer, the third in another
    two = lars.batch.BatchOutputs(b, TYPES, True, False, False, ring, allocate=False)
    a1, a2 = _ffi.DeviceBuffer(2 * two.plane_bytes), _ffi.DeviceBuffer(two.plane_bytes)
    two.adopt_two_arenas(a1, (0, two.plane_bytes), a2, (0,))
    assert two.index[2].ptr == a2.ptr and two.index[1].ptr == a1.ptr + two.plane_bytes
    rec_t = b.process(outputs=two)
    assert rec_t.tobytes() == rec_p.tobytes()
    for name in TYPES:
        assert plain.host_index(name, 0, ring).tobytes() == two.host_index(name, 0, ring).tobytes(), name
    two.free()
    assert two.arena is None and two.arena2 is None
    cross.free(); plain.free(); b.free()
