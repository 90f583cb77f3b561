"""CPU-only checks: the C-ABI library loads, exports every symbol include/lars_hip.h
declares, refuses to compute without a GPU, and its host-side fold matches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from lars_image_processing_amd import _ffi, batch
from oracle import index_oracle as orc


def header_symbols():
    text = open(os.path.join(ROOT, "include", "lars_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lars_[a-z0-9_]+)\s*\(", text)))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("lars_")})


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"liblars_hip.so lacks {n}"
    # and the binding declares a prototype for each of them
    assert set(names) == set(_ffi.SIGNATURES), set(names) ^ set(_ffi.SIGNATURES)
    assert lib.lars_abi_version() == 1


def test_product_exports_are_exactly_the_documented_set():
    """The product library exports the entry points include/lars_hip.h documents and nothing else: no probes, no
    pipeline, no allocation experiments (those live in liblars_lab.so, include/lars_lab.h), and it reads no environment
    variable that changes how it allocates."""
    got = exported_symbols(_ffi.LIB_PATH)
    assert got == header_symbols(), set(got) ^ set(header_symbols())
    for name in ("lars_d_probe", "lars_d_pipeline", "lars_pipeline_scratch_bytes", "lars_d_output_arena", "lars_lab_malloc"):
        assert name not in got
    blob = open(_ffi.LIB_PATH, "rb").read()
    for env in (b"LARS_MALLOC_KIND", b"LARS_VMM_CHUNK_MB", b"LARS_VMM_SHUFFLE", b"LARS_VMM_ALIGN_MB"):
        assert env not in blob, env
    lab = os.path.join(os.path.dirname(_ffi.LIB_PATH), "liblars_lab.so")
    if os.path.exists(lab):
        text = open(os.path.join(ROOT, "include", "lars_lab.h")).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared = sorted(set(re.findall(r"\b(lars_[a-z0-9_]+)\s*\(", text)))
        assert exported_symbols(lab) == declared, set(exported_symbols(lab)) ^ set(declared)


def test_the_loaded_library_is_the_product_build():
    """Laboratory builds of the kernel sources (plane-layout knob, IEEE division, other counter forms or launch geometries:
    csrc/Makefile `lablayout`, `EXTRA=-D...`) announce themselves through lars_build_flags(); what the package loads answers 0.
    The switches that produced wrong results on purpose (round 2's ablations, round 4's no-adds / fake-banks counting kernel) are
    gone from the sources altogether."""
    assert _ffi.load().lars_build_flags() == 0
    src = os.path.join(ROOT, "lars_image_processing_amd", "csrc")
    for name in os.listdir(src):
        if name.endswith((".hip", ".h", ".cpp")):
            text = open(os.path.join(src, name)).read()
            for gone in ("LARS_ABLATE", "LARS_JOINT_NO_ADDS", "LARS_JOINT_FAKE_BANKS"):
                assert gone not in text or name == "fused_v2.hip" and "#if" not in "".join(l for l in text.splitlines() if gone in l), (name, gone)


def test_struct_layouts_match_header():
    assert C.sizeof(_ffi.Stats) == 472 == _ffi.STATS_DTYPE.itemsize
    for name, _ in _ffi.Stats._fields_:
        if name != "hist":
            assert getattr(_ffi.Stats, name).offset == _ffi.STATS_DTYPE.fields[name][1]
    assert _ffi.Stats.hist.offset == _ffi.STATS_DTYPE.fields["hist"][1] == 72
    assert C.sizeof(_ffi.FusedArgs) == 8 + 8 + 8 + 4 + 4 + 8 + 4 + 4 + 24 + 8 + 24 + 24 + 8 + 8


@pytest.mark.skipif(_ffi.device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_gpu():
    import lars_image_processing_amd as lars
    img = np.zeros((8, 8, 3), np.uint8)
    for fn in (lambda: lars.fix_white_balance(img), lambda: lars.calculate_index(img, "NDVI"),
               lambda: lars.analyze_index(np.zeros((4, 4), np.float32), "NDVI"),
               lambda: lars.process_image(img)):
        with pytest.raises(_ffi.LarsError) as e:
            fn()
        assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_reference_contract_needs_no_gpu():
    """None/empty/unknown-type behaviour is decided on the host (process-images.py:427,452,485,495)."""
    import lars_image_processing_amd as lars
    assert lars.fix_white_balance(None) is None
    assert lars.fix_white_balance(np.zeros((0, 0, 3), np.uint8)) is None
    assert lars.calculate_index(None, "NDVI") is None
    assert lars.analyze_index(None, "NDVI") == {}
    assert lars.analyze_index(np.zeros((0,), np.float32), "NDWI") == {}
    with pytest.raises(ValueError, match="Unknown index type: EVI"):
        lars.calculate_index(np.zeros((2, 2, 3), np.uint8), "EVI")
    with pytest.raises(UnboundLocalError):
        lars.calculate_index(np.ones(1, np.float32), np.ones(1, np.float32), np.ones(1, np.float32), "EVI")
    with pytest.raises(IndexError):
        lars.calculate_index(np.zeros((4, 4), np.uint8), "NDVI")
    assert lars.correct_white_balance is lars.fix_white_balance
    assert lars.analyze_index_statistics is lars.analyze_index


def to_records(parts, index_id):
    rec = np.zeros(len(parts), dtype=_ffi.STATS_DTYPE)
    for i, p in enumerate(parts):
        rec[i]["sum"], rec[i]["sumsq"] = p["sum"], p["sumsq"]
        rec[i]["count"], rec[i]["above"] = p["count"], p["above"]
        rec[i]["min"], rec[i]["max"] = p["min"], p["max"]
        rec[i]["index_id"] = index_id
        rec[i]["hist"] = p["hist"]
    return rec


def test_stats_merge_matches_oracle_fold():
    rng = np.random.default_rng(3)
    tiles = [rng.integers(0, 256, (24, 40, 3), dtype=np.uint8) for _ in range(7)]
    for k, t in enumerate(("NDVI", "GNDVI", "NDWI")):
        parts = [orc.tile_partials(orc.index_app(x, t), t) for x in tiles]
        merged = batch.summarize(batch.merge_records(to_records(parts, k)))
        want = orc.merge_partials(parts)
        assert merged["count"] == want["count"]
        assert merged["min"] == want["min"] and merged["max"] == want["max"]
        assert merged["coverage"] == want["coverage"]
        np.testing.assert_array_equal(merged["hist"], want["hist"])
        assert abs(merged["mean"] - want["mean"]) <= 1e-15 * max(1.0, abs(want["mean"])) * len(tiles)
        assert abs(merged["std"] - want["std"]) <= 1e-12


def test_shard_range_partitions_tiles():
    for ntiles in (1, 7, 16, 1024, 16384):
        for world in (1, 2, 3, 4, 8):
            spans = [batch.shard_range(ntiles, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == ntiles
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_hist_edges_match_numpy_linspace():
    e32 = np.array([(np.float32(1.0) if i >= 50 else np.float32(np.float64(i) * (2.0 / 50.0) + (-1.0))) for i in range(51)])
    np.testing.assert_array_equal(e32, np.histogram_bin_edges(np.zeros(1, np.float32), bins=50, range=(-1, 1)))
    e64 = np.array([(1.0 if i >= 50 else np.float64(i) * (2.0 / 50.0) + (-1.0)) for i in range(51)])
    np.testing.assert_array_equal(e64, np.histogram_bin_edges(np.zeros(1, np.float64), bins=50, range=(-1, 1)))


def test_arena_placements():
    """The placements of the planes TileBatch.make_outputs tries inside one allocation (host logic, no GPU): packed first, then the second
    half of the planes further out in steps of 4 GiB up to 20 GiB; planes never overlap, never leave the allocation."""
    from lars_image_processing_amd import batch
    gib = 1 << 30
    got = batch.arena_placements(3, 4 * gib, 24 * gib)
    assert [tuple(o // gib for o in p) for p in got] == [(0, 4, 8), (0, 4, 12), (0, 4, 16), (0, 4, 20)]
    assert [tuple(o // gib for o in p) for p in batch.arena_placements(2, 4 * gib, 24 * gib)] == [(0, 4), (0, 8), (0, 12), (0, 16), (0, 20)]
    assert batch.arena_placements(1, 4 * gib, 24 * gib) == [(0,)]
    assert batch.arena_placements(3, 4 * gib, 12 * gib) == [(0, 4 * gib, 8 * gib)]                       # no room: packed only
    six = batch.arena_placements(6, 4 * gib, 32 * gib)
    assert [tuple(o // gib for o in p) for p in six] == [(0, 4, 8, 12, 16, 20), (0, 4, 8, 16, 20, 24), (0, 4, 8, 20, 24, 28)]
    for n, pb, nb in ((3, 4 * gib, 24 * gib), (2, 3 * gib + 256, 24 * gib), (5, 700 << 20, 22 * gib), (6, 4 * gib, 32 * gib), (4, 256, 1 << 20)):
        for p in batch.arena_placements(n, pb, nb):
            assert len(p) == n and p[0] == 0 and all(o % 256 == 0 for o in p)
            assert all(b - a >= pb for a, b in zip(p, p[1:])) and p[-1] + pb <= nb
