"""Batched, device-resident tile path (new entry point; BASELINE.json configs 2-5).

The reference processes one image per Python call (backend-process.py:92-95 is
its only batch loop).  Here a batch of equally sized RGNir tiles lives in HBM as
``[ntiles][H][W][C]`` and one launch sequence covers the whole batch:

    channel histograms -> white-balance tables -> fused indices + statistics

Tiles are independent, so a batch shards across GPUs by tile with no data-path
exchange; only the global statistics need one small collective (``dist.py``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import FusedArgs, INDEX_IDS, INDEX_NAMES, STATS_DTYPE, DeviceBuffer

MAX_TILES_PER_LAUNCH = 65535      # grid.y limit


def shard_range(ntiles, rank, world):
    """Contiguous block of tiles owned by ``rank`` (remainder to the low ranks)."""
    base, rem = divmod(int(ntiles), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class TileBatch:
    """A batch of interleaved uint8/uint16 tiles resident in HBM."""

    def __init__(self, ntiles, h, w, channels=3, dtype=np.uint8, first_tile=0):
        self.ntiles, self.h, self.w, self.channels = int(ntiles), int(h), int(w), int(channels)
        self.dtype = np.dtype(dtype)
        self.code = _ffi.dtype_code(self.dtype)
        if self.code is None:
            raise TypeError("tiles must be uint8 or uint16")
        if self.ntiles < 1 or self.ntiles > MAX_TILES_PER_LAUNCH:
            raise ValueError(f"1 <= ntiles <= {MAX_TILES_PER_LAUNCH} per batch")
        self.first_tile = int(first_tile)
        self.npix = self.h * self.w
        self.nvalues = 256 if self.code == _ffi.U8 else 65536
        self.table_bytes = int(_ffi.load().lars_wb_table_bytes(self.code))     # per tile
        self.tile_bytes = self.npix * self.channels * self.dtype.itemsize
        self.tiles = DeviceBuffer(self.ntiles * self.tile_bytes)
        self.hist = None
        self.table = None
        self.percentiles = None

    # -- construction -----------------------------------------------------
    @classmethod
    def from_host(cls, array, first_tile=0):
        arr = np.ascontiguousarray(array)
        if arr.ndim != 4 or arr.shape[3] < 3:
            raise ValueError("expected [ntiles, H, W, C>=3]")
        b = cls(arr.shape[0], arr.shape[1], arr.shape[2], arr.shape[3], arr.dtype, first_tile)
        b.tiles.upload(arr)
        return b

    @classmethod
    def synthetic(cls, ntiles, h, w, seed=1234, profile="uniform", first_tile=0, channels=3):
        """uint8 tiles generated in HBM by the counter hash (never cross PCIe)."""
        b = cls(ntiles, h, w, channels, np.uint8, first_tile)
        _ffi.call("lars_d_synth_u8", C.c_void_p(b.tiles.ptr), b.ntiles, b.first_tile, b.npix, b.channels,
                  int(seed) & 0xFFFFFFFF, {"uniform": 0, "vegetation": 1}[profile], None)
        _ffi.call("lars_synchronize", None)
        return b

    def host_tiles(self, start=0, count=None):
        count = self.ntiles - start if count is None else count
        return self.tiles.download(self.dtype, (count, self.h, self.w, self.channels), start * self.tile_bytes)

    # -- pass 1: white-balance tables --------------------------------------
    def compute_wb_tables(self, stream=None, rgn_variant=0):
        """np.percentile(ch, (2, 98)) per tile and channel -> 8-bit tables, on device."""
        if self.table is None:
            self.table = DeviceBuffer(self.ntiles * self.table_bytes)
            self.percentiles = DeviceBuffer(self.ntiles * 3 * 2 * 8)
        if self.code == _ffi.U8:
            if self.hist is None:
                self.hist = DeviceBuffer(self.ntiles * 3 * 256 * 4)
            _ffi.call("lars_d_channel_hist", C.c_void_p(self.tiles.ptr), self.ntiles, self.npix, self.channels,
                      self.code, C.c_void_p(self.hist.ptr), stream)
            _ffi.call("lars_d_wb_table", C.c_void_p(self.hist.ptr), self.ntiles, self.npix, self.code,
                      C.c_void_p(self.table.ptr), C.c_void_p(self.percentiles.ptr), int(rgn_variant), stream)
        else:
            # uint16: two-level radix percentiles, no 65536-bin histograms
            _ffi.call("lars_d_wb_prepare", C.c_void_p(self.tiles.ptr), self.ntiles, self.npix, self.channels,
                      self.code, C.c_void_p(self.table.ptr), C.c_void_p(self.percentiles.ptr), int(rgn_variant), stream)
        return self

    def host_tables(self):
        blob = self.table.download(np.uint8, (self.ntiles, self.table_bytes))
        return np.ascontiguousarray(blob[:, :3 * self.nvalues]).reshape(self.ntiles, 3, self.nvalues)

    def host_percentiles(self):
        return self.percentiles.download(np.float64, (self.ntiles, 3, 2))

    def host_hist(self):
        """uint8 batches only (uint16 percentiles come from two radix levels, no full histogram)."""
        return self.hist.download(np.uint32, (self.ntiles, 3, 256))

    # -- pass 2: the fused kernel ------------------------------------------
    def make_outputs(self, indices=INDEX_NAMES, index=False, wb=False, rgba=False, ring=None, placement_trials=0):
        """Allocate output planes.  ``ring`` < ntiles reuses a ring of that many
        tile slots (same HBM traffic, bounded footprint) -- see BatchOutputs.

        ``placement_trials`` = k > 1: allocate k candidate planes per index, time them and keep the fastest
        combination.  Where the driver happens to put a plane in HBM moves the write-bound fused kernel by up
        to 10 % -- reproducibly for the lifetime of the allocation -- so a long-lived
        ring is worth choosing once.  Costs k rings of memory for the duration of the trial."""
        if placement_trials <= 1 or not index:
            return BatchOutputs(self, indices, index, wb, rgba, ring)
        # Stage 1: a pool of candidate planes, each timed on its own (planes come out "fast" or "slow", ~5 % apart).
        # Stage 2: the combinations of the fastest few as a ring (planes also interact), the best one stays.
        import itertools
        ks = [INDEX_IDS[t] for t in indices]
        base = BatchOutputs(self, indices, False, wb, rgba, ring)          # everything but the index planes
        nbytes = base.slots * self.npix * 4
        pool = []
        for _ in range(int(placement_trials) * len(ks)):
            try:
                pool.append(DeviceBuffer(nbytes))
            except _ffi.LarsError:
                break                                       # out of memory: choose among what fits
        if len(pool) < len(ks):
            for p in pool:
                p.free()
            base.free()
            raise _ffi.LarsError(-2, "no memory for the output planes")
        solo = []
        for p in pool:
            base.index = [None] * 3
            base.index[ks[0]] = p
            solo.append(self._time_outputs(base, (indices[0],)))
        order = list(np.argsort(solo))
        short = order[:min(len(pool), len(ks) + 2)]
        best, best_ms, tried = None, None, []
        for combo in itertools.combinations(short, len(ks)):
            base.index = [None] * 3
            for k, j in zip(ks, combo):
                base.index[k] = pool[j]
            ms = self._time_outputs(base, indices)
            tried.append(ms)
            if best_ms is None or ms < best_ms:
                best, best_ms = combo, ms
        base.index = [None] * 3
        for k, j in zip(ks, best):
            base.index[k] = pool[j]
        for j, p in enumerate(pool):
            if j not in best:
                p.free()
        base.placement_ms = {"planes": [float(x) for x in solo], "rings": tried, "chosen": best_ms}
        return base

    def _time_outputs(self, outs, indices):
        """Milliseconds of fused launches that fill ``outs`` from up to three chunks of the batch (first, middle, last:
        the pairing with the input's placement matters too); the warm-up launch is not timed."""
        ev = [C.c_void_p(), C.c_void_p()]
        for e in ev:
            _ffi.call("lars_event_create", C.byref(e))
        stats = self.new_stats()
        count = min(outs.slots, self.ntiles)
        nchunks = max(1, self.ntiles // count)
        starts = sorted({0, (nchunks // 2) * count, (nchunks - 1) * count})
        launches = [self.fused_args(indices, self.table is not None, stats, False, outs, None, st, count) for st in starts]
        self.run_fused(launches[0])
        _ffi.call("lars_event_record", ev[0], None)
        for a in launches:
            self.run_fused(a)
        _ffi.call("lars_event_record", ev[1], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        for e in ev:
            _ffi.call("lars_event_destroy", e)
        stats.free()
        return float(ms.value) / len(launches)

    def fused_args(self, indices=INDEX_NAMES, white_balance=True, stats=None, hist=False, outputs=None,
                   stream=None, tile_start=0, tile_count=None, sumsq=False):
        tile_count = self.ntiles - tile_start if tile_count is None else tile_count
        a = FusedArgs()
        a.tiles = self.tiles.ptr + tile_start * self.tile_bytes
        a.ntiles, a.npix, a.channels, a.dtype = tile_count, self.npix, self.channels, self.code
        if white_balance:
            if self.table is None:
                raise RuntimeError("compute_wb_tables() first")
            a.wb_table = self.table.ptr + tile_start * self.table_bytes
        mask = 0
        for t in indices:
            mask |= 1 << INDEX_IDS[t]
        a.index_mask = mask
        a.flags = ((_ffi.F_STATS if stats is not None else 0) | (_ffi.F_HIST if (stats is not None and hist) else 0) |
                   (_ffi.F_SUMSQ if (stats is not None and sumsq) else 0))
        if stats is not None:
            a.stats = stats.ptr + tile_start * 3 * STATS_DTYPE.itemsize
        if outputs is not None:
            slot = tile_start % outputs.slots
            if slot + tile_count > outputs.slots:
                raise ValueError("tile range wraps the output ring")
            for k in range(3):
                if outputs.index[k] is not None:
                    a.out_index[k] = outputs.index[k].ptr + slot * self.npix * 4
                if outputs.rgba[k] is not None:
                    a.out_rgba[k] = outputs.rgba[k].ptr + slot * self.npix * 4
                    a.cmap_lut[k] = outputs.luts[k].ptr
            if outputs.wb is not None:
                a.out_wb = outputs.wb.ptr + slot * self.npix * self.channels
        a.stream = stream
        return a

    def new_stats(self):
        return DeviceBuffer(self.ntiles * 3 * STATS_DTYPE.itemsize)

    def run_fused(self, args):
        _ffi.call("lars_d_fused", C.byref(args))

    def process(self, indices=INDEX_NAMES, white_balance=True, hist=False, outputs=None, stream=None,
                recompute_tables=True, medians=False, sumsq=False):
        """Both passes over the whole batch; returns per-tile records
        (structured ndarray ``[ntiles, 3]`` of STATS_DTYPE; rows of indices not
        requested are zero).  ``hist`` adds the 50-bin histograms, ``sumsq`` the sums of squares
        (``summarize()['std']``).  ``medians=True`` also returns ``float64[ntiles, 3]``
        with np.median of each tile's index plane (exact: batched radix select on
        the float32 planes, which must then be written -- a small ring is
        allocated when ``outputs`` has none)."""
        if white_balance and (recompute_tables or self.table is None):
            self.compute_wb_tables(stream)
        stats = self.new_stats()
        stats.zero()
        if medians and outputs is None and self.code == _ffi.U8 and self.channels == 3 and (self.ntiles == 1 or self.npix % 4 == 0):
            # nothing to write: the statistics kernel also counts the select's bucket pass, two (rarely three) digit
            # passes follow -- 3 B per pixel each, everything on the device
            mask = 0
            for t in indices:
                mask |= 1 << INDEX_IDS[t]
            if mask in (1, 2, 4, 7):
                pairs_dev = DeviceBuffer(self.ntiles * 4 * 4)
                scratch = DeviceBuffer(int(_ffi.load().lars_quotient_median_scratch_bytes(self.ntiles)))
                args = self.fused_args(indices, white_balance, stats, hist, None, stream, sumsq=sumsq)
                _ffi.call("lars_d_stats_medians", C.byref(args), C.c_void_p(pairs_dev.ptr), C.c_void_p(scratch.ptr))
                _ffi.call("lars_synchronize", stream)
                med = self._medians_from_pairs(pairs_dev.download(np.float32, (self.ntiles, 2, 2)), indices)
                pairs_dev.free()
                scratch.free()
            else:                                           # two of the three indices: separate statistics pass
                self.run_fused(self.fused_args(indices, white_balance, stats, hist, None, stream, sumsq=sumsq))
                med = self.tile_medians(indices, white_balance, stream)
            rec = stats.download(STATS_DTYPE, (self.ntiles, 3))
            stats.free()
            return rec, med
        own_outputs = None
        if medians and (outputs is None or any(outputs.index[INDEX_IDS[t]] is None for t in indices)):
            own_outputs = outputs = self.make_outputs(indices=indices, index=True, ring=min(self.ntiles, 16))
        med_dev = sel = None
        if medians:
            med_dev = DeviceBuffer(self.ntiles * 3 * 2 * 4)
            med_dev.zero()
            sel = DeviceBuffer(outputs.slots * int(_ffi.load().lars_select_scratch_bytes()))
        chunk = self.ntiles if outputs is None else outputs.slots
        for start in range(0, self.ntiles, chunk):
            count = min(chunk, self.ntiles - start)
            self.run_fused(self.fused_args(indices, white_balance, stats, hist, outputs, stream, start, count, sumsq=sumsq))
            if medians:
                for t in indices:
                    k = INDEX_IDS[t]
                    # medians land in med_dev[k][tile][2]
                    _ffi.call("lars_d_median_pair_batch_f32", C.c_void_p(outputs.index[k].ptr), self.npix, count, self.npix,
                              C.c_void_p(med_dev.ptr + (k * self.ntiles + start) * 8), C.c_void_p(sel.ptr), stream)
        _ffi.call("lars_synchronize", stream)
        rec = stats.download(STATS_DTYPE, (self.ntiles, 3))
        stats.free()
        if not medians:
            return rec
        pairs = med_dev.download(np.float32, (3, self.ntiles, 2))
        med = ((pairs[:, :, 0] + pairs[:, :, 1]) / np.float32(2)).astype(np.float64).T.copy()   # float32 mean of the middles
        for t in INDEX_NAMES:
            if t not in indices:
                med[:, INDEX_IDS[t]] = np.nan
        med_dev.free()
        sel.free()
        if own_outputs is not None:
            own_outputs.free()
        return rec, med

    def free(self):
        for b in (self.tiles, self.hist, self.table, self.percentiles):
            if b is not None:
                b.free()

    def tile_medians(self, indices=INDEX_NAMES, white_balance=True, stream=None):
        """float64[ntiles, 3]: np.median of every tile's index planes, none of which is written
        (``lars_d_quotient_median_pairs``: per-tile radix select on recomputed values, all on the device)."""
        if white_balance and self.table is None:
            raise RuntimeError("compute_wb_tables() first")
        pairs_dev = DeviceBuffer(self.ntiles * 4 * 4)
        scratch = DeviceBuffer(int(_ffi.load().lars_quotient_median_scratch_bytes(self.ntiles)))
        _ffi.call("lars_d_quotient_median_pairs", C.c_void_p(self.tiles.ptr), self.ntiles, self.npix, self.channels, self.code,
                  C.c_void_p(self.table.ptr) if white_balance else None, C.c_void_p(pairs_dev.ptr), C.c_void_p(scratch.ptr), stream)
        _ffi.call("lars_synchronize", stream)
        pairs = pairs_dev.download(np.float32, (self.ntiles, 2, 2))
        pairs_dev.free()
        scratch.free()
        return self._medians_from_pairs(pairs, indices)

    def _medians_from_pairs(self, pairs, indices):
        """float32[ntiles][2 streams][2 middle values] -> float64[ntiles, 3] (np.median semantics; NDWI = -GNDVI)."""
        mid = ((pairs[:, :, 0] + pairs[:, :, 1]) / np.float32(2)).astype(np.float32)          # float32 mean of the middles
        med = np.full((self.ntiles, 3), np.nan, dtype=np.float64)
        for t in indices:
            k = INDEX_IDS[t]
            med[:, k] = mid[:, 0] if t == "NDVI" else (mid[:, 1] if t == "GNDVI" else np.float32(0) - mid[:, 1])
            if np.isnan(med[:, k]).any():
                raise RuntimeError("exact median select did not settle (values outside the uint8 quotient domain?)")
        return med

    # -- exact medians of the whole batch (all tiles, all ranks) ------------
    def digit_histogram(self, first, bias, shift, white_balance=True, stream=None):
        """One radix-select pass (``lars_d_quotient_digit_hist``): uint64[2 streams][2 tracks][2048]."""
        if self.code != _ffi.U8 or self.channels != 3:
            raise TypeError("exact batch medians need uint8 tiles with 3 channels")
        if white_balance and self.table is None:
            raise RuntimeError("compute_wb_tables() first")
        if getattr(self, "_selq", None) is None:
            self._selq = DeviceBuffer(2 * 2 * SELECT_BINS * 8)
        self._selq.zero()
        b = np.ascontiguousarray(bias, dtype=np.uint32).reshape(4)
        sh = np.ascontiguousarray(shift, dtype=np.uint32).reshape(4)
        _ffi.call("lars_d_quotient_digit_hist", C.c_void_p(self.tiles.ptr), self.ntiles, self.npix, self.channels, self.code,
                  C.c_void_p(self.table.ptr) if white_balance else None, int(bool(first)), _ffi.ptr(b), _ffi.ptr(sh),
                  C.c_void_p(self._selq.ptr), stream)
        _ffi.call("lars_synchronize", stream)
        return self._selq.download(np.uint64, (2, 2, SELECT_BINS))

    def global_medians(self, indices=INDEX_NAMES, white_balance=True, comm=None, recompute_tables=False):
        """``np.median`` of each index over ALL pixels of ALL tiles of ALL ranks, exactly, without writing a plane:
        three radix-select passes (2048 linear buckets, then two 11-bit digits of the key range of the chosen bucket)
        that recompute the index values from the tiles (3 bytes per pixel and pass) and one small all-reduce per pass
        (SURVEY.md 8(e))."""
        if white_balance and (recompute_tables or self.table is None):
            self.compute_wb_tables()
        n_local = self.ntiles * self.npix
        keys = select_order_statistics(lambda first, bias, shift: self.digit_histogram(first, bias, shift, white_balance),
                                       n_local, comm, min_abs=1.0 / 510.0)
        return medians_from_keys(keys, indices)


class BatchOutputs:
    """Device output planes of a batch (optionally a ring of ``slots`` tiles)."""

    def __init__(self, batch, indices, index, wb, rgba, ring=None):
        from .api import colormap_lut, _colormap_for
        self.slots = batch.ntiles if not ring else min(int(ring), batch.ntiles)
        self.index, self.rgba, self.luts = [None] * 3, [None] * 3, [None] * 3
        for t in indices:
            k = INDEX_IDS[t]
            if index:
                self.index[k] = DeviceBuffer(self.slots * batch.npix * 4)
            if rgba:
                self.rgba[k] = DeviceBuffer(self.slots * batch.npix * 4)
                self.luts[k] = DeviceBuffer(1024)
                self.luts[k].upload(colormap_lut(_colormap_for(t)))
        self.wb = DeviceBuffer(self.slots * batch.npix * batch.channels) if wb else None
        self.batch = batch

    def host_index(self, index_type, slot=0, count=1):
        k = INDEX_IDS[index_type]
        b = self.batch
        return self.index[k].download(np.float32, (count, b.h, b.w), slot * b.npix * 4)

    def host_rgba(self, index_type, slot=0, count=1):
        k = INDEX_IDS[index_type]
        b = self.batch
        return self.rgba[k].download(np.uint8, (count, b.h, b.w, 4), slot * b.npix * 4)

    def host_wb(self, slot=0, count=1):
        b = self.batch
        return self.wb.download(np.uint8, (count, b.h, b.w, b.channels), slot * b.npix * b.channels)

    def free(self):
        for b in self.index + self.rgba + self.luts + [self.wb]:
            if b is not None:
                b.free()


# ---------------------------------------------------------------------------
# folding records
# ---------------------------------------------------------------------------
def merge_records(records):
    """Fold a 1-D structured array of STATS_DTYPE records (one index) into one
    record, in order, through ``lars_stats_merge`` (host C; no GPU needed)."""
    rec = np.ascontiguousarray(records, dtype=STATS_DTYPE).reshape(-1)
    out = np.zeros(1, dtype=STATS_DTYPE)
    _ffi.call("lars_stats_merge", _ffi.ptr(rec), rec.size, _ffi.ptr(out))
    return out[0]


def summarize(record):
    """Global statistics of a merged record (SURVEY.md 8(e) semantics)."""
    count = int(record["count"])
    mean = float(record["sum"]) / count
    sumsq = float(record["sumsq"])
    # the fused kernel fills sumsq only with LARS_F_SUMSQ; without it std is unknown
    have_sq = sumsq != 0.0 or (float(record["min"]) == 0.0 and float(record["max"]) == 0.0)
    var = max(sumsq / count - mean * mean, 0.0)
    return {
        "count": count,
        "mean": mean,
        "std": var ** 0.5 if have_sq else None,
        "min": float(record["min"]),
        "max": float(record["max"]),
        "coverage": int(record["above"]) / count * 100.0,
        "hist": np.array(record["hist"], dtype=np.int64),
    }


def local_fold(tile_records, indices=INDEX_NAMES):
    """[ntiles, 3] per-tile records -> [3] per-index records of this rank."""
    out = np.zeros(3, dtype=STATS_DTYPE)
    for t in indices:
        k = INDEX_IDS[t]
        out[k] = merge_records(tile_records[:, k])
    return out


def timeseries_rows(records, medians, index_type, dates=None):
    """Rows of the reference's time-series table (process-images.py:646-658: ``Date``, ``Mean``, ``Median``,
    ``Min``, ``Max``, ``<feature> Coverage (%)``) for every tile of a processed batch -- one pass over
    the batch instead of one ``calculate_index`` + five NumPy reductions per image."""
    k = INDEX_IDS[index_type]
    feature = "Water" if index_type == "NDWI" else "Vegetation"
    rows = []
    for i in range(records.shape[0]):
        r = records[i, k]
        count = int(r["count"])
        rows.append({
            "Date": None if dates is None else dates[i],
            "Mean": float(r["sum"]) / count,
            "Median": float(medians[i, k]),
            "Min": float(r["min"]),
            "Max": float(r["max"]),
            f"{feature} Coverage (%)": int(r["above"]) / count * 100,
        })
    return rows


# ---------------------------------------------------------------------------
# exact order statistics across tiles and ranks (radix select on recomputed values)
# ---------------------------------------------------------------------------
SELECT_BINS = 2048
SELECT_DIGITS = 1984                                       # usable bins of a later pass (the kernel keeps 64 dummy words per row)
SELECT_DIGIT_BITS = 10                                     # 2^10 <= SELECT_DIGITS
KEY_MINUS1, KEY_PLUS1, KEY_ZERO = 0x407FFFFF, 0xBF800000, 0x80000000     # order-preserving keys of -1.0, +1.0, +0.0


def key_to_float32(key):
    """Inverse of the kernels' order-preserving key (x >= 0: bits | 2^31; x < 0: ~bits)."""
    key = np.uint32(key)
    bits = (key & np.uint32(0x7FFFFFFF)) if (key & np.uint32(0x80000000)) else ~key
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def float32_to_key(x):
    bits = int(np.array([x], dtype=np.float32).view(np.uint32)[0])
    return (~bits & 0xFFFFFFFF) if bits >> 31 else (bits | 0x80000000)


def select_bucket(x):
    """First-level bucket of x in [-1, 1], the kernels' ``selq_bucket``: the low 23 bits of
    float32(fma(x, 1023.5, 1023.5) + 2^23).  (The float64 product and sum are exact, so one rounding = the fma.)"""
    t = np.float32(np.float64(np.float32(x)) * 1023.5 + 1023.5)
    u = np.float32(t + np.float32(8388608.0))
    return int(np.array([u], dtype=np.float32).view(np.uint32)[0] & 0x7FFFFF)


def bucket_lower_key(b):
    """Smallest key in [key(-1), key(+1) + 1] whose bucket is >= b (the bucket function is monotone in x)."""
    lo, hi = KEY_MINUS1, KEY_PLUS1 + 1
    while lo < hi:
        mid = (lo + hi) // 2
        if select_bucket(key_to_float32(mid)) >= b:
            hi = mid
        else:
            lo = mid + 1
    return lo


def select_order_statistics(pass_fn, n_local, comm=None, min_abs=0.0, max_passes=8):
    """Keys of the two middle order statistics (ranks (N-1)//2 and N//2) of two value streams in [-1, 1].

    ``pass_fn(first, bias[4], shift[4]) -> uint64[2][2][SELECT_BINS]`` counts on this rank: first pass, the
    linear bucket of every value (under track 0); later passes, bin ``(key - bias) >> shift`` (below
    ``SELECT_DIGITS``) of the keys inside the chosen range -- under track 0 only when both streams' tracks share
    (bias, shift).
    Histograms are summed over ranks through ``comm.allreduce_f64`` (counts < 2^53 are exact in float64), every
    rank then picks the same bins.  ``min_abs``: the values are 0 or at least that large in magnitude (uint8
    quotients: 1/510), which cuts the bucket around zero down to the key of +0.0.  Returns uint32[2][2].
    """
    tot = np.array([float(n_local)])
    n_total = int((comm.allreduce_f64(tot, "sum") if comm is not None else tot)[0])
    ranks = np.array([[(n_total - 1) // 2, n_total // 2]] * 2, dtype=np.int64)     # [stream][track]
    bias = np.zeros((2, 2), dtype=np.uint32)
    shift = np.zeros((2, 2), dtype=np.uint32)
    first = True
    for _ in range(max_passes):
        local = np.asarray(pass_fn(first, bias.reshape(4), shift.reshape(4)), dtype=np.uint64).reshape(2, 2, SELECT_BINS)
        hist = local.astype(np.float64).reshape(-1)
        if comm is not None:
            hist = comm.allreduce_f64(hist, "sum")
        hist = np.asarray(hist).reshape(2, 2, SELECT_BINS).astype(np.int64)
        if first:
            hist[:, 0] += hist[:, 1]                        # bucket pass: counted under track 0
            hist[:, 1] = hist[:, 0]
        else:
            hist[:, :, SELECT_DIGITS:] = 0                  # not part of a later pass's histogram
            if (bias[:, 0] == bias[:, 1]).all() and (shift[:, 0] == shift[:, 1]).all():
                hist[:, 1] = hist[:, 0]                     # both streams' tracks shared: only track 0 was counted
        last = not first and not shift.any()
        for s in range(2):
            for t in range(2):
                cum = np.cumsum(hist[s, t])
                d = int(np.searchsorted(cum, ranks[s, t], side="right"))
                if d >= SELECT_BINS:
                    raise RuntimeError("radix select: rank beyond the histogram mass (inconsistent passes)")
                ranks[s, t] -= int(cum[d - 1]) if d else 0
                if first:
                    lo = bucket_lower_key(d)
                    hi = KEY_PLUS1 + 1 if d >= SELECT_BINS - 1 else bucket_lower_key(d + 1)
                    if min_abs > 0 and lo <= KEY_ZERO < hi and float32_to_key(-min_abs) < lo and hi <= float32_to_key(min_abs):
                        lo, hi = KEY_ZERO, KEY_ZERO + 1     # only +0.0 lives there
                    span = hi - lo - 1
                    sh = 0
                    while (span >> sh) >= SELECT_DIGITS:    # the first digit must fit a row
                        sh += 1
                    bias[s, t] = lo
                    shift[s, t] = sh
                else:
                    bias[s, t] = np.uint32(int(bias[s, t]) + (d << int(shift[s, t])))
                    shift[s, t] = max(int(shift[s, t]) - SELECT_DIGIT_BITS, 0)
        if last:
            return bias
        first = False
    raise RuntimeError("radix select did not finish")


def medians_from_keys(keys, indices=INDEX_NAMES):
    """np.median semantics (mean of the two middle values in float32); NDWI = -GNDVI shares GNDVI's statistics."""
    out = {}
    for t in indices:
        s = 0 if t == "NDVI" else 1
        a, b = key_to_float32(keys[s, 0]), key_to_float32(keys[s, 1])
        m = np.float32(np.float32(a + b) / np.float32(2))
        out[t] = float(np.float32(0) - m) if t == "NDWI" else float(m)
    return out
