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
import os
import time

import numpy as np

from . import _ffi
from ._ffi import FusedArgs, INDEX_IDS, INDEX_NAMES, STATS_DTYPE, DeviceBuffer, DeviceSlice

MAX_TILES_PER_LAUNCH = 65535      # grid.y limit
ARENA_MIN_BYTES = 2 << 30         # smaller arenas run alike wherever they land
ARENA_SPREAD_PLANE_BYTES = 2 << 30  # planes from this size on are worth a placement search (and its spare room)
ARENA_TRIALS = 4                  # allocations of the default search at most (each is tried with every placement of its planes)
ARENA_CLASS_GAP = 0.93            # the search ends once its best candidate is 7 % under its worst: both classes seen (they are ~18 % apart)
ARENA_WARM_MS = 30.0              # untimed launches before a candidate is timed: after an idle gap a fast arena needs ~22 ms to reach its level
ARENA_SPAN_BYTES = 20 << 30       # how far from the first planes the last ones may be placed inside an allocation (the memory changes kind every 6-16 GiB)
ARENA_SPAN_STEP = 4 << 30         # ... in steps of
ARENA_CROSS_TRIALS = 6            # pairs of allocations tried with the planes split between them when every allocation is of one kind
ARENA_EXTRA_BLOCKS = 24           # ... and after those, small allocations for the second half of the planes alone: stretches of one kind
                                  # reach 72 GiB in some processes (tools/lab/kindmap.py), and the allocations held so far have used most of one


# How statistics-only passes over uint8 RGNir batches run (no output planes):
#   "joint"   one read of the tiles: joint byte-pair histograms in LDS, everything else -- the white balance's percentiles
#             included -- derived from the counts (lars_d_stats_joint, csrc/joint.hip)
#   "classic" channel-histogram pass, then the per-pixel statistics kernel (and its median passes)
#   "auto"    with medians: joint; without: whichever of the two measures faster on the first tiles of the batch
#             (TileBatch.pick_stats_route; small batches: joint)
# Records are identical bit for bit, with ONE exception: the optional sum of squares (``sumsq=True`` / LARS_F_SUMSQ) is
# accumulated per pixel on the classic route and per cell (count x value^2) on the one-read route and may differ by a few
# units of 2^-32, so "auto" never decides by measurement when sumsq is requested (it takes the one-read route whenever that
# route can serve the batch): a given batch always gets the same bytes.  LARS_STATS_ROUTE presets the route.
_STATS_ROUTE = os.environ.get("LARS_STATS_ROUTE", "auto")


def set_stats_route(route):
    global _STATS_ROUTE
    if route not in ("auto", "joint", "classic"):
        raise ValueError("route must be auto, joint or classic")
    _STATS_ROUTE = route


def get_stats_route():
    return _STATS_ROUTE


def channels_of(indices, whole_image=False):
    """Channels (0 red, 1 green, 2 NIR) whose white-balance tables a pass over ``indices`` reads."""
    if whole_image:
        return {0, 1, 2}
    need = set()
    for t in indices:
        need |= {2, 0} if t == "NDVI" else {2, 1}
    return need


def arena_placements(nplanes, plane_bytes, nbytes):
    """Byte offsets of ``nplanes`` planes of ``plane_bytes`` inside an allocation of ``nbytes`` that ``TileBatch.make_outputs`` tries:
    packed back to back, then the first ceil(n / 2) planes packed at the start and the rest packed from 8, 12, 16, 20 GiB on
    (every multiple of ARENA_SPAN_STEP beyond the first cluster up to ARENA_SPAN_BYTES that still fits).  Device memory changes kind
    every 6-16 GiB along an allocation and a launch is fast when its planes are split between the kinds (make_outputs)."""
    n_first = (nplanes + 1) // 2
    first_bytes, second_bytes = n_first * plane_bytes, (nplanes - n_first) * plane_bytes
    packed = tuple(j * plane_bytes for j in range(nplanes))
    out = [packed]
    if nplanes < 2:
        return out
    start = ((first_bytes + ARENA_SPAN_STEP - 1) // ARENA_SPAN_STEP) * ARENA_SPAN_STEP
    for s0 in range(start, ARENA_SPAN_BYTES + 1, ARENA_SPAN_STEP):
        if s0 > first_bytes and s0 + second_bytes <= nbytes:
            out.append(packed[:n_first] + tuple(s0 + j * plane_bytes for j in range(nplanes - n_first)))
    return out


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
        self._table_channels = set()       # channels whose tables are valid (a one-read pass fills only those its indices read)
        self._hist_channels = set()        # channels whose 256-bin histograms are valid (a one-read pass fills them only on request)
        self._rgn_variant = 0              # flavour of the tables in force (process-rgn.py's extra inner clip = 1)

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
        self._table_channels = {0, 1, 2}
        self._hist_channels = {0, 1, 2} if self.code == _ffi.U8 else set()
        self._rgn_variant = int(rgn_variant)
        return self

    def _need_channels(self, partial):
        """The host_* getters hand out all three channels: after a one-read pass that served NDVI only (or GNDVI / NDWI
        only) the third channel's rows were never computed -- refuse unless the caller asks for them as they are."""
        if self.table is None:
            raise RuntimeError("compute_wb_tables() first")
        if not partial and self._table_channels != {0, 1, 2}:
            missing = sorted({0, 1, 2} - self._table_channels)
            raise RuntimeError(f"white-balance tables of channel(s) {missing} are not valid: the last pass computed only the "
                               f"channels its indices read (compute_wb_tables() for all three, or partial=True)")

    def host_tables(self, partial=False):
        self._need_channels(partial)
        blob = self.table.download(np.uint8, (self.ntiles, self.table_bytes))
        return np.ascontiguousarray(blob[:, :3 * self.nvalues]).reshape(self.ntiles, 3, self.nvalues)

    def host_percentiles(self, partial=False):
        self._need_channels(partial)
        return self.percentiles.download(np.float64, (self.ntiles, 3, 2))

    def host_hist(self, partial=False):
        """uint8 batches only (uint16 percentiles come from two radix levels, no full histogram).  After a one-read pass
        (``run_joint`` / ``process`` without planes) the histograms exist only if it was asked for them (``channel_hist=True``):
        without them the pass may count red and green clamped to a window around their percentiles (csrc/joint_win.hip)."""
        self._need_channels(partial)
        if self.hist is None or (not partial and self._hist_channels != {0, 1, 2}):
            missing = sorted({0, 1, 2} - self._hist_channels)
            raise RuntimeError(f"channel histograms of channel(s) {missing} are not valid: the last one-read pass was not asked for them "
                               f"(compute_wb_tables(), or run_joint(..., channel_hist=True); partial=True hands out what there is)")
        return self.hist.download(np.uint32, (self.ntiles, 3, 256))

    # -- pass 2: the fused kernel ------------------------------------------
    def make_outputs(self, indices=INDEX_NAMES, index=False, wb=False, rgba=False, ring=None, placement_trials=None, arena="auto",
                     pick="fastest"):
        """Allocate output planes.  ``ring`` < ntiles reuses a ring of that many
        tile slots (same HBM traffic, bounded footprint) -- see BatchOutputs.

        The float32 index planes (and RGBA8 planes) live in ONE allocation (an arena).  How fast the write-bound fused
        kernel runs into a multi-GiB arena depends on WHERE its planes lie: device memory comes in two kinds that alternate
        along an allocation in stretches of 6-16 GiB, and a launch whose write streams are split between the kinds runs 18 %
        faster than one whose planes all lie in one kind (2.49-2.52 against 3.01-3.15 ms per 64-tile launch of the headline
        kernel; profiles/r04_arena_two_kinds.txt).  Three 4 GiB planes packed into 12 GiB see a change of kind in three
        allocations of ten; with room to spare inside the allocation a placement that does can be found in three of four.
        ``arena``:
          "auto"      multi-GiB arenas: ONE allocation with up to ARENA_SPAN_BYTES of room beyond the first planes (24 GiB for
                      three planes of 4 GiB, less if the device is short of memory), and the planes are tried in a handful of
                      placements inside it -- packed, and with the second half of the planes further out in steps of 4 GiB up
                      to 20 GiB from the start -- each timed with the batch's own launches (``_probe_arena``).  If no placement is 7 % faster than another the
                      allocation is of one kind throughout: another one is taken (up to ``placement_trials``, default
                      ARENA_TRIALS), and the search ends as soon as both classes have been seen.  If every allocation was of one
                      kind (two fresh processes in ten on some boxes), the first half of the planes stays in one allocation and
                      the rest goes to another (allocations differ in kind among each other): up to ARENA_CROSS_TRIALS pairs;
                      such outputs keep BOTH allocations.  The fastest (allocation, placement) is kept, other allocations are freed.  Smaller arenas (they run alike wherever they land)
                      are one packed allocation
          "plain"     one packed allocation as it comes, unless ``placement_trials`` asks for a search among packed ones
        ``pick="slowest"`` keeps the slowest candidate instead (a diagnostic: what a process without a fast arena sees).
        ``outs.arena_report`` = {kind, search_ms, chosen_ms, post_free_ms, rejected, candidate_ms, placements, ...}."""
        outs = BatchOutputs(self, indices, index, wb, rgba, ring, allocate=False)
        nplanes = len(outs._index_ids) + len(outs._rgba_ids)
        report = {"kind": "none"} if not nplanes else None
        if nplanes and arena not in ("auto", "plain"):
            raise ValueError("arena must be auto or plain")
        if pick not in ("fastest", "slowest"):
            raise ValueError("pick must be fastest or slowest")
        packed_bytes = nplanes * outs.plane_bytes
        big = bool(nplanes) and nplanes * outs.slots * self.npix * 4 >= ARENA_MIN_BYTES
        # Room inside the allocation and several placements: only where it can pay -- two or more planes of ARENA_SPREAD_PLANE_BYTES
        # (2 GiB) or more each (planes of 1 GiB show no difference between placements: profiles/r04_arena_two_kinds.txt) -- and only
        # if the caller did not ask for one trial at most.  Such an arena KEEPS its spare room while the outputs live:
        # report["arena_bytes"] against the packed size (24 instead of 12 GiB for three planes of 4 GiB).
        spread = (arena == "auto" and big and nplanes >= 2 and outs.plane_bytes >= ARENA_SPREAD_PLANE_BYTES
                  and (placement_trials is None or int(placement_trials) > 1))
        if placement_trials is None:
            # one plane: nothing to split -- sixteen 4 GiB single-plane arenas measured within 1 % of each other (profiles/r04_ndvi_plane_step_ways.txt)
            placement_trials = ARENA_TRIALS if spread else 0
        if not nplanes or (placement_trials <= 1 and not spread):
            if nplanes:
                outs.adopt_arena(DeviceBuffer(packed_bytes))
                report = {"kind": "plain hipMalloc", "search_ms": 0.0, "chosen_ms": None, "post_free_ms": None, "rejected": 0}
            outs.arena_report = report
            return outs
        t_search = time.perf_counter()
        free_b, total_b = C.c_size_t(), C.c_size_t()
        # Two clusters: ceil(n / 2) planes at the start, the rest further out.  The other split of an odd number ((0, 16, 20) for three planes)
        # was measured too: 1 % slower than (0, 4, 16) although it balances the launch's four streams, the read included, more often -- and
        # the 28 GiB it needs came from one kind of memory throughout in three of six fresh processes, where the first 24 GiB of a fresh
        # process showed both classes in eleven of thirteen (profiles/r04_arena_fresh_processes.txt).
        n_first = (nplanes + 1) // 2                                       # planes of the first cluster; the rest form the second
        second_bytes = (nplanes - n_first) * outs.plane_bytes

        def placements_for(nbytes):
            return arena_placements(nplanes, outs.plane_bytes, nbytes)

        stats = self.new_stats()
        cands = []                                                         # (ms, allocation index, offsets)
        arenas, malloc_ms = [], []
        stopped = "placement_trials"
        try:
            while len(arenas) < max(1, int(placement_trials)):
                _ffi.call("lars_mem_info", C.byref(free_b), C.byref(total_b))
                want = packed_bytes
                if spread:
                    want = max(packed_bytes, min(ARENA_SPAN_BYTES + second_bytes, free_b.value - (16 << 30)))
                if free_b.value < want + (8 << 30):                        # keep 8 GiB of headroom for the caller
                    if arenas:
                        stopped = "device memory"
                        break
                    want = packed_bytes                                    # the first arena must exist whatever the headroom
                t0 = time.perf_counter()
                try:
                    buf = DeviceBuffer(want)
                except _ffi.LarsError:
                    if not arenas:
                        raise
                    stopped = "device memory"
                    break
                malloc_ms.append((time.perf_counter() - t0) * 1e3)
                arenas.append(buf)
                for k, offsets in enumerate(placements_for(want)):
                    outs.adopt_arena(buf, offsets)
                    cands.append((self._probe_arena(outs, indices, stats, warm_ms=ARENA_WARM_MS if k == 0 else 5.0), len(arenas) - 1, offsets))
                times = [c[0] for c in cands]
                if len(cands) >= 2 and min(times) <= ARENA_CLASS_GAP * max(times):
                    stopped = "both classes seen"
                    break
            # Every allocation of ONE kind throughout (no placement 7 % under another; seen for all four 24 GiB allocations of some
            # processes: profiles/r05_arena_first_process.txt): allocations differ in kind among each other -- their slow levels do,
            # 3.01 against 3.09 ms -- so the first cluster of planes stays in one and the rest goes to another, tried from the pair
            # whose levels lie furthest apart.  Both allocations are then kept.
            times = [c[0] for c in cands]
            if (spread and pick == "fastest" and stopped != "both classes seen" and len(arenas) >= 2
                    and min(times) > ARENA_CLASS_GAP * max(times)):
                level = [float(np.mean([t for t, a, _ in cands if a == j])) for j in range(len(arenas))]
                pairs = sorted(((abs(level[i] - level[j]), i, j) for i in range(len(arenas)) for j in range(len(arenas)) if i != j), reverse=True)
                first = tuple(j * outs.plane_bytes for j in range(n_first))
                second = tuple(j * outs.plane_bytes for j in range(nplanes - n_first))
                for _, i, j in pairs[:ARENA_CROSS_TRIALS]:
                    outs.adopt_two_arenas(arenas[i], first, arenas[j], second)
                    cands.append((self._probe_arena(outs, indices, stats, warm_ms=5.0), (i, j), first + second))
                    if cands[-1][0] <= ARENA_CLASS_GAP * max(times):
                        stopped = "both classes seen: planes split between two allocations"
                        break
                # ... and if all of them are of the SAME kind: small allocations for the second cluster alone (while the large ones are
                # held they come from other memory), next to the first cluster in the first allocation
                while stopped == "placement_trials" and len(arenas) < int(placement_trials) + ARENA_EXTRA_BLOCKS:
                    _ffi.call("lars_mem_info", C.byref(free_b), C.byref(total_b))
                    if free_b.value < second_bytes + (8 << 30):
                        break
                    t0 = time.perf_counter()
                    try:
                        arenas.append(DeviceBuffer(second_bytes))
                    except _ffi.LarsError:
                        break
                    malloc_ms.append((time.perf_counter() - t0) * 1e3)
                    outs.adopt_two_arenas(arenas[0], first, arenas[-1], second)
                    cands.append((self._probe_arena(outs, indices, stats, warm_ms=5.0), (0, len(arenas) - 1), first + second))
                    if cands[-1][0] <= ARENA_CLASS_GAP * max(times):
                        stopped = "both classes seen: second half of the planes in an allocation of its own"
        except BaseException:
            # a failed probe launch or allocation: nothing of the search may stay behind, and `outs` must not point into a freed arena
            _ffi.call("lars_synchronize", None)
            outs.arena = outs.arena2 = None
            for buf in arenas:
                buf.free()
            stats.free()
            raise
        times = [c[0] for c in cands]
        best = int(np.argmin(times) if pick == "fastest" else np.argmax(times))
        chosen_ms, chosen_alloc, chosen_offsets = cands[best]
        keep = set(chosen_alloc) if isinstance(chosen_alloc, tuple) else {chosen_alloc}
        if isinstance(chosen_alloc, tuple):
            outs.adopt_two_arenas(arenas[chosen_alloc[0]], chosen_offsets[:n_first], arenas[chosen_alloc[1]], chosen_offsets[n_first:])
        else:
            outs.adopt_arena(arenas[chosen_alloc], chosen_offsets)
        for j, buf in enumerate(arenas):
            if j not in keep:
                buf.free()
        _ffi.call("lars_synchronize", None)
        # the survivor once more, now that the rejected allocations are gone: the figure the steps should reproduce
        post_free = self._probe_arena(outs, indices, stats) if len(arenas) > 1 else float(chosen_ms)
        stats.free()
        gib = float(1 << 30)
        kept_bytes = sum(arenas[j].nbytes for j in keep)
        outs.placement_ms = {"arenas": [float(x) for x in times], "chosen": float(chosen_ms)}
        outs.arena_report = {
            "kind": f"plain hipMalloc of {kept_bytes / gib:.1f} GiB{' in two allocations' if len(keep) > 1 else ''}, the {pick} of {len(cands)} "
                    f"placements of the planes in {len(arenas)} allocation(s), timed with the batch's own launches (search ended by: {stopped})",
            "search_ms": (time.perf_counter() - t_search) * 1e3, "chosen_ms": float(chosen_ms), "post_free_ms": float(post_free),
            "rejected": len(arenas) - len(keep), "candidate_ms": [float(x) for x in times], "malloc_ms": [float(x) for x in malloc_ms],
            "placements": [{"allocation": list(a) if isinstance(a, tuple) else int(a), "offsets_gib": [round(o / gib, 3) for o in offs], "ms": float(t)}
                           for t, a, offs in cands],
            "chosen_offsets_gib": [round(o / gib, 3) for o in chosen_offsets],
            "arena_bytes": int(kept_bytes), "packed_bytes": int(packed_bytes), "allocations": len(arenas),
            "transient_bytes": int(sum(b.nbytes for b in arenas)),
            "probe": f">= {ARENA_WARM_MS:.0f} ms of untimed launches per allocation, then per placement one timed pass of launches over the batch's chunks"}
        return outs

    def _probe_arena(self, outs, indices, stats, warm_ms=ARENA_WARM_MS):
        """Milliseconds per fused launch into ``outs`` at the level a step sees: the step's own launch sequence (one launch
        per ring of tile slots over the batch, LARS_F_RAW records; sixteen evenly spaced chunks of a longer sequence).

        After an idle gap -- a 12 GiB hipMalloc takes 0.25-0.36 s once the first ~150 GiB are handed out -- launches into a
        FAST arena run at the slow class's level and come down over about eight launches / 22 ms (2.6, 3.1-3.2, 3.0, 2.83,
        2.7, 2.67, 2.63, 2.57, 2.53 -> 2.50 ms; slow arenas show no such ramp): profiles/r04_arena_probe_phases.txt, phases
        B and C.  Round 3's probe (one warm-up + three launches right after the allocation) therefore read 2.94-3.04 ms for
        every candidate whenever the allocations were slow, whatever their class.  Hence: untimed launches until
        ``warm_ms`` of device time have passed (less for a further placement in an allocation that has just been probed), then
        one timed pass; and the pass covers every chunk of the batch because the level also moves by 2-4 % with the input
        region a launch reads (the input changes kind along its length too)."""
        count = min(outs.slots, self.ntiles)
        starts = list(range(0, self.ntiles, count))
        if len(starts) > 16:
            starts = [starts[int(round(i * (len(starts) - 1) / 15.0))] for i in range(16)]
        mask = 0
        for t in indices:
            mask |= 1 << INDEX_IDS[t]
        wb_on = self.table is not None
        launches = [self.fused_args(indices, wb_on, stats, False, outs, None, st, min(count, self.ntiles - st), raw=True) for st in starts]
        while len(launches) < 4:
            launches = launches + launches
        ev = [C.c_void_p(), C.c_void_p()]
        for e in ev:
            _ffi.call("lars_event_create", C.byref(e))
        ms = C.c_float(0)

        def one_pass():
            _ffi.call("lars_event_record", ev[0], None)
            for a in launches:
                self.run_fused(a)
            _ffi.call("lars_event_record", ev[1], None)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))     # synchronises on ev[1]
            return float(ms.value)

        try:
            _ffi.call("lars_d_stats_begin", C.c_void_p(stats.ptr), self.ntiles, mask, None)
            warmed, passes = 0.0, 0
            while warmed < warm_ms and passes < 8:
                warmed += one_pass()
                passes += 1
            timed = one_pass()
            _ffi.call("lars_d_stats_end", C.c_void_p(stats.ptr), self.ntiles, mask, self.npix, None)
            _ffi.call("lars_synchronize", None)
        finally:
            for e in ev:                                                      # also when a launch failed: the events do not leak
                _ffi.call("lars_event_destroy", e)
        return timed / len(launches)

    def fused_args(self, indices=INDEX_NAMES, white_balance=True, stats=None, hist=False, outputs=None,
                   stream=None, tile_start=0, tile_count=None, sumsq=False, raw=False):
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
                   (_ffi.F_SUMSQ if (stats is not None and sumsq) else 0) | (_ffi.F_RAW if (stats is not None and raw) else 0))
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

    def fold_stats(self, stats, indices=INDEX_NAMES, out=None, stream=None):
        """Per-index fold of this batch's per-tile records ON THE DEVICE (``lars_d_stats_fold``): ``out`` (a DeviceBuffer of
        3 records, allocated when None) receives what ``local_fold`` computes on the host, bit for bit -- the three
        records a rank contributes to the global statistics, so that only 3 x 472 bytes leave the device."""
        mask = 0
        for t in indices:
            mask |= 1 << INDEX_IDS[t]
        if out is None:
            out = DeviceBuffer(3 * STATS_DTYPE.itemsize)
            out.zero(stream)
        _ffi.call("lars_d_stats_fold", C.c_void_p(stats.ptr), self.ntiles, mask, C.c_void_p(out.ptr), stream)
        return out

    def run_fused_chunks(self, indices, white_balance, stats, hist, outputs, stream=None, sumsq=False, launch_events=None):
        """``lars_d_fused`` over the whole batch in launches of ``outputs.slots`` tiles (one launch without a ring); the
        statistics records are opened and closed ONCE around the launches (LARS_F_RAW) instead of by two small kernels per
        launch.  Returns the number of fused launches.  ``launch_events``: event handles, [0] recorded before the first
        launch and [i + 1] after launch i (as many launches as there are handles for): per-launch times for diagnostics."""
        chunk = self.ntiles if outputs is None else outputs.slots
        if chunk >= self.ntiles:
            if launch_events:
                _ffi.call("lars_event_record", launch_events[0], stream)
            self.run_fused(self.fused_args(indices, white_balance, stats, hist, outputs, stream, sumsq=sumsq))
            if launch_events and len(launch_events) > 1:
                _ffi.call("lars_event_record", launch_events[1], stream)
            return 1
        mask = 0
        for t in indices:
            mask |= 1 << INDEX_IDS[t]
        if stats is not None:
            _ffi.call("lars_d_stats_begin", C.c_void_p(stats.ptr), self.ntiles, mask, stream)
        launches = 0
        if launch_events:
            _ffi.call("lars_event_record", launch_events[0], stream)
        for start in range(0, self.ntiles, chunk):
            count = min(chunk, self.ntiles - start)
            self.run_fused(self.fused_args(indices, white_balance, stats, hist, outputs, stream, start, count, sumsq=sumsq,
                                           raw=stats is not None))
            launches += 1
            if launch_events and launches < len(launch_events):
                _ffi.call("lars_event_record", launch_events[launches], stream)
        if stats is not None:
            _ffi.call("lars_d_stats_end", C.c_void_p(stats.ptr), self.ntiles, mask, self.npix, stream)
        return launches

    def run_fused(self, args):
        _ffi.call("lars_d_fused", C.byref(args))

    # -- statistics from one read: joint byte-pair histograms (csrc/joint.hip) ----
    def can_joint(self):
        """What ``lars_d_stats_joint`` serves: uint8 tiles with 3 (RGNir) or 4 (RGBA: alpha ignored) channels."""
        return self.code == _ffi.U8 and self.channels in (3, 4) and (self.ntiles == 1 or self.npix % 4 == 0)

    def run_joint(self, indices, white_balance, stats, hist=False, sumsq=False, pairs=None, stream=None, rgn_variant=0,
                  tile_count=None, channel_hist=False):
        """Enqueue ``lars_d_stats_joint``: final records into ``stats`` and, with ``pairs`` (a DeviceBuffer of
        ntiles * 4 floats), the two middle order statistics of every tile's NDVI / GNDVI values.  With white balance the
        call also leaves ``self.percentiles`` / ``self.table`` filled for the channels the indices read (NIR and red for
        NDVI, NIR and green for GNDVI / NDWI), as ``compute_wb_tables`` would -- and, with ``channel_hist=True``,
        ``self.hist``.  Without the histograms a pass over both value streams may count red and green clamped to windows
        around their percentiles, both pair tables of a tile chunk in ONE workgroup (csrc/joint_win.hip): same records,
        percentiles, tables and medians, one reader per byte instead of two."""
        mask = 0
        for t in indices:
            mask |= 1 << INDEX_IDS[t]
        need = int(_ffi.load().lars_joint_scratch_bytes(self.ntiles, self.npix, mask))
        if getattr(self, "_joint_scratch", None) is None or self._joint_scratch.nbytes < need:
            if getattr(self, "_joint_scratch", None) is not None:
                _ffi.call("lars_synchronize", stream)
                self._joint_scratch.free()
            self._joint_scratch = DeviceBuffer(need)
        if white_balance:
            if self.table is None:
                self.table = DeviceBuffer(self.ntiles * self.table_bytes)
                self.percentiles = DeviceBuffer(self.ntiles * 3 * 2 * 8)
                self.table.zero(stream)
                self.percentiles.zero(stream)
            if channel_hist and self.hist is None:
                self.hist = DeviceBuffer(self.ntiles * 3 * 256 * 4)
                self.hist.zero(stream)
                self._hist_channels = set()
            if tile_count is None or int(tile_count) == self.ntiles:
                # valid rows: those this pass writes, plus what an earlier pass of the same flavour left for the other channels
                keep = self._table_channels if int(rgn_variant) == self._rgn_variant else set()
                self._table_channels = keep | channels_of(indices)
                self._rgn_variant = int(rgn_variant)
                if channel_hist:
                    self._hist_channels = self._hist_channels | channels_of(indices)
            else:
                self._table_channels = set()                # a partial pass leaves the batch's tables in no usable state
                self._hist_channels = set()
        a = FusedArgs()
        a.tiles = self.tiles.ptr
        a.ntiles, a.npix, a.channels, a.dtype = (self.ntiles if tile_count is None else int(tile_count)), self.npix, self.channels, self.code
        a.wb_table = self.table.ptr if white_balance else None
        a.index_mask = mask
        a.flags = _ffi.F_STATS | (_ffi.F_HIST if hist else 0) | (_ffi.F_SUMSQ if sumsq else 0)
        a.stats = stats.ptr
        a.stream = stream
        self._joint_tiles = int(a.ntiles)                   # what the call covers: joint_window_report() reads that many window records
        _ffi.call("lars_d_stats_joint", C.byref(a), 1 if white_balance else 0, int(rgn_variant),
                  C.c_void_p(self.percentiles.ptr) if white_balance else None,
                  C.c_void_p(self.hist.ptr) if white_balance and channel_hist else None,
                  C.c_void_p(pairs.ptr) if pairs is not None else None, C.c_void_p(self._joint_scratch.ptr), self._joint_scratch.nbytes)

    def joint_window_report(self):
        """(tiles the last ``run_joint`` counted on windowed tables, tiles among them whose window missed and that were
        counted again) -- after the pass's stream has finished."""
        w, r = C.c_int64(0), C.c_int64(0)
        if getattr(self, "_joint_scratch", None) is None:
            return 0, 0
        _ffi.call("lars_joint_window_report", C.c_void_p(self._joint_scratch.ptr), getattr(self, "_joint_tiles", self.ntiles), C.byref(w), C.byref(r))
        return int(w.value), int(r.value)

    def joint_window_modes(self):
        """(tiles the last ``run_joint`` counted on full tables by two readers, on windowed red and green rows with NIR whole, on three
        windows) -- after the pass's stream has finished."""
        if getattr(self, "_joint_scratch", None) is None:
            return 0, 0, 0
        counts = (C.c_int64 * 3)()
        _ffi.call("lars_joint_window_modes", C.c_void_p(self._joint_scratch.ptr), getattr(self, "_joint_tiles", self.ntiles), counts)
        return int(counts[0]), int(counts[1]), int(counts[2])

    def check_joint(self, stream=None):
        """After a ``run_joint``: wait for ``stream`` and raise if the counting kernel reported a hand-over list overflow
        (its published counts would be truncated; cannot happen while a workgroup counts at most 2^24 pixels, which the
        chunking guarantees).  Every consumer of ``run_joint`` calls this before it trusts the records."""
        _ffi.call("lars_synchronize", stream)
        if getattr(self, "_joint_scratch", None) is not None and int(self._joint_scratch.download(np.uint32, (1,))[0]):
            raise RuntimeError("lars_d_stats_joint: a workgroup's hand-over list overflowed (a chunk of more than 2^24 pixels?)")

    def pick_stats_route(self, indices, white_balance=True, sample=256, min_pixels=1 << 29):
        """"joint" or "classic" for statistics WITHOUT medians over this batch, by measurement: both routes over the first
        ``sample`` tiles, once per (indices, white balance), remembered.  The one-read route counts byte pairs with LDS
        atomics, which queue up where the neighbouring pixels of a wave share cells (smooth imagery: profiles/
        r03_joint_hist_content.txt), while the per-pixel kernels do not care; with medians the one-read route wins on every
        content measured, so nothing is timed for those.  Small batches (fewer than ``min_pixels`` = 2^29 pixels in the sample)
        take the one-read route unmeasured.  The sample has to be large: the one-read route splits every tile into more chunks the fewer tiles a
        launch has, and each chunk publishes 256 KiB of counts -- over 32 tiles of 4096 x 4096 that overhead is 17 % of the
        input and made round 3's 32-tile sample pick the per-pixel route for content on which the one-read route is twice as
        fast over the batch (profiles/r04_bench_full.json, smooth_content).  The timing runs on tables, percentiles and
        histograms of its own: the batch's (and which of their channels are valid) are left exactly as they were."""
        key = (tuple(sorted(INDEX_IDS[t] for t in indices)), bool(white_balance))
        cache = self.__dict__.setdefault("_route_cache", {})
        if key in cache:
            return cache[key]
        sample = min(int(sample), self.ntiles)
        if sample * self.npix < int(min_pixels) or self.channels != 3:
            cache[key] = "joint"
            return "joint"
        saved = (self.table, self.percentiles, self.hist, set(self._table_channels), self._rgn_variant, set(self._hist_channels))
        self.table = self.percentiles = self.hist = None
        self._table_channels = set()
        self._hist_channels = set()
        ev = [C.c_void_p() for _ in range(3)]
        for e in ev:
            _ffi.call("lars_event_create", C.byref(e))
        stats = self.new_stats()
        ms = {}
        try:
            for _ in range(2):                                  # the first round warms both routes up
                _ffi.call("lars_event_record", ev[0], None)
                self.run_joint(indices, white_balance, stats, tile_count=sample)
                _ffi.call("lars_event_record", ev[1], None)
                if white_balance:
                    if self.hist is None:
                        self.hist = DeviceBuffer(self.ntiles * 3 * 256 * 4)
                    _ffi.call("lars_d_channel_hist", C.c_void_p(self.tiles.ptr), sample, self.npix, self.channels, self.code,
                              C.c_void_p(self.hist.ptr), None)
                    _ffi.call("lars_d_wb_table", C.c_void_p(self.hist.ptr), sample, self.npix, self.code, C.c_void_p(self.table.ptr),
                              C.c_void_p(self.percentiles.ptr), 0, None)
                self.run_fused(self.fused_args(indices, white_balance, stats, False, None, None, 0, sample))
                _ffi.call("lars_event_record", ev[2], None)
                t = C.c_float(0)
                _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(t)); ms["joint"] = t.value
                _ffi.call("lars_event_elapsed_ms", ev[1], ev[2], C.byref(t)); ms["classic"] = t.value
            self.check_joint()
        finally:
            _ffi.call("lars_synchronize", None)
            for e in ev:
                _ffi.call("lars_event_destroy", e)
            stats.free()
            for buf in (self.table, self.percentiles, self.hist):
                if buf is not None:
                    buf.free()
            self.table, self.percentiles, self.hist, self._table_channels, self._rgn_variant, self._hist_channels = saved
        cache[key] = "joint" if ms["joint"] <= ms["classic"] else "classic"
        self._route_ms = dict(ms)
        return cache[key]

    def process(self, indices=INDEX_NAMES, white_balance=True, hist=False, outputs=None, stream=None,
                recompute_tables=True, medians=False, sumsq=False, route=None, rgn_variant=None, channel_hist=False):
        """Both passes over the whole batch; returns per-tile records
        (structured ndarray ``[ntiles, 3]`` of STATS_DTYPE; rows of indices not
        requested are zero).  ``hist`` adds the 50-bin histograms, ``sumsq`` the sums of squares
        (``summarize()['std']``).  ``medians=True`` also returns ``float64[ntiles, 3]``
        with np.median of each tile's index plane, exact: uint8 RGNir tiles take the
        two-level select on recomputed values (planes or not); other tiles a batched radix
        select on the float32 planes, which must then be written -- a small ring is
        allocated when ``outputs`` has none.  ``route``: see ``set_stats_route`` (None = the module's setting).

        Tables: ``recompute_tables=False`` uses the white-balance tables the batch already holds (``compute_wb_tables``,
        any ``rgn_variant``) when they cover the channels this pass reads -- such a pass runs on the per-pixel route, the
        one-read route derives its tables from its own counts -- and computes them otherwise.  ``rgn_variant`` (None: 0
        for tables computed here, whatever the reused tables are) selects the flavour of tables computed here.
        ``channel_hist=True``: the 256-bin channel histograms (``host_hist``) are wanted too -- the per-pixel route always
        leaves them, the one-read route only on request (and then counts on full tables: ``run_joint``)."""
        route = _STATS_ROUTE if route is None else route
        variant = 0 if rgn_variant is None else int(rgn_variant)
        need = channels_of(indices, outputs is not None and outputs.wb is not None)
        reuse = (white_balance and not recompute_tables and self.table is not None and need <= self._table_channels
                 and (rgn_variant is None or variant == self._rgn_variant))
        if reuse and outputs is None and route == "joint":
            raise ValueError("route='joint' derives the tables from its own counts: it cannot honour recompute_tables=False")
        may_joint = outputs is None and route != "classic" and not reuse and self.can_joint() and bool(indices)
        if route == "auto" and may_joint and not medians and not sumsq:
            route = self.pick_stats_route(indices, white_balance)
            may_joint = route != "classic"
        if may_joint:
            # nothing to write: one read of the tiles serves the percentiles, the statistics and the medians
            stats = self.new_stats()
            stats.zero(stream)
            pairs_dev = DeviceBuffer(self.ntiles * 4 * 4) if medians else None
            self.run_joint(indices, white_balance, stats, hist, sumsq, pairs_dev, stream, rgn_variant=variant, channel_hist=channel_hist)
            self.check_joint(stream)
            rec = stats.download(STATS_DTYPE, (self.ntiles, 3))
            stats.free()
            if not medians:
                return rec
            med = self._medians_from_pairs(pairs_dev.download(np.float32, (self.ntiles, 2, 2)), indices)
            pairs_dev.free()
            return rec, med
        if route == "joint" and outputs is None:
            raise ValueError("route='joint' serves uint8 tiles with 3 channels (4-byte aligned) only")
        if (outputs is not None and outputs.wb is None and route != "classic" and not reuse and self.can_joint() and bool(indices)
                and (medians or hist or select_streams(indices) in (1, 2))):
            # Planes wanted (no white-balanced image: that needs all three tables).  Where the one-read pass costs no more than
            # the channel-histogram pass it replaces -- one value stream: 8.5 ms per 1024 tiles of 4096 x 4096 either way -- it
            # delivers the tables AND the statistics, and the plane-writing kernel runs without its statistics registers and
            # flush: 28.7 instead of 30.5 ms for the NDVI plane (profiles/r04_ndvi_plane_step_ways.txt).  With medians it wins
            # for any set of indices (the medians come with the same read instead of two more passes), and so it does with the 50-bin
            # histograms: they fall out of the counted cells, where the plane-writing kernel pays LDS atomics for them (0.76 against
            # 0.82 of the roofline): 48.9 instead of 51.7 ms per 1024 tiles on windowed tables, the same on full ones.
            stats = self.new_stats()
            stats.zero(stream)
            pairs_dev = DeviceBuffer(self.ntiles * 4 * 4) if medians else None
            self.run_joint(indices, white_balance, stats, hist, sumsq, pairs_dev, stream, rgn_variant=variant, channel_hist=channel_hist)
            self.run_fused_chunks(indices, white_balance, None, False, outputs, stream)
            self.check_joint(stream)
            rec = stats.download(STATS_DTYPE, (self.ntiles, 3))
            stats.free()
            if not medians:
                return rec
            med = self._medians_from_pairs(pairs_dev.download(np.float32, (self.ntiles, 2, 2)), indices)
            pairs_dev.free()
            return rec, med
        if white_balance and not reuse:
            self.compute_wb_tables(stream, rgn_variant=variant)
        stats = self.new_stats()
        stats.zero(stream)                                  # same stream as the kernels that accumulate into it
        # uint8 RGNir tiles: medians come from the two-level select on recomputed values (two passes over the 3-byte
        # pixels), whether or not planes are written; other tiles take the radix select over stored planes below
        select = (medians and self.code == _ffi.U8 and self.channels == 3 and self.npix * 6 < (1 << 30)
                  and (self.ntiles == 1 or self.npix % 4 == 0))
        if select:
            mask = 0
            for t in indices:
                mask |= 1 << INDEX_IDS[t]
            if outputs is None and mask in (1, 2, 4, 7):
                # nothing to write: the statistics kernel also counts the select's bucket pass, one slot pass follows --
                # 3 B per pixel each, everything on the device
                pairs_dev = DeviceBuffer(self.ntiles * 4 * 4)
                scratch = DeviceBuffer(int(_ffi.load().lars_quotient_median_scratch_bytes(self.ntiles)))
                args = self.fused_args(indices, white_balance, stats, hist, None, stream, sumsq=sumsq)
                _ffi.call("lars_d_stats_medians", C.byref(args), C.c_void_p(pairs_dev.ptr), C.c_void_p(scratch.ptr))
                _ffi.call("lars_synchronize", stream)
                med = self._medians_from_pairs(pairs_dev.download(np.float32, (self.ntiles, 2, 2)), indices)
                pairs_dev.free()
                scratch.free()
            else:                                           # planes wanted, or two of the three indices
                self.run_fused_chunks(indices, white_balance, stats, hist, outputs, stream, sumsq)
                med = self.tile_medians(indices, white_balance, stream)
            rec = stats.download(STATS_DTYPE, (self.ntiles, 3))
            stats.free()
            return rec, med
        own_outputs = None
        if medians and (outputs is None or any(outputs.index[INDEX_IDS[t]] is None for t in indices)):
            # a temporary ring: one plain allocation, never the arena search (seconds and several arenas' worth of memory per call)
            own_outputs = outputs = self.make_outputs(indices=indices, index=True, ring=min(self.ntiles, 16), arena="plain")
        med_dev = sel = None
        if medians:
            med_dev = DeviceBuffer(self.ntiles * 3 * 2 * 4)
            med_dev.zero(stream)
            sel = DeviceBuffer(outputs.slots * int(_ffi.load().lars_select_scratch_bytes()))
        chunk = self.ntiles if outputs is None else outputs.slots
        if not medians:
            self.run_fused_chunks(indices, white_balance, stats, hist, outputs, stream, sumsq)
        for start in range(0, self.ntiles if medians else 0, chunk):
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
        for b in (self.tiles, self.hist, self.table, self.percentiles, getattr(self, "_joint_scratch", None)):
            if b is not None:
                b.free()

    def tile_medians(self, indices=INDEX_NAMES, white_balance=True, stream=None):
        """float64[ntiles, 3]: np.median of every tile's index planes, none of which is written
        (``lars_d_quotient_median_pairs``: per-tile two-level select on recomputed values, all on the device)."""
        if white_balance and (self.table is None or not channels_of(indices) <= self._table_channels):
            raise RuntimeError("compute_wb_tables() first")
        pairs_dev = DeviceBuffer(self.ntiles * 4 * 4)
        scratch = DeviceBuffer(int(_ffi.load().lars_quotient_median_scratch_bytes(self.ntiles)))
        _ffi.call("lars_d_quotient_median_pairs", C.c_void_p(self.tiles.ptr), self.ntiles, self.npix, self.channels, self.code,
                  C.c_void_p(self.table.ptr) if white_balance else None, select_streams(indices), C.c_void_p(pairs_dev.ptr),
                  C.c_void_p(scratch.ptr), stream)
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
    def select_histogram(self, first, buckets, white_balance=True, stream=None, streams=3):
        """One pass of the select (``lars_d_quotient_select_hist``): uint64[2 streams][2 tracks][2048].  ``first``: 1 / True
        bucket pass, 0 / False slot pass, 3 bucket pass over a 1/16 subsample, 2 window pass (``buckets[2 s]`` = first slot).
        ``streams``: bit 0 NDVI, bit 1 GNDVI (NDWI shares it); a stream left out is not computed and stays zero."""
        if self.code != _ffi.U8 or self.channels != 3:
            raise TypeError("exact batch medians need uint8 tiles with 3 channels")
        if white_balance and self.table is None:
            raise RuntimeError("compute_wb_tables() first")
        if getattr(self, "_selq", None) is None:
            self._selq = DeviceBuffer(2 * 2 * SELECT_BINS * 8)
        self._selq.zero(stream)
        b = np.ascontiguousarray(np.asarray(buckets, dtype=np.int64) & 0xFFFFFFFF, dtype=np.uint32).reshape(4)    # window starts are int32
        _ffi.call("lars_d_quotient_select_hist", C.c_void_p(self.tiles.ptr), self.ntiles, self.npix, self.channels, self.code,
                  C.c_void_p(self.table.ptr) if white_balance else None, int(streams), int(first), _ffi.ptr(b),
                  C.c_void_p(self._selq.ptr), stream)
        _ffi.call("lars_synchronize", stream)
        return self._selq.download(np.uint64, (2, 2, SELECT_BINS))

    def global_medians(self, indices=INDEX_NAMES, white_balance=True, comm=None, recompute_tables=False):
        """``np.median`` of each index over ALL pixels of ALL tiles of ALL ranks, exactly, without writing a plane:
        two select passes (2048 linear buckets, then the 1024 slots of the chosen bucket, each of which holds one
        distinct quotient of bytes) that recompute the index values from the tiles (3 bytes per pixel and pass) and
        one small all-reduce per pass."""
        if white_balance and (recompute_tables or self.table is None or not channels_of(indices) <= self._table_channels):
            self.compute_wb_tables()
        n_local = self.ntiles * self.npix
        streams = select_streams(indices)
        values = select_order_statistics(lambda first, buckets: self.select_histogram(first, buckets, white_balance, None, streams),
                                         n_local, comm, streams)
        return medians_from_pairs(values, indices)


class BatchOutputs:
    """Device output planes of a batch (optionally a ring of ``slots`` tiles).  The float32 index planes and the RGBA8
    planes are slices of one allocation (``arena``): index planes in the order of INDEX_NAMES, then the RGBA planes."""

    def __init__(self, batch, indices, index, wb, rgba, ring=None, allocate=True):
        from .api import colormap_lut, _colormap_for
        self.slots = batch.ntiles if not ring else min(int(ring), batch.ntiles)
        self.index, self.rgba, self.luts = [None] * 3, [None] * 3, [None] * 3
        self.batch = batch
        # planes start on 256-byte boundaries inside the arena (the fast kernels want 16-byte aligned planes)
        self.plane_bytes = (self.slots * batch.npix * 4 + 255) & ~255
        self._index_ids = sorted(INDEX_IDS[t] for t in indices) if index else []
        self._rgba_ids = sorted(INDEX_IDS[t] for t in indices) if rgba else []      # RGBA8 planes: 4 bytes per pixel too
        self.arena = None
        self.arena2 = None                     # a second allocation holding the last planes (make_outputs' cross-allocation fall-back)
        self.arena_report = None
        if allocate and (self._index_ids or self._rgba_ids):
            self.adopt_arena(DeviceBuffer((len(self._index_ids) + len(self._rgba_ids)) * self.plane_bytes))
        for t in indices:
            k = INDEX_IDS[t]
            if rgba:
                self.luts[k] = DeviceBuffer(1024)
                self.luts[k].upload(colormap_lut(_colormap_for(t)))
        self.wb = DeviceBuffer(self.slots * batch.npix * batch.channels) if wb else None

    def adopt_arena(self, arena, offsets=None):
        """Point the planes at ``arena`` (the caller frees whatever arena was in use before).  ``offsets``: byte offset of every
        plane inside the arena (index planes in the order of INDEX_NAMES, then the RGBA planes; multiples of 256), packed
        back to back when None."""
        nplanes = len(self._index_ids) + len(self._rgba_ids)
        if offsets is None:
            offsets = tuple(j * self.plane_bytes for j in range(nplanes))
        offsets = tuple(int(o) for o in offsets)
        assert len(offsets) == nplanes and all(o % 256 == 0 and o + self.plane_bytes <= arena.nbytes for o in offsets)
        order = sorted(offsets)
        assert all(b - a >= self.plane_bytes for a, b in zip(order, order[1:])), "planes overlap"
        self.arena = arena
        self.arena2 = None
        self.plane_offsets = offsets
        for j, k in enumerate(self._index_ids):
            self.index[k] = DeviceSlice(arena, offsets[j], self.plane_bytes)
        for j, k in enumerate(self._rgba_ids):
            self.rgba[k] = DeviceSlice(arena, offsets[len(self._index_ids) + j], self.plane_bytes)

    def adopt_two_arenas(self, arena, offsets, arena2, offsets2):
        """The first ``len(offsets)`` planes inside ``arena``, the rest inside ``arena2`` (both stay owned by these outputs):
        what ``TileBatch.make_outputs`` falls back to when every allocation it took is of ONE kind of memory throughout."""
        offsets, offsets2 = tuple(int(o) for o in offsets), tuple(int(o) for o in offsets2)
        nplanes = len(self._index_ids) + len(self._rgba_ids)
        assert len(offsets) + len(offsets2) == nplanes
        for buf, offs in ((arena, offsets), (arena2, offsets2)):
            assert all(o % 256 == 0 and o + self.plane_bytes <= buf.nbytes for o in offs)
            order = sorted(offs)
            assert all(b - a >= self.plane_bytes for a, b in zip(order, order[1:])), "planes overlap"
        self.arena, self.arena2 = arena, arena2
        self.plane_offsets = offsets + offsets2
        where = [(arena, o) for o in offsets] + [(arena2, o) for o in offsets2]
        for j, k in enumerate(self._index_ids):
            self.index[k] = DeviceSlice(where[j][0], where[j][1], self.plane_bytes)
        for j, k in enumerate(self._rgba_ids):
            self.rgba[k] = DeviceSlice(*where[len(self._index_ids) + j], self.plane_bytes)

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
        for b in self.index + self.rgba + self.luts + [self.wb, self.arena, getattr(self, "arena2", None)]:
            if b is not None:
                b.free()
        self.index, self.rgba, self.arena, self.arena2 = [None] * 3, [None] * 3, None, None


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
# exact order statistics across tiles and ranks (two-level select on recomputed quotients of bytes)
# ---------------------------------------------------------------------------
SELECT_BINS = 2048
SELECT_SLOTS = 1024                                        # second level: slots inside one bucket
MAX_BYTE_SUM = 510                                         # largest denominator of a quotient of bytes


def select_position(x):
    """(bucket, slot) of float32 values in [-1, 1], the kernels' arithmetic: t = float32 fma(x, 1023.5, 3071.5) lies in
    [2048, 4095]; its 23 mantissa bits are 11 bits of bucket and 12 of fraction, slot = fraction >> 2.
    (The float64 product and sum are exact, so the single rounding to float32 is the fma's.)"""
    t = (np.asarray(x, dtype=np.float32).astype(np.float64) * 1023.5 + 3071.5).astype(np.float32)
    m = np.ascontiguousarray(t).view(np.uint32) - np.uint32(0x45000000)
    return (m >> np.uint32(12)).astype(np.int64), ((m & np.uint32(0xFFF)) >> np.uint32(2)).astype(np.int64)


def select_value(bucket, slot):
    """The one quotient of bytes n/d (d <= 510) whose position is (bucket, slot): any two such fractions differ by at
    least 1/(510 * 509) = 16 units of the fraction, a slot is 4 wide.  Every denominator is tried with n =
    rint(centre * d); float32 n/d is the correctly rounded quotient the kernels compute."""
    centre = ((2048.0 + bucket + (slot * 4.0 + 2.0) / 4096.0) - 3071.5) / 1023.5
    den = np.arange(1, MAX_BYTE_SUM + 1, dtype=np.float64)
    num = np.rint(centre * den)
    q = (num.astype(np.float32) / den.astype(np.float32)).astype(np.float32) + np.float32(0)      # -0/d -> +0.0
    b, s = select_position(q)
    hit = (np.abs(num) <= den) & (b == bucket) & (s == slot)
    if not hit.any():
        raise RuntimeError(f"select: no quotient of bytes at bucket {bucket}, slot {slot} (inconsistent passes)")
    vals = np.unique(q[hit])
    assert vals.size == 1
    return np.float32(vals[0])


def select_streams(indices):
    """Which of the two value streams a set of indices needs: bit 0 NDVI, bit 1 GNDVI (NDWI = -GNDVI rides on it)."""
    mask = 0
    for t in indices:
        mask |= 1 if t == "NDVI" else 2
    if not mask:
        raise ValueError("no index requested")
    return mask


WINDOW_SLOTS = 1920                    # SELQ_WIN_SLOTS (csrc/v2_device.h): a window is 3.75 buckets of 512 slots
WINDOW_SCALE = 524032.0               # slots per unit of the index: sigma(x) = round(x * 524032)
WINDOW_MAGIC = 12582912               # 1.5 * 2^23: floats in [2^23, 2^24) carry their integer value in the mantissa


def select_window_word(x, ws):
    """The kernels' word of float32 ``x`` in the row of a window that starts at slot ``ws``: bits(fma(x, 524032,
    1.5 * 2^23 - (ws - 64))) - bits(1.5 * 2^23); 64 .. 64 + 1919 inside the window.  (Product and sum are exact in float64,
    so the single rounding to float32 is the fma's.)"""
    u = (np.asarray(x, dtype=np.float32).astype(np.float64) * WINDOW_SCALE + float(WINDOW_MAGIC - (int(ws) - 64))).astype(np.float32)
    return np.ascontiguousarray(u).view(np.uint32).astype(np.int64) - 0x4B400000


def select_window_value(ws, slot):
    """The one quotient of bytes whose word in the window starting at ``ws`` is 64 + ``slot`` (see select_value)."""
    centre = (int(ws) + int(slot)) / WINDOW_SCALE
    den = np.arange(1, MAX_BYTE_SUM + 1, dtype=np.float64)
    num = np.rint(centre * den)
    q = (num.astype(np.float32) / den.astype(np.float32)).astype(np.float32) + np.float32(0)      # -0/d -> +0.0
    hit = (np.abs(num) <= den) & (select_window_word(q, ws) == 64 + int(slot))
    if not hit.any():
        raise RuntimeError(f"select: no quotient of bytes in slot {slot} of the window at {ws} (inconsistent passes)")
    vals = np.unique(q[hit])
    assert vals.size == 1
    return np.float32(vals[0])


def select_window_start(sample_hist):
    """First slot of the window for a stream whose sample fell into the 2048 buckets as ``sample_hist``: of the three
    windows of three whole buckets that contain the bucket of the sample's middle rank, the one that keeps that rank
    farthest (in ranks) from both ends (k_selq_predict's rule at bucket resolution; the batch-wide sample is large)."""
    c = np.cumsum(np.asarray(sample_hist, dtype=np.int64))
    total = int(c[-1])
    if total <= 0:
        return -int(WINDOW_SCALE)
    mid = total // 2
    m = int(np.searchsorted(c, mid, side="right"))
    best, best_margin = m, -1
    for b in (m - 2, m - 1, m):
        lo = min(max(b, 0), SELECT_BINS - 3)
        margin = min(mid - (int(c[lo - 1]) if lo else 0), int(c[lo + 2]) - 1 - mid)
        if margin > best_margin:
            best, best_margin = lo, margin
    ws = (best - 1023) * 512 - 256 - (WINDOW_SLOTS - 3 * 512) // 2
    return int(min(max(ws, -int(WINDOW_SCALE)), int(WINDOW_SCALE) + 1 - WINDOW_SLOTS))


def select_order_statistics(pass_fn, n_local, comm=None, streams=3, windowed=True):
    """The two middle order statistics (ranks (N-1)//2 and N//2) of two streams of quotients of bytes: float32[2][2].

    ``pass_fn(first, buckets[4]) -> uint64[2][2][SELECT_BINS]`` counts on this rank (lars_d_quotient_select_hist).
    Histograms are summed over ranks through ``comm.allreduce_f64`` (counts < 2^53 are exact in float64), every rank then
    picks the same bins.  ``streams`` (bit 0, bit 1): a stream that is not asked for is skipped and comes back as NaN.

    ``windowed``: first a bucket pass over a 1/16 subsample (``first=3``) predicts a window of 3.75 buckets per stream,
    then ONE full pass (``first=2``) counts the values below the window and its 1920 slots; ranks that fall inside are
    exact from those counts.  Only if a rank falls outside do the two classic passes follow: first pass, the bucket of
    every value (under track 0); second pass, the slot of the values inside ``buckets[stream * 2 + track]`` -- under
    track 0 only when both streams' tracks share their bucket.
    """
    tot = np.array([float(n_local)])
    n_total = int((comm.allreduce_f64(tot, "sum") if comm is not None else tot)[0])
    if n_total <= 0:
        raise ValueError("select: no values")

    def summed(first, buckets):
        local = np.asarray(pass_fn(first, buckets), dtype=np.uint64).reshape(2, 2, SELECT_BINS)
        hist = local.astype(np.float64).reshape(-1)
        if comm is not None:
            hist = comm.allreduce_f64(hist, "sum")
        return np.asarray(hist).reshape(2, 2, SELECT_BINS).astype(np.int64)

    ranks = np.array([[(n_total - 1) // 2, n_total // 2]] * 2, dtype=np.int64)     # [stream][track]
    values = np.full((2, 2), np.nan, dtype=np.float32)
    if windowed:
        sample = summed(3, np.zeros(4, dtype=np.int64))
        ws = [select_window_start(sample[s, 0]) for s in range(2)]
        rows = summed(2, np.array([ws[0], 0, ws[1], 0], dtype=np.int64))
        found = True
        for s in range(2):
            if not (streams >> s) & 1:
                continue
            below = int(rows[s, 0, :64].sum())
            cum = np.cumsum(rows[s, 0, 64:64 + WINDOW_SLOTS])
            for t in range(2):
                r = int(ranks[s, t]) - below
                if r < 0 or r >= int(cum[-1]):
                    found = False
                    continue
                values[s, t] = select_window_value(ws[s], int(np.searchsorted(cum, r, side="right")))
        if found:
            return values
        values[:] = np.nan
    buckets = np.zeros((2, 2), dtype=np.uint32)
    for first in (True, False):
        hist = summed(1 if first else 0, buckets.reshape(4))
        if first:
            hist[:, 1] = hist[:, 0]                         # bucket pass: counted under track 0
        else:
            hist[:, :, SELECT_SLOTS:] = 0
            if (buckets[:, 0] == buckets[:, 1]).all():
                hist[:, 1] = hist[:, 0]                     # both streams' tracks shared: only track 0 was counted
        for s in range(2):
            if not (streams >> s) & 1:
                continue
            for t in range(2):
                cum = np.cumsum(hist[s, t])
                d = int(np.searchsorted(cum, ranks[s, t], side="right"))
                if d >= (SELECT_BINS if first else SELECT_SLOTS):
                    raise RuntimeError("select: rank beyond the histogram mass (inconsistent passes)")
                ranks[s, t] -= int(cum[d - 1]) if d else 0
                if first:
                    buckets[s, t] = d
                else:
                    values[s, t] = select_value(int(buckets[s, t]), d)
    return values


def medians_from_pairs(values, indices=INDEX_NAMES):
    """np.median semantics (mean of the two middle values in float32); NDWI = -GNDVI shares GNDVI's statistics."""
    out = {}
    for t in indices:
        s = 0 if t == "NDVI" else 1
        m = np.float32(np.float32(values[s, 0] + values[s, 1]) / np.float32(2))
        out[t] = float(np.float32(0) - m) if t == "NDWI" else float(m)
    return out
