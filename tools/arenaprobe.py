#!/usr/bin/env python3
"""What decides the time of a plane-writing launch into an output arena -- the allocation, the moment, or the company?

VERDICT round 3, item 1: the driver's run timed sixteen candidate arenas at 2.94-3.04 ms per launch and then ran its steps
at 2.51 ms into the one it kept.  This tool takes the headline configuration (1024 tiles of 4096 x 4096 uint8, three
float32 planes, ring of 64) apart in ONE process, every launch timed by its own pair of events on the library stream:

  A  K candidate arenas allocated up front (all alive); R rounds of a step-length burst (16 launches through the whole
     batch, no allocation, no gap) into each: the steady-state level of every allocation, and whether it repeats
  B  the library's probe shape (one warm-up + three launches over first / middle / last chunk) into the same arenas,
     back to back and after an idle gap of G seconds
  C  a burst into a KNOWN arena right after a fresh 12 GiB hipMalloc (of another arena): does the allocation disturb
     what runs next, whatever it writes to?
  D  all candidates but one freed, then the same bursts into the survivor
  E  a whole step (histogram pass + tables + 16 fused launches) x 3 into the survivor, per-launch times of each

Clocks (sclk / mclk / fclk / socclk from sysfs, when readable) are sampled every 20 ms by a thread and summarised per phase.

    python tools/arenaprobe.py [--candidates 8] [--rounds 3] [--gap 0.3] [--tiles 1024]
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lars_image_processing_amd as lars  # noqa: E402
from lars_image_processing_amd import _ffi  # noqa: E402
from lars_image_processing_amd.batch import BatchOutputs  # noqa: E402

INDICES = ("NDVI", "GNDVI", "NDWI")


class Clocks(threading.Thread):
    """Samples the '*'-marked level of pp_dpm_{sclk,mclk,fclk,socclk} of every card that exposes them."""

    def __init__(self, period=0.02):
        super().__init__(daemon=True)
        self.period = period
        self.files = []
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
            for path in sorted(glob.glob(f"/sys/class/drm/card*/device/{name}")):
                self.files.append((name[7:], path))
        self.samples = []          # (t, {name: MHz})
        self.stop = False

    @staticmethod
    def _active(path):
        try:
            with open(path) as fh:
                for ln in fh:
                    if ln.rstrip().endswith("*"):
                        return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            return None
        return None

    def run(self):
        while not self.stop:
            row = {}
            for name, path in self.files:
                v = self._active(path)
                if v is not None:
                    row.setdefault(name, []).append(v)
            self.samples.append((time.perf_counter(), row))
            time.sleep(self.period)

    def summary(self, t0, t1):
        out = {}
        for t, row in self.samples:
            if t0 <= t <= t1:
                for name, vals in row.items():
                    out.setdefault(name, []).append(max(vals))
        return " ".join(f"{k} {min(v)}-{max(v)}" for k, v in sorted(out.items())) or "clocks unreadable"


class Timer:
    def __init__(self, n=40):
        self.ev = []
        for _ in range(n):
            e = C.c_void_p()
            _ffi.call("lars_event_create", C.byref(e))
            self.ev.append(e)

    def burst(self, batch, launches):
        """Per-launch milliseconds of ``launches`` (FusedArgs) issued back to back."""
        assert len(launches) < len(self.ev)
        _ffi.call("lars_event_record", self.ev[0], None)
        for i, a in enumerate(launches):
            batch.run_fused(a)
            _ffi.call("lars_event_record", self.ev[i + 1], None)
        _ffi.call("lars_synchronize", None)
        out = []
        ms = C.c_float(0)
        for i in range(len(launches)):
            _ffi.call("lars_event_elapsed_ms", self.ev[i], self.ev[i + 1], C.byref(ms))
            out.append(ms.value)
        return out


def fmt(ts):
    return " ".join(f"{t:.3f}" for t in ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--gap", type=float, default=0.3)
    ap.add_argument("--tiles", type=int, default=1024)
    ap.add_argument("--ring", type=int, default=64)
    args = ap.parse_args()

    clocks = Clocks()
    clocks.start()
    t_start = time.perf_counter()
    print(f"# device {_ffi.device_name()}; clock files: {[p for _, p in clocks.files]}", flush=True)
    b = lars.TileBatch.synthetic(args.tiles, 4096, 4096, seed=1234, profile="vegetation")
    b.compute_wb_tables()
    _ffi.call("lars_synchronize", None)
    stats = b.new_stats()
    stats.zero()
    outs = BatchOutputs(b, INDICES, True, False, False, args.ring, allocate=False)
    nbytes = 3 * outs.plane_bytes
    tm = Timer()
    print(f"# batch + tables ready after {time.perf_counter() - t_start:.1f} s; arena = {nbytes / 2**30:.1f} GiB", flush=True)

    def step_launches(raw=True):
        return [b.fused_args(INDICES, True, stats, False, outs, None, s, min(outs.slots, b.ntiles - s), raw=raw)
                for s in range(0, b.ntiles, outs.slots)]

    def probe_launches():
        count = outs.slots
        nchunks = max(1, b.ntiles // count)
        starts = sorted({0, (nchunks // 2) * count, (nchunks - 1) * count})
        return [b.fused_args(INDICES, True, stats, False, outs, None, st, count) for st in starts]

    # ---- A: candidates up front, step-length bursts --------------------------------------------------------------
    arenas, malloc_ms = [], []
    for _ in range(args.candidates):
        t0 = time.perf_counter()
        arenas.append(_ffi.DeviceBuffer(nbytes))
        malloc_ms.append((time.perf_counter() - t0) * 1e3)
    print(f"# A: {len(arenas)} arenas allocated up front, hipMalloc ms: {fmt(malloc_ms)}")
    print("# A: step-length bursts (16 launches, whole batch), per-launch ms; all candidates alive")
    level = np.zeros((args.rounds, len(arenas)))
    for r in range(args.rounds):
        for j, arena in enumerate(arenas):
            outs.adopt_arena(arena)
            t0 = time.perf_counter()
            ts = tm.burst(b, step_launches())
            level[r, j] = np.mean(ts[1:])
            print(f"A round {r} arena {j}: mean(2..16) {np.mean(ts[1:]):.3f}  first {ts[0]:.3f}  [{fmt(ts)}]  {clocks.summary(t0, time.perf_counter())}", flush=True)
    print("# A: level per arena (rows = rounds):")
    for r in range(args.rounds):
        print("A levels:", fmt(level[r]))
    order = np.argsort(level[-1])
    fast, slow = int(order[0]), int(order[-1])
    print(f"# fastest arena {fast} ({level[-1, fast]:.3f}), slowest {slow} ({level[-1, slow]:.3f})", flush=True)

    # ---- B: the probe's shape, back to back and after an idle gap ------------------------------------------------
    print(f"# B: probe shape (1 warm-up + 3 launches: first / middle / last chunk; not LARS_F_RAW), back to back, then after {args.gap} s idle")
    for label, j in (("fast", fast), ("slow", slow)):
        outs.adopt_arena(arenas[j])
        for rep in range(3):
            pl = probe_launches()
            ts = tm.burst(b, [pl[0]] + pl)
            print(f"B {label} arena {j} back-to-back {rep}: warm {ts[0]:.3f}  probe mean {np.mean(ts[1:]):.3f}  [{fmt(ts)}]")
        for rep in range(4):
            time.sleep(args.gap)
            t0 = time.perf_counter()
            pl = probe_launches()
            ts = tm.burst(b, [pl[0]] + pl)
            print(f"B {label} arena {j} after {args.gap} s idle {rep}: warm {ts[0]:.3f}  probe mean {np.mean(ts[1:]):.3f}  [{fmt(ts)}]  {clocks.summary(t0 - args.gap, time.perf_counter())}")
        time.sleep(args.gap)
        ts = tm.burst(b, step_launches())
        print(f"B {label} arena {j} 16-burst after {args.gap} s idle: mean(2..16) {np.mean(ts[1:]):.3f}  [{fmt(ts)}]", flush=True)
        time.sleep(2.0)
        ts = tm.burst(b, step_launches())
        print(f"B {label} arena {j} 16-burst after 2 s idle: mean(2..16) {np.mean(ts[1:]):.3f}  [{fmt(ts)}]", flush=True)

    # ---- C: a known arena right after somebody else's hipMalloc ---------------------------------------------------
    print("# C: bursts into KNOWN arenas directly after a fresh hipMalloc of another arena (the search's rhythm)")
    extra = []
    free_b, total_b = C.c_size_t(), C.c_size_t()
    for rep in range(3):
        _ffi.call("lars_mem_info", C.byref(free_b), C.byref(total_b))
        if free_b.value < nbytes + (8 << 30):
            print(f"C: only {free_b.value / 2**30:.0f} GiB free, stopping")
            break
        t0 = time.perf_counter()
        new = _ffi.DeviceBuffer(nbytes)
        ms_alloc = (time.perf_counter() - t0) * 1e3
        extra.append(new)
        for label, j in (("fast", fast), ("slow", slow)):
            outs.adopt_arena(arenas[j])
            pl = probe_launches()
            ts = tm.burst(b, [pl[0]] + pl)
            print(f"C rep {rep} (hipMalloc {ms_alloc:.0f} ms) then known {label} arena {j}: warm {ts[0]:.3f}  probe mean {np.mean(ts[1:]):.3f}  [{fmt(ts)}]")
        outs.adopt_arena(new)
        pl = probe_launches()
        ts = tm.burst(b, [pl[0]] + pl)
        p_ms = float(np.mean(ts[1:]))
        ts16 = tm.burst(b, step_launches())
        print(f"C rep {rep} the NEW arena: probe mean {p_ms:.3f} [{fmt(ts)}]  then 16-burst mean(2..16) {np.mean(ts16[1:]):.3f}  [{fmt(ts16)}]", flush=True)

    # ---- D: free everything but the fastest and the slowest, then each alone --------------------------------------
    print("# D: candidates freed; the survivor(s) again")
    for j, arena in enumerate(arenas):
        if j not in (fast, slow):
            arena.free()
    for e in extra:
        e.free()
    _ffi.call("lars_synchronize", None)
    for label, j in (("fast", fast), ("slow", slow)):
        outs.adopt_arena(arenas[j])
        for rep in range(2):
            ts = tm.burst(b, step_launches())
            print(f"D {label} arena {j} (others freed) 16-burst {rep}: mean(2..16) {np.mean(ts[1:]):.3f}  (A: {level[-1, j]:.3f})  [{fmt(ts)}]")
    if slow != fast:
        arenas[slow].free()
    _ffi.call("lars_synchronize", None)
    outs.adopt_arena(arenas[fast])
    ts = tm.burst(b, step_launches())
    print(f"D fast arena {fast} ALONE 16-burst: mean(2..16) {np.mean(ts[1:]):.3f}  [{fmt(ts)}]", flush=True)

    # ---- E: whole steps ---------------------------------------------------------------------------------------------
    print("# E: whole steps (histogram pass + tables, then 16 fused launches) into the survivor")
    for rep in range(3):
        t0 = time.perf_counter()
        b.compute_wb_tables()
        ts = tm.burst(b, step_launches())
        print(f"E step {rep}: mean {np.mean(ts):.3f}  [{fmt(ts)}]  wall {1e3 * (time.perf_counter() - t0):.1f} ms  {clocks.summary(t0, time.perf_counter())}", flush=True)

    # ---- F: a fresh allocation after all the freeing: which level does it get? -----------------------------------------
    print("# F: fresh allocations after the frees (one at a time, freed again): probe, then 16-burst")
    for rep in range(4):
        new = _ffi.DeviceBuffer(nbytes)
        outs.adopt_arena(new)
        pl = probe_launches()
        ts = tm.burst(b, [pl[0]] + pl)
        ts16 = tm.burst(b, step_launches())
        print(f"F rep {rep}: probe mean {np.mean(ts[1:]):.3f} [{fmt(ts)}]  16-burst mean(2..16) {np.mean(ts16[1:]):.3f}  first {ts16[0]:.3f}", flush=True)
        new.free()
        _ffi.call("lars_synchronize", None)
    clocks.stop = True
    print(f"# done after {time.perf_counter() - t_start:.1f} s")


if __name__ == "__main__":
    main()
