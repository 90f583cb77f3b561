#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the fused white-balance + NDVI/GNDVI/NDWI +
statistics path on 4096x4096 uint8 RGNir tiles (BASELINE.json), with the
achieved HBM GB/s of the fused kernel against the ~8 TB/s roofline and the
NumPy oracle timed on this box's host cores as a baseline.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # no launcher: this process starts the N ranks itself (launch_ranks)

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment).
Tiles shard by tile with no data-path collective; each step ends with the
RCCL fold of the global per-index statistics (csrc/comm.cpp).  No PyTorch is
imported: device memory, streams, events and RCCL all go through liblars_hip.so
(LARS_COMM=torch, or an error from the direct RCCL bootstrap, moves only that
fold onto torch.distributed's nccl backend: dist.TorchComm).

A step = one pass of the hot path over this rank's batch of synthetic tiles
that are already resident in HBM:
    channel histograms -> percentile white-balance tables -> fused kernel
    (band de-interleave, table lookup, three float32 index planes written,
    min/max/sum/coverage per index) -> fold of per-tile records -> global fold.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MODES = {
    # name: (indices, write index planes, hist, algorithmic bytes / pixel of the fused kernel)
    "wb3idx_out_stats": (("NDVI", "GNDVI", "NDWI"), True, False, 3 + 12),     # BASELINE configs[1]
    "wb3idx_out_stats_hist": (("NDVI", "GNDVI", "NDWI"), True, True, 3 + 12),  # configs[2]
    "wb_ndvi_out_stats": (("NDVI",), True, False, 3 + 4),
    "wb3idx_stats_only": (("NDVI", "GNDVI", "NDWI"), False, False, 3),
    "wb_ndvi_stats_only": (("NDVI",), False, False, 3),
    # statistics + the exact median of every tile (the reference's analyze_index / time-series table)
    "wb3idx_stats_medians": (("NDVI", "GNDVI", "NDWI"), False, False, 3),
    # the reference's whole job per image (process-images.py:424-513: fix_white_balance, three calculate_index planes,
    # three analyze_index dictionaries INCLUDING the median) -- what cpu_baseline times: one read for statistics, medians
    # and tables (joint byte-pair histograms), then the plane-writing kernel
    "wb3idx_out_stats_medians": (("NDVI", "GNDVI", "NDWI"), True, False, 3 + 12),
}
# statistics-only modes run on the one-read route (csrc/joint.hip) unless --stats-route classic; "<mode>_classic" in the
# line's `modes` is the same job on the two-pass route (channel-histogram pass + per-pixel statistics kernel)
STATS_ONLY = ("wb3idx_stats_only", "wb_ndvi_stats_only", "wb3idx_stats_medians")
MEDIAN_MODES = ("wb3idx_stats_medians", "wb3idx_out_stats_medians")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tiles", type=int, default=1024, help="tiles per GPU (weak scaling)")
    ap.add_argument("--tile", type=int, default=4096, help="tile edge in pixels")
    ap.add_argument("--ring", type=int, default=64, help="output ring, in tiles (same traffic, bounded footprint)")
    ap.add_argument("--mode", default="wb3idx_out_stats", choices=sorted(MODES))
    ap.add_argument("--all-modes", dest="all_modes", action="store_true", default=None,
                    help="time the other modes too (extra JSON field 'modes'): the default with one GPU; with --gpus N > 1 only "
                         "the headline mode is timed unless this flag is given")
    ap.add_argument("--no-all-modes", dest="all_modes", action="store_false",
                    help="skip timing the other modes")
    ap.add_argument("--no-probe", dest="probe", action="store_false", help="skip the streaming-roofline probes")
    ap.add_argument("--profile", default="vegetation", choices=["uniform", "vegetation"])
    ap.add_argument("--arena", default="auto", choices=["auto", "plain", "slowest"],
                    help="output arena of the plane-writing modes.  auto: the library's default (TileBatch.make_outputs) -- a multi-GiB "
                         "arena is ONE allocation with room to spare and the planes are tried in a handful of placements inside it, each "
                         "timed with the batch's own launches (device memory comes in two kinds; the launch is fast when its planes are "
                         "split between them); plain: one packed allocation as it comes; slowest: the same search but the SLOWEST "
                         "placement is kept (a diagnostic: the line of a process that finds no fast placement)")
    ap.add_argument("--placement-trials", type=int, default=-1,
                    help="candidate arenas of the search (-1: the library's default, 0/1: take the first)")
    ap.add_argument("--stats-route", default="joint", choices=["joint", "classic"],
                    help="statistics-only modes: one read through joint byte-pair histograms, or histogram pass + per-pixel kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tiles", type=int, default=8, help="tiles per run of the single-core CPU baseline (best of --cpu-runs; SURVEY.md 8(d): at least 8)")
    ap.add_argument("--cpu-runs", type=int, default=3)
    ap.add_argument("--cpu-workers", type=int, default=16,
                    help="process pool of the multi-core CPU baseline leg (16 = one GPU's share of the box's host cores; every "
                         "worker holds about 1 GiB of NumPy temporaries for a 4096 x 4096 tile)")
    ap.add_argument("--no-u16-leg", dest="u16_leg", action="store_false", help="skip the BASELINE configs[4] shape (uint16 8192 x 8192 tiles)")
    ap.add_argument("--no-smooth-leg", dest="smooth_leg", action="store_false",
                    help="skip the statistics-only modes on image-like (smooth, natural) content")
    ap.add_argument("--no-verify", dest="verify", action="store_false", help="skip the self-check after the timed region")
    return ap.parse_args()


class Runner:
    """Owns the resident batch and times steps of one mode."""

    def __init__(self, args, comm, rank, world):
        import lars_image_processing_amd as lars
        from lars_image_processing_amd import _ffi, batch as lb
        self.lars, self.ffi, self.lb = lars, _ffi, lb
        self.comm, self.rank, self.world = comm, rank, world
        self.args = args
        first = rank * args.tiles
        self.batch = lars.TileBatch.synthetic(args.tiles, args.tile, args.tile, seed=1234, profile=args.profile,
                                              first_tile=first)
        self.stats = self.batch.new_stats()
        self.folded = _ffi.DeviceBuffer(3 * _ffi.STATS_DTYPE.itemsize)     # this rank's three per-index records
        self.folded.zero()
        self.pairs = None
        self.ev = []
        for _ in range(4):
            e = C.c_void_p()
            _ffi.call("lars_event_create", C.byref(e))
            self.ev.append(e)
        self.outputs = {}
        self.launch_ev = []                 # per-launch events of ONE step (the first timed one of the headline mode)
        self.first_step_launch_ms = None

    def outputs_for(self, indices, write):
        if not write:
            return None
        key = tuple(indices)
        if key not in self.outputs:
            if self.batch.table is None:
                self.batch.compute_wb_tables()
            self.outputs[key] = self.batch.make_outputs(indices=indices, index=True, ring=self.args.ring,
                                                        arena="plain" if self.args.arena == "plain" else "auto",
                                                        pick="slowest" if self.args.arena == "slowest" else "fastest",
                                                        placement_trials=None if self.args.placement_trials < 0 else self.args.placement_trials)
        return self.outputs[key]

    def joint_then_planes(self, base_mode):
        """Modes that take the one-read statistics pass and then the plane-writing kernel WITHOUT statistics (TileBatch.process does the
        same): medians and the 50-bin histograms come out of the counted cells for any set of indices; for one value stream the read
        costs what the channel-histogram pass costs."""
        if not (self.args.stats_route == "joint" and self.batch.can_joint()):
            return base_mode == "wb3idx_out_stats_medians"
        return base_mode in ("wb3idx_out_stats_medians", "wb3idx_out_stats_hist", "wb_ndvi_out_stats")

    def step(self, mode, timed=None, launch_events=None):
        """One pass over the batch.  ``timed`` collects (pre_ms, main_ms, launches of the main kernel): pre = the channel-
        histogram pass + tables (or, in the like-for-like mode, the one-read statistics pass), main = the fused kernel(s)
        (or the whole one-read pass of a statistics-only mode)."""
        classic = mode.endswith("_classic")
        base_mode = mode[:-len("_classic")] if classic else mode
        indices, write, hist, _ = MODES[base_mode]
        b, ffi = self.batch, self.ffi
        outs = self.outputs_for(indices, write)
        joint = base_mode in STATS_ONLY and not classic and self.args.stats_route == "joint" and b.can_joint()
        if base_mode in MEDIAN_MODES and self.pairs is None:
            self.pairs = ffi.DeviceBuffer(b.ntiles * 4 * 4)
        ffi.call("lars_event_record", self.ev[0], None)
        if joint:
            ffi.call("lars_event_record", self.ev[1], None)
            b.run_joint(indices, True, self.stats, hist, False, self.pairs if base_mode in MEDIAN_MODES else None)
            launches = 1
        elif self.joint_then_planes(base_mode):
            # one read for statistics, (medians,) and the tables the planes need; then the plane-writing kernel without statistics.
            # For ONE value stream (the NDVI plane) this read costs what the channel-histogram pass costs and the planes-only kernel
            # is 8 % faster than with statistics: TileBatch.process takes the same route (profiles/r04_ndvi_plane_step_ways.txt)
            b.run_joint(indices, True, self.stats, hist, False, self.pairs if base_mode in MEDIAN_MODES else None)
            ffi.call("lars_event_record", self.ev[1], None)
            launches = b.run_fused_chunks(indices, True, None, False, outs, launch_events=launch_events)      # planes only
        else:
            b.compute_wb_tables()
            ffi.call("lars_event_record", self.ev[1], None)
            if base_mode == "wb3idx_stats_medians":
                if getattr(self, "_med_scratch", None) is None:
                    self._med_scratch = ffi.DeviceBuffer(int(ffi.load().lars_quotient_median_scratch_bytes(b.ntiles)))
                args = b.fused_args(indices, True, self.stats, hist, None)
                ffi.call("lars_d_stats_medians", C.byref(args), C.c_void_p(self.pairs.ptr), C.c_void_p(self._med_scratch.ptr))
                launches = 1
            else:
                # one launch without output planes, one per ring of `outs.slots` tiles with them; the statistics records are
                # opened and closed once around the launches (lars_d_stats_begin / _end), not by two small kernels per launch
                launches = b.run_fused_chunks(indices, True, self.stats, hist, outs, launch_events=launch_events)
        ffi.call("lars_event_record", self.ev[2], None)
        # per-index fold of the per-tile records on the device, then the exchange of 3 x 472 bytes (RCCL all-gather + fold in
        # rank order inside the library, or nothing at all with one rank): no per-tile record leaves the device in a step
        b.fold_stats(self.stats, indices, self.folded)
        glob = self.comm.allreduce_stats_device(self.folded, 3)
        if timed is not None:
            ms = C.c_float(0)
            ffi.call("lars_event_elapsed_ms", self.ev[0], self.ev[1], C.byref(ms))
            pre_ms = ms.value
            ffi.call("lars_event_elapsed_ms", self.ev[1], self.ev[2], C.byref(ms))
            timed.append((pre_ms, ms.value, launches))
        return glob

    def run(self, mode, steps, warmup, per_launch=False):
        """``per_launch``: the fused launches of the FIRST timed step are bracketed by events of their own (17 records on the
        stream, no synchronisation: the step is timed like the others) -> ``self.first_step_launch_ms``."""
        for _ in range(warmup):
            self.step(mode)
        per_launch = per_launch and MODES[mode][1] and mode != "wb3idx_stats_medians"     # plane-writing modes only
        if per_launch and not self.launch_ev:
            outs = self.outputs_for(MODES[mode][0], MODES[mode][1])
            n = 1 + (-(-self.batch.ntiles // outs.slots) if outs is not None else 1)
            for _ in range(min(n, 65)):
                e = C.c_void_p()
                self.ffi.call("lars_event_create", C.byref(e))
                self.launch_ev.append(e)
        timed = []
        self.ffi.call("lars_synchronize", None)
        self.comm.barrier()
        t0 = time.perf_counter()
        glob = None
        for i in range(steps):
            glob = self.step(mode, timed, self.launch_ev if (per_launch and i == 0) else None)
        self.ffi.call("lars_synchronize", None)
        self.comm.barrier()
        dt_local = time.perf_counter() - t0
        if per_launch and self.launch_ev and steps:
            ms = C.c_float(0)
            out = []
            for i in range(min(len(self.launch_ev) - 1, timed[0][2])):
                self.ffi.call("lars_event_elapsed_ms", self.launch_ev[i], self.launch_ev[i + 1], C.byref(ms))
                out.append(float(ms.value))
            self.first_step_launch_ms = out
        dt = float(self.comm.allreduce_f64([dt_local], "max")[0])
        self.last_local_dt = dt_local
        return dt, timed, glob

    # ---- self-check after the timed region -------------------------------------------------------------------
    def verify(self, mode, glob_by_mode):
        """What the timed configuration left behind against single-tile runs of the same library: the per-tile records of
        three tiles (first, middle, last) and, where planes are written, the ring planes of three tiles of the LAST chunk
        (what the ring holds after a step), bit for bit; the medians of those tiles where the mode has them; and the global
        statistics of all modes that share an index.  Returns a dict of booleans + what was compared."""
        ffi, b = self.ffi, self.batch
        indices, write, hist, _ = MODES[mode]
        self.step(mode)
        b.check_joint()                     # synchronises; raises if a one-read pass of this step reported an overflow
        rec = self.stats.download(ffi.STATS_DTYPE, (b.ntiles, 3))
        outs = self.outputs_for(indices, write)
        ring = outs.slots if outs is not None else b.ntiles
        last_chunk = ((b.ntiles - 1) // ring) * ring
        rec_tiles = sorted({0, b.ntiles // 2, b.ntiles - 1})
        plane_tiles = sorted({last_chunk, (last_chunk + b.ntiles - 1) // 2, b.ntiles - 1}) if write else []
        out = {"mode": mode, "record_tiles": rec_tiles, "plane_tiles": plane_tiles, "records": True, "planes": True if write else None}
        one = self.lars.TileBatch(1, b.h, b.w, 3, np.uint8)
        one_out = one.make_outputs(indices=indices, index=True) if write else None
        for t in sorted(set(rec_tiles) | set(plane_tiles)):
            ffi.call("lars_memcpy_d2d", C.c_void_p(one.tiles.ptr), C.c_void_p(b.tiles.ptr + t * b.tile_bytes), b.tile_bytes, None)
            r1 = one.process(indices=indices, hist=hist, outputs=one_out, route="classic")
            if t in rec_tiles and r1[0].tobytes() != rec[t].tobytes():
                out["records"] = False
            if t in plane_tiles:
                for name in indices:
                    got = outs.host_index(name, t % ring, 1)
                    want = one_out.host_index(name, 0, 1)
                    if got.tobytes() != want.tobytes():
                        out["planes"] = False
        if mode in MEDIAN_MODES:
            pairs = self.pairs.download(np.float32, (b.ntiles, 2, 2))
            med = b._medians_from_pairs(pairs, indices)
            out["medians"] = True
            for t in rec_tiles:
                ffi.call("lars_memcpy_d2d", C.c_void_p(one.tiles.ptr), C.c_void_p(b.tiles.ptr + t * b.tile_bytes), b.tile_bytes, None)
                _, m1 = one.process(indices=indices, medians=True, route="classic")
                if not np.array_equal(m1[0], med[t]):
                    out["medians"] = False
        if one_out is not None:
            one_out.free()
        one.free()
        # global statistics: every mode that ran reports the same record for an index they share (hist only where counted)
        same = True
        names = sorted(glob_by_mode)
        for k in range(3):
            seen = None
            for m in names:
                base = m[:-len("_classic")] if m.endswith("_classic") else m
                if self.ffi.INDEX_NAMES[k] not in MODES[base][0]:
                    continue
                r = glob_by_mode[m][k].copy()
                r["hist"] = 0
                if seen is None:
                    seen = r.tobytes()
                elif seen != r.tobytes():
                    same = False
        out["global_stats_identical_across_modes"] = same
        out["modes_compared"] = names
        out["ok"] = bool(out["records"] and same and out["planes"] is not False and out.get("medians", True))
        return out


def _cpu_tile_job(job):
    """One tile through the NumPy oracle, the reference's whole job per image (process-images.py:424-513): white balance,
    three calculate_index planes, three analyze_index dictionaries incl. the median.  Returns the seconds of each step:
    [wb, index NDVI, index GNDVI, index NDWI, stats NDVI, stats GNDVI, stats NDWI] (worker of cpu_baseline)."""
    import warnings
    from oracle import index_oracle as orc
    tile, edge, profile = job
    img = orc.synth_tile_u8(1234, tile, edge, edge, profile=profile)
    steps = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        wb = orc.wb_app(img)
        steps.append(time.perf_counter() - t0)
        planes = []
        for t in ("NDVI", "GNDVI", "NDWI"):
            t0 = time.perf_counter()
            planes.append(orc.index_app(wb, t))
            steps.append(time.perf_counter() - t0)
        for t, plane in zip(("NDVI", "GNDVI", "NDWI"), planes):
            t0 = time.perf_counter()
            orc.stats_app(plane, t)
            steps.append(time.perf_counter() - t0)
    return steps


CPU_STEP_NAMES = ("wb", "index_NDVI", "index_GNDVI", "index_NDWI", "stats_NDVI", "stats_GNDVI", "stats_NDWI")


def cpu_baseline(args):
    """The NumPy oracle (oracle/index_oracle.py == the reference's expressions) on a bounded sample, as BASELINE.md section 4
    asks: (i) one process / one core (how the reference runs), best of ``--cpu-runs`` runs over ``--cpu-tiles`` tiles with
    the per-step breakdown; (ii) a process pool over tiles on this GPU's share of the host cores, best of the same number
    of runs.  The job is the one the GPU mode ``wb3idx_out_stats_medians`` does (planes + statistics incl. medians)."""
    n = max(1, args.cpu_tiles)
    runs = max(1, args.cpu_runs)
    pix_tile = args.tile * args.tile
    best, all_runs = None, []
    for _ in range(runs):
        per_tile = [_cpu_tile_job((i, args.tile, args.profile)) for i in range(n)]
        total = float(np.sum(per_tile))
        all_runs.append(total)
        if best is None or total < best[0]:
            best = (total, np.sum(per_tile, axis=0))
    dt1, steps = best
    out = {
        "value": n * pix_tile / dt1 / 1e6, "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": f"{n} tiles {args.tile}x{args.tile} uint8 ({args.profile}): fix_white_balance + 3x calculate_index "
                  f"+ 3x analyze_index (incl. median), NumPy {np.__version__}, single process, best of {runs} runs "
                  f"({', '.join(f'{t:.1f}' for t in all_runs)} s)",
        "same_job_as_gpu_mode": "wb3idx_out_stats_medians",
        "steps_s_per_tile": {name: float(v) / n for name, v in zip(CPU_STEP_NAMES, steps)},
        "runs_s": all_runs,
        "host_cpus": os.cpu_count(),
    }
    workers = max(1, min(args.cpu_workers, os.cpu_count() or 1))
    if workers > 1:
        import multiprocessing as mp
        jobs = [(i, args.tile, args.profile) for i in range(workers)]
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.map(_cpu_tile_job, jobs)                        # warm the workers (imports, page faults)
            times = []
            for _ in range(runs):
                t0 = time.perf_counter()
                pool.map(_cpu_tile_job, jobs)
                times.append(time.perf_counter() - t0)
        dtp = min(times)
        out["pool"] = {"value": len(jobs) * pix_tile / dtp / 1e6, "unit": "Mpix/s", "cores": workers,
                       "sample": f"{len(jobs)} tiles over a pool of {workers} processes, best of {runs} runs "
                                 f"({', '.join(f'{t:.1f}' for t in times)} s)",
                       "why_not_all_cpus": f"os.cpu_count() = {os.cpu_count()} is the whole box; {workers} is one GPU's share of "
                                           "it, and every worker holds about 1 GiB of NumPy temporaries per 4096 x 4096 tile"}
    return out


def device_probe(runner):
    """Plain streaming kernels with the hot path's access shapes: what this device sustains (GB/s).  The probe kernels live
    in the laboratory library (liblars_lab.so, tools/lab/lablib.py); None when it has not been built."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "lab"))
    try:
        import lablib
        if not lablib.available():
            return None
        lablib.load()
    except (ImportError, OSError, AttributeError):
        return None
    ffi = runner.ffi
    nbytes = min(runner.batch.ntiles, 256) * runner.batch.tile_bytes
    nbytes -= nbytes % 960
    dst = ffi.DeviceBuffer(nbytes)
    src = runner.batch.tiles.ptr
    out = {}
    for kind, name, mult in ((1, "read_12B_per_lane", 1), (3, "write_16B_per_lane", 1), (2, "copy_16B_per_lane", 2),
                             (5, "mix_12B_read_48B_write", 1), (7, "mix_12B_read_16B_write", 1)):
        ts = []
        for _ in range(4):
            ffi.call("lars_event_record", runner.ev[0], None)
            lablib.probe(kind, 1, 65536, src, dst.ptr, nbytes)
            ffi.call("lars_event_record", runner.ev[1], None)
            ms = C.c_float(0)
            ffi.call("lars_event_elapsed_ms", runner.ev[0], runner.ev[1], C.byref(ms))
            ts.append(ms.value)
        out[name] = nbytes * mult / float(np.median(ts[1:])) / 1e6
    dst.free()
    # the same 12 B / 48 B mix with the three planes where the headline's output arena has them (its chosen placement: the planes split
    # between the two kinds of device memory) and packed at the start of the same arena: the bare pattern's ceiling for the headline kernel
    outs = runner.outputs.get(("NDVI", "GNDVI", "NDWI"))
    if outs is not None and getattr(outs, "plane_offsets", None) is not None and hasattr(lablib, "probe_mix3"):
        nquads = min(outs.slots, runner.batch.ntiles) * runner.batch.npix // 4
        # (the planes' own pointers: outputs whose search ended with the planes split between two allocations have no common base)
        for name, ptrs in (("mix_12B_read_48B_write_planes_as_placed", tuple(outs.index[k].ptr for k in range(3))),
                           ("mix_12B_read_48B_write_planes_packed", tuple(outs.arena.ptr + j * outs.plane_bytes for j in range(3)))):
            if name.endswith("packed") and outs.arena.nbytes < 3 * outs.plane_bytes:
                continue
            ts = []
            for _ in range(5):
                ffi.call("lars_event_record", runner.ev[0], None)
                lablib.probe_mix3(src, *ptrs, nquads)
                ffi.call("lars_event_record", runner.ev[1], None)
                ms = C.c_float(0)
                ffi.call("lars_event_elapsed_ms", runner.ev[0], runner.ev[1], C.byref(ms))
                ts.append(ms.value)
            out[name] = nquads * 60 / float(np.median(ts[2:])) / 1e6
    return out


KERNEL_SOURCES = ("fused.hip", "fused_device.h", "fused_v2.hip", "joint.hip", "joint_win.hip", "joint_device.h", "select_q.hip", "device_common.h",
                  "v2_device.h", "common.h")


def kernel_sources_sha():
    """Identity of the kernel sources the PMC traffic figures belong to (tools/collect_profiles.py stores it next to them)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "lars_image_processing_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def traffic_from_profiles(mode, pixels_per_launch):
    """(bytes per launch | None, provenance string).  HBM bytes of the fused kernel from the committed PMC summary
    (profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, corrected as
    MI355X_MICROARCH.md says).  The figure is only reported while the kernel sources still are the ones it was
    measured on; otherwise traffic is null and the provenance says why."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
    except (OSError, ValueError) as exc:
        return None, f"null: cannot read profiles/traffic.json ({exc.__class__.__name__})"
    measured_on, now = doc.get("_kernel_sources_sha"), kernel_sources_sha()
    if measured_on != now:
        return None, (f"null: profiles/traffic.json was measured on kernel sources {measured_on}, this build is {now} "
                      "(re-run tools/profile_round.sh + tools/collect_profiles.py)")
    if mode not in doc:
        return None, f"null: profiles/traffic.json@{now} has no row for mode {mode}"
    return doc[mode]["bytes_per_pixel"] * pixels_per_launch, f"profiles/traffic.json@{now} ({doc.get('_round', '?')}, rocprofv3 --pmc)"


def config4_leg(tiles=64, edge=8192, ring=16):
    """BASELINE configs[4] shape on this GPU (not the headline): uint16 8192 x 8192 tiles, percentile white balance,
    float32 NDVI + RdYlGn RGBA written (ring of 16 tile slots, arena chosen like the headline's) + statistics.  14 algorithmic bytes per
    pixel (6 read, 8 written)."""
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch(tiles, edge, edge, 3, np.uint16)
    _ffi.call("lars_d_synth_u8", C.c_void_p(b.tiles.ptr), tiles, 0, b.npix * 2, 3, 1234, 0, None)   # random 16-bit samples
    b.compute_wb_tables()
    outs = b.make_outputs(indices=("NDVI",), index=True, rgba=True, ring=ring)
    stats = b.new_stats()
    ev = [C.c_void_p() for _ in range(3)]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))
    prep, fused = [], []
    launches = 1
    for _ in range(4):
        _ffi.call("lars_event_record", ev[0], None)
        b.compute_wb_tables()
        _ffi.call("lars_event_record", ev[1], None)
        launches = b.run_fused_chunks(("NDVI",), True, stats, False, outs)
        _ffi.call("lars_event_record", ev[2], None)
        _ffi.call("lars_synchronize", None)
        ms = C.c_float(0)
        _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms)); prep.append(ms.value)
        _ffi.call("lars_event_elapsed_ms", ev[1], ev[2], C.byref(ms)); fused.append(ms.value)
    p_ms, f_ms = float(np.median(prep[1:])), float(np.median(fused[1:]))
    npix = tiles * edge * edge
    report = outs.arena_report
    outs.free(); stats.free(); b.free()
    return {"workload": f"{tiles} tiles of {edge}x{edge} uint16, white balance + float32 NDVI + RGBA8 written (ring of {ring}) + statistics",
            "Mpix_s": npix / ((p_ms + f_ms) * 1e-3) / 1e6, "wb_prepare_ms": p_ms, "fused_ms": f_ms, "launches": launches,
            "whole_step_frac": npix * 14 / ((p_ms + f_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "fused_GBs_algorithmic": npix * 14 / (f_ms * 1e-3) / 1e9,
            "fused_frac_of_8TBs": npix * 14 / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "arena": report}


def smooth_tile(edge, seed=7):
    """One image-like tile: slow gradients per channel plus two levels of noise (tools/jointbench.py's `smooth`): the 64 pixels a
    wave counts at a time fall into a 5 x 5 neighbourhood of byte pairs instead of 64 independent ones."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:edge, 0:edge].astype(np.float32)
    out = np.empty((edge, edge, 3), np.uint8)
    for c in range(3):
        g = 60 + 50 * c + 40 * np.sin(xx / 900.0 + c) + 30 * np.cos(yy / 700.0) + rng.integers(-2, 3, (edge, edge))
        out[:, :, c] = np.clip(g, 0, 255).astype(np.uint8)
    return out


def natural_tile(edge, seed=7, scale=55.0, centre=(120.0, 135.0, 150.0)):
    """One photograph-like tile (tools/jointbench.py's `natural`): 1 / f noise -- structure at every scale --, channels that share most
    of it, sensor noise on top, stretched over the whole 8-bit range so that both ends clip (0.3-0.7 % of the pixels sit on the
    (255, 255) byte pairs): broad histograms with piled-up ends, what a processed JPEG looks like to the counting kernels."""
    rng = np.random.default_rng(seed)
    fy, fx = np.meshgrid(np.fft.fftfreq(edge), np.fft.rfftfreq(edge), indexing="ij")
    amp = 1.0 / np.maximum(np.hypot(fy, fx), 1.0 / edge)
    common = np.fft.irfft2(amp * np.exp(2j * np.pi * rng.random(amp.shape)), s=(edge, edge))
    out = np.empty((edge, edge, 3), np.uint8)
    for c in range(3):
        own = np.fft.irfft2(amp * np.exp(2j * np.pi * rng.random(amp.shape)), s=(edge, edge))
        f = 0.8 * common + 0.6 * own
        f = (f - f.mean()) / f.std()
        out[:, :, c] = np.clip(centre[c] + scale * f + rng.normal(0, 1.5, (edge, edge)), 0, 255).astype(np.uint8)
    return out


def smooth_leg(tiles=256, edge=4096, rounds=4, kind="smooth"):
    """The statistics-only jobs on IMAGE-LIKE content (the reference's inputs are photographs, process-images.py:1441-1457; the
    bench's counter-hash tiles are iid): both routes timed, and what ``route="auto"`` picks (TileBatch.pick_stats_route; with
    medians the library always takes the one-read route).  One tile (``kind``: "smooth" gradients or a "natural" 1 / f scene); every
    other tile of the batch is that tile rolled by a pseudo-random number of rows."""
    import lars_image_processing_amd as lars
    from lars_image_processing_amd import _ffi
    b = lars.TileBatch(tiles, edge, edge, 3, np.uint8)
    # "natural": histograms over the whole range, clipped at both ends (two readers); "natural_mid": p2 .. p98 over 165 values per channel, the
    # range in which three windows (NIR as well) still share one workgroup's LDS
    tile = (smooth_tile(edge) if kind == "smooth" else natural_tile(edge) if kind == "natural" else
            natural_tile(edge, scale=40.0, centre=(120.0, 128.0, 136.0)))
    b.tiles.upload(tile[None])
    row = edge * 3
    for i in range(1, tiles):
        r = (i * 997) % edge
        dst = b.tiles.ptr + i * b.tile_bytes
        _ffi.call("lars_memcpy_d2d", C.c_void_p(dst), C.c_void_p(b.tiles.ptr + r * row), b.tile_bytes - r * row, None)
        if r:
            _ffi.call("lars_memcpy_d2d", C.c_void_p(dst + b.tile_bytes - r * row), C.c_void_p(b.tiles.ptr), r * row, None)
    _ffi.call("lars_synchronize", None)
    stats = b.new_stats()
    pairs = _ffi.DeviceBuffer(b.ntiles * 4 * 4)
    scratch = _ffi.DeviceBuffer(int(_ffi.load().lars_quotient_median_scratch_bytes(b.ntiles)))
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _ffi.call("lars_event_create", C.byref(e))

    def timed(fn):
        ts = []
        for _ in range(rounds + 1):
            _ffi.call("lars_event_record", ev[0], None)
            fn()
            _ffi.call("lars_event_record", ev[1], None)
            ms = C.c_float(0)
            _ffi.call("lars_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            ts.append(ms.value)
        return float(np.median(ts[1:]))

    npix = tiles * edge * edge
    what = ("gradients + two levels of noise per channel" if kind == "smooth" else
            "1/f noise with correlated channels + sensor noise over the whole 8-bit range, clipped at both ends" if kind == "natural" else
            "1/f noise with correlated channels + sensor noise, p2 .. p98 over 165 values per channel")
    out = {"workload": f"{tiles} row-rolled copies of one {edge}x{edge} uint8 tile: {what} (image-like; the headline's tiles are iid)",
           "algorithmic_bytes_per_pixel": 3}
    for name, indices, med in (("wb_ndvi_stats_only", ("NDVI",), False), ("wb3idx_stats_only", ("NDVI", "GNDVI", "NDWI"), False),
                               ("wb3idx_stats_medians", ("NDVI", "GNDVI", "NDWI"), True)):
        def classic():
            b.compute_wb_tables()
            a = b.fused_args(indices, True, stats)
            if med:
                _ffi.call("lars_d_stats_medians", C.byref(a), C.c_void_p(pairs.ptr), C.c_void_p(scratch.ptr))
            else:
                b.run_fused(a)
        c_ms = timed(classic)
        _ffi.call("lars_synchronize", None)
        want = stats.download(_ffi.STATS_DTYPE, (tiles, 3)).tobytes()             # every tile's records, not only the first's
        want_med = pairs.download(np.float32, (tiles, 2, 2)).tobytes() if med else None
        j_ms = timed(lambda: b.run_joint(indices, True, stats, pairs=pairs if med else None))
        b.check_joint()
        windowed, recounted = b.joint_window_report()
        three = b.joint_window_modes()[2]
        same = stats.download(_ffi.STATS_DTYPE, (tiles, 3)).tobytes() == want
        if med:
            same = same and pairs.download(np.float32, (tiles, 2, 2)).tobytes() == want_med
        b.__dict__.pop("_route_cache", None)
        auto = "joint" if med else b.pick_stats_route(indices, True)
        chosen = j_ms if auto == "joint" else c_ms
        out[name] = {"one_read_ms": j_ms, "per_pixel_ms": c_ms, "auto_route": "one read" if auto == "joint" else "per pixel",
                     "ms_per_step": chosen, "whole_step_frac": npix * 3 / (chosen * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "one_read_frac": npix * 3 / (j_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "per_pixel_frac": npix * 3 / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "records_identical": bool(same), "tiles_on_windowed_tables": windowed, "tiles_recounted": recounted,
                     "tiles_on_three_windows": three}
    for e in ev:
        _ffi.call("lars_event_destroy", e)
    stats.free(); pairs.free(); scratch.free(); b.free()
    return out


def _visible_devices():
    """Device count seen by a CHILD process: the launching parent must never initialise HIP itself (its children
    are new processes, and a process that has touched the GPU may not be replaced or forked into ranks)."""
    code = "from lars_image_processing_amd import _ffi; print(_ffi.device_count())"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit(f"bench.py: cannot count GPUs: {out.stderr[-500:]}")


def run_rank_processes(n, command, base_env):
    """Start ``n`` copies of ``command`` as ranks 0..n-1 (RANK / LOCAL_RANK added to ``base_env``), wait for all of
    them and return rank 0's stdout.  If any rank exits non-zero the others are ended (exactly the pids started here)
    and SystemExit is raised: a half-dead group would otherwise sit in its rendezvous until a timeout."""
    procs = []
    for r in range(n):
        env = dict(base_env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(command, env=env, cwd=ROOT, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    failed, pending = None, set(range(n))
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        else:
            time.sleep(0.05)
    if failed is not None:
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=15)
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
        procs[0].stdout.close()
        raise SystemExit(f"bench.py --gpus {n}: rank {failed[0]} exited with status {failed[1]}")
    out = procs[0].stdout.read()
    procs[0].stdout.close()
    return out


def launch_ranks(args):
    """``python bench.py --gpus N`` with no launcher environment: start N rank processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* and one shared rendezvous token), relay rank 0's JSON line, fail if any rank fails.
    Runs before anything of this process has loaded liblars_hip.so or touched HIP."""
    import secrets
    import shutil
    import socket
    import tempfile
    n = args.gpus
    rehearsal = os.environ.get("LARS_COMM") == "gloo"            # every rank on GPU LARS_DEVICE, statistics over gloo
    have = _visible_devices()
    if have < 1 or (have < n and not rehearsal):
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node (one rank per GPU; "
                         "LARS_COMM=gloo LARS_DEVICE=0 rehearses N ranks on one GPU)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    rdzv_dir = tempfile.mkdtemp(prefix="lars_rdzv_")
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                LARS_RDZV_TOKEN=secrets.token_hex(16), LARS_RDZV_DIR=rdzv_dir, LARS_BENCH_LAUNCHER="self")
    # The ranks of one node share device memory through dmabuf IPC handles (RCCL's intra-node transports, and any device buffer a
    # rank hands to another).  This image's host driver supports ONLY dmabuf IPC: with the legacy mode (the runtime's default when the
    # variable is unset) hipIpcGetMemHandle fails with "invalid argument" and ncclCommInitRank with nranks > 1 never comes up.  The
    # image exports the variable already; a launcher that builds its ranks' environment itself has to keep it.
    # setdefault: an operator's own setting wins.
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        out = run_rank_processes(n, [sys.executable, os.path.abspath(__file__), *sys.argv[1:]], base)
    finally:
        shutil.rmtree(rdzv_dir, ignore_errors=True)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        raise SystemExit(f"bench.py --gpus {n}: expected one JSON line from rank 0, got {len(lines)}")
    line = json.loads(lines[0])
    if line.get("n_gpus") != n or line.get("config", {}).get("ranks_seen") != n:
        raise SystemExit(f"bench.py --gpus {n}: the ranks report n_gpus={line.get('n_gpus')}, "
                         f"ranks_seen={line.get('config', {}).get('ranks_seen')}")
    print(lines[0])
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    from lars_image_processing_amd import _ffi
    from lars_image_processing_amd import dist
    rank, local_rank, world = dist.env_rank_world()
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, start it as "
                         f"`python bench.py --gpus N` or under torch.distributed.run with --nproc-per-node N")
    collective = "none (single process)"
    voted = False
    if world > 1 or os.environ.get("LARS_FORCE_RCCL"):
        # Every rank makes the same choice of transport: from the environment, or -- by default -- from a pre-flight vote:
        # each rank checks that librccl loads (lars_comm_available) and all ranks exchange that verdict through marker files
        # (dist.agree) BEFORE anybody enters a blocking bootstrap; only a unanimous yes takes the library's own communicator,
        # anything else takes torch.distributed's.  A bootstrap error after that ends the rank with a non-zero status and the
        # launcher tears the group down (no per-rank fallback, no second change of transport).
        #   LARS_COMM unset / auto: vote, then rccl or torch
        #   LARS_COMM=rccl: the library's own RCCL communicator (csrc/comm.cpp), no vote
        #   LARS_COMM=torch: the statistics exchange through torch.distributed (nccl backend = RCCL)
        #   LARS_COMM=gloo: rehearsal of the N > 1 flow on fewer GPUs than ranks -- statistics over torch.distributed's
        #                   gloo backend on the host, every rank on GPU LARS_DEVICE (default LOCAL_RANK)
        flavour = os.environ.get("LARS_COMM", "auto")
        if flavour == "auto":
            flavour = "rccl" if dist.agree(rank, world, _ffi.load().lars_comm_available() == 0) else "torch"
            voted = True
            if flavour == "torch":
                print(f"[bench rank {rank}] librccl is not usable on every rank ({_ffi.load().lars_last_error().decode()!r} here): "
                      "all ranks use torch.distributed", file=sys.stderr)
        comm = None
        try:
            if flavour == "gloo":
                _ffi.call("lars_set_device", int(os.environ.get("LARS_DEVICE", local_rank)))
                comm = dist.TorchComm.from_env("gloo")
                collective = "torch.distributed all_gather (gloo, host) + rank-order fold -- rehearsal transport"
            elif flavour == "torch":
                comm = dist.TorchComm.from_env("nccl")
                collective = "torch.distributed all_gather (nccl backend = RCCL) + rank-order fold"
            elif flavour == "rccl":
                comm = dist.Comm.from_env()
                collective = "RCCL ncclAllGather of packed records + rank-order fold (csrc/comm.cpp)"
            else:
                raise SystemExit(f"LARS_COMM={flavour}: expected rccl, torch or gloo")
        except (_ffi.LarsError, TimeoutError, OSError, RuntimeError) as exc:
            print(f"[bench rank {rank}] {flavour} bootstrap failed: {exc}", file=sys.stderr)
            raise SystemExit(3)
    else:
        _ffi.call("lars_set_device", 0)
        comm = dist.SingleProcessComm()
    if world > 1 and voted:
        # the markers of the vote may only go once EVERY rank has left agree(): the ranks meet in a barrier first
        dist.release_agreement(comm, rank, world, phases=("pre",))
    ranks_seen = comm.ranks_seen()
    if ranks_seen != world:
        print(f"[bench rank {rank}] the communicator reports {ranks_seen} ranks, WORLD_SIZE={world}", file=sys.stderr)
        raise SystemExit(4)

    runner = Runner(args, comm, rank, world)
    npix_rank = args.tiles * args.tile * args.tile
    if args.all_modes is None:
        args.all_modes = world == 1          # N > 1: only the headline mode is timed unless --all-modes (the line stays cheap)
    dt, timed, glob = runner.run(args.mode, args.steps, args.warmup, per_launch=True)
    total_pix = npix_rank * world * args.steps
    value = total_pix / dt / 1e6
    local_step_ms = runner.last_local_dt / args.steps * 1e3

    indices, write, hist, bpp = MODES[args.mode]
    fused_ms = float(np.mean([t[1] for t in timed]))
    hist_ms = float(np.mean([t[0] for t in timed]))
    launches = timed[0][2]
    bytes_per_launch = npix_rank * bpp / launches
    achieved = bytes_per_launch / (fused_ms / launches * 1e-3) / 1e9
    traffic, traffic_source = traffic_from_profiles(args.mode, npix_rank / launches)
    step_ms = dt / args.steps * 1e3
    glob_by_mode = {args.mode: glob}

    extra = {}
    if args.all_modes:
        names = [m for m in MODES if m != args.mode]
        if args.stats_route == "joint" and runner.batch.can_joint():
            names += [m + "_classic" for m in STATS_ONLY]
        for m in names:
            base = m[:-len("_classic")] if m.endswith("_classic") else m
            k_steps = max(2, args.steps // 2)
            d, tm, g_m = runner.run(m, k_steps, 1)
            glob_by_mode[m] = g_m
            f_ms = float(np.mean([t[1] for t in tm]))
            m_step_ms = d / k_steps * 1e3
            one_read = base in STATS_ONLY and not m.endswith("_classic") and args.stats_route == "joint" and runner.batch.can_joint()
            extra[m] = {
                "Mpix_s": npix_rank * world * k_steps / d / 1e6,
                "ms_per_step": m_step_ms,
                "whole_step_frac": npix_rank * MODES[base][3] / (m_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "route": ("one read: joint byte-pair histograms (windowed tables, one reader per tile chunk: k_joint_predict + "
                          "k_joint_count_win + k_joint_finish; full tables, two readers, where the windows do not fit or one value "
                          "stream is counted: k_joint_count)" if one_read else
                          "one-read statistics pass (k_joint_count_win / k_joint_count + k_joint_finish), then k_fused_u8c3 planes only"
                          if runner.joint_then_planes(base) else
                          "channel-histogram pass + tables, then the fused kernel"),
                "fused_ms": f_ms, "hist_pass_ms": float(np.mean([t[0] for t in tm])),
                "fused_GBs_algorithmic": npix_rank * MODES[base][3] / (f_ms * 1e-3) / 1e9,
                "fused_frac_of_8TBs": npix_rank * MODES[base][3] / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            }
            if one_read:
                # how the one-read pass counted this rank's tiles: on windowed pair tables (one reader per tile chunk), and how many
                # of those had to be counted again on full tables because a window missed
                extra[m]["tiles_on_windowed_tables"], extra[m]["tiles_recounted"] = runner.batch.joint_window_report()

    # self-check, after the timed region: the timed configuration's records / planes against single-tile runs, and the
    # global statistics of every mode that ran against each other
    verified = None
    if args.verify:
        verified = runner.verify(args.mode, glob_by_mode)
        if args.all_modes:
            for m in MEDIAN_MODES:
                v = runner.verify(m, {})
                verified[m] = {k: v[k] for k in ("records", "planes", "medians", "ok")}
                verified["ok"] = bool(verified["ok"] and v["ok"])
        ok_all = float(comm.allreduce_f64([1.0 if verified["ok"] else 0.0], "min")[0]) == 1.0
        verified["ok_on_every_rank"] = ok_all

    # exact global medians (all tiles of all ranks) by radix select on recomputed values: informational, untimed region
    medians, median_ms = None, None
    if args.all_modes:
        _ffi.call("lars_synchronize", None)
        comm.barrier()
        t0 = time.perf_counter()
        medians = runner.batch.global_medians(indices, white_balance=True, comm=comm)
        median_ms = (time.perf_counter() - t0) * 1e3

    # per-rank figures, gathered over the communicator: which rank sets the step time, and with what arena
    outs_main = runner.outputs.get(tuple(indices))
    placement = getattr(outs_main, "placement_ms", None) or {}
    arena_report = getattr(outs_main, "arena_report", None) or {}
    first_ms = runner.first_step_launch_ms or [0.0]
    mine = [rank, local_step_ms, fused_ms, hist_ms, fused_ms / launches, float(arena_report.get("chosen_ms") or 0.0),
            float(arena_report.get("search_ms") or 0.0), float(arena_report.get("rejected") or 0),
            float(arena_report.get("post_free_ms") or 0.0), float(min(first_ms)), float(max(first_ms))]
    per_rank = comm.allgather_f64(mine)

    probe = device_probe(runner) if (args.probe and rank == 0) else None
    if args.all_modes and world == 1 and args.u16_leg:
        extra["u16_8192_ndvi_rgba_out_stats"] = config4_leg()
    if args.all_modes and world == 1 and args.smooth_leg:
        extra["smooth_content"] = smooth_leg(tiles=min(256, args.tiles), edge=args.tile)
        extra["natural_content"] = smooth_leg(tiles=min(256, args.tiles), edge=args.tile, kind="natural")
        extra["natural_mid_content"] = smooth_leg(tiles=min(256, args.tiles), edge=args.tile, kind="natural_mid")
    if rank == 0:
        cpu = None if args.no_cpu_baseline or world > 1 else cpu_baseline(args)
        g = {t: runner.lb.summarize(glob[_ffi.INDEX_IDS[t]]) for t in indices}
        line = {
            "metric": "Mpixels/sec NDVI+stats on 4096x4096 RGNir tiles; achieved HBM GB/s",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.tiles}-tile batch per GPU of {args.tile}x{args.tile} uint8 RGNir ({args.profile} "
                            f"counter-hash tiles generated in HBM), mode {args.mode}: percentile white balance + "
                            f"{'/'.join(indices)}" + (" float32 planes written" if write else " (stats only)") +
                            " + min/max/mean/coverage" + ("/50-bin histogram" if hist else "") + " per tile, "
                            "then global fold" + (" over RCCL" if world > 1 else ""),
                "tiles_per_gpu": args.tiles, "tile": [args.tile, args.tile, 3], "input_dtype": "u8", "mode": args.mode,
                "output_ring_tiles": args.ring if write else 0,
                # ms per launch into each candidate arena (the fastest was kept); outside the timed region, like the warm-up
                "output_arena_trial_ms": placement.get("arenas"),
                # how the output arena came about (kind, search_ms, chosen_ms = ms per probe launch over one group of tile
                # slots, rejected candidates); outside the timed region, like the warm-up
                "arena": arena_report or None,
                "stats_route": args.stats_route,
                "parallelism": f"tile-sharded x{world}",
                "collective": collective, "ranks_seen": ranks_seen,
                # True only when the exchange of the global statistics ran on the library's own RCCL communicator (csrc/comm.cpp:
                # ncclAllGather over xGMI) -- a fall-back to torch.distributed or the gloo rehearsal shows here, not only in the text above
                "rccl_used": collective.startswith("RCCL ncclAllGather"),
                # which BASELINE.json configuration the line is: [1] per GPU by default (1024 tiles); [3] is 16384 tiles over 8 GPUs
                "baseline_config": ("configs[3]: 16384 tiles over 8 GPUs" if world == 8 and args.tiles == 2048 and args.mode == "wb3idx_out_stats"
                                    else f"configs[1] per GPU ({args.tiles} tiles each; configs[3] = --gpus 8 --tiles 2048)"
                                    if args.mode == "wb3idx_out_stats" else f"mode {args.mode} (the headline is wb3idx_out_stats)"),
                "build_flags": int(_ffi.load().lars_build_flags()),
                # the headline's output arena, flat (the nested report above may be dropped by a recorder): allocations taken,
                # bytes held during the search, bytes kept
                "arena_allocations": (arena_report or {}).get("allocations"),
                "arena_transient_bytes": (arena_report or {}).get("transient_bytes"),
                "arena_bytes": (arena_report or {}).get("arena_bytes"),
                "statistics_fold": "per-index fold on the device (lars_d_stats_fold), 3 x 472 B per rank exchanged",
                "launcher": os.environ.get("LARS_BENCH_LAUNCHER", "external" if world > 1 else "none"),
                "device": _ffi.device_name(),
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_fused_u8c3 (outputs) / k_joint_count (statistics only)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                # the whole step against the same peak: algorithmic bytes of the step (each input byte once, each output
                # byte once -- the percentile pre-pass's second read of the input is not algorithmic) / ms_per_step
                "whole_step_frac": npix_rank * bpp / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_pixel": bpp, "bytes_per_launch": bytes_per_launch,
                "launches_per_step": launches, "avg_launch_ms": fused_ms / launches,
                # the fused launches of the first timed step, one by one (rank 0): what the arena probe's post_free_ms predicts
                "first_step_launch_ms": runner.first_step_launch_ms,
                "arena_probe_vs_steps": (None if not arena_report.get("post_free_ms") else
                                         {"post_free_ms": arena_report["post_free_ms"], "avg_launch_ms": fused_ms / launches,
                                          "rel_diff": arena_report["post_free_ms"] / (fused_ms / launches) - 1.0}),
            },
            "passes_ms": {"histogram+tables": hist_ms, "fused": fused_ms, "rest_of_step": step_ms - hist_ms - fused_ms},
            "ranks": [{"rank": int(r[0]), "ms_per_step": r[1], "fused_ms": r[2], "hist_ms": r[3], "avg_launch_ms": r[4],
                       "arena_ms": r[5], "arena_search_ms": r[6], "arena_rejected": int(r[7]), "arena_post_free_ms": r[8],
                       "first_step_launch_ms_min_max": [r[9], r[10]]} for r in per_rank],
            "cpu_baseline": cpu,
            "verified": verified,
            "global_stats": {t: {k: v for k, v in s.items() if k != "hist"} for t, s in g.items()},
        }
        if medians is not None:
            for t in indices:
                line["global_stats"][t]["median"] = medians[t]
            line["global_median_ms"] = median_ms
        if extra:
            line["modes"] = extra
        if probe:
            line["device_probe_GBs"] = probe
        print(json.dumps(line))
    for e in runner.launch_ev:
        _ffi.call("lars_event_destroy", e)
    runner.launch_ev = []
    comm.destroy()
    if verified is not None and not verified["ok_on_every_rank"]:
        print(f"[bench rank {rank}] self-check FAILED: {verified}", file=sys.stderr)
        raise SystemExit(5)


if __name__ == "__main__":
    main()
