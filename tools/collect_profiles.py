#!/usr/bin/env python3
"""profiles/<round>_* from what tools/profile_round.sh left under gpurun_out/prof_<round>/ (run in the container)."""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read(rnd, counter):
    out, name = {}, None
    for line in open(f"{ROOT}/profiles/{rnd}_pmc_{counter}.txt"):
        if not line.startswith(" "):
            name = line.split(" grid=")[0].strip()
        else:
            m = re.search(r"mean=\s*([\d.]+)", line)
            if m:
                out[name] = float(m.group(1))
    return out


def headline_launches(rnd, src):
    """The headline kernel's launches of the profiled bench run, split by what they were: the arena search (untimed and timed
    passes of the probe into every candidate, and once more into the survivor) and the steps (warm-up, timed and self-check
    steps).  A step is recognised by its channel-histogram pass: the `launches_per_step` ring launches of the headline kernel
    that follow a full-batch k_chan_hist_u8c3_v2 launch.  Written to profiles/<round>_bench_headline_launches.csv so that
    bytes / mean duration / 8 TB/s of the 'step' row can be held against roofline.frac of <round>_bench_under_rocprof.json."""
    import csv
    line = json.loads(open(f"{src}/bench_under_rocprof.json").read().strip().splitlines()[-1])
    ncand = len((line["config"].get("arena") or {}).get("candidate_ms") or [])
    per_step = int(line["roofline"]["launches_per_step"])
    trace = max(glob.glob(f"{src}/trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    allrows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    head = "void lars::k_fused_u8c3<unsigned char, 7u, true, 1, 3>"

    def grid(r):
        return (r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y", ""))
    rows = [r for r in allrows if r["Kernel_Name"].startswith(head)]
    ring_grid = max(set(grid(r) for r in rows), key=lambda g: sum(1 for r in rows if grid(r) == g))
    hist = [r for r in allrows if "k_chan_hist_u8c3_v2<3>" in r["Kernel_Name"]]
    hist_grid = max(set(grid(r) for r in hist), key=lambda g: int(g[1] or 1))        # grid.y = tiles: the pass over the whole batch
    # groups of headline launches between a k_stats_init and the next k_stats_finalize: a step's group has exactly `launches_per_step`
    # of them and a full-batch histogram pass since the previous group; the arena probe's groups are longer (untimed + timed passes)
    steps, search, group, saw_hist = [], [], [], False
    for r in allrows:
        name = r["Kernel_Name"]
        if "k_chan_hist_u8c3_v2<3>" in name and grid(r) == hist_grid:
            saw_hist = True
        elif name.startswith(head) and grid(r) == ring_grid:
            group.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        elif "k_stats_finalize" in name and group:
            (steps if (len(group) == per_step and saw_hist) else search).extend(group)
            group, saw_hist = [], False
    search.extend(group)
    nbytes = line["roofline"]["bytes_per_launch"]
    out = f"{ROOT}/profiles/{rnd}_bench_headline_launches.csv"
    with open(out, "w") as fh:
        fh.write("class,launches,mean_ns,GBs_algorithmic,frac_of_8TBs\n")
        for name, d in (("arena_search", search), ("step", steps)):
            if d:
                mean = sum(d) / len(d)
                fh.write(f"{name},{len(d)},{mean:.1f},{nbytes / mean:.1f},{nbytes / mean / 8000:.4f}\n")
        fh.write(f"# bench line of the same run: roofline.frac {line['roofline']['frac']:.4f}, avg_launch_ms {line['roofline']['avg_launch_ms']:.4f}, "
                 f"{ncand} candidate arenas (the search's launches include the untimed passes of its probe and every candidate's class); "
                 f"other launches of this kernel (single tiles of the self-check): {len(rows) - len(steps) - len(search)}\n")
    print(open(out).read())


def joint_launches(rnd, src):
    """The one-read statistics route's launches of the profiled bench run, by kernel and grid (= batch size): count, mean and
    extreme durations, and for the counting kernels the algorithmic rate (3 B per pixel: every input byte once).  Launches whose
    workgroups found nothing to do (the recount of tiles whose window missed, when none did; the full-table kernel beside a
    fully windowed batch) are listed on a line of their own.  -> profiles/<round>_joint_launches.csv"""
    import csv
    trace = max(glob.glob(f"{src}/trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    groups = {}
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"]
        if "k_joint_" not in name:
            continue
        short = name.split("(")[0].replace("void lars::", "").replace("lars::", "")
        wgs = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
        wgs *= max(1, int(r.get("Grid_Size_Y", 1) or 1))
        groups.setdefault((short, wgs), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = f"{ROOT}/profiles/{rnd}_joint_launches.csv"
    with open(out, "w") as fh:
        fh.write("kernel,workgroups,launches,mean_us,min_us,max_us,idle_launches,GBs_algorithmic,frac_of_8TBs\n")
        for (short, wgs), d in sorted(groups.items()):
            top = max(d)
            work = [x for x in d if x >= 0.05 * top]
            mean = sum(work) / len(work)
            rate = ""
            if short.startswith("k_joint_count"):
                # tiles of 4096 x 4096: windowed kernel one workgroup per (tile, chunk); full-table kernel two per unit with two streams
                tiles = {1024: 1024, 512: 256, 2048: 1024}.get(wgs)
                if tiles and top > 1e6:
                    gbs = tiles * 4096 * 4096 * 3 / mean
                    rate = f"{gbs:.1f},{gbs / 8000:.4f}"
            fh.write(f"{short},{wgs},{len(work)},{mean / 1e3:.1f},{min(work) / 1e3:.1f},{max(work) / 1e3:.1f},{len(d) - len(work)},{rate or ','}\n")
    print(open(out).read())


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = f"{ROOT}/gpurun_out/prof_{rnd}"
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        with open(f"{ROOT}/profiles/{rnd}_pmc_{c}.txt", "w") as fh:
            subprocess.run([sys.executable, f"{ROOT}/tools/pmc_summary.py", f"{src}/pmc_{c}", "k_"], stdout=fh, check=True)
    # gpurun merges every call's files into gpurun_out/: take the newest trace (and clear prof_<round>/ between runs)
    stats = max(glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    shutil.copy(stats, f"{ROOT}/profiles/{rnd}_bench_kernel_stats.csv")
    shutil.copy(f"{src}/bench_under_rocprof.json", f"{ROOT}/profiles/{rnd}_bench_under_rocprof.json")
    headline_launches(rnd, src)
    joint_launches(rnd, src)
    f, w = read(rnd, "FETCH_SIZE"), read(rnd, "WRITE_SIZE")
    px64, px256 = 64 * 4096 * 4096, 256 * 4096 * 4096
    # kernel names as rocprofv3 prints them, up to the template arguments that only tuning knobs change (matched by prefix)
    rows = {
        "wb3idx_out_stats": ("k_fused_u8c3<unsigned char, 7u, true, 1, 3>", px64),
        "wb3idx_out_stats_hist": ("k_fused_u8c3<unsigned char, 7u, true, 2, 3>", px64),
        "wb_ndvi_out_stats": ("k_fused_u8c3<unsigned char, 1u, true, 0, 3>", px64),       # planes only; its statistics pass is k_joint_count
        "wb3idx_out_stats_medians": ("k_fused_u8c3<unsigned char, 7u, true, 0, 3>", px64),      # planes only; its statistics pass is k_joint_count
        # the one-read statistics route: one counting launch per step over the whole batch.  Two value streams on the bench's tiles:
        # windowed tables, one reader per tile chunk (k_joint_count_win); one stream: the full table (k_joint_count).  The launches of
        # lars_d_stats_joint whose workgroups find nothing to do are left out of the means (tools/pmc_summary.py)
        "wb3idx_stats_only": ("k_joint_count_win<", px256),
        "wb_ndvi_stats_only": ("k_joint_count<6, 3>", px256),
        "wb3idx_stats_medians": ("k_joint_count_win<", px256),
        "joint_predict": ("k_joint_predict<3>", px256),
        "joint_finish": ("k_joint_finish", px256),
        "wb3idx_stats_only_classic": ("k_fused_v2<7u, true, 1, false, false, 0,", px256),
        "wb_ndvi_stats_only_classic": ("k_fused_v2<1u, true, 1, false, false, 0,", px256),
        "channel_hist": ("k_chan_hist_u8c3_v2<3>", px256),
    }

    def find(table, prefix):
        hits = [k for k in table if k.startswith(prefix)]
        return hits[0] if len(hits) >= 1 else None
    t = {"_comment": "HBM bytes per pixel from rocprofv3 PMC (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes): "
                     "(2*FETCH_SIZE + WRITE_SIZE)*1024 / pixels per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md "
                     "(HBM) prescribes for wide coalesced streaming reads on gfx950.  _kernel_sources_sha identifies "
                     "the kernel sources the counters were collected on (bench.kernel_sources_sha); bench.py reports "
                     "roofline.traffic only while it still matches."}
    from bench import kernel_sources_sha
    t["_kernel_sources_sha"] = kernel_sources_sha()
    t["_round"] = rnd
    for mode, (prefix, px) in rows.items():
        kf, kw = find(f, prefix), find(w, prefix)
        if kf and kf == kw:
            t[mode] = {"bytes_per_pixel": (2 * f[kf] + w[kw]) * 1024 / px, "kernel": kf}
    json.dump(t, open(f"{ROOT}/profiles/traffic.json", "w"), indent=1)
    print({k: round(v["bytes_per_pixel"], 4) for k, v in t.items() if not k.startswith("_")})


if __name__ == "__main__":
    main()
