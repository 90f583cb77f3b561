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
    f, w = read(rnd, "FETCH_SIZE"), read(rnd, "WRITE_SIZE")
    px64, px256 = 64 * 4096 * 4096, 256 * 4096 * 4096
    # kernel names as rocprofv3 prints them, up to the template arguments that only tuning knobs change (matched by prefix)
    rows = {
        "wb3idx_out_stats": ("k_fused_u8c3<unsigned char, 7u, true, 1, 1>", px64),
        "wb3idx_out_stats_hist": ("k_fused_u8c3<unsigned char, 7u, true, 2, 1>", px64),
        "wb_ndvi_out_stats": ("k_fused_u8c3<unsigned char, 1u, true, 1, 1>", px64),
        "wb3idx_stats_only": ("k_fused_v2<7u, true, 1, false, false, 0,", px256),
        "wb_ndvi_stats_only": ("k_fused_v2<1u, true, 1, false, false, 0,", px256),
        "channel_hist": ("k_chan_hist_u8c3_v2", px256),
    }

    def find(table, prefix):
        hits = [k for k in table if k.startswith(prefix)]
        return hits[0] if len(hits) == 1 else None
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
