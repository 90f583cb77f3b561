#!/usr/bin/env bash
# Regenerates the rocprofv3 evidence under profiles/ for round $1 (default r01) on a GPU box:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# 1. kernel-trace --stats of bench.py (per-kernel average durations)
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (HBM traffic per launch)
# 3. profiles/traffic.json (bytes per pixel, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)
set -euo pipefail
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p "$OUT" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 3 --warmup 1 \
    --no-cpu-baseline --no-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 "$R/bench.py" --steps 2 --warmup 1 --tiles 256 \
        --no-cpu-baseline --no-probe > /dev/null 2> "$OUT/pmc_$c.err"
    python3 "$R/tools/pmc_summary.py" "$OUT/pmc_$c" k_ > "$R/profiles/${ROUND}_pmc_$c.txt"
done
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$R/profiles/${ROUND}_bench_kernel_stats.csv"
cp "$OUT/bench_under_rocprof.json" "$R/profiles/${ROUND}_bench_under_rocprof.json"
python3 - "$R" "$ROUND" <<'PY'
import json, re, sys
root, rnd = sys.argv[1], sys.argv[2]
def read(counter):
    out, name = {}, None
    for line in open(f"{root}/profiles/{rnd}_pmc_{counter}.txt"):
        if not line.startswith(" "):
            name = line.split(" grid=")[0].strip()
        else:
            m = re.search(r"mean=\s*([\d.]+)", line)
            if m:
                out[name] = float(m.group(1))
    return out
f, w = read("FETCH_SIZE"), read("WRITE_SIZE")
px64, px256 = 64 * 4096 * 4096, 256 * 4096 * 4096
rows = {
    "wb3idx_out_stats": ("k_fused_u8c3<unsigned char, 7u, true, 1>", px64),
    "wb3idx_out_stats_hist": ("k_fused_u8c3<unsigned char, 7u, true, 2>", px64),
    "wb_ndvi_out_stats": ("k_fused_u8c3<unsigned char, 1u, true, 1>", px64),
    "wb3idx_stats_only": ("k_fused_v2<7u, true, 1, false, false>", px256),
    "wb_ndvi_stats_only": ("k_fused_v2<1u, true, 1, false, false>", px256),
    "channel_hist": ("k_chan_hist_u8c3_v2", px256),
}
t = {"_comment": "HBM bytes per pixel from rocprofv3 PMC (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes): "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024 / pixels per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md "
                 "(HBM) prescribes for wide coalesced streaming reads on gfx950."}
for mode, (kernel, px) in rows.items():
    if kernel in f and kernel in w:
        t[mode] = {"bytes_per_pixel": (2 * f[kernel] + w[kernel]) * 1024 / px, "kernel": kernel}
json.dump(t, open(f"{root}/profiles/traffic.json", "w"), indent=1)
print({k: round(v["bytes_per_pixel"], 4) for k, v in t.items() if k != "_comment"})
PY
echo "profiles/${ROUND}_* written"
