#!/usr/bin/env bash
# Collects the rocprofv3 evidence of round $1 (default r01) on a GPU box into gpurun_out/prof_<round>/ (the only
# directory that travels back); tools/collect_profiles.py then writes the summaries into profiles/:
#   rm -rf gpurun_out/prof_r03; gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03' && python tools/collect_profiles.py r03
# 1. kernel-trace --stats of bench.py (per-kernel average durations)
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (HBM traffic per launch)
# 3. profiles/traffic.json (bytes per pixel, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)
set -euo pipefail
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p "$OUT" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
if [ "${PMC_ONLY:-0}" != "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 3 --warmup 1 \
    --no-cpu-baseline --no-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
fi
# (the smooth-content leg stays out of the counter passes: its batch is fetched with another mix of 64- and 128-byte requests, which the
#  FETCH_SIZE formula counts differently -- 9.38 M instead of 6.29 M KiB for the same 12 GiB in the same time -- and the summaries average per kernel name)
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 "$R/bench.py" --steps 2 --warmup 1 --tiles 256 \
        --no-cpu-baseline --no-probe --no-verify --no-u16-leg --no-smooth-leg --arena plain --placement-trials 0 > /dev/null 2> "$OUT/pmc_$c.err"
done
echo "collected under gpurun_out/prof_$ROUND; back in the container: python tools/collect_profiles.py $ROUND"
