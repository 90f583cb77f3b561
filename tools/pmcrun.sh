#!/usr/bin/env bash
# rocprofv3 evidence for any of the tools, on a GPU box (run from the repository root):
#   gpurun --timeout 900 -- 'bash tools/pmcrun.sh u16 tools/u16bench.py 32'      -> gpurun_out/prof_u16/{trace,pmc_1..4,pmc_summary.txt}
# 1. kernel trace + stats   2. PMC passes, one group of counters per run, never together with a trace flag:
#    HBM traffic (FETCH_SIZE / WRITE_SIZE), instruction mix, LDS conflicts and waits
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
SCRIPT=$R/$1; shift
OUT=$R/gpurun_out/prof_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$SCRIPT" "$@" > "$OUT/trace.log" 2> "$OUT/trace.err"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_$i.log" 2> "$OUT/pmc_$i.err"
    echo "pmc group $i done: $grp"
done
python3 "$R/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_summary.txt" || true
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kernel_stats.csv"
cut -c1-170 "$f" | head -24
