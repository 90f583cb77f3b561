#!/usr/bin/env bash
# rocprofv3 evidence for the one-read statistics route (csrc/joint.hip), on a GPU box:
#   gpurun --timeout 900 -- 'bash tools/jointprof.sh'        -> gpurun_out/prof_joint/
# 1. kernel trace + stats of tools/jointbench.py (per-kernel durations: k_joint_count, k_joint_finish and the classic kernels)
# 2. PMC passes, one group of counters per run (no trace flags): HBM traffic, instruction mix, LDS conflicts
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_joint
TILES=${TILES:-256}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/jointbench.py" --tiles $TILES --rounds 3 \
    --content vegetation --depths ${DEPTHS:-6} > "$OUT/trace.log" 2> "$OUT/trace.err"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 "$R/tools/jointbench.py" --tiles $TILES --rounds 1 \
        --content vegetation --depths ${DEPTHS:-6} > "$OUT/pmc_$i.log" 2> "$OUT/pmc_$i.err"
    echo "pmc group $i done: $grp"
done
python3 "$R/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_summary.txt" || true
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cut -c1-160 "$f" | head -20
