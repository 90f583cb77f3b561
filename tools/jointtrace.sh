#!/usr/bin/env bash
# Kernel trace of tools/jointbench.py (one-read statistics route) and the launch-by-launch timeline of every windowed call:
#   gpurun -- 'bash tools/jointtrace.sh NAME [jointbench arguments]'   -> gpurun_out/trace_NAME/{timeline.txt,kernel_stats.csv,bench.txt}
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/trace_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/jointbench.py" "$@" > "$OUT/bench.txt" 2> "$OUT/trace.err"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
python3 - "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" > "$OUT/timeline.txt" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void lars::", "").replace("lars::", "")
i = 0
while i < len(rows):
    if "k_joint_predict" in rows[i]["Kernel_Name"]:
        t0 = int(rows[i]["Start_Timestamp"])
        seq = []
        for r in rows[i:i + 6]:
            seq.append("%s @%.0f %.0f us" % (short(r["Kernel_Name"]), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        print(" | ".join(seq))
        i += 6
    else:
        i += 1
PY
rm -rf "$OUT/trace"
cat "$OUT/bench.txt" | grep -v classic
awk 'NR % 4 == 1' "$OUT/timeline.txt"
