#!/usr/bin/env bash
# Every kernel launch of a python tool, in order, with its duration:
#   gpurun -- 'bash tools/ktrace.sh NAME tools/jointbench.py --tiles 128 ...'   -> gpurun_out/ktrace_NAME/{launches.txt,out.txt}
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
PROG=$R/$1; shift
OUT=$R/gpurun_out/ktrace_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$PROG" "$@" > "$OUT/out.txt" 2> "$OUT/trace.err"
python3 - "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" > "$OUT/launches.txt" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void lars::", "").replace("lars::", "")
    print("%10.0f us  %8.1f us  grid %-8s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                              r.get("Grid_Size", "?"), name))
PY
rm -rf "$OUT/trace"
