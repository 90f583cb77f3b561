#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name.

    python tools/pmc_summary.py gpurun_out/pmc_dir [substring filter]

lars_d_stats_joint launches its counting and finish kernels a second time for the tiles whose window missed (and the full-table
counting kernel once for the tiles without a window): launches whose workgroups find nothing to do and leave at once.  A dispatch
is counted as such -- and left out of its kernel's mean -- when its value is below 1 % of the largest dispatch of the same kernel,
grid and counter; their number is printed.
"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if filt and filt not in name:
                    continue
                short = name.split("(")[0].replace("void lars::", "").replace("lars::", "")
                key = (short, row.get("Grid_Size", ""), row.get("LDS_Block_Size", ""), row.get("VGPR_Count", ""))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key in sorted(acc):
        print(f"{key[0]} grid={key[1]} lds={key[2]} vgpr={key[3]}")
        for c in sorted(acc[key]):
            v = acc[key][c]
            top = max(v)
            work = [x for x in v if x >= 0.01 * top] if top > 0 and "k_joint" in key[0] else v
            idle = len(v) - len(work)
            print(f"    {c:28s} mean={sum(work)/len(work):16.1f} n={len(work)}" + (f"  (+{idle} launches that found nothing to do)" if idle else ""))

if __name__ == "__main__":
    main()
