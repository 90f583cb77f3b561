#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name.

    python tools/pmc_summary.py gpurun_out/pmc_dir [substring filter]
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
            print(f"    {c:28s} mean={sum(v)/len(v):16.1f} n={len(v)}")

if __name__ == "__main__":
    main()
