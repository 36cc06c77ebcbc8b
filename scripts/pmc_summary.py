#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per kernel (substring match), the per-launch average of every counter and of the
dispatch duration.  Usage: python scripts/pmc_summary.py <kernel substring> dir1 [dir2 ...]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    pat, dirs = sys.argv[1], sys.argv[2:]
    vals, durs = defaultdict(list), []
    regs = None
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            seen = set()
            for row in csv.DictReader(open(f)):
                if pat not in row["Kernel_Name"]:
                    continue
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                regs = (row["VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print(f"kernel ~ '{pat}': VGPR {regs[0]}, SGPR {regs[1]}, LDS {regs[2]} B, scratch {regs[3]}; "
          f"{len(durs)} profiled launches, average {sum(durs) / len(durs) / 1e6:.3f} ms")
    print("| counter | per launch |\n|---|---|")
    for k in sorted(vals):
        print(f"| {k} | {sum(vals[k]) / len(vals[k]):.4g} |")


if __name__ == "__main__":
    main()
