#!/usr/bin/env python3
"""HBM bytes per frame of a bench configuration from two rocprofv3 PMC passes (WRITE_SIZE, FETCH_SIZE; KiB units, FETCH_SIZE
doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B).  Production kernels only (COUNT = 0
instantiations, wavefront kernels, tile-order kernels); a frame = one dispatch of the primary kernel.
Usage: python scripts/hbm_summary.py <config> <dir with WRITE_SIZE/ and FETCH_SIZE/>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def production(name):
    if "wf_" in name or "tile_hist" in name or "tile_scatter" in name:
        return True
    m = re.search(r"render_kernel<(\w+), (\d+)", name)
    return bool(m) and m.group(2) == "0"


def primary(name):
    return "wf_march_kernel<0" in name or re.search(r"render_kernel<\w+, 0", name) is not None


def main():
    cfg, root = sys.argv[1], sys.argv[2]
    out = {}
    frames = None
    per_kernel = {}
    for counter in ("WRITE_SIZE", "FETCH_SIZE"):
        tot = defaultdict(float)
        disp = defaultdict(set)
        for f in glob.glob(f"{root}/{counter}/**/*_counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                n = row["Kernel_Name"]
                if row["Counter_Name"] != counter or not production(n):
                    continue
                tot[n] += float(row["Counter_Value"])
                disp[n].add(row["Dispatch_Id"])
        fr = sum(len(v) for k, v in disp.items() if primary(k))
        frames = fr
        out[counter] = sum(tot.values()) * 1024.0 / max(fr, 1)
        for k in tot:
            short = k.split("(")[0].replace("void ", "")
            per_kernel.setdefault(short, {})[counter] = round(tot[k] * 1024.0 / max(fr, 1) / 1e6, 2)
    total = out["WRITE_SIZE"] + 2.0 * out["FETCH_SIZE"]
    print(json.dumps({cfg: {"bytes_per_launch": round(total), "write_bytes": round(out["WRITE_SIZE"]), "fetch_bytes_x2": round(2 * out["FETCH_SIZE"]),
                            "frames_profiled": frames, "per_kernel_MB": per_kernel}}))


if __name__ == "__main__":
    main()
