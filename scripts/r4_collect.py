#!/usr/bin/env python3
"""Copy the summaries of scripts/gpu_r4_final.sh (gpurun_out/r04_final/) into profiles/ under round-4 names and derive the two JSON
files bench.py reads (profiles/r04_hbm_traffic.json, profiles/r04_pmc.json).  Prints DESIGN.md's table of configurations.
Runs in the container (no GPU).  Usage: python scripts/r4_collect.py [tag]   (tag: the letter of the session, default g)"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r04_final")
PRO = os.path.join(ROOT, "profiles")
SIZES = {"c1": (256, 256), "c2": (1920, 1080), "c3": (3840, 2160), "c4": (3840, 2160), "c5": (7680, 4320)}


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "g"
    traffic, rows = {}, []
    pmc = json.load(open(os.path.join(SRC, "pmc.json"))) if os.path.exists(os.path.join(SRC, "pmc.json")) else {}
    # a partial re-run (scripts/gpu_r4_final.sh c5) refreshes its configurations only: the others keep what profiles/ holds
    old_pmc = json.load(open(os.path.join(PRO, "r04_pmc.json"))) if os.path.exists(os.path.join(PRO, "r04_pmc.json")) else {}
    old_traffic = json.load(open(os.path.join(PRO, "r04_hbm_traffic.json"))) if os.path.exists(os.path.join(PRO, "r04_hbm_traffic.json")) else {}
    for c, (W, H) in SIZES.items():
        b = os.path.join(SRC, f"{c}_bench.json")
        if not os.path.exists(b):
            continue
        for suffix in ("bench.json", "bench_profiled.json", "kernel_stats.csv"):
            s = os.path.join(SRC, f"{c}_{suffix}")
            if os.path.exists(s):
                shutil.copy(s, os.path.join(PRO, f"r04_{tag}_{c}_{suffix}"))
        d = last_json_line(b)
        hj = os.path.join(SRC, f"{c}.json")
        fresh = os.path.exists(hj) and os.path.getmtime(hj) >= os.path.getmtime(b) - 3600
        if c not in pmc and c in old_pmc:
            pmc[c] = old_pmc[c]
        if not fresh and c in old_traffic:
            traffic[c] = old_traffic[c]
        elif os.path.exists(hj):
            t = json.load(open(hj))[c]
            if not t.get("frames_profiled"):
                # scripts/hbm_summary.py did not recognise the primary kernel's name (fixed since: `wf_march_kernel<0, false>`): the
                # totals are those of the run's 4 frames (--steps 3 --warmup 1)
                for k in ("bytes_per_launch", "write_bytes", "fetch_bytes_x2"):
                    t[k] = round(t[k] / 4)
                t["per_kernel_MB"] = {k: {kk: round(vv / 4, 2) for kk, vv in v.items()} for k, v in t["per_kernel_MB"].items()}
                t["frames_profiled"] = 4
            alg = W * H * 16
            t["algorithmic_bytes"] = alg
            t["ratio"] = round(t["bytes_per_launch"] / alg, 2)
            t["source"] = (f"rocprofv3 PMC, WRITE_SIZE + 2 x FETCH_SIZE in separate passes over `bench.py --config {c} --steps 3 --warmup 1` "
                           f"(scripts/gpu_hbm_all.sh, profiles/r04_hbm_traffic.json): {t['bytes_per_launch'] / 1e6:.0f} MB per frame = "
                           f"{t['ratio']} x the 16 B/pixel")
            traffic[c] = t
        if c in pmc:
            pmc[c]["source"] = (f"rocprofv3 PMC (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_WAVE_CYCLES; three passes over `bench.py --config {c} "
                                "--steps 3 --warmup 1`, scripts/gpu_pmc_configs3.sh), all production kernels of a frame together; profiled clocks")
        r = d["roofline"]
        ex = r.get("executed", {}).get("frac")
        tr = traffic.get(c)
        rows.append(f"| {c} | {d['ms_per_step']:.3f} ({r['kernel_ms']:.3f}) | **{d['value']:.0f}** | {r['frac']:.3f}"
                    + (f"; executed {ex:.3f}" if ex is not None else "")
                    + (f"; as written {r['shader_as_written']['frac']:.2f}" if "shader_as_written" in r else "")
                    + f" | {pmc[c]['valu_issue_frac']:.2f} ({pmc[c]['lanes_live']:.0%} lanes, {pmc[c]['cycles_per_valu']:.2f} cycles per instr, {pmc[c]['resident_waves']:.1f} waves)" * (c in pmc)
                    + f" | {tr['bytes_per_launch'] / 1e6:.0f} MB = {tr['ratio']} ×" * (tr is not None)
                    + f" | {d.get('cpu_baseline', {}).get('value', float('nan')):.2f} | "
                    + ", ".join(f"{k} {v['value']:.0f}" for k, v in d.get("variants", {}).items()) + " |")
    if traffic:
        json.dump(dict(sorted(traffic.items())), open(os.path.join(PRO, "r04_hbm_traffic.json"), "w"), indent=1)
    if pmc:
        json.dump(dict(sorted(pmc.items())), open(os.path.join(PRO, "r04_pmc.json"), "w"), indent=1)
    t = os.path.join(SRC, "pmc_table.md")
    md = os.path.join(PRO, f"r04_{tag}_configs_pmc.md")
    if os.path.exists(t) and os.path.exists(md):
        new_rows = [ln for ln in open(t).read().splitlines() if ln.startswith("| c")]
        cfgs = {ln.split("|")[1].strip() for ln in new_rows}
        old = open(md).read().splitlines()
        keep = [ln for ln in old if not (ln.startswith("| c") and ln.split("|")[1].strip() in cfgs)]
        body = [ln for ln in keep if ln.startswith("| c")] + new_rows
        head = [ln for ln in keep if not ln.startswith("| c")]
        with open(md, "w") as f:
            f.write("\n".join(head).rstrip("\n") + "\n" + "\n".join(sorted(body)) + "\n")
    elif os.path.exists(t):
        with open(md, "w") as f:
            f.write("# PMC per kernel of every bench configuration, round-4 final build (`scripts/gpu_r4_final.sh` → `scripts/gpu_pmc_configs3.sh`: "
                    "rocprofv3 --pmc over `bench.py --config cN --steps 3 --warmup 1`, three counter sets in separate passes; production kernels; "
                    "per-launch averages incl. the cold first frame, profiled clocks)\n\nlanes live = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU); "
                    "cycles per instruction = ms x 2.39 GHz x 1024 SIMDs / SQ_INSTS_VALU; resident waves = 4 x SQ_WAVE_CYCLES / (ms x 2.39 GHz x 1024).\n\n")
            f.write(open(t).read())
    print("| configuration | ms per frame (kernel) | Mpixels/s | roofline.frac (algorithmic) | issue (PMC) | HBM bytes per frame ÷ 16 B/pixel | CPU oracle, 16 cores, Mpixels/s | variants (Mpixels/s) |")
    print("|---|---|---|---|---|---|---|---|")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
