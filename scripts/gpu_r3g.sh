#!/bin/bash
# every BASELINE configuration as a bench line
set -e
for c in c1 c2 c3 c4 c5; do
  python bench.py --config $c > gpurun_out/r03_e_bench_$c.json 2> gpurun_out/r03_e_bench_$c.err || { tail -20 gpurun_out/r03_e_bench_$c.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_e_bench_$c.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$c", d["value"], "Mpix/s", d["ms_per_step"], "ms; roofline frac", r["frac"], "kernel_ms", r["kernel_ms"], "sched", r["kernel"][:28], "| cpu", d.get("cpu_baseline",{}).get("value"), "parity", d.get("parity_check",{}).get("mismatched_words"), "| variants", {k:v["value"] for k,v in d["variants"].items()})
PY
done
