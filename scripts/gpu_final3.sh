#!/bin/bash
# round 3, final build: every BASELINE configuration as a bench line, and the same command under rocprofv3 --kernel-trace --stats
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final3
cd $R
for c in c1 c2 c3 c4 c5; do
  python bench.py --config $c > gpurun_out/final3/${c}_bench.json 2> gpurun_out/final3/${c}_bench.err || { tail -20 gpurun_out/final3/${c}_bench.err; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
for c in c1 c2 c3 c4 c5; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final3/prof_$c -o p --output-format csv -- python3 $R/bench.py --config $c --no-variants --no-cpu-baseline > $R/gpurun_out/final3/${c}_bench_profiled.json 2> $R/gpurun_out/final3/${c}_prof.err
  cp $(find $R/gpurun_out/final3/prof_$c -name '*kernel_stats.csv' | head -1) $R/gpurun_out/final3/${c}_kernel_stats.csv
done
cd $R
python - <<'PY'
import json
for c in ["c1","c2","c3","c4","c5"]:
    d=json.loads(open(f"gpurun_out/final3/{c}_bench.json").read().strip().splitlines()[-1])
    p=json.loads(open(f"gpurun_out/final3/{c}_bench_profiled.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print(c, d["value"], "Mpix/s", d["ms_per_step"], "ms | roofline", r["frac"], "kernel_ms", r["kernel_ms"], "| profiled kernel_ms", p["roofline"]["kernel_ms"], "| cpu", d["cpu_baseline"]["value"], "parity", d["parity_check"]["mismatched_words"], "| variants", {k:v["value"] for k,v in d["variants"].items()})
PY
