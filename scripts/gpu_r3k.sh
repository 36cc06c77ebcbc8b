#!/bin/bash
# full GPU suite, then the bench lines of the two configurations the lockstep instantiation serves (c1, c2) and the headline
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3k_tests.log 2>&1 || { tail -30 gpurun_out/r3k_tests.log; exit 1; }
tail -2 gpurun_out/r3k_tests.log
for c in c1 c2 c3; do
  timeout -k 10 400 python bench.py --config $c > gpurun_out/r3k_$c.json 2> gpurun_out/r3k_$c.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3k_$c.json").read().strip().splitlines()[-1])
print("$c", d["value"], "ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], {k:v.get("ms_per_step") for k,v in d["variants"].items()}, d["parity_check"]["mismatched_words"], d["cpu_baseline"]["value"])
PY
done
